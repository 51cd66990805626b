"""``create_model(env_state_shape, env_n_actions, config)`` — mirrors
``/root/reference/prism/factory/model_factory.py:49-153`` for the MinAtar model family.

Reference quirks reproduced on purpose:
  * ``q_loss_fn="huber"`` still yields an MSE loss (model_factory.py:39-46 has no ``return`` in the
    huber branch); the HIP Q-head kernel is MSE-only.
  * ``config.embedding_model_final_dim`` is overwritten with the embed's output size
    (model_factory.py:83).
  * IQN gradients reach the embedding iff ``not use_ids or ids_allow_distributional_gradients``
    (model_factory.py:87).
"""
import torch

from prism_amd.agents.modules import CompositeModel, IQNHead, MinAtarEmbed, QHeads


class UnsupportedConfig(NotImplementedError):
    pass


def check_supported(config):
    """The HIP hot path covers the MinAtar model family of BASELINE.json's configs; anything else
    fails loudly here instead of silently running somewhere slower."""
    bad = []
    if config.embedding_model_type != "minatar_cnn":
        bad.append("embedding_model_type=%r (only 'minatar_cnn')" % config.embedding_model_type)
    if config.embedding_model_act_fn_id != "relu":
        bad.append("embedding_model_act_fn_id != 'relu'")
    if config.sparse_init_p != 0.0:
        bad.append("sparse_init_p != 0")
    if config.frame_stack_size != 1:
        bad.append("frame_stack_size != 1")
    if not config.use_adam:
        bad.append("optimizer other than Adam")
    if config.use_c51:
        bad.append("use_c51 (empty in the reference too)")
    if not (config.use_iqn or config.use_ids or config.use_dqn):
        bad.append("no loss head enabled")
    if bad:
        raise UnsupportedConfig("prism_amd HIP path does not support: " + "; ".join(bad))


def create_model(env_state_shape, env_n_actions, config):
    check_supported(config)
    dev = config.device
    embed = MinAtarEmbed(in_channels=int(env_state_shape[-1]), device=dev)
    config.embedding_model_final_dim = embed.output_dim

    iqn = None
    if config.use_iqn:
        propagate = (config.ids_allow_distributional_gradients and config.use_ids) or not config.use_ids
        iqn = IQNHead(n_in=embed.output_dim, n_actions=env_n_actions,
                      n_basis=config.iqn_n_basis_elements, use_layer_norm=config.use_layer_norm,
                      n_layers=config.iqn_quantile_model_layers,
                      width=config.iqn_quantile_model_feature_dim,
                      n_tau=config.iqn_n_current_state_quantile_samples,
                      n_tau_next=config.iqn_n_next_state_quantile_samples,
                      n_tau_act=config.iqn_quantile_samples_per_action,
                      huber_k=config.iqn_huber_loss_kappa, double_q=config.use_double_q_learning,
                      loss_weight=config.distributional_loss_weight, propagate_grad=propagate,
                      device=dev)
    q = None
    if config.use_ids:
        q = QHeads(embed.output_dim, env_n_actions, config.ids_n_q_heads, config.use_layer_norm,
                   config.ids_n_q_head_model_layers, config.ids_q_head_feature_dim,
                   config.use_double_q_learning, config.q_loss_weight,
                   config.ids_ensemble_variation_coef, device=dev)
    elif config.use_dqn:
        q = QHeads(embed.output_dim, env_n_actions, 1, config.use_layer_norm, config.dqn_n_model_layers,
                   config.dqn_n_model_feature_dim, config.use_double_q_learning, config.q_loss_weight,
                   0, device=dev)
    model = CompositeModel(embed, iqn, q, device=dev)
    n = sum(p.numel() for p in model.parameters())
    print("Built model with {} parameters:".format(n))
    print(model)
    return model
