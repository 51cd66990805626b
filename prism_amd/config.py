"""Configuration surface of the learner — field-for-field the reference's ``Config``.

The reference reads one flat dataclass everywhere (``/root/reference/prism/config/
algorithm_configuration.py:5-123``) and ships presets as module-level instances
(``default_config.py:3-112``, ``minatar_config.py:3-52``).  Existing experiment files construct
configs by ``Config(**BASE.__dict__)`` and then assign attributes, so this mirror keeps the same
field names, order, "all fields required" constructor and JSON round trip.  It is generated from
one table instead of a hand-written class body.

Knobs that only this MI355X build understands are NOT dataclass fields (old presets must keep
constructing unchanged); they are read with ``getattr(config, name, default)``:

    per_mass_rng      "philox" (device RNG, default) | "numpy" (host np.random, parity with a
                      seeded reference run; costs one D2H sync per sample)
    tau_rng           "philox" (in-kernel, default)  | "torch" (torch.rand on the model device)
    hip_graph         bool, capture Learner.step() into a hipGraph (default True on GPU)
"""
import dataclasses
import json

# (name, type, DEFAULT value, MINATAR override or ...)  — ``...`` means "same as DEFAULT".
_FIELDS = [
    ("env_name", str, "ALE/Defender-v5", "MinAtar/Freeway-v1"),
    ("num_processes", int, 17, 8),
    ("shared_memory_num_floats_per_process", int, 100_000, ...),
    ("training_reward_ema", float, 0.9, ...),
    ("timestep_limit", int, 50_000_000, 5_000_000),
    ("timesteps_per_iteration", int, 4, 1),
    ("timesteps_between_evaluations", int, 1_000_000, 10_000),
    ("evaluation_timestep_horizon", int, 125_000, 100_000),
    ("timesteps_per_report", int, 100_000, 10_000),
    ("episode_timestep_limit", int, 108_000, ...),
    ("run_through_redis", bool, False, ...),
    ("redis_host", str, "localhost", ...),
    ("redis_port", int, 6379, ...),
    ("redis_side", str, "server", ...),
    ("distributional_loss_weight", float, 1, ...),
    ("q_loss_weight", float, 1, ...),
    ("batch_size", int, 32, ...),
    ("gamma", float, 0.99, ...),
    ("learning_rate", float, 6.25e-5, 0.00025),
    ("max_grad_norm", float, 10.0, ...),
    ("sparse_init_p", float, 0.0, ...),
    ("reward_clipping_type", str, "dopamine_clip", "none"),
    ("loss_squish_fn_id", str, "none", "none"),
    ("use_adam", bool, True, True),
    ("adam_beta1", float, 0.9, ...),
    ("adam_beta2", float, 0.999, ...),
    ("adam_epsilon", float, 1.5e-4, ...),
    ("use_rmsprop", bool, False, False),
    ("rmsprop_alpha", float, 0.95, ...),
    ("rmsprop_epsilon", float, 0.01, ...),
    ("q_loss_fn", str, "mse", "huber"),
    ("embedding_model_final_dim", int, 3136, ...),
    ("embedding_model_layer_sizes", int, 512, ...),
    ("embedding_model_num_layers", int, 0, ...),
    ("embedding_model_type", str, "nature_atari_cnn", "minatar_cnn"),
    ("embedding_model_act_fn_id", str, "relu", "relu"),
    ("use_ids", bool, True, True),
    ("ids_use_random_samples", bool, False, False),
    ("ids_beta", float, 1, 0.8),
    ("ids_lambda", float, 0.1, ...),
    ("ids_n_q_heads", int, 10, ...),
    ("ids_q_head_feature_dim", int, 512, 128),
    ("ids_n_q_head_model_layers", int, 2, ...),
    ("ids_allow_distributional_gradients", bool, True, ...),
    ("ids_rho_lower_bound", float, 0.25, ...),
    ("ids_epsilon", float, 1e-10, ...),
    ("ids_ensemble_variation_coef", float, 1e-6, ...),
    ("use_e_greedy", bool, False, False),
    ("e_greedy_initial_epsilon", float, 1.0, ...),
    ("e_greedy_final_epsilon", float, 0.01, 0.1),
    ("e_greedy_decay_timesteps", int, 50_000_000, 100_000),
    ("n_step_returns_length", int, 3, 3),
    ("use_layer_norm", bool, True, True),
    ("use_experience_replay", bool, True, ...),
    ("experience_replay_capacity", int, 1_000_000, 100_000),
    ("num_initial_random_timesteps", int, 20_000, 5000),
    ("use_per", bool, False, True),
    ("per_alpha", float, 0.5, ...),
    ("per_beta_start", float, 0.5, ...),
    ("per_beta_end", float, 0.5, ...),
    ("per_beta_anneal_timesteps", int, 1, 5_000_000),
    ("use_iqn", bool, True, True),
    ("iqn_n_current_state_quantile_samples", int, 8, ...),
    ("iqn_n_next_state_quantile_samples", int, 8, ...),
    ("iqn_quantile_samples_per_action", int, 200, ...),
    ("iqn_n_basis_elements", int, 64, ...),
    ("iqn_quantile_model_feature_dim", int, 512, 128),
    ("iqn_quantile_model_layers", int, 1, ...),
    ("iqn_huber_loss_kappa", float, 1.0, ...),
    ("iqn_risk_policy_id", str, "neutral", ...),
    ("use_dqn", bool, False, False),
    ("dqn_n_model_layers", int, 1, 1),
    ("dqn_n_model_feature_dim", int, 512, 128),
    ("use_c51", bool, False, ...),
    ("use_double_q_learning", bool, False, False),
    ("use_target_network", bool, False, False),
    ("target_update_period", int, 8_000, 1000),
    ("seed", int, 123, ...),
    ("hours_per_checkpoint", float, 1.0, ...),
    ("checkpoint_dir", str, "data/checkpoints", ...),
    ("log_to_wandb", bool, False, ...),
    ("wandb_group_name", str, "debug", ...),
    ("wandb_run_name", str, "null", ...),
    ("wandb_project_name", str, "Prism", ...),
    ("device", str, "cuda:0", ...),
    ("env_device", str, "cpu", ...),
    ("use_cuda_graph", bool, True, True),
    ("render", bool, False, ...),
    ("frame_stack_size", int, 4, 1),
    ("atari_sticky_actions_prob", float, 0, 0.1),
    ("atari_noops", int, 30, ...),
]


def _serialize(self):
    return json.dumps(self.__dict__)


def _deserialize(cls, serialized_config):
    raw = serialized_config.decode("utf-8") if isinstance(serialized_config, (bytes, bytearray)) \
        else serialized_config
    return cls(**dict(json.loads(raw)))


Config = dataclasses.make_dataclass(
    "Config", [(name, typ) for name, typ, _d, _m in _FIELDS],
    namespace={"serialize": _serialize, "deserialize": classmethod(_deserialize)})
Config.__module__ = __name__
Config.__doc__ = "Flat learner configuration; see module docstring."

DEFAULT_CONFIG = Config(**{name: d for name, _t, d, _m in _FIELDS})
MINATAR_CONFIG = Config(**{name: (d if m is ... else m) for name, _t, d, m in _FIELDS})


# The two ablation presets the reference's MinAtar experiments start from
# (prism/config/additive_ablation_base_config.py:2-52, subtractive_ablation_base_config.py:2-52), as the
# fields that differ from DEFAULT_CONFIG.  The one-element tuples are the reference's own values (a trailing
# comma after the string there); every consumer treats them as "none" / no clipping.
_ABLATION_COMMON = dict(
    embedding_model_type="minatar_cnn", atari_sticky_actions_prob=0.1, evaluation_timestep_horizon=100_000,
    timesteps_per_report=50_000, timestep_limit=3_000_000, experience_replay_capacity=3_000_000,
    learning_rate=0.0001, batch_size=64, num_initial_random_timesteps=4_000, frame_stack_size=1,
    adam_epsilon=0.0003125, loss_squish_fn_id=("none",), reward_clipping_type=("dopamine clamp",),
    log_to_wandb=True, iqn_n_current_state_quantile_samples=32, iqn_n_next_state_quantile_samples=32,
    iqn_quantile_samples_per_action=32, iqn_quantile_model_feature_dim=256)
_ADDITIVE = dict(
    _ABLATION_COMMON, use_ids=False, use_layer_norm=False, dqn_n_model_feature_dim=256, dqn_n_model_layers=2,
    use_target_network=True, use_e_greedy=True, e_greedy_decay_timesteps=250_000, n_step_returns_length=1,
    target_update_period=4_000, num_processes=1, wandb_group_name="Second Prism Additive Ablation Experiment")
_SUBTRACTIVE = dict(
    _ABLATION_COMMON, ids_q_head_feature_dim=256, use_per=True, num_processes=16,
    wandb_group_name="Prism Subtractive Ablation Experiment")


def derive(base, **overrides):
    """``Config(**base.__dict__)`` + attribute assignment, the reference's preset idiom
    (minatar_config.py:3, additive_ablation_experiment.py:36).  Unknown names become plain
    attributes (the optional MI355X knobs)."""
    known = {f.name for f in dataclasses.fields(Config)}
    cfg = Config(**{k: v for k, v in base.__dict__.items() if k in known})
    for k, v in base.__dict__.items():
        if k not in known:
            setattr(cfg, k, v)
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


ADDITIVE_ABLATION_BASE_CONFIG = derive(DEFAULT_CONFIG, **_ADDITIVE)
SUBTRACTIVE_ABLATION_BASE_CONFIG = derive(DEFAULT_CONFIG, **_SUBTRACTIVE)


def baseline_config(i, **overrides):
    """The five workload configurations BASELINE.json names (``configs[i]``), on MinAtar/Breakout
    shapes (obs 10x10x4, 6 actions)."""
    common = dict(env_name="MinAtar/Breakout-v1", use_cuda_graph=False)
    table = [
        dict(use_ids=False, use_iqn=False, use_dqn=True, use_per=False, use_layer_norm=False,
             n_step_returns_length=1, batch_size=32, use_e_greedy=True),
        dict(use_ids=False, use_iqn=False, use_dqn=True, use_per=True, use_layer_norm=False,
             n_step_returns_length=1, batch_size=256, use_e_greedy=True),
        dict(use_ids=False, use_iqn=True, use_dqn=False, use_per=True, use_layer_norm=True,
             n_step_returns_length=3, batch_size=256),
        dict(use_ids=True, use_iqn=True, use_dqn=False, use_per=True, use_layer_norm=True,
             n_step_returns_length=3, batch_size=512, use_target_network=True),
        dict(use_ids=True, use_iqn=True, use_dqn=False, use_per=True, use_layer_norm=True,
             n_step_returns_length=3, batch_size=512, use_target_network=True,
             experience_replay_capacity=10_000_000),
    ]
    kw = dict(common)
    kw.update(table[i])
    kw.update(overrides)
    return derive(MINATAR_CONFIG, **kw)
