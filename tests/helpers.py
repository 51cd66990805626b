"""Shared test helpers: golden-fixture decoding and oracle construction."""
import contextlib
import io
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

UPDATE_CASES = ["iqn_small", "iqn_c3", "dqn_c2", "dqn_ln", "dqn_target_c2", "full_c4", "full_small",
                "full_notarget", "full_doubleq", "iqn_target", "iqn_doubleq", "iqn_tau32",
                # the reference's ablation presets and the stages its experiment files derive from them
                "abl_iqn", "abl_ln_notarget", "abl_doubleq", "abl_ids", "abl_ids_var", "abl_sub"]


def load_case(name):
    return np.load(os.path.join(GOLDEN, f"update_{name}.npz"))


def case_overrides(g):
    return {str(k): eval(str(v), {}, {}) for k, v in zip(g["overrides_keys"], g["overrides_vals"])}


def case_config(g, device="cpu", **extra):
    from prism_amd import config as C
    base = {"minatar": C.MINATAR_CONFIG, "additive": C.ADDITIVE_ABLATION_BASE_CONFIG,
            "subtractive": C.SUBTRACTIVE_ABLATION_BASE_CONFIG}[str(g["base"]) if "base" in g else "minatar"]
    kw = dict(device=device, use_cuda_graph=False, use_e_greedy=False)
    kw.update(case_overrides(g))
    kw.update(extra)
    return C.derive(base, **kw)


def case_batch(g, step):
    B, C = int(g["B"]), int(g["C"])
    pre = f"s{step}/"
    n = B * 100 * C
    obs = np.unpackbits(g[pre + "obs_bits"])[:n].reshape(B, 10, 10, C).astype(np.float32)
    nobs = np.unpackbits(g[pre + "next_obs_bits"])[:n].reshape(B, 10, 10, C).astype(np.float32)
    batch = dict(obs=torch.from_numpy(obs), next_obs=torch.from_numpy(nobs),
                 reward=torch.from_numpy(g[pre + "reward"]).flatten(),
                 nonterminal=torch.from_numpy(g[pre + "nonterminal"]).flatten(),
                 gamma=torch.from_numpy(g[pre + "gamma"]).flatten(),
                 action=torch.from_numpy(g[pre + "action"]).flatten())
    w = torch.from_numpy(g[pre + "w"])
    taus = [torch.from_numpy(g[pre + f"tau{i}"]).reshape(-1, 1) for i in range(int(g[pre + "n_taus"]))]
    return batch, w, taus


def spec_from_config(cfg, C=4, A=6):
    from oracle.learner_ref import ModelSpec
    propagate = (cfg.ids_allow_distributional_gradients and cfg.use_ids) or not cfg.use_ids
    if cfg.use_ids:
        heads, hl, coef = cfg.ids_n_q_heads, cfg.ids_n_q_head_model_layers, cfg.ids_ensemble_variation_coef
    elif cfg.use_dqn:
        heads, hl, coef = 1, cfg.dqn_n_model_layers, 0.0
    else:
        heads, hl, coef = 0, 0, 0.0
    return ModelSpec(in_channels=C, n_actions=A, use_iqn=cfg.use_iqn, use_layer_norm=cfg.use_layer_norm,
                     n_basis=cfg.iqn_n_basis_elements, iqn_layers=cfg.iqn_quantile_model_layers,
                     n_tau=cfg.iqn_n_current_state_quantile_samples,
                     n_tau_next=cfg.iqn_n_next_state_quantile_samples, huber_k=cfg.iqn_huber_loss_kappa,
                     dist_loss_weight=cfg.distributional_loss_weight, propagate_grad=propagate,
                     n_heads=heads, head_layers=hl, q_loss_weight=cfg.q_loss_weight, theil_coef=coef,
                     double_q=cfg.use_double_q_learning, max_grad_norm=cfg.max_grad_norm,
                     lr=cfg.learning_rate, beta1=cfg.adam_beta1, beta2=cfg.adam_beta2,
                     adam_eps=cfg.adam_epsilon)


def build_init_state(cfg, seed, C=4, A=6):
    """Initial weights by construction (same seed, same layer creation order as the reference);
    returns (state_dict, target_state_dict or None)."""
    from prism_amd.factory.model_factory import create_model
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = create_model((10, 10, C), A, cfg)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        tgt = None
        if cfg.use_target_network:
            create_model((10, 10, C), A, cfg)      # advances the RNG like agent_factory.py:15-18
            tgt = {k: v.clone() for k, v in sd.items()}
    return sd, tgt


def checksums(sd):
    s = np.array([float(v.double().sum()) for v in sd.values()])
    l2 = np.array([float(v.double().norm()) for v in sd.values()])
    return s, l2


def case_act(g):
    """Acting inputs / the reference's outputs of an update fixture: (obs (n,10,10,C), taus (T*n,1) or None, dict)."""
    n, C = int(g["act/n"]), int(g["C"])
    obs = np.unpackbits(g["act/obs_bits"])[:n * 100 * C].reshape(n, 10, 10, C).astype(np.float32)
    taus = torch.from_numpy(g["act/tau"]).reshape(-1, 1) if "act/tau" in g.files else None
    return torch.from_numpy(obs), taus, {k[4:]: g[k] for k in g.files if k.startswith("act/")}
