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
                "abl_iqn", "abl_ln_notarget", "abl_doubleq", "abl_ids", "abl_ids_var", "abl_sub",
                # value squish of the TD target (loss_squish_fn_id): symlog / obs_look_further
                "iqn_symlog", "full_olf", "dqn_symlog"]


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
                     squish=str(cfg.loss_squish_fn_id),
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


def jitter_grads(sd, tgt, spec, batch, w, taus, seed, draws=6, ulps=2.4e-7):
    """The oracle's fp32 gradient at parameters jittered by about two units in the last place (`draws` times): where
    these disagree with the plain fp32 gradient, a ReLU unit sits within rounding distance of zero and two correct
    evaluations may differ by its whole contribution."""
    from oracle.learner_ref import LearnerOracle
    gen = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(draws):
        jit = {k: v * (1.0 + ulps * torch.randn(v.shape, generator=gen)) for k, v in sd.items()}
        pj = LearnerOracle(jit, spec, tgt)
        pj.update(batch, w, taus, apply=False)
        out.append(pj.last["grads"])
    return out


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


# ---- Philox4x32-10 on the host (Salmon et al., SC'11) -- restates prism_amd/csrc/common.h::Philox so that a test can
# re-draw what a kernel drew on the device (PER masses, quantile samples) from (seed, counter, stream key)
def philox4x32(seed, ctr, stream):
    """ctr: uint64 array of counters; returns uint32 array [n, 4]."""
    ctr = np.asarray(ctr, dtype=np.uint64)
    m32 = np.uint64(0xFFFFFFFF)
    c0, c1 = ctr & m32, ctr >> np.uint64(32)
    c2 = np.full_like(ctr, np.uint64(stream) & m32)
    c3 = np.full_like(ctr, np.uint64(stream) >> np.uint64(32))
    a, b = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        h0, l0, h1, l1 = p0 >> np.uint64(32), p0 & m32, p1 >> np.uint64(32), p1 & m32
        c0, c1, c2, c3 = h1 ^ c1 ^ a, l1, h0 ^ c3 ^ b, l0
        a = (a + np.uint64(0x9E3779B9)) & m32
        b = (b + np.uint64(0xBB67AE85)) & m32
    return np.stack([c0, c1, c2, c3], axis=1).astype(np.uint32)


def philox_per_mass(seed, offset, batch, p_sum):
    """The masses step_front_kernel / per_sample_kernel draw: U(0, p_sum) in float64 (53 random bits, as numpy's
    random_sample), narrowed to fp32 (replay_kernels.h, key "PERM")."""
    r = philox4x32(seed, np.uint64(offset) + np.arange(batch, dtype=np.uint64), 0x5045524D)
    a, b = (r[:, 0] >> np.uint32(5)).astype(np.float64), (r[:, 1] >> np.uint32(6)).astype(np.float64)
    u = (a * 67108864.0 + b) / 9007199254740992.0
    return (0.0 + (np.float64(np.float32(p_sum)) - 0.0) * u).astype(np.float32)


def philox_uniform_index(seed, offset, batch, size):
    """The slots uniform_sample_kernel / step_front_kernel draw for uniform replay (torchrl RandomSampler,
    exp_buffer_factory.py:30-33): floor(x * size / 2**64) of a 64-bit Philox word, key "UNIF"."""
    r = philox4x32(seed, np.uint64(offset) + np.arange(batch, dtype=np.uint64), 0x554E4946)
    return np.array([((int(a) << 32 | int(b)) * int(size)) >> 64 for a, b in zip(r[:, 0], r[:, 1])], dtype=np.int64)
