"""Step rate of the reference's ablation presets on the HIP path (GPU box): python tools/bench_presets.py"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prism_amd import config as C
from prism_amd.learner import Learner
from prism_amd.synthetic import fill_replay

CASES = {
    "additive base (IQN, width 256, T=32, no LN, uniform replay, B=64)": (C.ADDITIVE_ABLATION_BASE_CONFIG, {}),
    "additive + PER + n-step + LN": (C.ADDITIVE_ABLATION_BASE_CONFIG, dict(use_per=True, n_step_returns_length=3, use_layer_norm=True)),
    "additive + IDS (10 two-layer heads of width 256)": (C.ADDITIVE_ABLATION_BASE_CONFIG,
                                                         dict(use_ids=True, ids_n_q_head_model_layers=2, ids_n_q_heads=10,
                                                              ids_q_head_feature_dim=256, ids_ensemble_variation_coef=0)),
    "subtractive base (everything on)": (C.SUBTRACTIVE_ABLATION_BASE_CONFIG, {}),
}
only = sys.argv[1] if len(sys.argv) > 1 else None        # substring of a case name: run just that one
for name, (base, over) in CASES.items():
    if only and only not in name:
        continue
    cfg = C.derive(base, device="cuda:0", experience_replay_capacity=100_000, log_to_wandb=False, **over)
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
    ln.time_phases = False
    fill_replay(ln.experience_buffer, ln.experience_buffer.capacity, seed=0)
    n = int(os.environ.get("PRESET_STEPS", "1000"))            # (counter passes: PRESET_STEPS=40 PRESET_EAGER=1)
    eager = os.environ.get("PRESET_EAGER") == "1"
    for _ in range(min(100, n)):
        ln.step(eager=eager)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ln.step(eager=eager)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:72s} {n / dt:9.0f} steps/s  {dt / n * 1e6:7.1f} us/step  params {ln.agent.flat.numel()}")
    # per-kernel HIP-event times of 64 eager steps
    import ctypes
    from prism_amd import _native as N
    L = N.lib()
    ms = (ctypes.c_double * N.N_KERNEL_IDS)()
    cnt = (ctypes.c_int64 * N.N_KERNEL_IDS)()
    L.prism_profile_collect(ms, cnt)
    for i in range(N.N_KERNEL_IDS):
        ms[i], cnt[i] = 0.0, 0
    L.prism_profile_enable(1)
    for _ in range(64):
        ln.step(eager=True)
    L.prism_profile_enable(0)
    torch.cuda.synchronize()
    L.prism_profile_collect(ms, cnt)
    print("      ", {L.prism_profile_kernel_name(i).decode(): round(ms[i] / cnt[i] * 1e3, 1) for i in range(N.N_KERNEL_IDS) if cnt[i]})
