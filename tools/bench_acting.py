"""Latency of Agent.forward on the HIP acting path (GPU box): python tools/bench_acting.py"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prism_amd.config import baseline_config
from prism_amd.learner import Learner
for cfg_i, name in ((2, "IQN (greedy on the quantile mean)"), (3, "IDS + IQN, ten heads")):
    cfg = baseline_config(cfg_i, device="cuda:0", log_to_wandb=False)
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
    ag = ln.agent
    rng = np.random.default_rng(0)
    for n in (1, 4, 16):
        obs = (rng.random((n, 10, 10, 4)) < 0.1).astype(np.float32)
        for _ in range(20):
            ag.forward(obs).cpu()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            a = ag.forward(obs).cpu()
        dt = (time.perf_counter() - t0) / 300
        print(f"{name:36s} n={n:3d}: {dt * 1e6:7.1f} us per Agent.forward (host -> actions on the host)")
