"""Latency of Agent.forward on the HIP acting path (GPU box): python tools/bench_acting.py
host -> actions on the host, per call, for the hipGraph form (default) and the eager launches (config.act_graph = False):
  numpy in      a host array in, .cpu() out (evaluators: sync_agent_evaluator.py:43)
  device in     the reference collector's pattern: one persistent device inference buffer in, .cpu().numpy() out
                (multiprocessing_experience_collection/experience_collector.py:77-78,127)"""
import contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prism_amd.config import baseline_config
from prism_amd.learner import Learner


def measure(cfg_i, graph, reps=400):
    cfg = baseline_config(cfg_i, device="cuda:0", log_to_wandb=False, act_graph=graph)
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
    ag = ln.agent
    rng = np.random.default_rng(0)
    out = {}
    for n in (1, 4, 16):
        obs = (rng.random((n, 10, 10, 4)) < 0.1).astype(np.float32)
        dev_buf = torch.zeros((n, 10, 10, 4), device="cuda:0")
        host_t = torch.from_numpy(obs)
        for mode in ("numpy in", "device in"):
            def call():
                if mode == "numpy in":
                    return ag.forward(obs).cpu()
                dev_buf.copy_(host_t, non_blocking=True)
                return ag.forward(dev_buf).cpu().numpy()
            for _ in range(30):
                call()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                call()
                ts.append(time.perf_counter() - t0)
            out[f"n={n} {mode}"] = round(float(np.median(ts)) * 1e6, 1)
    return out


if __name__ == "__main__":
    res = {}
    for cfg_i, name in ((2, "IQN (greedy on the quantile mean)"), (3, "IDS + IQN, ten heads")):
        for graph in (True, False):
            r = measure(cfg_i, graph)
            res[f"configs[{cfg_i}] {'graph' if graph else 'eager'}"] = r
            for k, v in r.items():
                print(f"{name:36s} {'hipGraph' if graph else 'eager   '} {k:18s}: {v:7.1f} us per Agent.forward (host -> actions on the host)")
    print(json.dumps(res))
