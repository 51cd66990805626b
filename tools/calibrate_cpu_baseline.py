#!/usr/bin/env python3
"""BASELINE.md §3 calibration: the CPU restatement bench.py times on the GPU box (oracle/learner_ref.py)
against the reference's own Agent.update (imported from /root/reference) on the same container, inputs and
thread counts.  Build-container only -- the reference does not travel.

    python tools/calibrate_cpu_baseline.py [--calls 30] > gpurun_out/calibration.md
"""
import argparse
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden as G          # noqa: E402  (reference import shims + synthetic batches)


def ref_agent(i):
    from prism.config import Config, MINATAR_CONFIG
    from prism.factory import agent_factory
    from prism_amd.config import baseline_config
    cfg = Config(**MINATAR_CONFIG.__dict__)
    ours = baseline_config(i)
    for k in cfg.__dict__:               # same field values as the configuration bench.py runs
        if hasattr(ours, k):
            setattr(cfg, k, getattr(ours, k))
    cfg.device, cfg.use_cuda_graph = "cpu", False
    torch.manual_seed(cfg.seed)
    with contextlib.redirect_stdout(io.StringIO()):
        return cfg, agent_factory.build_agent(cfg, (10, 10, 4), 6)


def oracle(i):
    from oracle.learner_ref import LearnerOracle
    from prism_amd.config import baseline_config, derive
    from tests import helpers as H
    cfg = derive(baseline_config(i), device="cpu")
    with contextlib.redirect_stdout(io.StringIO()):
        sd, tgt = H.build_init_state(cfg, cfg.seed)
    return cfg, LearnerOracle(sd, H.spec_from_config(cfg), tgt)


def timed(fn, calls, warm=5):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    return (time.perf_counter() - t0) / calls * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=30)
    args = ap.parse_args()
    G._import_reference()
    rows = []
    for i in range(4):
        rcfg, agent = ref_agent(i)
        ocfg, orc = oracle(i)
        B = rcfg.batch_size
        rng = np.random.default_rng(7)
        b = G.synth_batch(rng, B)
        rb, w = G.to_ref_batch(b), torch.from_numpy(b["w"])
        ob = dict(obs=torch.from_numpy(b["obs"]).view(B, 10, 10, 4), next_obs=torch.from_numpy(b["next_obs"]).view(B, 10, 10, 4),
                  reward=torch.from_numpy(b["reward"]).flatten(), nonterminal=torch.from_numpy(b["nonterminal"]).flatten(),
                  gamma=torch.from_numpy(b["gamma"]).flatten(), action=torch.from_numpy(b["action"]).flatten())
        T, Tn = rcfg.iqn_n_current_state_quantile_samples, rcfg.iqn_n_next_state_quantile_samples
        n_tau = (1 + (1 if (not rcfg.use_target_network or rcfg.use_double_q_learning) else 0)
                 + (1 if rcfg.use_target_network else 0)) if rcfg.use_iqn else 0

        def ref_step():
            agent.update(rb, per_weights=w)

        def orc_step():
            taus = [torch.rand(T * B, 1)] + [torch.rand(Tn * B, 1) for _ in range(n_tau - 1)] if n_tau else []
            orc.update(ob, w, taus)

        res = {}
        for th in (8, 1):
            torch.set_num_threads(th)
            calls = args.calls if th == 8 else max(5, args.calls // 3)
            res[th] = (timed(ref_step, calls), timed(orc_step, calls))
        rows.append((i, B, res))
        print(f"c{i + 1}: ref {res[8][0]:.2f} / restatement {res[8][1]:.2f} ms at 8 threads; "
              f"{res[1][0]:.2f} / {res[1][1]:.2f} ms at 1", file=sys.stderr)
    print("| Config | Reference `Agent.update` ms (8 thr) | Restatement ms (8 thr) | Δ | Reference ms (1 thr) | Restatement ms (1 thr) | Δ |")
    print("|---|---|---|---|---|---|---|")
    for i, B, res in rows:
        d8 = (res[8][1] / res[8][0] - 1) * 100
        d1 = (res[1][1] / res[1][0] - 1) * 100
        print(f"| c{i + 1} (B={B}) | {res[8][0]:.2f} | {res[8][1]:.2f} | {d8:+.0f} % | {res[1][0]:.2f} | {res[1][1]:.2f} | {d1:+.0f} % |")


if __name__ == "__main__":
    main()
