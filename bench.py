#!/usr/bin/env python3
"""Learner grad-steps/sec on the replay-sample -> TD-update -> priority-writeback hot path.

    python bench.py --gpus 1 --steps K --warmup W            # BASELINE.json metric config (c3)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = Learner.step(): PER sample + n-step gather + IQN TD update (fwd, loss, bwd, clip, Adam)
+ priority writeback, on synthetic MinAtar/Breakout-shaped transitions already resident in HBM.
N > 1: one process per GPU, replay sharded by capacity (each rank samples its own shard), one RCCL
all-reduce of the flat gradient per step (weak scaling: per-GPU batch fixed).

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel,
HIP-event timed on the launch stream inside the timed region) and `cpu_baseline` (the CPU oracle
port timed on this box's host cores, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBS = 8000.0             # HBM3E spec


def algorithmic_bytes(cfg, P, P_tgt, cap2_levels):
    """SURVEY.md §8d 'algorithmic (compulsory) bytes per step'."""
    B, O = cfg.batch_size, 400
    L = cap2_levels
    rows = B * (8 * O + 17)
    sample = B * (4 * L + 4 + 12)
    wb = B * 4 + 2 * B * (4 + 12 * L)
    return rows + sample + wb + 24 * P + 4 * P_tgt + 4 * B + 4


def kernel_flops(cfg, B, A=6):
    """ALGORITHMIC flops per launch of the GEMM kernels: forward = phi + trunk + head of every
    row; backward = dW of phi, dW and dX of the trunk (the phi columns the backward kernel recomputes instead of
    reading them back are not algorithmic work and are not counted)."""
    T, Tn, E, K = cfg.iqn_n_current_state_quantile_samples, cfg.iqn_n_next_state_quantile_samples, 1024, 64
    H = cfg.iqn_quantile_model_feature_dim
    n_next = 2 if (cfg.use_target_network and cfg.use_double_q_learning) else 1
    fwd = bwd = qbwd = 0
    if cfg.use_iqn:
        fwd += (B * T + n_next * B * Tn) * (2 * K * E + 2 * E * H + 2 * H * A)
        bwd += B * T * 2 * E * (K + H + H)
    heads = cfg.ids_n_q_heads if cfg.use_ids else (1 if cfg.use_dqn and cfg.dqn_n_model_layers == 2 else 0)
    if heads:
        H = cfg.ids_q_head_feature_dim if cfg.use_ids else cfg.dqn_n_model_feature_dim
        fwd += (1 + n_next) * B * heads * (2 * E * H + 2 * H * A)
        qbwd += B * heads * 2 * E * (H + H)
    fl = {"fwd_tile_kernel": fwd}
    if bwd:
        fl["iqn_bwd_kernel"] = bwd
    if qbwd:
        fl["qh_bwd_kernel"] = qbwd
    return fl


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def collate_python_loop_ms(rp, B, n_batches=40):
    """The reference's collate cost class: `_timesteps_to_batch` (/root/reference/prism/experience/timestep_buffer.py:
    79-196) is a per-sample Python loop -- for each of the B sampled timesteps one row assignment into each of the six
    static-batch tensors (plus the n-step walk of timestep_buffer.py:198-238 while it is not cached).  Restated here over
    the port's ring arrays (cached n-step values, the steady state): same number of Python-level tensor row stores.
    The port itself collates with one vectorised gather, so this is reported beside it, not inside it."""
    rng = np.random.default_rng(1)
    obs = torch.zeros(B, 1, 10, 10, 4)
    next_obs = torch.zeros(B, 1, 10, 10, 4)
    rewards, gammas = torch.zeros(B, 1), torch.ones(B, 1)
    nonterminals, actions = torch.zeros(B, 1, dtype=torch.bool), torch.zeros(B, 1, dtype=torch.long)
    ring_obs, ring_succ = rp.obs.reshape(-1, 10, 10, 4), rp.succ_obs.reshape(-1, 10, 10, 4)
    t_total = 0.0
    for _ in range(n_batches):
        idx = rng.integers(0, rp.length, B)
        g = rp.gather(idx)                      # the cached n-step scalars of these timesteps
        last = idx                              # (next observation: the successor row of the slot)
        ret, gam, done, act = g["reward"], g["gamma"], 1 - g["nonterminal"], g["action"]
        t0 = time.perf_counter()
        for i in range(B):
            s = int(idx[i])
            if done[i]:
                obs[i] = torch.from_numpy(ring_obs[s])
                next_obs[i] = obs[i]
            else:
                obs[i] = torch.from_numpy(ring_obs[s])
                next_obs[i] = torch.from_numpy(ring_succ[int(last[i])])
            rewards[i] = float(ret[i])
            nonterminals[i] = 1 - int(done[i])
            gammas[i] = float(gam[i])
            actions[i] = int(act[i])
        t_total += time.perf_counter() - t0
    return t_total / n_batches * 1e3


def cpu_baseline(cfg, seconds=20.0):
    """The CPU oracle port of the same step (C sum tree + n-step gather, torch-CPU TD update)."""
    import contextlib
    import io
    from oracle import per_ref
    from oracle.learner_ref import LearnerOracle
    from tests import helpers as H
    from prism_amd.config import derive
    ccfg = derive(cfg, device="cpu")
    cap = min(cfg.experience_replay_capacity, 100_000)
    rng = np.random.default_rng(0)
    rp = per_ref.ReplayOracle(cap, 400, cfg.n_step_returns_length, cfg.gamma, cfg.per_alpha, 0.5, use_per=True)
    rp.obs[:] = rng.random((cap, 400), dtype=np.float32) < 0.1
    rp.succ_obs[:] = rng.random((cap, 400), dtype=np.float32) < 0.1
    rp.reward[:] = rng.standard_normal(cap).astype(np.float32)
    rp.action[:] = rng.integers(0, 6, cap)
    done = rng.random(cap) < 0.017
    rp.flags[:] = done * 1 + (~done) * 4
    link = np.arange(cap) + 8
    rp.link[:] = np.where((link < cap) & ~done, link, -1)
    rp.length = cap
    prio = (np.abs(rng.standard_normal(cap)) ** 0.5 + 1e-8).astype(np.float32)
    rp.sampler.sum_tree.update(np.arange(cap), prio)
    rp.sampler.min_tree.update(np.arange(cap), prio)
    torch.manual_seed(cfg.seed)
    with contextlib.redirect_stdout(io.StringIO()):
        sd, tgt = H.build_init_state(ccfg, cfg.seed)
    orc = LearnerOracle(sd, H.spec_from_config(ccfg), tgt)
    B, T, Tn = cfg.batch_size, cfg.iqn_n_current_state_quantile_samples, cfg.iqn_n_next_state_quantile_samples
    n_tau = 1 + (1 if (not cfg.use_target_network or cfg.use_double_q_learning) else 0) + \
        (1 if cfg.use_target_network else 0)

    def one():
        mass = rp.sampler.draw_mass(cap, B)
        idx, w, _, _ = rp.sampler.sample(cap, mass)
        g = rp.gather(idx)
        batch = dict(obs=torch.from_numpy(g["obs"]).view(B, 10, 10, 4), next_obs=torch.from_numpy(g["next_obs"]).view(B, 10, 10, 4),
                     reward=torch.from_numpy(g["reward"]), nonterminal=torch.from_numpy(g["nonterminal"].astype(bool)),
                     gamma=torch.from_numpy(g["gamma"]), action=torch.from_numpy(g["action"]))
        taus = [torch.rand(T * B, 1)] + [torch.rand(Tn * B, 1) for _ in range(n_tau - 1)]
        td = orc.update(batch, torch.from_numpy(w), taus)
        rp.sampler.update_priority(idx, td.abs().numpy())

    # thread sweep (oversubscribing a many-core host thrashes: the rate at nproc threads is NOT the best one)
    nproc = os.cpu_count() or 1
    sweep = sorted({t for t in (1, 8, 16, 32, min(nproc, 64)) if t <= nproc})     # (256 threads: 12 s per step, pure thrash)
    per = max(1.5, seconds / len(sweep))
    rates = {}
    for th in sweep:
        torch.set_num_threads(th)
        for _ in range(2):
            one()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < per:
            one()
            n += 1
        rates[th] = n / (time.perf_counter() - t0)
    best = max(rates, key=rates.get)
    torch.set_num_threads(1)
    collate_ms = collate_python_loop_ms(rp, B)
    torch.set_num_threads(nproc)
    return {"value": round(rates[best], 3), "unit": "steps/s", "cores": nproc, "threads": best, "kind": "port",
            "host_cpu": host_cpu_model(),
            "collate_python_loop_ms": round(collate_ms, 3),
            "collate_note": "per-sample Python collate loop of the reference (timestep_buffer.py:79-196), restated over the "
                            "port's arrays and timed on its own: NOT part of `value` (the port gathers vectorised); a "
                            "reference learner pays it on top of every update",
            "one_thread": round(rates[1], 3), "by_threads": {str(k): round(v, 3) for k, v in rates.items()},
            "sample": f"the same workload (B={B}, replay {cap}) for {per:.1f} s at each of {sweep} torch threads: C sum-tree "
                      f"sample + n-step gather + torch-CPU TD update + priority writeback; value = best, one_thread = 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json configs[i]; 2 = the metric's config")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-acting", action="store_true", help="skip the Agent.forward latency leg (profiling runs: its launches "
                    "share kernel names with the learner's)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed blocks of --steps steps (0: as many as fill --min-seconds of timed GPU work, at least 5)")
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="with --repeats 0: total duration of the timed blocks (the GPU is busy for this long back to back)")
    ap.add_argument("--profile-steps", type=int, default=128, help="eager, HIP-event instrumented steps after the timing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # PRISM_BENCH_REHEARSAL=1: run the N > 1 code path with all ranks on GPU 0 and gloo over host memory
    # (a one-GPU box cannot host two RCCL ranks); the numbers mean nothing, the control flow is the point
    rehearsal = os.environ.get("PRISM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
            from prism_amd import dist as pdist

            def host_allreduce(flat, group=None):
                h = flat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                flat.copy_(h)
                return 1.0 / dist.get_world_size(group)
            pdist.allreduce_grads = host_allreduce
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))
        pg = dist.group.WORLD

    from prism_amd import _native as N
    from prism_amd.config import baseline_config
    from prism_amd.learner import Learner
    from prism_amd.synthetic import fill_replay
    import contextlib
    import io

    cfg = baseline_config(args.config, device=dev)
    if args.config == 4:
        from prism_amd.dist import shard_capacity
        cfg.experience_replay_capacity = shard_capacity(10_000_000, world)
    cfg.per_seed_offset = rank
    cfg.hip_graph = not args.no_graph
    learner = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        learner.configure(cfg, obs_shape=(10, 10, 4), n_actions=6, process_group=pg)
    learner.time_phases = False
    buf, agent = learner.experience_buffer, learner.agent
    # which matrix pipe the GEMMs run on: the config knob, else PRISM_GEMM, else the library default
    gemm_mode = str(getattr(cfg, "gemm_mode", "auto"))
    if gemm_mode not in ("fp32", "bf16x3"):
        gemm_mode = os.environ.get("PRISM_GEMM", "")
    if gemm_mode not in ("fp32", "bf16x3"):
        gemm_mode = "bf16x3"          # PRISM_GEMM_DEFAULT (include/prism_hip.h)
    from prism_amd.dist import rank_seeds
    _, buf.seed, agent.seed = rank_seeds(cfg.seed, rank)
    fill_replay(buf, buf.capacity, seed=rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    L = N.lib()
    for _ in range(args.warmup):
        learner.step()
    sync()
    import ctypes
    ms = (ctypes.c_double * N.N_KERNEL_IDS)()
    cnt = (ctypes.c_int64 * N.N_KERNEL_IDS)()
    L.prism_profile_collect(ms, cnt)
    for i in range(N.N_KERNEL_IDS):
        ms[i], cnt[i] = 0.0, 0
    # timed region: `repeats` blocks of EXACTLY `steps` steps each, every block bracketed by barrier + synchronize;
    # the reported rate is the median block (a 20-step block is 2 ms: one block alone is mostly noise).  With --repeats 0
    # the number of blocks is sized from the first one so that the timed blocks add up to >= --min-seconds of back-to-back
    # GPU work (the same count on every rank: it comes from the MAX-over-ranks time of block 0)
    def timed_block():
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            learner.step()
        sync()
        own = time.perf_counter() - t0
        dt = own
        if world > 1:
            t = torch.tensor([own], device="cpu" if rehearsal else dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return own, dt

    own0, dt0 = timed_block()
    repeats = args.repeats if args.repeats > 0 else int(min(20000, max(5, -(-args.min_seconds // max(dt0, 1e-6)))))
    blocks, own_blocks = [dt0], [own0]
    for _ in range(repeats - 1):
        own, dt = timed_block()
        own_blocks.append(own)
        blocks.append(dt)
    # a step that hit an abandoned grid barrier or a collective time-out applied no (or half an) update: such a run has
    # no rate to report (sticky status words of the learner workspace, the replay and the direct collective)
    agent.check_status()
    buf.check_status()
    elapsed = float(np.median(blocks))
    dp_info = None
    if world > 1:
        # self-check of the data-parallel run: who took part, how the collective was launched, every rank's own rate
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, args.steps / float(np.median(own_blocks)))
        dp_info = {"rccl_ranks": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                   "collective_in_graph": bool(getattr(agent, "collective_in_graph", False)),
                   "graph_capture_fallback": getattr(agent, "_capture_error", None),
                   "per_rank_steps_per_s": [round(float(x), 2) for x in per_rank],
                   "replicas_identical": None}
        from prism_amd.dist import assert_replicas_identical
        dp_info["replicas_identical"] = bool(assert_replicas_identical(agent.flat if not rehearsal else agent.flat.cpu(), pg))
    # per-kernel durations: HIP events on the launch stream around every launch of `profile_steps` eager steps
    # (outside the timed blocks: the events cost ~2 us per launch)
    L.prism_profile_enable(1)
    for _ in range(args.profile_steps):
        learner.step(eager=True)
    L.prism_profile_enable(0)
    sync()
    L.prism_profile_collect(ms, cnt)

    if rank == 0:
        kern = {}
        for i in range(N.N_KERNEL_IDS):
            if cnt[i]:
                kern[L.prism_profile_kernel_name(i).decode()] = ms[i] / cnt[i] * 1e3     # us per launch
        fl = {k: v for k, v in kernel_flops(cfg, cfg.batch_size).items() if v}
        # the instrumentation names launches by role; the IQN backward that RUNS at width 128 on the bf16 pipe is
        # iqn_bwd3_kernel (bwd3_kernels.h), elsewhere iqn_bwd_kernel
        bwd_name = "iqn_bwd3_kernel" if (gemm_mode == "bf16x3" and cfg.iqn_quantile_model_feature_dim == 128) else "iqn_bwd_kernel"
        for d_ in (kern, fl):
            if "iqn_bwd_kernel" in d_:
                d_[bwd_name] = d_.pop("iqn_bwd_kernel")
        P = agent.flat.numel()
        P_tgt = P if cfg.use_target_network else 0
        levels = int(np.log2(buf.tree_capacity))
        abytes = algorithmic_bytes(cfg, P, P_tgt, levels)
        step_s = elapsed / args.steps
        Bsz = cfg.batch_size
        wb_bytes = (4 * Bsz + 2 * Bsz * (4 + 12 * levels)) if cfg.use_per else 0
        kbytes = {"step_front_kernel": Bsz * (8 * 400 + 17) + (Bsz * (4 * levels + 16) if cfg.use_per else 8 * Bsz),
                  "step_back_kernel": 24 * P + wb_bytes}
        kbytes["step_tail_kernel"] = kbytes["step_back_kernel"]      # the fused tail: the same compulsory bytes
        # the gradient reduction as a launch of its own (c4/c5): what this decomposition has to move -- every gradient
        # slab read once, the flat gradient written once (SURVEY 8d counts gradients as on-chip: 0 compulsory bytes)
        Hq = cfg.ids_q_head_feature_dim if cfg.use_ids else cfg.dqn_n_model_feature_dim
        n_heads = cfg.ids_n_q_heads if cfg.use_ids else (1 if cfg.use_dqn and cfg.dqn_n_model_layers == 2 else 0)
        if cfg.use_iqn or n_heads:
            Hi = cfg.iqn_quantile_model_feature_dim
            ln = 2048 if cfg.use_layer_norm else 0
            slab = (1024 * 64 + 1024 + ln + Hi * 1024) * (8 if Hi == 128 else 4) if cfg.use_iqn else 0
            qslab = n_heads * (ln + Hq * 1024)
            kbytes["iqn_post_kernel"] = 4 * (slab + qslab) + 4 * P
        traffic_all, traffic_src, rocprof_us, rocprof_tag = {}, None, {}, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic_all = tj.get(f"configs{args.config}", {})
            rocprof_tag = tj.get("_meta", {}).get(f"configs{args.config}", "r02e")
            traffic_src = {"file": "profiles/traffic.json", "tag": rocprof_tag, "measured_in_run": False,
                           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of an earlier run of this command (NOT measured "
                                   "in this run): (2*FETCH + WRITE)*1024 per launch"}
            # the rocprofv3 --kernel-trace --stats averages committed with that tag (no launch boundary in them)
            spath = os.path.join(ROOT, "profiles", f"{rocprof_tag}_configs{args.config}_kernel_stats.csv")
            if os.path.exists(spath):
                import csv
                for r in csv.DictReader(open(spath)):
                    try:
                        if "prism::" in r["Name"] and int(r["Calls"]) > 100:
                            rocprof_us[r["Name"].split("(")[0].replace("void ", "").replace("prism::", "").split("<")[0]] = \
                                float(r["AverageNs"]) / 1e3
                    except (KeyError, ValueError, TypeError):
                        pass
        roof = None
        if kern:
            mf = [k for k in fl if kern.get(k)]
            if mf:      # a GEMM kernel dominates: fp32-MFMA roofline
                dom = max(mf, key=lambda k: kern[k])
                ach = fl[dom] / (kern[dom] * 1e-6) / 1e12
                roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": FP32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": traffic_all.get(dom), "flops_per_launch": fl[dom]}
                if rocprof_us.get(dom):
                    # the same kernel by rocprofv3's own duration (committed profile of an earlier run of this command: no
                    # launch boundary inside): printed beside the event-timed figure so the two cannot drift apart silently
                    roof["frac_rocprof"] = round(fl[dom] / (rocprof_us[dom] * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
                    roof["rocprof"] = {"tag": rocprof_tag, "us_per_launch": round(rocprof_us[dom], 3),
                                       "file": f"profiles/{rocprof_tag}_configs{args.config}_kernel_stats.csv",
                                       "measured_in_run": False}
                if gemm_mode == "bf16x3":
                    # the same fp32-accurate products, executed as six bf16 piece products each on the bf16 pipe
                    roof["pipe"] = {"gemm_mode": "bf16x3", "executed_TFLOPs": round(6 * ach, 1), "bf16_dense_peak": BF16_MFMA_PEAK_TFLOPS,
                                    "executed_frac_of_bf16_peak": round(6 * ach / BF16_MFMA_PEAK_TFLOPS, 4),
                                    "note": "achieved / peak / frac above count ALGORITHMIC fp32 flops against the fp32-MFMA peak "
                                            "(the rate an exact-fp32 path is priced at); every fp32 operand is three bf16 pieces, six "
                                            "piece products per product, fp32 accumulate (DESIGN.md 4.1)"}
            else:       # DQN configurations: no GEMM; the longest kernel is a latency-bound byte mover
                dom = max((k for k in kern if k in kbytes), key=lambda k: kern[k])
                ach = kbytes[dom] / (kern[dom] * 1e-6) / 1e9
                roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic_all.get(dom),
                        "bytes_per_launch": kbytes[dom]}
            # the genuinely memory-bound launches: achieved GB/s on their algorithmic bytes and, where a counter
            # profile exists, on the HBM-side traffic they actually caused
            alias = {"step_tail_kernel": "iqn_post_kernel"}       # (traffic.json names the fused tail by its kernel)
            mem = {}
            for k in ("step_front_kernel", "step_tail_kernel", "step_back_kernel", "iqn_post_kernel"):
                if kern.get(k) and kbytes.get(k):
                    us = kern[k]
                    e = {"bytes": int(kbytes[k]), "us_event_incl_boundary": round(us, 3),
                         "GBps": round(kbytes[k] / us / 1e3, 1), "frac_of_8TBps": round(kbytes[k] / us / 1e3 / HBM_PEAK_GBS, 4)}
                    tr = traffic_all.get(alias.get(k, k))
                    if tr:
                        e.update({"traffic_bytes": int(tr), "traffic_GBps": round(tr / us / 1e3, 1),
                                  "traffic_frac_of_8TBps": round(tr / us / 1e3 / HBM_PEAK_GBS, 4)})
                    mem[k] = e
            roof.update({"us_per_launch": round(kern[dom], 3), "traffic_source": traffic_src, "memory_bound_kernels": mem,
                         "kernel_us_event_incl_boundary": {k: round(v, 3) for k, v in kern.items()},
                         "kernel_us_note": "HIP-event pairs around each launch of eager steps: each figure includes ~3 us of "
                                           "launch boundary, so they sum to more than ms_per_step (graph replay); rocprofv3 "
                                           "durations are in profiles/*_kernel_stats.csv",
                         "step_hbm": {"algorithmic_bytes": abytes, "achieved_GBps": round(abytes / step_s / 1e9, 2),
                                      "peak_GBps": HBM_PEAK_GBS, "frac": round(abytes / step_s / 1e9 / HBM_PEAK_GBS, 5)},
                         "step_mfma_frac": round(sum(fl.values()) / step_s / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)})
        out = {"metric": "learner grad-steps/sec, IQN+PER batch=256, 1/2/4/8 MI355X" if args.config == 2
               else f"learner grad-steps/sec, BASELINE configs[{args.config}]",
               "value": round(args.steps * world / elapsed, 2), "unit": "steps/s", "n_gpus": world,
               "per_replica_steps_per_s": round(args.steps / elapsed, 2),
               "value_note": "value = grad steps of all replicas per second (per_replica_steps_per_s x n_gpus; every replica "
                             "steps on its own batch of `batch_per_gpu`); SURVEY 8d's per-replica step rate is per_replica_steps_per_s",
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 5),
               "repeats": repeats, "timed_seconds": round(float(sum(blocks)), 3),
               "timing": "median of `repeats` blocks of `steps` steps; min/max block ms_per_step: "
               f"{min(blocks) / args.steps * 1e3:.5f}/{max(blocks) / args.steps * 1e3:.5f}",
               "status": "ok (sticky status words of learner workspace, replay and collective read after the timed blocks)",
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "dtype_note": f"fp32 operands, fp32 accumulation, fp32-accurate results (gemm_mode {gemm_mode}: "
                             + ("each product as six bf16 piece products on the bf16 matrix pipe" if gemm_mode == "bf16x3" else "fp32 MFMA") + ")",
               "data": "synthetic",
               "config": {"workload": f"configs[{args.config}]: " + ["DQN + uniform replay, batch=32",
                                                                     "DQN + PER, batch=256",
                                                                     "IQN (N_tau=8) + PER + 3-step returns, batch=256",
                                                                     "IDS + IQN + LayerNorm + 3-step + target net, batch=512",
                                                                     "full default, 10M-transition replay sharded"][args.config],
                          "batch_per_gpu": cfg.batch_size, "global_batch": cfg.batch_size * world,
                          "replay_capacity_per_gpu": buf.capacity, "obs": "10x10x4 fp32", "n_actions": 6,
                          "parallelism": f"dp{world}", "hip_graph": any(isinstance(g, tuple) for g in getattr(agent, "_graphs", {}).values())},
               "roofline": roof}
        if dp_info:
            out["data_parallel"] = dp_info
        if world == 1 and not args.no_acting:
            # the loop's other half: one Agent.forward per update in the MinAtar presets -- host array in,
            # actions on the host out, replayed from its hipGraph; median over 300 calls each
            act = {}
            rng_a = np.random.default_rng(7)
            for n_obs in (1, 16):
                o = (rng_a.random((n_obs, 10, 10, 4)) < 0.1).astype(np.float32)
                for _ in range(20):
                    agent.forward(o).cpu()
                ts = []
                for _ in range(300):
                    t0 = time.perf_counter()
                    agent.forward(o).cpu()
                    ts.append(time.perf_counter() - t0)
                act[f"n_obs_{n_obs}"] = round(float(np.median(ts)) * 1e6, 1)
            out["acting_latency_us"] = dict(act, what="Agent.forward host -> host (agent.py:31-41), hipGraph replay, median of 300")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
