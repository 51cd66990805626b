"""Diagnostic (GPU box): per-tensor gradient error of a golden case against the oracle, every step.
usage: python tools/dbg_case.py <case> [gemm_mode]      (PRISM_NO_BWD4 / PRISM_NO_BWD3 / PRISM_GEMM select kernels)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers as H
from tests.test_gpu_learner import to_hip_batch, build_hip_agent
from oracle.learner_ref import LearnerOracle
name = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
dev = "cuda:0"
g = H.load_case(name)
cfg, agent = build_hip_agent(g, dev, gemm_mode=mode)
cpu_cfg = H.case_config(g)
sd, tgt = H.build_init_state(cpu_cfg, int(g["seed"]), C=int(g["C"]), A=int(g["A"]))
orc = LearnerOracle(sd, H.spec_from_config(cpu_cfg, C=int(g["C"]), A=int(g["A"])), tgt)
for step in range(int(g["steps"])):
    batch, w, taus = H.case_batch(g, step)
    g64 = orc.grads_fp64(batch, w, taus)
    # kink probe: the oracle's own gradient at parameters jittered by ~two ulps (tests/test_gpu_fullsize_parity.py)
    sd_now = {k: v.clone() for k, v in orc.state_dict().items()}
    tg_now = None if orc.p_tgt is None else {k: v.clone() for k, v in orc.p_tgt.items()}
    gen = torch.Generator().manual_seed(1234 + step)
    jit_g = []
    for _ in range(6):
        jit = {k: v * (1.0 + 2.4e-7 * torch.randn(v.shape, generator=gen)) for k, v in sd_now.items()}
        pj = LearnerOracle(jit, H.spec_from_config(cpu_cfg, C=int(g["C"]), A=int(g["A"])), tg_now)
        pj.update(batch, w, taus, apply=False)
        jit_g.append({k: v.clone() for k, v in pj.last["grads"].items()})
    td_o = orc.update(batch, w, taus)
    td = agent.update(to_hip_batch(batch, dev), per_weights=w.to(dev), taus=[t.to(dev) for t in taus])
    torch.cuda.synchronize()
    off, gflat = 0, agent.grads.cpu()
    print("step", step, "td err", float((td.cpu() - td_o).abs().max()))
    for k in sd:
        n = sd[k].numel(); go = orc.last["grads"][k].reshape(-1); gh = gflat[off:off + n]
        d = (gh.double() - g64[k].reshape(-1)).abs()
        i = int(d.argmax())
        jk = max(float((jg[k].reshape(-1) - go).abs().max()) for jg in jit_g)
        print(f"   {k:44s} max|g| {float(go.abs().max()):.3e} err64 {float(d.max()):.3e} rel {float(d.max()) / (float(go.abs().max()) + 1e-30):.2e}"
              f" n>1e-5: {int((d > 1e-5).sum())} of {n} at {i}: {float(gh[i]):.6e} vs {float(g64[k].reshape(-1)[i]):.6e}  jitter-kink {jk:.2e}")
        off += n
    if cfg.use_target_network and step == 0:
        agent.sync_target_model(); orc.sync_target()
