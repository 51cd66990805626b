import sys, contextlib, io
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests import helpers as H
from tests.test_gpu_variants import CASES, _config, _batch
from tests.test_gpu_learner import to_hip_batch
from oracle.learner_ref import LearnerOracle
from prism_amd.factory import agent_factory
name = sys.argv[1] if len(sys.argv) > 1 else "tau16"
dev, A, seed = "cuda:0", 6, 11
case = CASES[name]; B, C = case["B"], case["C"]
cfg = _config(dev, case["over"])
torch.manual_seed(seed)
with contextlib.redirect_stdout(io.StringIO()):
    agent = agent_factory.build_agent(cfg, (10, 10, C), A)
cpu_cfg = _config("cpu", case["over"])
sd, tgt = H.build_init_state(cpu_cfg, seed, C=C, A=A)
orc = LearnerOracle(sd, H.spec_from_config(cpu_cfg, C=C, A=A), tgt)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
for step in range(3):
    batch, w, taus = _batch(rng, B, C, A, cfg)
    td_o = orc.update(batch, w, taus)
    td = agent.update(to_hip_batch(batch, dev), per_weights=w.to(dev), taus=[t.to(dev) for t in taus])
    torch.cuda.synchronize()
    off, gflat = 0, agent.grads.cpu()
    for k in sd:
        n = sd[k].numel(); go = orc.last["grads"][k].reshape(-1); gh = gflat[off:off+n]
        err = float((gh-go).abs().max())
        if err > 1e-4*float(go.abs().max())+1e-7:
            i = int((gh-go).abs().argmax())
            print(step, k, "err", err, "max", float(go.abs().max()), "at", i, float(gh[i]), float(go[i]), "n_bad", int(((gh-go).abs() > 1e-5).sum()))
        off += n
    print("step", step, "td err", float((td.cpu()-td_o).abs().max()))
