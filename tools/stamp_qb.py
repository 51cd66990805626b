import os, sys, contextlib, io
os.environ["PRISM_DBG"] = "8"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from prism_amd import config as C
from prism_amd.learner import Learner
from prism_amd.synthetic import fill_replay
cfg = C.derive(C.SUBTRACTIVE_ABLATION_BASE_CONFIG, device="cuda:0", experience_replay_capacity=100_000, log_to_wandb=False)
cfg.hip_graph = False
ln = Learner()
with contextlib.redirect_stdout(io.StringIO()):
    ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
fill_replay(ln.experience_buffer, 20000, seed=0)
for _ in range(5): ln.step(eager=True)
st = torch.zeros(4096 * 64, dtype=torch.int64, device="cuda:0")
ln.agent._desc.dbg_stamps = st.data_ptr()
st.zero_(); ln.step(eager=True); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(4096, 64)
rows = np.arange(640)
t0 = s[rows, 32 + 27].astype(float).min()
f = lambda k: (s[rows, 32 + k].astype(float) - t0) / 100.0
print("qh_bwd<256> real time us: start med/max %.2f/%.2f | loop done %.2f/%.2f | past barrier %.2f/%.2f | end %.2f/%.2f" % (
    np.median(f(27)), f(27).max(), np.median(f(28)), f(28).max(), np.median(f(29)), f(29).max(), np.median(f(30)), f(30).max()))
d = (s[rows, 30] - s[rows, 27]).astype(float)
print("per-workgroup ticks: total med %d max %d; loads+loop med %d; barrier wait med %d; fold+store med %d" % (
    np.median(d), d.max(), np.median(s[rows, 28] - s[rows, 27]), np.median(s[rows, 29] - s[rows, 28]), np.median(s[rows, 30] - s[rows, 29])))
order = np.argsort(f(27))
print("start times (us) of every 64th workgroup by start order:", [round(float(f(27)[i]), 2) for i in order[::64]])
