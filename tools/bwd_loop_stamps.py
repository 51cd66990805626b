"""Diagnostic (GPU box, PRISM_DBG=24): issue-time stamps inside the first four tiles of the backward kernel's loop,
workgroups 256.. only (their stamp rows are not shared with the forward kernel)."""
import os, sys, contextlib, io
os.environ["PRISM_DBG"] = "24"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prism_amd.config import baseline_config
from prism_amd.learner import Learner
from prism_amd.synthetic import fill_replay
cfg = baseline_config(2, device="cuda:0")
cfg.hip_graph = False
ln = Learner()
with contextlib.redirect_stdout(io.StringIO()):
    ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
fill_replay(ln.experience_buffer, 20000, seed=0)
for _ in range(5): ln.step(eager=True)
st = torch.zeros(4096 * 64, dtype=torch.int64, device="cuda:0")
ln.agent._desc.dbg_stamps = st.data_ptr()
st.zero_(); ln.step(eager=True); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(4096, 64)[256:512].astype(np.float64)
for ti in range(4):
    b = 16 + 4 * ti
    ph1, ph2, tail = s[:, b + 1] - s[:, b], s[:, b + 2] - s[:, b + 1], s[:, b + 3] - s[:, b + 2]
    nxt = (s[:, b + 4] - s[:, b + 3]) if ti < 3 else np.zeros(len(s))
    print(f"tile {ti}: phi + dX products {np.median(ph1):6.0f} | elementwise + dW products {np.median(ph2):6.0f} | next loads, d e, shuffles {np.median(tail):6.0f}"
          f" | to next tile {np.median(nxt):6.0f}   (issue ticks, wave 0)")
print("loop (stamp 9 -> 10):", np.median(s[:, 10] - s[:, 9]))
