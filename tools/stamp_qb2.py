"""Diagnostic: phase stamps of qh_bwd2_kernel (GPU box; PRISM_DBG=32).  Stamps land in the shared buffer; the post kernel
overwrites some slots afterwards, so the kernel's own slots 0..3 / 32..35 are read (post uses 7.. / 13..)."""
import os, sys, contextlib, io
os.environ["PRISM_DBG"] = "32"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prism_amd.config import baseline_config
from prism_amd.learner import Learner
from prism_amd.synthetic import fill_replay
cfg = baseline_config(3, device="cuda:0")
cfg.hip_graph = False
ln = Learner()
with contextlib.redirect_stdout(io.StringIO()):
    ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
fill_replay(ln.experience_buffer, 20000, seed=0)
ag = ln.agent
for _ in range(5): ln.step(eager=True)
st = torch.zeros(4096 * 64, dtype=torch.int64, device="cuda:0")
ag._desc.dbg_stamps = st.data_ptr()
st.zero_(); ln.step(eager=True); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(4096, 64)
ng = 160
for name, rows in (("G role", s[:ng]), ("S role", s[ng:ng + 256])):
    f = rows[:, :4].astype(np.float64)
    rt = rows[:, 32:36].astype(np.float64)
    print(name, "ticks: first stage %.0f  loop %.0f  epilogue %.0f   | real-time span first start -> last end %.2f us, median wg %.2f us" % (
        np.median(f[:, 1] - f[:, 0]), np.median(f[:, 2] - f[:, 1]), np.median(f[:, 3] - f[:, 2]),
        (rt[:, 3].max() - rt[:, 0].min()) / 100.0, np.median(rt[:, 3] - rt[:, 0]) / 100.0))
