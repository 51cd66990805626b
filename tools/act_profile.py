import contextlib, io, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from prism_amd.config import baseline_config
from prism_amd.learner import Learner
cfg = baseline_config(int(sys.argv[1]), device="cuda:0", log_to_wandb=False)
ln = Learner()
with contextlib.redirect_stdout(io.StringIO()):
    ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
ag = ln.agent
obs = (np.random.default_rng(0).random((1, 10, 10, 4)) < 0.1).astype(np.float32)
for _ in range(2000):
    ag.forward(obs).cpu()
