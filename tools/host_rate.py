import contextlib, io, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from prism_amd.config import baseline_config
from prism_amd.learner import Learner
from prism_amd.synthetic import fill_replay
cfg = baseline_config(0, device="cuda:0", log_to_wandb=False)     # c1: the shortest GPU step (30 us)
ln = Learner()
with contextlib.redirect_stdout(io.StringIO()):
    ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
ln.time_phases = False
fill_replay(ln.experience_buffer, ln.experience_buffer.capacity, seed=0)
for _ in range(200): ln.step()
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
for _ in range(3000): ln.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6*(t1-t0)/3000:.1f} us/step, total {1e6*(t2-t0)/3000:.1f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): ln.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
