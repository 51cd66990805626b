"""Soak run (GPU box): many fused hipGraph steps with periodic consistency checks -- the priority tree's
parent = op(children) invariant on the device, finite parameters, self-resetting flags back at zero.
Usage: python tools/soak.py [config index] [steps] [fill fraction: < 1 keeps the step on the eager (graph-free) path]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prism_amd.config import baseline_config
from prism_amd.learner import Learner
from prism_amd.synthetic import fill_replay

ci = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
cfg = baseline_config(ci, device="cuda:0")
if ci == 4:
    cfg.experience_replay_capacity = 1_250_000
ln = Learner()
with contextlib.redirect_stdout(io.StringIO()):
    ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
ln.time_phases = False
buf, ag = ln.experience_buffer, ln.agent
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
fill_replay(buf, max(int(buf.capacity * frac), cfg.batch_size), seed=0)


def check(tag):
    torch.cuda.synchronize()
    if buf.use_per:
        t, cap2 = buf.tree, buf.tree_capacity
        kids = t[2:2 * cap2].view(cap2 - 1, 2, 2)
        assert torch.equal(t[1:cap2, 0], kids[:, 0, 0] + kids[:, 1, 0]), f"{tag}: sum tree broken"
        assert torch.equal(t[1:cap2, 1], torch.minimum(kids[:, 0, 1], kids[:, 1, 1])), f"{tag}: min tree broken"
        assert float(t[1, 0]) > 0 and float(t[1, 1]) > 0
    assert bool(torch.isfinite(ag.flat).all()), f"{tag}: non-finite parameters"
    assert int(buf.status.item()) == 0, f"{tag}: sticky status {int(buf.status.item())}"
    ag.check_status()          # the fused tail's grid barrier never timed out (workspace status word)


t0 = time.time()
done = 0
chunk = 20_000
while done < steps:
    n = min(chunk, steps - done)
    for _ in range(n):
        ln.step()
    done += n
    check(f"after {done} steps")
    print(f"{done:8d} steps ok  ({done / (time.time() - t0):8.0f} steps/s incl. checks)  loss {float(ag._static_total_loss):.5f}", flush=True)
print("soak OK")
