"""Soak run (GPU box): many fused hipGraph steps with periodic consistency checks -- the priority tree's
parent = op(children) invariant on the device, finite parameters, self-resetting flags back at zero.
Usage: python tools/soak.py [config index | add | add_per | add_ids | sub] [steps] [fill fraction: < 1 keeps the step on the
eager (graph-free) path] [twin]
`twin`: a second learner with the same configuration and seeds steps beside the first; their flat parameters must stay
BIT-IDENTICAL at every check -- every kernel is deterministic by design (fixed summation orders, no float atomics), so a race
anywhere (an LDS hand-off, a grid barrier, a ticket) shows up as a divergence long before it shows up as a wrong number."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prism_amd.config import baseline_config
from prism_amd.learner import Learner
from prism_amd.synthetic import fill_replay

_arg = sys.argv[1] if len(sys.argv) > 1 else "2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
twin = len(sys.argv) > 4 and sys.argv[4] == "twin"


def make():
    if _arg in ("add", "add_per", "add_ids", "sub"):          # the reference's ablation presets (tools/bench_presets.py)
        from prism_amd import config as C
        base, over = {"add": (C.ADDITIVE_ABLATION_BASE_CONFIG, {}),
                      "add_per": (C.ADDITIVE_ABLATION_BASE_CONFIG, dict(use_per=True, n_step_returns_length=3, use_layer_norm=True)),
                      "add_ids": (C.ADDITIVE_ABLATION_BASE_CONFIG, dict(use_ids=True, ids_n_q_head_model_layers=2, ids_n_q_heads=10,
                                                                        ids_q_head_feature_dim=256, ids_ensemble_variation_coef=0)),
                      "sub": (C.SUBTRACTIVE_ABLATION_BASE_CONFIG, {})}[_arg]
        cfg = C.derive(base, device="cuda:0", experience_replay_capacity=100_000, log_to_wandb=False, **over)
    else:
        cfg = baseline_config(int(_arg), device="cuda:0")
        if int(_arg) == 4:
            cfg.experience_replay_capacity = 1_250_000
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
    ln.time_phases = False
    fill_replay(ln.experience_buffer, max(int(ln.experience_buffer.capacity * frac), cfg.batch_size), seed=0)
    return ln, cfg


ln, cfg = make()
buf, ag = ln.experience_buffer, ln.agent
ln2 = make()[0] if twin else None


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
    if ln2 is not None:
        a2 = ln2.agent
        if os.environ.get("SOAK_DEEP") == "1":
            # (diagnosis) everything the step leaves behind: Adam moments, the whole workspace word for word
            for nm, x, y in (("exp_avg", ag.optimizer.exp_avg, a2.optimizer.exp_avg), ("exp_avg_sq", ag.optimizer.exp_avg_sq, a2.optimizer.exp_avg_sq),
                             ("grads", ag.grads, a2.grads), ("workspace", ag.workspace, a2.workspace)):
                xi, yi = x.view(torch.int32), y.view(torch.int32)
                if not torch.equal(xi, yi):
                    w = torch.nonzero(xi != yi).flatten()
                    print(f"{tag}: {nm} differs in {w.numel()} words; first at float offset {int(w[0])}, last {int(w[-1])}: "
                          f"{float(x.flatten()[w[0]])!r} vs {float(y.flatten()[w[0]])!r}", flush=True)
                    if nm == "workspace":
                        import numpy as np
                        wn = w.cpu().numpy()
                        brk = np.nonzero(np.diff(wn) > 4096)[0]
                        starts = [int(wn[0])] + [int(wn[i + 1]) for i in brk]
                        ends = [int(wn[i]) for i in brk] + [int(wn[-1])]
                        print("      differing workspace ranges (float offsets):", list(zip(starts, ends))[:24], "of", x.numel(), flush=True)
                        lo_ = max(0, int(wn[0]) - 6)
                        print("      around the first:", [round(float(v), 6) for v in x[lo_:lo_ + 16]], "|", [round(float(v), 6) for v in y[lo_:lo_ + 16]], flush=True)
                        if os.environ.get("SOAK_DEEP_CONTINUE") == "1":
                            return
                        raise AssertionError(f"{tag}: workspace differs")
        if not torch.equal(ag.grads, ln2.agent.grads):
            print(f"{tag}: gradients differ in {int((ag.grads != ln2.agent.grads).sum())} elements; scalars",
                  ag.scalars.tolist(), ln2.agent.scalars.tolist())
        if not torch.equal(ag.flat, ln2.agent.flat):
            off, bad = 0, []
            for k, v in ag.model.state_dict().items():
                d = int((ag.flat[off:off + v.numel()] != ln2.agent.flat[off:off + v.numel()]).sum())
                if d:
                    bad.append(f"{k}: {d} of {v.numel()}")
                off += v.numel()
            raise AssertionError(f"{tag}: the twin learner's parameters differ -- " + "; ".join(bad[:12]))
        if buf.use_per:
            assert torch.equal(buf.tree, ln2.experience_buffer.tree), f"{tag}: the twin's priority tree differs"


t0 = time.time()
done = 0
chunk = int(os.environ.get("SOAK_CHUNK", "20000"))
fine_from = int(os.environ.get("SOAK_FINE_FROM", "0"))       # (diagnosis: check after EVERY step from this step on)
while done < steps:
    n = min(chunk, steps - done)
    if fine_from and done >= fine_from:
        n = 1
    elif fine_from and done + n > fine_from:
        n = fine_from - done
    for _ in range(n):
        ln.step()
        if ln2 is not None:
            ln2.step()
    done += n
    check(f"after {done} steps")
    if n > 1 or done % 1000 == 0:
        print(f"{done:8d} steps ok  ({done / (time.time() - t0):8.0f} steps/s incl. checks)  loss {float(ag._static_total_loss):.5f}", flush=True)
print("soak OK")
