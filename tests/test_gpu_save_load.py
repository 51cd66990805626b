"""GPU: persistence round trips of the two objects the reference checkpoints
(`TimestepBuffer.save/load` timestep_buffer.py:259-318 — its own `complex_save_load_test` :397-475 is the
model: a chain with done / truncated steps, priorities moved by updates, state compared after reload —
and `Agent.save/load` agent.py:179-231)."""
import contextlib
import io
import weakref

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_gpu_learner import build_hip_agent, to_hip_batch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return "cuda:0"


def test_replay_save_load_round_trip(tmp_path):
    dev = _need_gpu()
    from prism_amd.experience import HipReplayBuffer, Timestep
    cap, B = 100, 5                       # the reference test's ListStorage(100), batch_size=5
    rng = np.random.default_rng(0)

    def mk():
        return HipReplayBuffer(cap, B, device=dev, n_step=3, gamma=0.99, use_per=True, alpha=0.5, beta=0.5, seed=3)

    buf = mk()
    steps = [Timestep(id=i, obs=torch.from_numpy((rng.random((10, 10, 4)) < 0.1).astype(np.float32))) for i in range(27)]
    for i in range(26):
        t = steps[i]
        t.reward, t.action = float(i % 3), i % 6
        t.done, t.truncated = (i % 9 == 8), (i % 7 == 6 and i % 9 != 8)
        if not t.done:
            t.next = steps[i + 1] if t.truncated else weakref.ref(steps[i + 1])
        buf.extend(t)
    buf.flush()
    idx = torch.tensor([0, 3, 3, 7, 25], device=dev)
    buf.update_priority(idx, torch.tensor([0.5, 2.0, 0.1, 4.0, 1.5], device=dev))
    buf.save(str(tmp_path))

    new = mk()
    new.load(str(tmp_path))
    torch.cuda.synchronize()
    assert len(new) == len(buf) == 26 and new.buffer._writer._cursor == buf.buffer._writer._cursor
    for name in ("obs", "succ_obs", "reward", "action", "flags", "link", "back", "per_state"):
        a, b = getattr(buf, name), getattr(new, name)
        n = min(a.shape[0], 26) if name != "per_state" else a.shape[0]
        assert torch.equal(a[:n].cpu(), b[:n].cpu()), name
    assert torch.equal(buf.tree.cpu(), new.tree.cpu())          # every node, sum and min
    # the restored buffer samples what the original samples (same device RNG stream position)
    new._draws = buf._draws
    b0, i0 = buf.sample(return_info=True)
    b1, i1 = new.sample(return_info=True)
    torch.cuda.synchronize()
    assert torch.equal(i0["index"].cpu(), i1["index"].cpu())
    assert torch.equal(i0["_weight"].cpu(), i1["_weight"].cpu())
    for k in ("observation", "nonterminal", "gamma", "action"):
        assert torch.equal(b0[k].cpu(), b1[k].cpu()), k
    assert torch.equal(b0["next"]["reward"].cpu(), b1["next"]["reward"].cpu())
    assert torch.equal(b0["next"]["observation"].cpu(), b1["next"]["observation"].cpu())


@pytest.mark.parametrize("name", ["iqn_small", "full_small"])
def test_agent_save_load_round_trip(tmp_path, name):
    dev = _need_gpu()
    g = H.load_case(name)
    _, agent = build_hip_agent(g, dev)
    for step in range(2):
        batch, w, taus = H.case_batch(g, step)
        agent.update(to_hip_batch(batch, dev), per_weights=w.to(dev), taus=[t.to(dev) for t in taus])
    agent.save(str(tmp_path))
    _, other = build_hip_agent(g, dev)
    other.load(str(tmp_path))
    torch.cuda.synchronize()
    assert torch.equal(agent.flat.cpu(), other.flat.cpu())
    assert torch.equal(agent.optimizer.exp_avg.cpu(), other.optimizer.exp_avg.cpu())
    assert torch.equal(agent.optimizer.exp_avg_sq.cpu(), other.optimizer.exp_avg_sq.cpu())
    assert int(agent.optimizer.step_t.item()) == int(other.optimizer.step_t.item()) == 2
    # the checkpoint has the reference's layout and key names
    sd = torch.load(str(tmp_path / "agent" / "model.pt"), map_location="cpu")
    assert list(sd.keys()) == [str(k) for k in g["param_names"]]
    assert (tmp_path / "agent" / "optimizer.pt").exists() and (tmp_path / "agent" / "state.pkl").exists()
    # and training continues identically
    batch, w, taus = H.case_batch(g, 2 % int(g["steps"]))
    td0 = agent.update(to_hip_batch(batch, dev), per_weights=w.to(dev), taus=[t.to(dev) for t in taus])
    td1 = other.update(to_hip_batch(batch, dev), per_weights=w.to(dev), taus=[t.to(dev) for t in taus])
    torch.cuda.synchronize()
    assert torch.equal(td0.cpu(), td1.cpu())
    assert torch.equal(agent.flat.cpu(), other.flat.cpu())
