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
    for name in ("obs", "succ_obs", "reward", "action", "link", "back", "per_state"):
        a, b = getattr(buf, name), getattr(new, name)
        n = min(a.shape[0], 26) if name != "per_state" else a.shape[0]
        assert torch.equal(a[:n].cpu(), b[:n].cpu()), name
    # flags: the last stored step's successor had not been stored -> written truncated, exactly as the reference's
    # save does (timestep_buffer.py:276-291: the collectors' state cannot be recovered); every other slot unchanged
    f0, f1 = buf.flags[:26].cpu().numpy(), new.flags[:26].cpu().numpy()
    np.testing.assert_array_equal(f0[:25], f1[:25])
    assert f1[25] == f0[25] | 2 and (f0[25] & 4) and int(buf.link[25]) == -1
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


def test_agent_loads_a_checkpoint_written_by_the_reference(tmp_path):
    """tests/golden/ref_checkpoint was written by the reference's Agent.save (tools/gen_golden.py checkpoint)."""
    import os
    import shutil
    dev = _need_gpu()
    from prism_amd.agents import action_selectors as S
    from prism_amd.config import MINATAR_CONFIG, derive
    from prism_amd.factory import agent_factory
    exp = np.load(os.path.join(H.GOLDEN, "ref_checkpoint_expected.npz"))
    cfg = derive(MINATAR_CONFIG, device=dev, use_cuda_graph=False, use_ids=False, use_iqn=False, use_dqn=True,
                 use_layer_norm=False, use_e_greedy=True, use_target_network=True, seed=999)     # other initial weights
    with contextlib.redirect_stdout(io.StringIO()):
        agent = agent_factory.build_agent(cfg, (10, 10, 4), 6)
    agent.load(os.path.join(H.GOLDEN, "ref_checkpoint"))
    sd = agent.model.state_dict()
    assert list(sd.keys()) == [str(k) for k in exp["param_names"]]
    np.testing.assert_array_equal(np.array([float(v.cpu().double().sum()) for v in sd.values()]), exp["sum"])
    np.testing.assert_array_equal(np.array([float(v.cpu().double().norm()) for v in sd.values()]), exp["l2"])
    assert agent.n_updates == int(exp["n_updates"]) and agent.max_grad_norm == float(exp["max_grad_norm"])
    assert int(agent.optimizer.step_t.item()) == int(exp["adam_step"])
    off = 0
    for i, n in enumerate(agent.optimizer.numels):
        assert abs(float(agent.optimizer.exp_avg[off:off + n].cpu().double().norm()) - float(exp["exp_avg_l2"][i])) < 1e-12
        off += n
    assert isinstance(agent.action_selector, S.EGreedyActionSelector)
    assert agent.action_selector.epsilon.get_state() == int(exp["epsilon_step"])
    tgt = agent.target_model.state_dict()
    assert all(torch.isfinite(v).all() for v in tgt.values())
    # written back out, the directory has the reference's layout and its state.pkl names the reference's classes
    agent.save(str(tmp_path))
    names = sorted(os.listdir(tmp_path / "agent"))
    assert names == ["model.pt", "optimizer.pt", "state.pkl", "target_model.pt"]
    raw = open(tmp_path / "agent" / "state.pkl", "rb").read()
    assert b"prism_amd" not in raw and b"cprism.agents.action_selectors\nEGreedyActionSelector\n" in raw
    ref_sd = torch.load(os.path.join(H.GOLDEN, "ref_checkpoint", "agent", "model.pt"), map_location="cpu")
    new_sd = torch.load(str(tmp_path / "agent" / "model.pt"), map_location="cpu")
    for k in ref_sd:
        assert torch.equal(ref_sd[k], new_sd[k]), k
    shutil.rmtree(tmp_path / "agent")


def test_replay_loads_a_buffer_file_written_by_the_reference():
    """tests/golden/ref_buffer/experience_buffer/timesteps.pkl was written by the reference's TimestepBuffer.save; the
    expected batch is what the reference collates from its OWN load of that file (tools/gen_golden.py nstep)."""
    import os
    dev = _need_gpu()
    from prism_amd.experience import HipReplayBuffer
    exp = np.load(os.path.join(H.GOLDEN, "ref_buffer_expected.npz"))
    n = int(exp["n"])
    buf = HipReplayBuffer(100, n, device=dev, n_step=3, gamma=0.99, use_per=True, alpha=0.5, beta=0.5, seed=1)
    buf.load(os.path.join(H.GOLDEN, "ref_buffer"))
    assert len(buf) == n
    np.testing.assert_array_equal(buf._slot_id[:n], exp["ids"])
    batch = buf.gather(torch.arange(n, device=dev))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(batch["observation"].cpu().numpy().reshape(exp["obs"].shape), exp["obs"])
    import pickle
    from prism_amd.experience import ref_format
    from tests.test_ref_formats import check_next_obs
    flat = pickle.load(open(os.path.join(H.GOLDEN, "ref_buffer", "experience_buffer", "timesteps.pkl"), "rb"))
    check_next_obs(batch["next"]["observation"].cpu().numpy().reshape(n, -1), ref_format.ring_from_timesteps(flat),
                   ref_format.parse_timesteps(flat), exp)
    np.testing.assert_array_equal(batch["next"]["reward"].cpu().numpy(), exp["reward"])
    np.testing.assert_array_equal(batch["nonterminal"].cpu().numpy(), exp["nonterminal"])
    np.testing.assert_array_equal(batch["gamma"].cpu().numpy(), exp["gamma"])
    np.testing.assert_array_equal(batch["action"].cpu().numpy(), exp["action"])
    # every loaded slot starts at the sampler's default priority and can be sampled
    b, info = buf.sample(return_info=True)
    torch.cuda.synchronize()
    assert int(info["index"].max()) < n and torch.all(info["_weight"] == 1.0)
