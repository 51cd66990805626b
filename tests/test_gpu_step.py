"""GPU: the fused four-/five-launch step and its hipGraph replay produce exactly what the unfused
sample() -> update() -> update_priority() sequence produces (same Philox streams)."""
import contextlib
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(dev, fused, graph, B=32, cap=4096, base=2, fuse_tail=True, **over):
    from prism_amd.config import baseline_config
    from prism_amd.learner import Learner
    from prism_amd.synthetic import fill_replay
    cfg = baseline_config(base, device=dev, batch_size=B, experience_replay_capacity=cap, **over)
    cfg.fused_step, cfg.hip_graph, cfg.fuse_tail = fused, graph, fuse_tail
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
    fill_replay(ln.experience_buffer, cap, seed=3)
    return ln


@pytest.mark.parametrize("over", [dict(), dict(use_target_network=True, target_update_period=2),
                                  dict(base=0), dict(base=1), dict(base=3, target_update_period=3),
                                  dict(base=1, use_layer_norm=True, use_double_q_learning=True, use_target_network=True)])
def test_fused_and_graph_equal_unfused(over):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = "cuda:0"
    ref, fus, gra = _mk(dev, False, False, **over), _mk(dev, True, False, **over), _mk(dev, True, True, **over)
    # (fus / gra: four launches, the gradient reduction and clip + Adam behind a grid barrier in one; spl: the five-launch
    # form a data-parallel step uses, without the all-reduce)
    spl = _mk(dev, True, True, fuse_tail=False, **over)
    for step in range(6):
        outs = []
        for ln in (ref, fus, gra, spl):
            td = ln.step(timesteps_this_iteration=1).clone()
            torch.cuda.synchronize()
            buf, ag = ln.experience_buffer, ln.agent
            tree = buf.sum_tree.cpu().numpy() if buf.use_per else np.zeros(1)
            outs.append((td.cpu().numpy(), buf._index.cpu().numpy(), buf._weight.cpu().numpy(),
                         ag.flat.cpu().numpy(), tree, float(ag.scalars[0]),
                         buf._obs.cpu().numpy(), buf._reward.cpu().numpy()))
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                np.testing.assert_array_equal(x, y)
    assert int(gra.agent.optimizer.step_t.item()) == 6
    assert any(isinstance(g, tuple) for g in gra.agent._graphs.values())      # a graph really was captured
    assert int(gra.agent.rng_counters[0].item()) == 6 * 32


def _scramble_links(buf, seed):
    """Chains the stride predictor of the fused front launch cannot guess: links to random later slots, truncations,
    open chains (has-next with no stored link), episode ends."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    n = buf._size
    kind = torch.rand(n, generator=g)
    link = torch.arange(n) + 8
    jump = torch.randint(1, 40, (n,), generator=g)
    link = torch.where(kind < 0.35, torch.arange(n) + jump, link)            # irregular stride
    done = (kind >= 0.35) & (kind < 0.45)
    trunc = (kind >= 0.45) & (kind < 0.55)
    open_ = (kind >= 0.55) & (kind < 0.62)
    link = torch.where((link < n) & ~done & ~open_, link, torch.full_like(link, -1))
    flags = done.to(torch.uint8) * 1 + trunc.to(torch.uint8) * 2 + (~done).to(torch.uint8) * 4
    buf.link[:n] = link.to(torch.int32).to(buf.link.device)
    buf.flags[:n] = flags.to(buf.flags.device)
    buf.back.fill_(-1)


def test_front_walk_on_irregular_chains():
    """The fused front launch predicts the n-step chain from its first link; whatever the chain really does, the batch
    must be the unfused gather's, bit for bit (n_step 1, 3 and 5)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = "cuda:0"              # (flag bits: include/prism_hip.h PRISM_FLAG_DONE 1, _TRUNC 2, _HAS_NEXT 4)
    for n_step in (1, 3, 5):
        ref, fus = _mk(dev, False, False, n_step_returns_length=n_step), _mk(dev, True, False, n_step_returns_length=n_step)
        for ln in (ref, fus):
            _scramble_links(ln.experience_buffer, 11)
        stops = 0
        for step in range(4):
            outs = []
            for ln in (ref, fus):
                ln.step(timesteps_this_iteration=1)
                torch.cuda.synchronize()
                buf = ln.experience_buffer
                outs.append([t.cpu().numpy().copy() for t in (buf._index, buf._obs, buf._next_obs, buf._reward, buf._gamma,
                                                             buf._nonterminal, buf._action, buf._weight)])
            for x, y in zip(*outs):
                np.testing.assert_array_equal(x, y)
            stops += int((outs[0][5] == 0).sum())
        assert stops > 0            # some walks did end at an episode end
