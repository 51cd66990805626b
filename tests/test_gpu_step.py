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
