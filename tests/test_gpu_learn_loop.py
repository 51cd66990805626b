"""GPU: the whole learner loop end to end -- ``Learner.configure`` / ``learn()`` as the reference's experiment files call
them (`/root/reference/prism/learner.py:60-93,127-160`), fed by a stub of the collector surface (the reference's collectors
talk to environment processes over Redis: out of scope here).  The stub acts with ``Agent.forward`` (the HIP acting path),
builds linked ``Timestep`` objects the way the reference's collectors do and hands them to ``buffer.extend``; the loop then
samples, updates, writes priorities back, synchronises the target network on its timestep period and checkpoints.
Checked: the loop runs on the HIP path (four-launch fused step + graph), parameters move and stay finite, the target
network follows the online one, the replay holds what was collected, checkpoints load back into a fresh learner."""
import contextlib
import io
import os
import weakref

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class StubCollector:
    def __init__(self, n_env=4, C=4, n_actions=6, seed=0, p_done=0.05):
        self.n_env, self.C, self.A, self.p_done = n_env, C, n_actions, p_done
        self.rng = np.random.default_rng(seed)
        self.ids = iter(range(10 ** 9))
        self.cur = None
        self.started = self.closed = False
        self.n_forward = 0
        self.actions = []

    def _obs(self):
        return (self.rng.random((10, 10, self.C)) < 0.1).astype(np.float32)

    def get_env_info(self):
        return (10, 10, self.C), self.A, 1

    def signal_processes_start_collecting(self, agent):
        self.started = True

    def collect_timesteps(self, n_timesteps, agent, exp_buffer, random=False):
        from prism_amd.experience import Timestep
        if self.cur is None:
            self.cur = [Timestep(id=next(self.ids), obs=self._obs()) for _ in range(self.n_env)]
        done = 0
        while done < n_timesteps:
            if random:
                acts = self.rng.integers(0, self.A, self.n_env)
            else:
                acts = agent.forward(np.stack([t.obs for t in self.cur])).cpu().numpy()
                self.n_forward += 1
                assert acts.shape == (self.n_env,) and acts.min() >= 0 and acts.max() < self.A
                self.actions.extend(int(a) for a in acts)
            for e in range(self.n_env):
                t = self.cur[e]
                nxt = Timestep(id=next(self.ids), obs=self._obs())
                t.action, t.reward = int(acts[e]), float(np.float32(self.rng.standard_normal()))
                t.done, t.truncated = bool(self.rng.random() < self.p_done), False
                if not t.done:
                    t.next = weakref.ref(nxt)
                    nxt.prev = weakref.ref(t)
                exp_buffer.extend(t)          # one completed timestep at a time (timestep_buffer.py:32-33)
                self.cur[e] = nxt
            done += self.n_env
        return done

    def log(self, logger):
        logger.log_data(data=0.0, group_name="Report/Rewards", var_name="Training Reward")

    def close(self):
        self.closed = True


def _config(dev, tmp, **over):
    from prism_amd.config import baseline_config
    kw = dict(device=dev, batch_size=32, experience_replay_capacity=2048, num_initial_random_timesteps=256,
              timesteps_per_iteration=4, timestep_limit=256 + 4 * 150, timesteps_per_report=400,
              timesteps_between_evaluations=300, target_update_period=36, checkpoint_dir=str(tmp), log_to_wandb=False)
    kw.update(over)
    extra = {k: kw.pop(k) for k in ("fused_step", "hip_graph") if k in kw}      # MI355X-only knobs: plain attributes
    cfg = baseline_config(kw.pop("base", 2), **kw)
    for k, v in extra.items():
        setattr(cfg, k, v)
    return cfg


@pytest.mark.parametrize("over", [dict(), dict(use_target_network=True), dict(base=3),
                                  dict(base=0), dict(base=1, use_target_network=True), dict(fused_step=False),
                                  dict(base=3, hip_graph=False)])
def test_learn_runs_end_to_end_and_checkpoints_reload(tmp_path, over):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from prism_amd.learner import Learner
    dev = "cuda:0"
    cfg = _config(dev, tmp_path, **over)
    col = StubCollector(seed=5)
    ln = Learner()
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        ln.configure(cfg, collector=col)
        agent, buf = ln.agent, ln.experience_buffer
        p0 = agent.flat.clone()
        real_empty = buf.empty
        buf.empty = lambda: None              # learn() empties the buffer on exit: keep it for the checks below
        ln.learn()
        buf.empty = real_empty
    torch.cuda.synchronize()
    n_iter = 150
    assert col.started and col.closed and col.n_forward == n_iter          # every iteration acted through Agent.forward
    assert ln.cumulative_timesteps == 256 + 4 * n_iter and ln.cumulative_model_updates == n_iter
    assert int(agent.optimizer.step_t.item()) == n_iter
    assert len(set(col.actions)) > 1                                       # (not a constant policy by accident)
    assert torch.isfinite(agent.flat).all() and not torch.equal(agent.flat, p0)
    assert buf._size == ln.cumulative_timesteps                            # nothing lost between extend() and the ring
    if cfg.use_per:
        tree = buf.sum_tree.cpu().numpy()
        assert np.isfinite(tree).all() and tree[1] > 0
        leaves = tree[buf.tree_capacity:buf.tree_capacity + buf._size]
        assert (leaves > 0).all() and len(np.unique(leaves)) > 10          # sampled slots carry |td|-priorities
    if cfg.use_target_network:
        # the last synchronisation was at most target_update_period timesteps ago: close to, not equal to, the online net
        d = float((agent.flat - agent.flat_target).abs().max())
        assert 0 < d < float((agent.flat - p0).abs().max())
    assert "iteration" in out.getvalue()                                   # the report ran
    # a checkpoint written by the loop loads into a fresh learner and gives the same actions
    root = os.path.join(str(tmp_path), cfg.env_name)
    cks = sorted(d for d in os.listdir(root) if d.startswith("agent_checkpoint_"))
    assert cks, os.listdir(root)
    with contextlib.redirect_stdout(io.StringIO()):
        agent.save(os.path.join(root, "final"))
        ln2 = Learner()
        ln2.configure(_config(dev, tmp_path / "b", **over), collector=StubCollector(seed=6))
        ln2.agent.load(os.path.join(root, "final"))
    assert torch.equal(ln2.agent.flat, agent.flat)
    obs = np.stack([col._obs() for _ in range(7)])
    agent.eval(), ln2.agent.eval()
    assert torch.equal(agent.forward(obs), ln2.agent.forward(obs))
