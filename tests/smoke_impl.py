"""One tiny pass of the hot path on the GPU, checked against the CPU oracle (used by
__graft_entry__.smoke() and tests/test_gpu_smoke.py)."""
import contextlib
import io

import numpy as np
import torch


def run_smoke(dev="cuda:0", B=16, capacity=2048, verbose=True):
    from oracle.learner_ref import LearnerOracle
    from prism_amd.config import baseline_config
    from prism_amd.learner import Learner
    from prism_amd.synthetic import fill_replay
    from tests import helpers as H

    cfg = baseline_config(2, device=dev, batch_size=B, experience_replay_capacity=capacity)
    learner = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        learner.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
    buf, agent = learner.experience_buffer, learner.agent
    fill_replay(buf, capacity, seed=1)
    sd0 = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
    td = learner.step(eager=True)
    torch.cuda.synchronize()

    # replay the same minibatch / quantile samples through the oracle
    b = buf.get_static_batch()
    batch = dict(obs=b["observation"].squeeze(1).cpu(), next_obs=b["next"]["observation"].squeeze(1).cpu(),
                 reward=b["next"]["reward"].flatten().cpu(), nonterminal=b["nonterminal"].flatten().cpu(),
                 gamma=b["gamma"].flatten().cpu(), action=b["action"].flatten().cpu())
    T = cfg.iqn_n_current_state_quantile_samples
    taus = [agent.tau_out[0, :T * B].cpu().reshape(-1, 1), agent.tau_out[1, :T * B].cpu().reshape(-1, 1)]
    orc = LearnerOracle(sd0, H.spec_from_config(H.case_config({"overrides_keys": [], "overrides_vals": []},
                                                                **{k: getattr(cfg, k) for k in (
                                                                    "use_ids", "use_iqn", "use_dqn", "use_layer_norm",
                                                                    "use_target_network", "batch_size")})), None)
    td_o = orc.update(batch, buf._weight.cpu(), taus)
    err = float((td.cpu() - td_o).abs().max())
    assert err < 1e-5, f"smoke: td error {err}"
    post = agent.model.state_dict()
    perr = max(float((post[k].cpu() - v).abs().max()) for k, v in orc.state_dict().items())
    assert perr < 2e-6, f"smoke: parameter error after Adam {perr}"
    # priority writeback: leaf = sqrt(|td| + 1e-8) for the (last occurrence of each) sampled slot
    idx = buf._index.cpu().numpy()
    leaves = buf.sum_tree[buf.tree_capacity:buf.tree_capacity + capacity].cpu().numpy()
    want = np.sqrt(np.abs(td.cpu().numpy()) + np.float32(1e-8)).astype(np.float32)
    last = {int(i): w for i, w in zip(idx, want)}
    for i, w in last.items():
        assert leaves[i] == w, (i, leaves[i], w)
    # acting forward on the updated weights (Agent.forward's estimates) vs the oracle on the same quantile samples
    from oracle.learner_ref import act_forward
    rng = np.random.default_rng(3)
    obs = torch.from_numpy((rng.random((3, 10, 10, 4)) < 0.1).astype(np.float32))
    Ta = cfg.iqn_quantile_samples_per_action
    ataus = torch.from_numpy(rng.random((Ta * 3, 1)).astype(np.float32))
    q, dist = agent.act_estimates(obs.to(dev), taus=ataus.to(dev))
    sd1 = {k: v.detach().cpu() for k, v in agent.model.state_dict().items()}
    qo, do = act_forward(sd1, orc.spec, obs, ataus)
    aerr = max(float((q.cpu() - qo).abs().max()), float((dist.cpu() - do).abs().max()))
    assert aerr < 1e-5, f"smoke: acting estimates differ by {aerr}"
    act = agent.forward(obs.to(dev))
    assert tuple(act.shape) == (3,) and int(act.min()) >= 0 and int(act.max()) < 6
    if verbose:
        print(f"smoke OK: td err {err:.2e}, param err {perr:.2e}, {len(last)} priorities written, acting err {aerr:.2e}")
    return err, perr
