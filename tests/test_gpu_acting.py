"""GPU parity of the acting path: ``Agent.forward`` = CompositeModel.forward(for_action=True) +
the action selector, run by ``prism_act_forward`` / ``prism_ids_select`` on the weights the golden update steps left
behind, against the live reference's recorded outputs (tests/golden/update_*.npz, ``act/*``) and the CPU oracle.

Tolerance (fp32): estimates within 1e-5 of the reference's; IDS scores within 1e-3 relative (a ratio of a squared
regret and a logarithm of variances: each factor carries the estimates' 1e-6 relative error several times over);
the chosen actions equal."""
import ctypes

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_gpu_learner import IQN_CASES, build_hip_agent, to_hip_batch

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return "cuda:0"


def updated_agent(name, dev, with_oracle=False, **extra):
    g = H.load_case(name)
    cfg, agent = build_hip_agent(g, dev, **extra)
    orc = None
    if with_oracle:
        from oracle.learner_ref import LearnerOracle
        cpu_cfg = H.case_config(g)
        sd, tgt = H.build_init_state(cpu_cfg, int(g["seed"]), C=int(g["C"]), A=int(g["A"]))
        orc = LearnerOracle(sd, H.spec_from_config(cpu_cfg, C=int(g["C"]), A=int(g["A"])), tgt)
    for step in range(int(g["steps"])):
        batch, w, taus = H.case_batch(g, step)
        agent.update(to_hip_batch(batch, dev), per_weights=w.to(dev), taus=[t.to(dev) for t in taus])
        if orc is not None:
            orc.update(batch, w, taus)
        if cfg.use_target_network and step == 0:
            agent.sync_target_model()
            if orc is not None:
                orc.sync_target()
    return (g, cfg, agent, orc) if with_oracle else (g, cfg, agent)


@pytest.mark.parametrize("gemm_mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", IQN_CASES)
def test_acting_matches_reference(dev, name, gemm_mode):
    g, cfg, agent, orc = updated_agent(name, dev, with_oracle=True, gemm_mode=gemm_mode)
    obs, taus, ref = H.case_act(g)
    # (1) on the parameters the device's own updates left: the packed weight copies follow the update (stale ones would be
    # off by the size of two Adam steps, ~1e-2).  Not held to 1e-5: where an update met a ReLU unit within rounding distance
    # of zero (tests/test_gpu_learner.py: abl_ln_notarget, step 1) two correct updates differ by up to lr per element.
    q, dist = agent.act_estimates(obs.to(dev), taus=None if taus is None else taus.to(dev))
    torch.cuda.synchronize()
    np.testing.assert_allclose(q.cpu().numpy(), ref["q"], rtol=0, atol=2e-4)
    # (2) the acting forward itself, on the oracle's parameters after the same updates (pinned to the reference's:
    # tests/test_oracle_golden.py), at the parity tolerance
    agent.model.load_state_dict(orc.state_dict())
    agent._params_replaced()
    q, dist = agent.act_estimates(obs.to(dev), taus=None if taus is None else taus.to(dev))
    torch.cuda.synchronize()
    assert tuple(q.shape) == ref["q"].shape
    np.testing.assert_allclose(q.cpu().numpy(), ref["q"], rtol=0, atol=TOL)
    if "dist" in ref:
        assert tuple(dist.shape) == ref["dist"].shape
        np.testing.assert_allclose(dist.cpu().numpy(), ref["dist"], rtol=0, atol=TOL)
    else:
        assert dist is None
    # the selector on these estimates: the reference's choice
    sel = agent.action_selector
    probs = sel.generate_action_probs(dist, q)
    np.testing.assert_array_equal(sel.select_action(probs).cpu().numpy(), ref["action"])
    if cfg.use_ids:
        # the fused scoring kernel, on the buffers the forward left on the device
        z, qb, n, n_pad, T = agent._act_raw
        from prism_amd import _native as N
        A = int(g["A"])
        scores = torch.empty((n, A), device=dev)
        aux = torch.empty((n, 4, A), device=dev)
        action = torch.empty(n, dtype=torch.int64, device=dev)
        from prism_amd.agents.squish_functions import SQUISH_IDS
        N.check(N.lib().prism_ids_select(N.ptr(z), N.ptr(qb), n, n_pad, T, A, cfg.ids_n_q_heads, cfg.ids_lambda,
                                         cfg.ids_epsilon, cfg.ids_rho_lower_bound, SQUISH_IDS.get(str(cfg.loss_squish_fn_id), 0),
                                         N.ptr(scores), N.ptr(aux), N.ptr(action),
                                         None, N.current_stream_handle()), "prism_ids_select")
        torch.cuda.synchronize()
        np.testing.assert_array_equal(action.cpu().numpy(), ref["action"])
        np.testing.assert_allclose(scores.cpu().numpy(), ref["ids/IDS Scores"], rtol=1e-3, atol=1e-6)
        a = aux.cpu().numpy()
        np.testing.assert_allclose(a[:, 0], ref["ids/Q Estimate Ensemble Mean"], rtol=0, atol=TOL)
        np.testing.assert_allclose(a[:, 1], ref["ids/Q Estimate Ensemble Variance"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(a[:, 2], ref["ids/Return Distribution Variance"], rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(a[:, 3], ref["ids/Information Gain"], rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("name", ["full_small", "abl_ids", "iqn_c3"])
def test_agent_forward_in_pieces_and_with_device_draws(dev, name):
    """Agent.forward end to end (in-kernel Philox quantile samples): more observations than fit the workspace at once
    are served in pieces; the same counter gives the same estimates; every action is a valid index and agrees with
    the torch selector run on the same estimates."""
    from oracle.learner_ref import act_forward
    g, cfg, agent = updated_agent(name, dev)
    C, A = int(g["C"]), int(g["A"])
    rng = np.random.default_rng(5)
    n = int(g["B"]) + 7
    obs = torch.from_numpy((rng.random((n, 10, 10, C)) < 0.1).astype(np.float32)).to(dev)
    T = cfg.iqn_quantile_samples_per_action
    taus = torch.from_numpy(rng.random((T * n, 1)).astype(np.float32))
    q, dist = agent.act_estimates(obs, taus=taus.to(dev))
    sd = {k: v.cpu() for k, v in agent.model.state_dict().items()}
    qo, do = act_forward(sd, H.spec_from_config(H.case_config(g), C=C, A=A), obs.cpu(), taus)
    np.testing.assert_allclose(q.cpu().numpy(), qo.numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(dist.cpu().numpy(), do.numpy(), rtol=0, atol=TOL)
    # device draws: reproducible from the counter, uniform, and a new draw each call
    agent._act_draws = 1000
    q1, d1 = agent.act_estimates(obs[:5])
    agent._act_draws = 1000
    q2, d2 = agent.act_estimates(obs[:5])
    assert torch.equal(d1, d2) and torch.equal(q1, q2)
    q3, d3 = agent.act_estimates(obs[:5])
    assert not torch.equal(d1, d3)
    if cfg.use_ids:
        assert torch.equal(q1, q3)          # the ensemble heads take no quantile samples
    act = agent.forward(obs[:5])
    assert act.dtype == torch.int64 and tuple(act.shape) == (5,) and int(act.min()) >= 0 and int(act.max()) < A
    agent.eval()
    greedy = agent.forward(obs[:5])
    agent.train()
    assert tuple(greedy.shape) == (5,)


def test_ids_select_against_oracle_on_random_estimates(dev):
    from oracle.learner_ref import ids_scores
    from prism_amd import _native as N
    rng = np.random.default_rng(11)
    for n, T, A, heads in ((1, 200, 6, 10), (7, 32, 3, 2), (33, 200, 16, 10), (4, 1, 6, 1)):
        dist = torch.from_numpy(rng.standard_normal((T, n, A)).astype(np.float32) * 2.0 + 1.0)
        q = torch.from_numpy(rng.standard_normal((n, A, heads)).astype(np.float32))
        n_pad = (n + 15) // 16 * 16
        z = dist.permute(1, 0, 2).contiguous().to(dev)
        qb = torch.zeros((heads, n_pad, A))
        qb[:, :n] = q.permute(2, 0, 1)
        qb = qb.to(dev)
        scores = torch.empty((n, A), device=dev)
        action = torch.empty(n, dtype=torch.int64, device=dev)
        N.check(N.lib().prism_ids_select(N.ptr(z), N.ptr(qb), n, n_pad, T, A, heads, 0.1, 1e-10, 0.25, 0, N.ptr(scores), None,
                                         N.ptr(action), None, N.current_stream_handle()), "prism_ids_select")
        torch.cuda.synchronize()
        if T == 1 or heads == 1:
            continue          # torch.var / torch.std of a single element are NaN in the reference too; only "does not fault"
        r = ids_scores(dist, q, 0.1, 1e-10, 0.25)
        np.testing.assert_allclose(scores.cpu().numpy(), r["scores"].numpy(), rtol=2e-4, atol=1e-7)
        np.testing.assert_array_equal(action.cpu().numpy(), r["action"].numpy())


def test_act_forward_rejects_bad_arguments(dev):
    from prism_amd import _native as N
    g, cfg, agent = updated_agent("iqn_small", dev)
    obs = torch.zeros((4, 10, 10, int(g["C"])), device=dev)
    agent.act_estimates(obs)
    L, d = N.lib(), agent._desc
    z = torch.empty((1024, 6), device=dev)
    assert L.prism_act_forward(ctypes.byref(d), N.ptr(obs), 0, 8, None, 1, 0, N.ptr(z), None, None) == N.PRISM_ERR_INVALID
    assert L.prism_act_forward(ctypes.byref(d), None, 4, 8, None, 1, 0, N.ptr(z), None, None) == N.PRISM_ERR_INVALID
    assert L.prism_act_forward(ctypes.byref(d), N.ptr(obs), 4, 8, None, 1, 0, None, None, None) == N.PRISM_ERR_INVALID
    assert L.prism_act_forward(ctypes.byref(d), N.ptr(obs), agent._B + 1, 8, None, 1, 0, N.ptr(z), None, None) == N.PRISM_ERR_INVALID


def test_acting_before_the_first_update(dev):
    """The collector acts before the learner has updated once: no minibatch is bound to the descriptor yet."""
    from oracle.learner_ref import act_forward
    g = H.load_case("full_small")
    cfg, agent = build_hip_agent(g, dev)
    obs, taus, _ = H.case_act(g)
    q, dist = agent.act_estimates(obs.to(dev), taus=taus.to(dev))
    sd = {k: v.cpu() for k, v in agent.model.state_dict().items()}
    qo, do = act_forward(sd, H.spec_from_config(H.case_config(g), C=int(g["C"]), A=int(g["A"])), obs, taus)
    np.testing.assert_allclose(q.cpu().numpy(), qo.numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(dist.cpu().numpy(), do.numpy(), rtol=0, atol=TOL)
    assert tuple(agent.forward(obs.to(dev)).shape) == (obs.shape[0],)


@pytest.mark.parametrize("name", ["full_small", "iqn_c3", "dqn_c2", "dqn_ln"])
def test_forward_selects_natively_also_in_pieces(dev, name):
    """Agent.forward with MORE observations than the learner batch (the collector may pass any number): the actions come
    from prism_ids_select / prism_greedy_select piece by piece and must have the full length and equal the selector's
    torch code run on the same estimates (same Philox counters).  Covers deterministic IDS, greedy (eval mode) and
    epsilon-greedy (both branches), with no ATen math in the product path for these selectors."""
    from prism_amd.agents.action_selectors import EGreedyActionSelector, GreedyActionSelector
    g, cfg, agent = updated_agent(name, dev)
    C, A = int(g["C"]), int(g["A"])
    rng = np.random.default_rng(9)
    n = 2 * int(g["B"]) + 5
    obs = torch.from_numpy((rng.random((n, 10, 10, C)) < 0.1).astype(np.float32)).to(dev)

    def both(selector_for_torch):
        agent._act_draws = 5000
        act = agent.forward(obs)
        agent._act_draws = 5000
        q, dist = agent.act_estimates(obs)
        want = selector_for_torch.select_action(selector_for_torch.generate_action_probs(dist, q))
        torch.cuda.synchronize()
        assert act.dtype == torch.int64 and tuple(act.shape) == (n,)
        np.testing.assert_array_equal(act.cpu().numpy(), want.cpu().numpy())

    both(agent.action_selector)                     # IDS (full_*) or greedy (use_e_greedy False in the fixtures)
    agent.eval()
    both(GreedyActionSelector())
    agent.train()
    # epsilon-greedy: epsilon 1 -> the host generator's actions; epsilon 0 -> the greedy kernel
    agent.action_selector = EGreedyActionSelector(1.0, 1.0, 0, seed=7)
    twin = np.random.RandomState(7)
    twin.uniform(0, 1)
    draws0 = agent._act_draws
    np.testing.assert_array_equal(agent.forward(obs).cpu().numpy(), twin.randint(A, size=(n,)))
    # the explore branch consumes the forward's quantile draws like the reference, which runs the model BEFORE the coin
    # (agent.py:33-36): the streams of explore-then-greedy sequences stay aligned
    if agent.dims.use_iqn:
        assert agent._act_draws == draws0 + n * int(agent.model.distribution_model.n_quantile_samples_per_action)
    else:
        assert agent._act_draws == draws0
    agent.action_selector = EGreedyActionSelector(0.0, 0.0, 0, seed=7)
    agent._act_draws = 5000
    act = agent.forward(obs)
    agent._act_draws = 5000
    q, dist = agent.act_estimates(obs)
    g_sel = GreedyActionSelector()
    np.testing.assert_array_equal(act.cpu().numpy(), g_sel.select_action(g_sel.generate_action_probs(dist, q)).cpu().numpy())


@pytest.mark.parametrize("name", ["full_small", "iqn_c3", "dqn_ln"])
def test_forward_from_the_hipgraph_equals_the_eager_launches(dev, name):
    """Agent.forward replayed from its hipGraph (the default: H2D of the observations, prism_act_forward, the selector kernel
    and the D2H of the actions in one launch; quantile draws from the device counter rng_counters[2]) against the eager
    launches on the same counters: the same actions call after call, from a host array, from a persistent device buffer
    (the reference collector's pattern) and from fresh device tensors; ``.cpu()`` of the result is the graph's own copy."""
    g, cfg, agent = updated_agent(name, dev)
    C = int(g["C"])
    rng = np.random.default_rng(3)
    n = 5
    assert agent.act_graph
    frames = [(rng.random((n, 10, 10, C)) < 0.1).astype(np.float32) for _ in range(7)]
    agent.act_graph = False
    agent._act_draws = 777
    want = [agent.forward(f).cpu().numpy().copy() for f in frames]
    end_draws = agent._act_draws
    agent.act_graph = True
    # host arrays in
    agent._act_draws = 777
    got = [agent.forward(f) for f in frames]
    assert type(got[-1]).__name__ == "_Actions" and got[-1].is_cuda and got[-1].dtype == torch.int64
    agent._act_draws = 777
    for i, f in enumerate(frames):
        a = agent.forward(f)
        np.testing.assert_array_equal(a.cpu().numpy(), want[i])             # the pinned copy of the graph's D2H node
        np.testing.assert_array_equal(torch.Tensor.cpu(a.as_subclass(torch.Tensor)).numpy(), want[i])      # ... and the device tensor
    assert agent._act_draws == end_draws
    assert int(agent.rng_counters[2].item()) == end_draws                    # the device counter mirrors the host's
    # one persistent device buffer in (read in place, keyed by its address)
    buf = torch.zeros((n, 10, 10, C), device=dev)
    agent._act_draws = 777
    for i, f in enumerate(frames):
        buf.copy_(torch.from_numpy(f))
        np.testing.assert_array_equal(agent.forward(buf).cpu().numpy(), want[i])
    # fresh device tensors every call (more than three addresses: staged through one device block)
    agent._act_draws = 777
    keep = []
    for i, f in enumerate(frames):
        t = torch.from_numpy(f).to(dev)
        keep.append(t)
        np.testing.assert_array_equal(agent.forward(t).cpu().numpy(), want[i])
    assert any(k[2] == "dev" for k in agent._act_graphs)
    assert any(st["g"] is not None for st in agent._act_graphs.values()), "no call was replayed from a graph"
