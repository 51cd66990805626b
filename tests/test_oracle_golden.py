"""Pins the CPU oracle (oracle/learner_ref.py) to golden vectors captured from the live reference
(tools/gen_golden.py).  Runs without a GPU and without /root/reference."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from oracle.learner_ref import LearnerOracle, act_forward, ids_scores

LOSS_TOL = 1e-5       # BASELINE.json: loss parity to reference within 1e-5


@pytest.mark.parametrize("name", H.UPDATE_CASES)
def test_oracle_matches_reference_update(name):
    g = H.load_case(name)
    cfg = H.case_config(g)
    sd, tgt = H.build_init_state(cfg, int(g["seed"]), C=int(g["C"]), A=int(g["A"]))
    s0, l0 = H.checksums(sd)
    assert list(sd.keys()) == list(g["param_names"])
    np.testing.assert_array_equal(s0, g["init_sum"])           # init parity is exact
    np.testing.assert_array_equal(l0, g["init_l2"])
    if "init/" + list(sd.keys())[0] in g.files:
        for k in sd:
            np.testing.assert_array_equal(sd[k].numpy(), g["init/" + k])

    orc = LearnerOracle(sd, H.spec_from_config(cfg, C=int(g["C"]), A=int(g["A"])), tgt)
    for step in range(int(g["steps"])):
        batch, w, taus = H.case_batch(g, step)
        td = orc.update(batch, w, taus)
        pre = f"s{step}/"
        np.testing.assert_allclose(td.numpy(), g[pre + "td"], rtol=0, atol=LOSS_TOL)
        if pre + "dl" in g.files:
            np.testing.assert_allclose(orc.last["dl"].numpy(), g[pre + "dl"], rtol=0, atol=LOSS_TOL)
        if pre + "ql" in g.files:
            np.testing.assert_allclose(orc.last["ql"].numpy(), g[pre + "ql"], rtol=0, atol=LOSS_TOL)
        assert abs(float(orc.last["total"]) - float(g[pre + "total"])) < LOSS_TOL
        if pre + "theil" in g.files:
            assert abs(float(orc.last["theil"]) - float(g[pre + "theil"])) < 1e-6
        # clipped per-tensor grad norms as left on the reference's parameters
        coef = min(1.0, cfg.max_grad_norm / (float(orc.last["grad_norm"]) + 1e-6))
        gl2 = np.array([float(orc.last["grads"][k].double().norm()) * coef for k in orc.p])
        np.testing.assert_allclose(gl2, g[pre + "clipped_grad_l2"], rtol=2e-4, atol=1e-7)
        s, l2 = H.checksums(orc.state_dict())
        np.testing.assert_allclose(l2, g[pre + "post_l2"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(s, g[pre + "post_sum"], rtol=0, atol=2e-4)
        if pre + "post/" + list(sd.keys())[0] in g.files:
            for k, v in orc.state_dict().items():
                np.testing.assert_allclose(v.numpy(), g[pre + "post/" + k], rtol=0, atol=2e-6)
        if cfg.use_target_network and step == 0:
            orc.sync_target()
    # acting on the updated weights: CompositeModel.forward(for_action=True) and the IDS selector's intermediates
    obs, taus, ref = H.case_act(g)
    q, dist = act_forward(orc.state_dict(), orc.spec, obs, taus)
    np.testing.assert_allclose(q.numpy(), ref["q"], rtol=0, atol=LOSS_TOL)
    if dist is not None:
        np.testing.assert_allclose(dist.numpy(), ref["dist"], rtol=0, atol=LOSS_TOL)
    if cfg.use_ids:
        from oracle.learner_ref import squish_pair
        r = ids_scores(dist, q, cfg.ids_lambda, cfg.ids_epsilon, cfg.ids_rho_lower_bound,
                       squish_pair(H.spec_from_config(cfg, C=int(g["C"]), A=int(g["A"])))[1])
        np.testing.assert_allclose(r["scores"].numpy(), ref["ids/IDS Scores"], rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(r["var_z"].numpy(), ref["ids/Return Distribution Variance"], rtol=1e-3, atol=1e-7)
        np.testing.assert_array_equal(r["action"].numpy(), ref["action"])
