"""GPU parity: the fused HIP TD update (through the C ABI) vs the torch-CPU oracle on the SAME
minibatches and quantile samples as the golden reference runs.

Tolerances (fp32, BASELINE.json): per-sample losses / TD errors / total loss within 1e-5 of the
reference's recorded values; gradients within 1e-4 * max|g| + 1e-7 of oracle autograd; parameters after
each Adam step within 2e-6."""
import contextlib
import io

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
LOSS_TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return "cuda:0"


def build_hip_agent(g, dev, **extra):
    from prism_amd.factory import agent_factory
    cfg = H.case_config(g, device=dev, **extra)
    torch.manual_seed(int(g["seed"]))
    with contextlib.redirect_stdout(io.StringIO()):
        agent = agent_factory.build_agent(cfg, (10, 10, int(g["C"])), int(g["A"]))
    return cfg, agent


def to_hip_batch(batch, dev):
    from prism_amd.experience import Batch
    B = batch["obs"].shape[0]
    return Batch({"observation": batch["obs"].unsqueeze(1).to(dev),
                  "next": Batch({"observation": batch["next_obs"].unsqueeze(1).to(dev),
                                 "reward": batch["reward"].view(B, 1).to(dev)}),
                  "nonterminal": batch["nonterminal"].view(B, 1).to(dev),
                  "gamma": batch["gamma"].view(B, 1).to(dev),
                  "action": batch["action"].view(B, 1).to(dev)}, B, dev)


IQN_CASES = ["iqn_small", "iqn_c3", "iqn_tau32", "iqn_target", "iqn_doubleq",
             "full_small", "full_notarget", "full_doubleq", "full_c4",
             "dqn_c2", "dqn_ln", "dqn_target_c2",
             # the reference's ablation presets / experiment stages: width 256, T = 32, LayerNorm off or on
             "abl_iqn", "abl_ln_notarget", "abl_doubleq", "abl_ids", "abl_ids_var", "abl_sub",
             # value squish of the TD target (loss_squish_fn_id: symlog, obs_look_further) in every loss kernel that forms one
             "iqn_symlog", "full_olf", "dqn_symlog"]


# Both ways of multiplying the forward GEMMs (include/prism_hip.h gemm_mode) are held to the same fixtures at the same
# tolerances: the exact fp32 MFMA chain, and fp32 operands as three bf16 pieces on the bf16 matrix pipe.
@pytest.mark.parametrize("gemm_mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", IQN_CASES)
def test_iqn_update_matches_reference_and_oracle(dev, name, gemm_mode):
    from oracle.learner_ref import LearnerOracle
    g = H.load_case(name)
    cfg, agent = build_hip_agent(g, dev, gemm_mode=gemm_mode)
    # init parity by construction
    s0 = np.array([float(v.double().sum()) for v in agent.model.state_dict().values()])
    np.testing.assert_array_equal(s0, g["init_sum"])
    cpu_cfg = H.case_config(g)
    sd, tgt = H.build_init_state(cpu_cfg, int(g["seed"]), C=int(g["C"]), A=int(g["A"]))
    orc = LearnerOracle(sd, H.spec_from_config(cpu_cfg, C=int(g["C"]), A=int(g["A"])), tgt)
    names = list(sd.keys())
    kink_total = 0
    for step in range(int(g["steps"])):
        batch, w, taus = H.case_batch(g, step)
        g64 = orc.grads_fp64(batch, w, taus)             # same gradient in float64, at the pre-update parameters
        pre_sd = orc.state_dict()
        pre_tgt = None if orc.p_tgt is None else {k: v.clone() for k, v in orc.p_tgt.items()}
        td_o = orc.update(batch, w, taus)
        td = agent.update(to_hip_batch(batch, dev), per_weights=w.to(dev), taus=[t.to(dev) for t in taus])
        torch.cuda.synchronize()
        pre = f"s{step}/"
        # losses vs the live reference's recorded outputs
        np.testing.assert_allclose(td.cpu().numpy(), g[pre + "td"], rtol=0, atol=LOSS_TOL)
        if pre + "dl" in g.files:
            np.testing.assert_allclose(agent._static_distribution_loss.cpu().numpy(), g[pre + "dl"], rtol=0,
                                       atol=LOSS_TOL)
        else:
            assert agent._static_distribution_loss is None
        if pre + "ql" in g.files:
            np.testing.assert_allclose(agent._static_q_loss.cpu().numpy(), g[pre + "ql"], rtol=0, atol=LOSS_TOL)
            assert abs(float(agent.scalars[4]) - float(g[pre + "theil"])) < 1e-6
        assert abs(float(agent._static_total_loss) - float(g[pre + "total"])) < LOSS_TOL
        np.testing.assert_allclose(td.cpu().numpy(), td_o.numpy(), rtol=0, atol=LOSS_TOL)
        # gradients vs oracle autograd (unclipped).  A ReLU input within fp32 rounding distance of zero makes two
        # correct evaluations of the SAME gradient differ by that unit's whole contribution: the oracle's own fp32 and
        # fp64 results disagree wherever such units exist (abl_iqn, LayerNorm off: 16-19 of the 524 288 trunk
        # pre-activations of a step sit within 1e-6 of zero and 2-3 change sign between fp32 and fp64; 1.7e-5 on the
        # conv weight, 0.2 % of its largest element), and a kernel with another summation order may land on yet
        # another side.  So the tolerance of a tensor grows by twice the oracle's own fp32/fp64 disagreement on it
        # (1e-8-ish where no unit is kink-adjacent), and the (tensor, step) pairs that needed the allowance are counted.
        off = 0
        gflat = agent.grads.cpu()
        kinked, allowance = {}, {}
        jitter = None          # the oracle's gradient at parameters jittered by ~two ulps: evaluated only when a tensor needs it
        for k in names:
            n = sd[k].numel()
            go = orc.last["grads"][k].reshape(-1)
            gh = gflat[off:off + n]
            tol = 1e-4 * float(go.abs().max()) + 1e-7
            kink = 2.0 * float((go.double() - g64[k].reshape(-1)).abs().max())
            if min(float((gh - go).abs().max()), float((gh.double() - g64[k].reshape(-1)).abs().max())) > tol + kink:
                # A unit may sit closer to zero than the oracle's fp32 / fp64 pair resolves (both on the same side, the
                # device's K = 1024 sum -- another summation order -- on the other; seen at width 256 with LayerNorm ON: one
                # hidden unit of one row, abl_ln_notarget step 1, bf16x3 forward).  The oracle measures that on itself
                # too: its fp32 gradient at parameters jittered by about two units in the last place, six draws
                # (tests/test_gpu_fullsize_parity.py).  Where a jitter moves the oracle's gradient, a correct kernel may too.
                if jitter is None:
                    jitter = H.jitter_grads(pre_sd, pre_tgt, H.spec_from_config(cpu_cfg, C=int(g["C"]), A=int(g["A"])), batch, w,
                                            taus, seed=1234 + step)
                kink = max(kink, 2.0 * max(float((jg[k].reshape(-1) - go).abs().max()) for jg in jitter))
            allowance[k] = kink
            err32 = float((gh - go).abs().max())
            err64 = float((gh.double() - g64[k].reshape(-1)).abs().max())
            err = min(err32, err64)
            if err > tol:
                kinked[k] = err
            assert err <= tol + kink, f"step {step} grad {k}: max err {err32:.3e} (vs fp64: {err64:.3e}) > {tol:.3e} + {kink:.3e}"
            off += n
        kink_total += len(kinked)
        assert abs(float(agent.scalars[3]) - float(orc.last["grad_norm"])) < 1e-4 * max(1.0, float(orc.last["grad_norm"]))
        # parameters after the Adam step: vs oracle and vs the reference's checksums
        post = agent.model.state_dict()
        for k, v in orc.state_dict().items():
            # (where units are kink-adjacent, the Adam step of an element differs by up to lr * |dg| / adam_eps)
            atol = 2e-6 + cfg.learning_rate * allowance[k] / cfg.adam_epsilon
            np.testing.assert_allclose(post[k].cpu().numpy(), v.numpy(), rtol=0, atol=atol, err_msg=k)
        l2 = np.array([float(v.double().norm()) for v in post.values()])
        np.testing.assert_allclose(l2, g[pre + "post_l2"], rtol=2e-6, atol=1e-7)
        if cfg.use_target_network and step == 0:
            agent.sync_target_model()
            orc.sync_target()
    assert int(agent.optimizer.step_t.item()) == int(g["steps"])
    # Width 128 never needed the allowance; at width 256 (half a million trunk ReLU units per step) a unit within rounding
    # distance of zero turns up now and then -- every use is bounded above by what the oracle shows on itself.
    assert kink_total == 0 or max(cfg.iqn_quantile_model_feature_dim if cfg.use_iqn else 0, cfg.ids_q_head_feature_dim if cfg.use_ids else 0) >= 256, f"{kink_total} (tensor, step) pairs needed the kink allowance"


def test_philox_taus_are_uniform_and_recorded(dev):
    g = H.load_case("iqn_c3")
    cfg, agent = build_hip_agent(g, dev)
    batch, w, _ = H.case_batch(g, 0)
    hb = to_hip_batch(batch, dev)
    td1 = agent.update(hb, per_weights=w.to(dev)).clone()
    t1 = agent.tau_out.clone()
    torch.cuda.synchronize()
    B, T = int(g["B"]), 8
    cur = t1[0, :T * B].cpu().numpy()
    assert 0.0 <= cur.min() and cur.max() < 1.0 and abs(cur.mean() - 0.5) < 0.03
    # replaying the recorded taus through the oracle reproduces the in-kernel-RNG step
    from oracle.learner_ref import LearnerOracle
    cpu_cfg = H.case_config(g)
    sd, tgt = H.build_init_state(cpu_cfg, int(g["seed"]), C=int(g["C"]), A=int(g["A"]))
    orc = LearnerOracle(sd, H.spec_from_config(cpu_cfg, C=int(g["C"]), A=int(g["A"])), tgt)
    taus = [t1[0, :T * B].cpu().reshape(-1, 1), t1[1, :T * B].cpu().reshape(-1, 1)]
    td_o = orc.update(batch, w, taus)
    np.testing.assert_allclose(td1.cpu().numpy(), td_o.numpy(), rtol=0, atol=LOSS_TOL)
    agent.update(hb, per_weights=w.to(dev))
    assert not torch.equal(agent.tau_out, t1)          # fresh draws every step


def test_unsupported_config_fails_loudly(dev):
    from prism_amd.factory.model_factory import UnsupportedConfig
    g = H.load_case("iqn_c3")
    cfg, agent = build_hip_agent(g, dev, iqn_quantile_model_feature_dim=512)     # widths covered: 128, 256
    batch, w, taus = H.case_batch(g, 0)
    with pytest.raises(UnsupportedConfig):
        agent.update(to_hip_batch(batch, dev), per_weights=w.to(dev))
