"""Run-to-run determinism of the fused step: two learners built from one seed, stepping side by side, must stay BIT-identical
-- parameters, Adam moments, gradients and the priority tree.  Every kernel sums in a fixed order and uses no float atomics,
so any difference is a race or an arrival-order dependence.  Round 4's tools/soak.py `twin` mode found one: the post launch's
conv-backward role reported its sum of squared gradients in the slot of WHICHEVER workgroup arrived last, so the order in which
the partial norms were added changed from run to run and, once the global norm clipped, every parameter differed in the last
bit (step_kernels.h `norm_slot`).  Clipping is forced here (max_grad_norm far below the norm) so that such a thing shows within
a few steps; the reference's own loop (/root/reference/prism/learner.py:95-125) is deterministic given its seeds, too."""
import contextlib
import io

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    # the full default (IDS heads + IQN + target + PER): conv backward as a post-launch role, ticketed fold
    "full_b64": dict(base=3, B=64),
    # IQN + PER (conv taps inside the backward kernel, fused tail behind the grid barrier)
    "iqn_per": dict(base=2, B=256),
    # the ablation presets' width and T (bf16 forward at H = 256, iqn_bwd4_kernel, small-batch Q backward)
    "w256_full": dict(base=3, B=64, iqn_quantile_model_feature_dim=256, ids_q_head_feature_dim=256,
                      iqn_n_current_state_quantile_samples=32, iqn_n_next_state_quantile_samples=32),
}


def _learner(base, B, **over):
    from prism_amd.config import baseline_config
    from prism_amd.learner import Learner
    from prism_amd.synthetic import fill_replay
    cfg = baseline_config(base, device="cuda:0", batch_size=B, experience_replay_capacity=20_000, max_grad_norm=0.05, **over)
    cfg.fused_step, cfg.hip_graph = True, True
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
    fill_replay(ln.experience_buffer, 20_000, seed=3)
    return ln


@pytest.mark.parametrize("name", sorted(CASES))
def test_two_learners_of_one_seed_stay_bit_identical(name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    kw = dict(CASES[name])
    a, b = _learner(**kw), _learner(**kw)
    for chunk in range(6):
        for _ in range(50):
            a.step()
            b.step()
        torch.cuda.synchronize()
        where = f"{name}, after {50 * (chunk + 1)} steps"
        assert float(a.agent.scalars[5]) < 1.0, f"{where}: the global norm is not clipping -- the case would prove nothing"
        for what, x, y in (("parameters", a.agent.flat, b.agent.flat), ("Adam m", a.agent.optimizer.exp_avg, b.agent.optimizer.exp_avg),
                           ("Adam v", a.agent.optimizer.exp_avg_sq, b.agent.optimizer.exp_avg_sq), ("gradients", a.agent.grads, b.agent.grads)):
            assert torch.equal(x, y), f"{where}: {what} differ in {int((x != y).sum())} elements"
        if a.experience_buffer.use_per:
            assert torch.equal(a.experience_buffer.tree, b.experience_buffer.tree), f"{where}: priority trees differ"
    a.agent.check_status()
    b.agent.check_status()
