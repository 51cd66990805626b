"""GPU: LayerNorm(1024) numerics under stress.  The forward tiles apply LayerNorm BEHIND the trunk GEMM from row statistics
gathered on the side (fwd_kernels.h).  In its naive form -- var = E[x^2] - mean^2, pre = rstd * (x.W - mean * u) -- both
differences cancel catastrophically when a row's mean is large against its spread.  The rows here are made so: the phi
bias and the conv bias are raised until mean / std of the trunk input reaches 9, 24 and 50 (it is 0.3 at initialisation;
ReLU(phi) * e is non-negative, so long training can only push it up).  Reference semantics: nn.LayerNorm(eps=1e-5) of
/root/reference/prism/agents/models/ffnn_model.py:61-76 (torch computes it with a two-pass / Welford row moment).

What the kernel does about it: every wave shifts its slice of the row by a constant near the row's mean before it squares,
sums and multiplies, the row moments are combined as shifted moments, and the phi / conv biases are added to the finished
products instead of being accumulated onto.  Measured on MI355X, per-sample loss error against float64
(torch's own fp32 evaluation in brackets): ratio 9: 1.0e-5 (0.9e-5), ratio 24: 1.4e-5 (1.1e-5), ratio 50: 2.2e-5 (2.3e-5);
the one-pass form this replaced: 1.4e-5, 3.8e-5, 10.4e-5.

Bar: within 1e-5 of the fp32 oracle up to ratio 9 (the regime training can plausibly reach) with parameters after the Adam
step within 2e-6; beyond it the loss error against float64 must stay within twice torch's own fp32 error (+3e-6) -- the
regime's noise floor -- which the one-pass form misses by a factor of two to four."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_gpu_learner import build_hip_agent, to_hip_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("phi_bias,conv_bias", [(0.0, 0.0), (10.0, 2.0), (30.0, 5.0), (100.0, 10.0)])
@pytest.mark.parametrize("gemm_mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", ["iqn_c3", "full_c4"])
def test_layernorm_rows_with_large_mean(name, phi_bias, conv_bias, gemm_mode):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.learner_ref import LearnerOracle, composite_losses
    dev = "cuda:0"
    g = H.load_case(name)
    cfg, agent = build_hip_agent(g, dev, gemm_mode=gemm_mode)
    with torch.no_grad():
        sd = agent.model.state_dict()            # views of the flat parameter buffer: written in place
        sd["distribution_model.phi.0.bias"] += phi_bias
        sd["embedding_model.model.0.bias"] += conv_bias
        if agent.target_model is not None:
            agent.target_model.load_state_dict(agent.model.state_dict())
            agent._target_changed()
    sd0 = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
    tg0 = None if agent.target_model is None else {k: v.clone() for k, v in sd0.items()}
    batch, w, taus = H.case_batch(g, 0)
    td = agent.update(to_hip_batch(batch, dev), per_weights=w.to(dev), taus=[t.to(dev) for t in taus])
    torch.cuda.synchronize()
    spec = H.spec_from_config(cfg, C=int(g["C"]), A=int(g["A"]))
    orc = LearnerOracle(sd0, spec, tg0)
    orc.update(batch, w, taus)
    dl32 = orc.last["dl"]
    p64 = {k: v.double() for k, v in sd0.items()}
    t64 = None if tg0 is None else {k: v.double() for k, v in tg0.items()}
    b64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in batch.items()}
    dl64 = composite_losses(p64, t64, spec, b64, [t.double() for t in taus])[0]
    dl = agent.out_dl.cpu()
    floor = float((dl32.double() - dl64).abs().max())            # what fp32 arithmetic costs torch itself here
    err64 = float((dl.double() - dl64).abs().max())
    err32 = float((dl - dl32).abs().max())
    print(f"{gemm_mode} {name} phi_bias {phi_bias} conv_bias {conv_bias}: |dl - fp64| {err64:.2e} (torch fp32: {floor:.2e}), |dl - fp32 oracle| {err32:.2e}")
    assert err64 <= 2.0 * floor + 3e-6, f"kernel loss error vs fp64 {err64} against torch's own fp32 error {floor}"
    if phi_bias <= 10.0:
        assert err32 <= 1e-5, f"loss error vs the fp32 oracle {err32}"
        post = agent.model.state_dict()
        perr = max(float((post[k].cpu() - v).abs().max()) for k, v in orc.state_dict().items())
        assert perr <= 2e-6, f"parameter error after Adam {perr}"
    assert np.isfinite(td.cpu().numpy()).all()
