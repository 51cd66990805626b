"""ORACLE — test infrastructure only.  Never imported by ``prism_amd``.

torch-CPU fp32 restatement of the reference's TD-update arithmetic, written functionally over a
``{state_dict key: tensor}`` mapping (keys as in SURVEY.md Appendix B).  Gradients come from
autograd, so this is independent of the hand-derived backward in the HIP kernels.  Pinned against
golden vectors captured from the live reference (tools/gen_golden.py -> tests/golden/update_*.npz,
checked by tests/test_oracle_golden.py).

Follows (paths under /root/reference):
  prism/agents/models/minatar_cnn_model.py:14-16,43-46   conv embed
  prism/agents/models/iqn_model.py:48-93                 IQN forward
  prism/agents/models/iqn_model.py:95-201                IQN target + quantile-Huber loss
  prism/agents/models/ffnn_model.py:61-81                LN/Linear/act stack
  prism/agents/models/q_ensemble.py:44-92                Q-ensemble / DQN loss (always MSE:
                                                         prism/factory/model_factory.py:39-46)
  prism/agents/models/composite_model.py:94-144          batch unpack, td errors
  prism/agents/agent.py:53-79                            loss reduce, clip, optimizer step
  prism/factory/agent_factory.py:44-47                   Adam flags
  prism/agents/models/composite_model.py:51-70           acting forward (act_forward)
  prism/agents/action_selectors.py:125-176               IDS scores (ids_scores)
Taus are explicit inputs, in the reference's draw order (iqn_model.py:104,112-126):
current -> online-next (no target, or double-Q) -> target-next (target present).
"""
from dataclasses import dataclass
import math

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class ModelSpec:
    in_channels: int = 4
    n_actions: int = 6
    use_iqn: bool = True
    use_layer_norm: bool = True
    n_basis: int = 64
    iqn_layers: int = 1           # iqn_quantile_model_layers
    n_tau: int = 8                # current-state quantile samples
    n_tau_next: int = 8
    huber_k: float = 1.0
    squish: str = "none"          # loss_squish_fn_id: "obs_look_further" | "symlog" | anything else = none (model_factory.py:16-23)
    dist_loss_weight: float = 1.0
    propagate_grad: bool = True
    n_heads: int = 0              # 0: no q model; 1: DQN; 10: IDS ensemble
    head_layers: int = 0          # ids_n_q_head_model_layers / dqn_n_model_layers
    q_loss_weight: float = 1.0
    theil_coef: float = 0.0       # ids_ensemble_variation_coef (0 for DQN)
    double_q: bool = False
    max_grad_norm: float = 10.0
    lr: float = 2.5e-4
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1.5e-4


def squish_pair(spec):
    """(squish, unsquish) of the TD target, restating /root/reference/prism/agents/squish_functions.py:4-18 (same torch
    operations in the same order, so fp32 and fp64 evaluations round as the reference's do); (None, None) without one."""
    if spec.squish == "symlog":
        return (lambda x: torch.sign(x) * torch.log(x.abs() + 1)), (lambda x: torch.sign(x) * (torch.exp(x.abs()) - 1))
    if spec.squish == "obs_look_further":
        return (lambda x: torch.sign(x) * (torch.sqrt(x.abs() + 1) - 1) + 0.01 * x), \
               (lambda y: torch.sign(y) * (torch.square((torch.sqrt(1 + 4 * 0.01 * (torch.abs(y) + 1 + 0.01)) - 1) / (2 * 0.01)) - 1))
    return None, None


def conv_embed(p, obs):
    """(B,10,10,C) -> (B,1024), channel-major flatten."""
    x = obs.permute(0, 3, 1, 2).to(p["embedding_model.model.0.weight"].dtype)      # (.float() in the reference)
    y = F.relu(F.conv2d(x, p["embedding_model.model.0.weight"], p["embedding_model.model.0.bias"]))
    return y.flatten(1)


def _stack(p, prefix, x, n_layers, use_ln, ln_first, final_act):
    """FFNNModel (ffnn_model.py:61-76): [LN] Linear [ReLU] ... ; indices advance as in nn.Sequential."""
    idx = 0
    for i in range(n_layers):
        if use_ln and (i != 0 or ln_first):
            x = F.layer_norm(x, (x.shape[-1],), p[f"{prefix}.{idx}.weight"], p[f"{prefix}.{idx}.bias"], 1e-5)
            idx += 1
        x = F.linear(x, p[f"{prefix}.{idx}.weight"], p[f"{prefix}.{idx}.bias"])
        idx += 1
        if i != n_layers - 1:
            x = F.relu(x)
            idx += 1
    if final_act:
        x = F.relu(x)
    return x


def iqn_forward(p, spec, e, taus):
    """e (B,E); taus (T*B,1) -> Z (T*B, A).  Rows are tau-major: row = t*B + b."""
    B = e.shape[0]
    T = taus.shape[0] // B
    if not spec.propagate_grad:
        e = e.detach()
    basis = torch.arange(1, spec.n_basis + 1, 1)
    c = torch.cos(torch.tile(taus, [1, spec.n_basis]) * basis * np.pi)
    phi = F.relu(F.linear(c, p["distribution_model.phi.0.weight"], p["distribution_model.phi.0.bias"]))
    h = phi * torch.tile(e, [T, 1])
    if spec.iqn_layers > 0:
        h = _stack(p, "distribution_model.model.model", h, spec.iqn_layers, spec.use_layer_norm, True, True)
    if spec.use_layer_norm:
        h = F.layer_norm(h, (h.shape[-1],), p["distribution_model.embedding_to_quantile_layer.0.weight"],
                         p["distribution_model.embedding_to_quantile_layer.0.bias"], 1e-5)
        z = F.linear(h, p["distribution_model.embedding_to_quantile_layer.1.weight"],
                     p["distribution_model.embedding_to_quantile_layer.1.bias"])
    else:
        z = F.linear(h, p["distribution_model.embedding_to_quantile_layer.weight"],
                     p["distribution_model.embedding_to_quantile_layer.bias"])
    return z


def iqn_loss(p, p_tgt, spec, e_cur, e_next, acts, returns, dg, taus):
    """Returns per-sample quantile-Huber loss (B,).  taus: list in draw order."""
    B = acts.shape[0]
    T, Tn, k = spec.n_tau, spec.n_tau_next, spec.huber_k
    taus = list(taus)
    tau_cur = taus.pop(0)
    z_cur = iqn_forward(p, spec, e_cur, tau_cur)
    with torch.no_grad():
        if p_tgt is None:
            z_on = iqn_forward(p, spec, e_next, taus.pop(0))
            z_tg = z_on
        elif spec.double_q:
            z_on = iqn_forward(p, spec, e_next, taus.pop(0))
            z_tg = iqn_forward(p_tgt, spec, e_next, taus.pop(0))
        else:
            z_tg = iqn_forward(p_tgt, spec, e_next, taus.pop(0))
            z_on = z_tg
        a_star = z_on.view(Tn, B, -1).mean(dim=0).argmax(dim=-1).view(-1, 1)
        zsel = torch.gather(z_tg, 1, torch.tile(a_star, [Tn, 1]))
        squish, unsquish = squish_pair(spec)                         # iqn_model.py:141-148
        if unsquish is not None:
            zsel = unsquish(zsel)
        y = torch.tile(returns.view(-1, 1), [Tn, 1]) + zsel * torch.tile(dg.view(-1, 1), [Tn, 1])
        if squish is not None:
            y = squish(y)
        y = y.view(Tn, B, 1).transpose(1, 0)                         # (B, T', 1)
    q = torch.gather(z_cur, 1, torch.tile(acts.view(-1, 1), [T, 1])).view(T, B, 1).transpose(1, 0)
    delta = y[:, :, None] - q[:, None, :]                             # (B, T', T, 1)
    le = torch.le(delta.abs(), k).float()
    hub = le * 0.5 * delta.square() + (1 - le) * k * (delta.abs() - 0.5 * k)
    tq = tau_cur.view(T, B, 1).transpose(1, 0)
    tq = torch.tile(tq[:, None, :, :], [1, Tn, 1, 1]).float()
    ind = torch.where(delta < 0, 1, 0).float().detach()
    rho = (torch.abs(tq - ind) * hub) / k
    return rho.sum(dim=2).mean(dim=1).view(-1) * spec.dist_loss_weight


def qens_forward(p, spec, e):
    """(B,E) -> (B, A, n_heads)."""
    outs = []
    for h in range(spec.n_heads):
        if spec.head_layers > 0:
            outs.append(_stack(p, f"q_function_model.q_heads.{h}.model", e, spec.head_layers,
                               spec.use_layer_norm, True, False))
        elif spec.use_layer_norm:
            x = F.layer_norm(e, (e.shape[-1],), p[f"q_function_model.q_heads.{h}.0.weight"],
                             p[f"q_function_model.q_heads.{h}.0.bias"], 1e-5)
            outs.append(F.linear(x, p[f"q_function_model.q_heads.{h}.1.weight"],
                                 p[f"q_function_model.q_heads.{h}.1.bias"]))
        else:
            outs.append(F.linear(e, p[f"q_function_model.q_heads.{h}.weight"],
                                 p[f"q_function_model.q_heads.{h}.bias"]))
    return torch.stack(outs, dim=-1)


def qens_loss(p, p_tgt, spec, e_cur, e_next, acts, returns, dg):
    B = acts.shape[0]
    ar = torch.arange(B)
    q_cur = qens_forward(p, spec, e_cur)
    with torch.no_grad():
        if p_tgt is None:
            q_on = q_tg = qens_forward(p, spec, e_next)
        elif spec.double_q:
            q_on = qens_forward(p, spec, e_next)
            q_tg = qens_forward(p_tgt, spec, e_next)
        else:
            q_tg = q_on = qens_forward(p_tgt, spec, e_next)
        best = q_on.argmax(dim=-2)                                    # (B, heads)
        nxt = q_tg[ar[:, None], best, torch.arange(spec.n_heads)[None, :]]
        squish, unsquish = squish_pair(spec)                         # q_ensemble.py:77-82
        if unsquish is not None:
            nxt = unsquish(nxt)
        target = returns.view(-1, 1) + nxt * dg.view(-1, 1)
        if squish is not None:
            target = squish(target)
    ql = F.mse_loss(q_cur[ar, acts, :], target, reduction="none").mean(dim=-1)
    theil = torch.tensor(0.0)
    if spec.theil_coef != 0:
        l2 = torch.stack([torch.cat([p[k].reshape(-1) for k in _head_keys(p, h)]).norm()
                          for h in range(spec.n_heads)])
        ratio = l2 / l2.mean()
        theil = (ratio * torch.log(ratio)).mean()
    return spec.q_loss_weight * (ql - theil * spec.theil_coef), theil


def act_forward(p, spec, obs, taus):
    """CompositeModel.forward(x, for_action=True) (composite_model.py:51-70): (q (n,A,heads), dist (T,n,A) or None).
    taus (T*n,1) is the draw iqn_model.py:66-68 makes with T = n_quantile_samples_per_action."""
    with torch.no_grad():
        e = conv_embed(p, obs)
        dist = None
        if spec.use_iqn:
            dist = iqn_forward(p, spec, e, taus).view(taus.shape[0] // e.shape[0], -1, spec.n_actions)
        if spec.n_heads > 0:
            q = qens_forward(p, spec, e)
        else:
            q = dist.mean(dim=0).unsqueeze(-1)
    return q, dist


def ids_scores(dist, q, lmbda, epsilon, rho_lower_bound, unsquish=None):
    """IDSActionSelector.generate_action_probs without random sampling (action_selectors.py:125-176):
    dict of the logged intermediates + the chosen action.  `unsquish`: the selector's unsquish function (:128-130)."""
    if unsquish is not None:
        dist, q = unsquish(dist), unsquish(q)
    mean, variance = q.mean(dim=-1), q.std(dim=-1)
    std = torch.sqrt(variance)
    regret = torch.max(mean + lmbda * std, dim=-1).values.view(-1, 1) - (mean - lmbda * std)
    var_z = dist.var(dim=0)
    rho = torch.clamp(var_z / (epsilon + var_z.mean(dim=-1).unsqueeze(-1)), min=rho_lower_bound)
    gain = torch.log(1 + variance / rho) + epsilon
    scores = torch.square(regret) / gain
    return dict(mean=mean, variance=variance, var_z=var_z, gain=gain, scores=scores, action=torch.argmin(scores, dim=-1))


def _head_keys(p, h):
    pre = f"q_function_model.q_heads.{h}."
    return [k for k in p.keys() if k.startswith(pre)]


def composite_losses(p, p_tgt, spec, batch, taus):
    """batch: dict(obs (B,10,10,C), next_obs, reward (B,), nonterminal (B,) bool, gamma (B,), action (B,))."""
    obs, nobs = batch["obs"], batch["next_obs"]
    R = batch["reward"].flatten()
    dg = batch["gamma"].flatten().float() * batch["nonterminal"].flatten().float()
    acts = batch["action"].flatten().long()
    e_cur = conv_embed(p, obs)
    with torch.no_grad():
        e_next = conv_embed(p_tgt if p_tgt is not None else p, nobs)
    dl = ql = td = None
    aux = {}
    if spec.use_iqn:
        dl = iqn_loss(p, p_tgt, spec, e_cur, e_next, acts, R, dg, taus)
    if spec.n_heads > 0:
        ql, aux["theil"] = qens_loss(p, p_tgt, spec, e_cur, e_next, acts, R, dg)
    if dl is not None and ql is not None:
        td = dl.detach() * 0.5 + ql.detach() * 0.5
    elif dl is not None:
        td = dl.detach()
    elif ql is not None:
        td = ql.abs().detach()
    return dl, ql, td, aux


class LearnerOracle:
    """Holds leaf parameter tensors + torch.optim.Adam; one call == Agent._update_without_cuda_graph."""

    def __init__(self, state_dict, spec, target_state_dict=None):
        self.spec = spec
        self.p = {k: v.detach().clone().float().requires_grad_(True) for k, v in state_dict.items()}
        self.p_tgt = None
        if target_state_dict is not None:
            self.p_tgt = {k: v.detach().clone().float() for k, v in target_state_dict.items()}
        self.opt = torch.optim.Adam(list(self.p.values()), lr=spec.lr, betas=(spec.beta1, spec.beta2),
                                    eps=spec.adam_eps)
        self.last = {}

    def grads_fp64(self, batch, per_weights, taus):
        """The same gradient evaluated in float64 at the current (fp32) parameters.  Where a ReLU input sits
        within fp32 rounding distance of zero, the fp32 autograd result and this one differ by that unit's
        whole contribution; a correct fp32 kernel may land on either side (tests accept both)."""
        p64 = {k: v.detach().double().requires_grad_(True) for k, v in self.p.items()}
        pt64 = None if self.p_tgt is None else {k: v.double() for k, v in self.p_tgt.items()}
        b64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in batch.items()}
        dl, ql, _, _ = composite_losses(p64, pt64, self.spec, b64, [t.double() for t in taus])
        w = per_weights.double() if torch.is_tensor(per_weights) else per_weights
        total = 0
        if dl is not None:
            total = total + (dl * w).mean()
        if ql is not None:
            total = total + (ql * w).mean()
        total.backward()
        return {k: (v.grad.detach() if v.grad is not None else torch.zeros_like(v)) for k, v in p64.items()}

    def update(self, batch, per_weights, taus, apply=True):
        dl, ql, td, aux = composite_losses(self.p, self.p_tgt, self.spec, batch, taus)
        total = 0
        if dl is not None:
            total = total + (dl * per_weights).mean()
        if ql is not None:
            total = total + (ql * per_weights).mean()
        self.opt.zero_grad()
        total.backward()
        grads = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v))
                 for k, v in self.p.items()}
        gnorm = torch.nn.utils.clip_grad_norm_(list(self.p.values()), self.spec.max_grad_norm)
        if apply:
            self.opt.step()
        self.last = dict(dl=None if dl is None else dl.detach(), ql=None if ql is None else ql.detach(),
                         td=td, total=total.detach(), grads=grads, grad_norm=gnorm.detach(), **aux)
        return td

    def sync_target(self):
        for k in self.p_tgt:
            self.p_tgt[k].copy_(self.p[k].detach())

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.p.items()}
