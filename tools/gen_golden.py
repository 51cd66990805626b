#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference (imported from /root/reference).

Build-container only: the reference does not exist on the GPU box, and nothing in tests/ or the
product imports it.  The fixtures hold DATA only — seeds, inputs, captured quantile samples and the
reference's outputs — never reference source.

    python tools/gen_golden.py            # writes tests/golden/{update_*,nstep_*}.npz

What is captured:
  update_<case>.npz  Agent.update() (non-graph path, prism/agents/agent.py:53-79) for N steps on
                     seeded synthetic MinAtar-shaped batches: inputs, taus (torch.rand patched to
                     record, order current -> next -> [target]), per-sample losses / td errors,
                     total loss, per-tensor grad L2 norms, global grad norm, per-tensor parameter
                     checksums after every step (+ full tensors for the small cases).
  nstep_chain.npz    TimestepBuffer._compute_n_step / _timesteps_to_batch
                     (prism/experience/timestep_buffer.py:79-238) on hand-built Timestep chains.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.dont_write_bytecode = True


def _import_reference():
    sys.path.insert(0, REF)
    # stand-ins for absent third-party modules the experience half imports at module scope
    td = types.ModuleType("tensordict")

    class TensorDict(dict):
        def __init__(self, d=None, batch_size=None, device=None):
            super().__init__(d or {})
            self.batch_size, self.device = batch_size, device

    td.TensorDict = TensorDict
    sys.modules["tensordict"] = td
    sys.modules["wandb"] = types.ModuleType("wandb")


def synth_batch(rng, B, C=4, A=6, p_term=0.05):
    obs = (rng.random((B, 1, 10, 10, C)) < 0.1).astype(np.float32)
    nobs = (rng.random((B, 1, 10, 10, C)) < 0.1).astype(np.float32)
    rew = rng.standard_normal((B, 1)).astype(np.float32)
    nonterm = rng.random((B, 1)) >= p_term
    g = np.full((B, 1), 0.99 ** 3, np.float32)
    sel = rng.random(B)
    g[sel < 0.05, 0] = np.float32(0.99 ** 2)
    g[sel < 0.025, 0] = np.float32(0.99)
    act = rng.integers(0, A, size=(B, 1)).astype(np.int64)
    w = (rng.random(B).astype(np.float32) * 0.9 + 0.1)
    return dict(obs=obs, next_obs=nobs, reward=rew, nonterminal=nonterm, gamma=g, action=act, w=w)


def to_ref_batch(b):
    t = torch.from_numpy
    return {"observation": t(b["obs"]), "next": {"observation": t(b["next_obs"]), "reward": t(b["reward"])},
            "nonterminal": t(b["nonterminal"]), "gamma": t(b["gamma"]), "action": t(b["action"])}


def tensor_stats(sd):
    names = list(sd.keys())
    s = np.array([float(sd[k].double().sum()) for k in names])
    l2 = np.array([float(sd[k].double().norm()) for k in names])
    return names, s, l2


CASES = {
    # name: (config overrides, B, steps, store_full_params)
    "iqn_small": (dict(use_ids=False, use_iqn=True, use_dqn=False, iqn_n_current_state_quantile_samples=4,
                       iqn_n_next_state_quantile_samples=4), 8, 3, True),
    "iqn_c3": (dict(use_ids=False, use_iqn=True, use_dqn=False), 256, 3, False),
    "dqn_c2": (dict(use_ids=False, use_iqn=False, use_dqn=True, use_layer_norm=False), 256, 3, True),
    "dqn_ln": (dict(use_ids=False, use_iqn=False, use_dqn=True, use_layer_norm=True), 32, 2, False),
    "dqn_target_c2": (dict(use_ids=False, use_iqn=False, use_dqn=True, use_layer_norm=False,
                           use_target_network=True), 64, 2, False),
    "full_c4": (dict(use_ids=True, use_iqn=True, use_target_network=True), 512, 2, False),
    "full_small": (dict(use_ids=True, use_iqn=True, use_target_network=True), 16, 2, False),
    "full_notarget": (dict(use_ids=True, use_iqn=True, use_target_network=False), 16, 2, False),
    "full_doubleq": (dict(use_ids=True, use_iqn=True, use_target_network=True, use_double_q_learning=True),
                     16, 2, False),
    "iqn_target": (dict(use_ids=False, use_iqn=True, use_target_network=True), 32, 2, False),
    "iqn_doubleq": (dict(use_ids=False, use_iqn=True, use_target_network=True, use_double_q_learning=True),
                    32, 2, False),
    "iqn_tau32": (dict(use_ids=False, use_iqn=True, iqn_n_current_state_quantile_samples=32,
                       iqn_n_next_state_quantile_samples=32), 16, 2, False),
    # ---- the ablation presets the reference ships (prism/config/{additive,subtractive}_ablation_base_config.py)
    # and the stages its experiment files derive from them (additive_ablation_experiment.py:31-162,
    # subtractive_ablation_experiment.py:30-52), at their own batch size (64), T = 32, width 256.
    # "base" picks the preset; the remaining keys are the stage's overrides.  "C" = observation channels
    # (MinAtar games: Breakout/Asterix 4, SpaceInvaders 6, Freeway 7, Seaquest 10).
    "abl_iqn": (dict(base="additive"), 64, 2, False),
    "abl_ln_notarget": (dict(base="additive", use_layer_norm=True, use_target_network=False), 64, 2, False),
    "abl_doubleq": (dict(base="additive", use_target_network=True, use_double_q_learning=True, C=7), 64, 2, False),
    "abl_ids": (dict(base="additive", use_ids=True, ids_n_q_head_model_layers=2, ids_n_q_heads=10,
                     ids_ensemble_variation_coef=0, ids_q_head_feature_dim=256), 64, 2, False),
    "abl_ids_var": (dict(base="additive", use_dqn=False, use_iqn=True, use_ids=True, ids_n_q_head_model_layers=2,
                         ids_n_q_heads=10, ids_q_head_feature_dim=256, ids_ensemble_variation_coef=1e-6, C=10),
                    64, 2, False),
    "abl_sub": (dict(base="subtractive", use_layer_norm=True, use_per=False, use_target_network=True,
                     target_update_period=4_000), 64, 2, False),
    # ---- the value-squish hooks of the TD target (loss_squish_fn_id; model_factory.py:16-23, iqn_model.py:141-148,
    # q_ensemble.py:77-82; the IDS selector scores unsquished estimates, action_selectors.py:128-130)
    "iqn_symlog": (dict(use_ids=False, use_iqn=True, use_dqn=False, loss_squish_fn_id="symlog"), 32, 2, False),
    "full_olf": (dict(use_ids=True, use_iqn=True, use_target_network=True, loss_squish_fn_id="obs_look_further"), 16, 2, False),
    "dqn_symlog": (dict(use_ids=False, use_iqn=False, use_dqn=True, use_layer_norm=False, loss_squish_fn_id="symlog"),
                   32, 2, False),
}


def gen_update_case(name, overrides, B, steps, store_full):
    from prism.config import Config, MINATAR_CONFIG, ADDITIVE_ABLATION_BASE_CONFIG, SUBTRACTIVE_ABLATION_BASE_CONFIG
    from prism.factory import agent_factory

    overrides = dict(overrides)
    base = overrides.get("base", "minatar")
    C = overrides.pop("C", 4)
    A = 6
    preset = {"minatar": MINATAR_CONFIG, "additive": ADDITIVE_ABLATION_BASE_CONFIG,
              "subtractive": SUBTRACTIVE_ABLATION_BASE_CONFIG}[base]
    cfg = Config(**preset.__dict__)
    cfg.device, cfg.use_cuda_graph, cfg.use_e_greedy = "cpu", False, False
    for k, v in overrides.items():
        if k != "base":
            setattr(cfg, k, v)
    torch.manual_seed(cfg.seed)
    with contextlib.redirect_stdout(io.StringIO()):
        agent = agent_factory.build_agent(cfg, (10, 10, C), A)
    out = {"seed": cfg.seed, "B": B, "steps": steps, "C": C, "A": A}
    ov = {k: v for k, v in overrides.items() if k != "base"}
    out["base"] = base
    out["overrides_keys"] = np.array(list(ov.keys()))
    out["overrides_vals"] = np.array([repr(v) for v in ov.values()])
    names, s0, l0 = tensor_stats(agent.model.state_dict())
    out["param_names"] = np.array(names)
    out["init_sum"], out["init_l2"] = s0, l0
    out["n_params"] = sum(p.numel() for p in agent.model.parameters())
    if store_full:
        for k, v in agent.model.state_dict().items():
            out["init/" + k] = v.numpy().copy()

    rng = np.random.default_rng(1000 + sum(map(ord, name)))
    real_rand = torch.rand
    for step in range(steps):
        b = synth_batch(rng, B, C, A)
        taus = []

        def rec(*a, **k):
            r = real_rand(*a, **k)
            taus.append(r.clone())
            return r

        torch.rand = rec
        try:
            td = agent.update(to_ref_batch(b), per_weights=torch.from_numpy(b["w"]))
        finally:
            torch.rand = real_rand
        pre = f"s{step}/"
        out[pre + "obs_bits"] = np.packbits(b["obs"].astype(np.uint8).reshape(-1))
        out[pre + "next_obs_bits"] = np.packbits(b["next_obs"].astype(np.uint8).reshape(-1))
        for k in ("reward", "nonterminal", "gamma", "action", "w"):
            out[pre + k] = b[k]
        out[pre + "n_taus"] = len(taus)
        for i, t in enumerate(taus):
            out[pre + f"tau{i}"] = t.numpy().reshape(-1)
        out[pre + "td"] = td.detach().numpy()
        if agent._static_distribution_loss is not None:
            out[pre + "dl"] = agent._static_distribution_loss.detach().numpy()
        if agent._static_q_loss is not None:
            out[pre + "ql"] = agent._static_q_loss.detach().numpy()
        out[pre + "total"] = float(agent._static_total_loss.detach())
        # grads left on the parameters are the CLIPPED ones; recover the global norm from the
        # reference's own arithmetic: clip multiplies by min(1, max/(norm+1e-6))
        gl2 = np.array([float(p.grad.double().norm()) if p.grad is not None else 0.0
                        for p in agent.model.parameters()])
        out[pre + "clipped_grad_l2"] = gl2
        n2, s, l2 = tensor_stats(agent.model.state_dict())
        out[pre + "post_sum"], out[pre + "post_l2"] = s, l2
        if agent.model.q_function_model is not None:
            out[pre + "theil"] = float(agent.model.q_function_model.theil.detach())
        if store_full and step == steps - 1:
            for k, v in agent.model.state_dict().items():
                out[pre + "post/" + k] = v.numpy().copy()
        if store_full and step == 0:
            for (k, p) in agent.model.named_parameters():
                out[pre + "clipped_grad/" + k] = p.grad.numpy().copy()
        if cfg.use_target_network and step == 0:
            agent.sync_target_model()       # exercise hard sync between steps
    # acting on the updated weights: Agent.forward's pieces (agent.py:31-41) for a handful of observations --
    # CompositeModel.forward(for_action=True), then the training-time selector's scores / choice
    n_act = 5
    act_obs = (rng.random((n_act, 10, 10, C)) < 0.1).astype(np.float32)
    torch.manual_seed(4242)
    with torch.no_grad():
        q, dist = agent.model(torch.from_numpy(act_obs), for_action=True)
    out["act/obs_bits"] = np.packbits(act_obs.astype(np.uint8).reshape(-1))
    out["act/n"] = n_act
    out["act/q"] = q.numpy()
    if cfg.use_iqn:
        T_act = cfg.iqn_quantile_samples_per_action
        torch.manual_seed(4242)
        out["act/tau"] = torch.rand([T_act * n_act, 1]).float().numpy().reshape(-1)      # the draw iqn_model.py:66-68 made
        out["act/dist"] = dist.numpy()
    sel = agent.action_selector
    with torch.no_grad():
        probs = sel.generate_action_probs(dist, q, for_log=True) if cfg.use_ids else sel.generate_action_probs(dist, q)
        out["act/action"] = sel.select_action(probs).numpy()
    out["act/probs"] = probs.numpy()
    if cfg.use_ids:
        for k, v in sel.loggables.items():
            out["act/ids/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, f"update_{name}.npz"), **out)
    print(f"update_{name}: P={out['n_params']} total[-1]={out[pre + 'total']:.6f} act={out['act/action'].tolist()}")


def gen_nstep():
    """Hand-built Timestep chains pushed through the reference's n-step + collate."""
    from prism.experience import Timestep, TimestepBuffer
    import weakref

    class FakeRB:
        _batch_size = 0

    rng = np.random.default_rng(7)
    n_step, gamma = 3, 0.99
    keep = []            # strong refs (links are weak)
    rows = []            # per stored timestep: dict of scalars
    env_streams = 3
    O = (10, 10, 4)
    next_id = [0]

    def new_ts():
        t = Timestep(id=next_id[0])
        next_id[0] += 1
        return t

    # interleave several env streams, as the collector does (experience_collector.py:94-120)
    current = []
    for e in range(env_streams):
        t = new_ts()
        t.obs = torch.from_numpy((rng.random(O) < 0.1).astype(np.float32))
        current.append(t)
    stored = []          # completed timesteps in insertion order
    steps_per_stream = [0] * env_streams
    for it in range(67):
        e = it % env_streams
        cur = current[e]
        steps_per_stream[e] += 1
        k = steps_per_stream[e]
        done = (k % 7 == 0)
        trunc = (not done) and (k % 5 == 0)
        cur.action = int(rng.integers(0, 6))
        cur.reward = float(np.float32(rng.standard_normal()))
        cur.done, cur.truncated = bool(done), bool(trunc)
        nxt = new_ts()
        nxt.obs = torch.from_numpy((rng.random(O) < 0.1).astype(np.float32))
        if trunc:
            tr = new_ts()
            tr.obs = torch.from_numpy((rng.random(O) < 0.1).astype(np.float32))
            tr.prev = weakref.ref(cur)
            cur.next = tr
        elif not done:
            nxt.prev = weakref.ref(cur)
            cur.next = weakref.ref(nxt)
        current[e] = nxt
        stored.append(cur)
        keep.append(cur)
    keep.extend(current)

    buf = TimestepBuffer(FakeRB(), frame_stack=1, device="cpu", n_step=n_step, gamma=gamma)
    N = len(stored)
    batch = buf._timesteps_to_batch(stored, N)
    id_to_row = {t.id: i for i, t in enumerate(stored)}
    out = {"n_step": n_step, "gamma": gamma, "N": N}
    out["obs"] = np.stack([t.obs.numpy() for t in stored])
    succ = np.zeros_like(out["obs"])
    link = np.full(N, -1, np.int32)
    has_next = np.zeros(N, bool)
    for i, t in enumerate(stored):
        nx = t.next
        if nx is None:
            continue
        node = nx if isinstance(nx, Timestep) else nx()
        has_next[i] = True
        succ[i] = node.obs.numpy()
        if not isinstance(nx, Timestep) and node.reward is not None and node.id in id_to_row:
            link[i] = id_to_row[node.id]
    out["succ_obs"], out["link"], out["has_next"] = succ, link, has_next
    out["reward"] = np.array([t.reward for t in stored], np.float64)
    out["done"] = np.array([t.done for t in stored])
    out["truncated"] = np.array([t.truncated for t in stored])
    out["action"] = np.array([t.action for t in stored], np.int64)
    out["exp_n_step_return"] = np.array([t.n_step_return for t in stored], np.float64)
    out["exp_n_step_gamma"] = np.array([t.n_step_gamma for t in stored], np.float64)
    out["exp_n_step_done"] = np.array([bool(t.n_step_done) for t in stored])
    out["exp_needs_n_step"] = np.array([bool(t.needs_n_step) for t in stored])
    out["exp_batch_obs"] = batch["observation"].numpy()
    out["exp_batch_next_obs"] = batch["next"]["observation"].numpy()
    out["exp_batch_reward"] = batch["next"]["reward"].numpy()
    out["exp_batch_nonterminal"] = batch["nonterminal"].numpy()
    out["exp_batch_gamma"] = batch["gamma"].numpy()
    out["exp_batch_action"] = batch["action"].numpy()
    np.savez_compressed(os.path.join(OUT, "nstep_chain.npz"), **out)
    print("nstep_chain:", N, "timesteps; needs_n_step:", int(out["exp_needs_n_step"].sum()),
          "terminal rows:", int((~out["exp_batch_nonterminal"]).sum()))

    # ---- the same chain through the reference's buffer checkpoint: TimestepBuffer.save -> timesteps.pkl, then its
    # own load and collate again (timestep_buffer.py:259-318).  The torchrl sampler / writer dumps are absent here:
    # stand-ins that write nothing.
    class Dumper:
        def dumps(self, path):
            pass

        def loads(self, path):
            pass

    rb = FakeRB()
    rb._storage = [[t] for t in stored]
    rb._sampler, rb._writer = Dumper(), Dumper()
    buf.buffer = rb
    ck = os.path.join(OUT, "ref_buffer")
    with contextlib.redirect_stdout(io.StringIO()):
        buf.save(ck)
        rb2 = FakeRB()
        rb2._storage = [None] * N
        rb2._sampler, rb2._writer = Dumper(), Dumper()
        buf2 = TimestepBuffer(rb2, frame_stack=1, device="cpu", n_step=n_step, gamma=gamma)
        buf2.load(ck)
    kept = [s_[0] for s_ in rb2._storage if s_ is not None]
    batch2 = buf2._timesteps_to_batch(kept, len(kept))
    np.savez_compressed(os.path.join(OUT, "ref_buffer_expected.npz"), n=len(kept), ids=np.array([t.id for t in kept]),
                        obs=batch2["observation"].numpy(), next_obs=batch2["next"]["observation"].numpy(),
                        reward=batch2["next"]["reward"].numpy(), nonterminal=batch2["nonterminal"].numpy(),
                        gamma=batch2["gamma"].numpy(), action=batch2["action"].numpy())
    print("ref_buffer: saved", N, "timesteps, the reference's own load keeps", len(kept))


def gen_checkpoint():
    """A checkpoint directory written by the reference's Agent.save (agent.py:179-203) after two updates of a small
    DQN agent with an epsilon-greedy selector, plus a state dict holding an IDS selector: files as the reference
    wrote them (tensors + pickled selector objects)."""
    import pickle
    import shutil
    from prism.config import Config, MINATAR_CONFIG
    from prism.factory import agent_factory
    from prism.agents.action_selectors import IDSActionSelector
    cfg = Config(**MINATAR_CONFIG.__dict__)
    cfg.device, cfg.use_cuda_graph = "cpu", False
    for k, v in dict(use_ids=False, use_iqn=False, use_dqn=True, use_layer_norm=False, use_e_greedy=True,
                     use_target_network=True).items():
        setattr(cfg, k, v)
    torch.manual_seed(cfg.seed)
    with contextlib.redirect_stdout(io.StringIO()):
        agent = agent_factory.build_agent(cfg, (10, 10, 4), 6)
    rng = np.random.default_rng(99)
    for _ in range(2):
        b = synth_batch(rng, 32)
        agent.update(to_ref_batch(b), per_weights=torch.from_numpy(b["w"]))
    agent.action_selector.epsilon.update(1234)
    ck = os.path.join(OUT, "ref_checkpoint")
    shutil.rmtree(ck, ignore_errors=True)
    agent.save(ck)
    names, s, l2 = tensor_stats(agent.model.state_dict())
    opt = agent.optimizer.state_dict()
    np.savez_compressed(os.path.join(OUT, "ref_checkpoint_expected.npz"), param_names=np.array(names), sum=s, l2=l2,
                        n_updates=agent.n_updates, max_grad_norm=agent.max_grad_norm, epsilon_step=1234,
                        adam_step=float(opt["state"][0]["step"]),
                        exp_avg_l2=np.array([float(opt["state"][i]["exp_avg"].double().norm()) for i in range(len(names))]))
    ids = IDSActionSelector(lmbda=0.1, random_sample=False, epsilon=1e-10, ids_rho_lower_bound=0.25, beta=0.8)
    with open(os.path.join(ck, "state_ids.pkl"), "wb") as f:
        pickle.dump({"action_selector": ids, "n_updates": 7}, f)
    print("ref_checkpoint:", sorted(os.listdir(os.path.join(ck, "agent"))))


def main():
    os.makedirs(OUT, exist_ok=True)
    _import_reference()
    which = sys.argv[1:] or (list(CASES) + ["nstep", "config", "checkpoint"])
    for name in which:
        if name == "nstep":
            gen_nstep()
        elif name == "checkpoint":
            gen_checkpoint()
        elif name == "config":
            gen_config_snapshot()
        else:
            gen_update_case(name, *CASES[name])




def gen_config_snapshot():
    """Field names / order / preset values of the reference's Config, as data."""
    import dataclasses
    import json
    from prism.config import (Config, DEFAULT_CONFIG, MINATAR_CONFIG, ADDITIVE_ABLATION_BASE_CONFIG,
                              SUBTRACTIVE_ABLATION_BASE_CONFIG)
    snap = {"fields": [f.name for f in dataclasses.fields(Config)],
            "DEFAULT_CONFIG": DEFAULT_CONFIG.__dict__, "MINATAR_CONFIG": MINATAR_CONFIG.__dict__,
            "ADDITIVE_ABLATION_BASE_CONFIG": ADDITIVE_ABLATION_BASE_CONFIG.__dict__,
            "SUBTRACTIVE_ABLATION_BASE_CONFIG": SUBTRACTIVE_ABLATION_BASE_CONFIG.__dict__}
    # every configuration the two experiment files generate, as {group name: fields that differ from the base}
    # (seed 0 of each group; seeds only change `seed` / names / checkpoint_dir)
    from prism.experiments.experiment_files.additive_ablation_experiment import AdditiveAblationExperiment
    from prism.experiments.experiment_files.subtractive_ablation_experiment import SubtractiveAblationExperiment
    skip = {"seed", "wandb_project_name", "wandb_group_name", "wandb_run_name", "checkpoint_dir"}
    for key, exp_cls, base in (("ADDITIVE_STAGES", AdditiveAblationExperiment, ADDITIVE_ABLATION_BASE_CONFIG),
                               ("SUBTRACTIVE_STAGES", SubtractiveAblationExperiment, SUBTRACTIVE_ABLATION_BASE_CONFIG)):
        stages = {}
        for cfg in exp_cls(num_seeds=1).configs:
            diff = {k: v for k, v in cfg.__dict__.items() if k not in skip and base.__dict__.get(k) != v}
            stages[cfg.wandb_group_name] = diff
        snap[key] = stages
    with open(os.path.join(OUT, "config_presets.json"), "w") as f:
        json.dump(snap, f, indent=1, sort_keys=True)
    print("config_presets:", len(snap["fields"]), "fields")


if __name__ == "__main__":
    main()
