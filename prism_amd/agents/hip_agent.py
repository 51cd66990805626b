"""``HipAgent`` — duck-types the reference ``Agent`` (``/root/reference/prism/agents/agent.py``)
with the TD update running in the hand-written gfx950 kernels behind ``prism_learner_fwd_bwd`` /
``prism_learner_clip_adam``.

Memory model: ONE flat fp32 parameter buffer per network in ``model.parameters()`` order
(SURVEY.md Appendix B); every ``nn.Parameter`` of the container modules is a view into it, so
``state_dict()`` / checkpoints interchange with the reference while the kernels and the RCCL
all-reduce see a single contiguous buffer.  Adam moments and the gradient are flat buffers too.

Data-parallel: ``torch.distributed`` all-reduce (RCCL over xGMI) of the flat gradient sits
between the two native calls; clip + Adam then run redundantly on every rank so replicas stay
bit-identical.
"""
import ctypes
import os

import numpy as np
import torch

from prism_amd import _native as N
from prism_amd import dist as pdist
from prism_amd.util import ref_pickle


class HipAdam:
    """Flat Adam state with a ``torch.optim.Adam``-compatible ``state_dict`` (the update itself is
    the fused kernel).  Mirrors the options of agent_factory.py:44-47."""

    def __init__(self, named_params, flat_params, lr, betas, eps):
        self.names = [n for n, _ in named_params]
        self.shapes = [tuple(p.shape) for _, p in named_params]
        self.numels = [p.numel() for _, p in named_params]
        self.flat_params = flat_params
        dev = flat_params.device
        self.exp_avg = torch.zeros_like(flat_params)
        self.exp_avg_sq = torch.zeros_like(flat_params)
        self.step_t = torch.zeros(1, dtype=torch.int64, device=dev)
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False,
                                  maximize=False, foreach=None, capturable=False, differentiable=False,
                                  fused=None, params=list(range(len(self.names))))]

    def zero_grad(self, set_to_none=True):
        pass

    def state_dict(self):
        state, off = {}, 0
        step = float(self.step_t.item())
        for i, (n, shp) in enumerate(zip(self.numels, self.shapes)):
            if step > 0:
                state[i] = {"step": torch.tensor(step), "exp_avg": self.exp_avg[off:off + n].view(shp).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(shp).clone()}
            off += n
        return {"state": state, "param_groups": [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd):
        off, step = 0, 0
        for i, (n, shp) in enumerate(zip(self.numels, self.shapes)):
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = int(float(st["step"]))
            off += n
        self.step_t.fill_(step)
        if sd.get("param_groups"):
            g = sd["param_groups"][0]
            for k in ("lr", "betas", "eps"):
                if k in g:
                    self.param_groups[0][k] = g[k]


def _flatten_into(model, device):
    """Move every parameter into one contiguous fp32 buffer (parameters() order) and re-point the
    parameters at views of it."""
    params = list(model.parameters())
    flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=device)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view(p.shape)
        off += n
    return flat


def _offsets(model):
    """state_dict key -> flat offset, mapped onto the C ABI's prism_param_offsets."""
    off, table = 0, {}
    for name, p in model.named_parameters():
        table[name] = off
        off += p.numel()
    o = N.ParamOffsets()
    for f, _ in N.ParamOffsets._fields_:
        setattr(o, f, -1)
    o.n_params = off
    g = table.get
    o.conv_w, o.conv_b = g("embedding_model.model.0.weight", -1), g("embedding_model.model.0.bias", -1)
    d = "distribution_model."
    o.phi_w, o.phi_b = g(d + "phi.0.weight", -1), g(d + "phi.0.bias", -1)
    if d + "model.model.1.weight" in table:          # LayerNorm on: [LN, Linear, ReLU] / [LN, Linear]
        o.iqn_ln1_g, o.iqn_ln1_b = g(d + "model.model.0.weight", -1), g(d + "model.model.0.bias", -1)
        o.iqn_w1, o.iqn_b1 = g(d + "model.model.1.weight", -1), g(d + "model.model.1.bias", -1)
        o.iqn_ln2_g = g(d + "embedding_to_quantile_layer.0.weight", -1)
        o.iqn_ln2_b = g(d + "embedding_to_quantile_layer.0.bias", -1)
        o.iqn_w2 = g(d + "embedding_to_quantile_layer.1.weight", -1)
        o.iqn_b2 = g(d + "embedding_to_quantile_layer.1.bias", -1)
    else:                                            # use_layer_norm=False: [Linear, ReLU] / Linear (iqn_model.py:42-46)
        o.iqn_w1, o.iqn_b1 = g(d + "model.model.0.weight", -1), g(d + "model.model.0.bias", -1)
        o.iqn_w2 = g(d + "embedding_to_quantile_layer.weight", -1)
        o.iqn_b2 = g(d + "embedding_to_quantile_layer.bias", -1)
    heads = sorted({int(k.split(".")[2]) for k in table if k.startswith("q_function_model.q_heads.")})
    if heads:
        pre = "q_function_model.q_heads.0."
        names = [k[len(pre):] for k in table if k.startswith(pre)]
        o.head_base = min(table[pre + n] for n in names)
        if len(heads) > 1:
            o.head_stride = table["q_function_model.q_heads.1." + names[0]] - table[pre + names[0]]
        else:
            o.head_stride = 0
        rel = {n: table[pre + n] - o.head_base for n in names}
        # FFNN-style head: model.{0:LN,1:Linear,3:LN,4:Linear} | model.{0:Linear} | model.{0:LN,1:Linear}
        order = sorted(rel, key=lambda n: rel[n])
        lin = [n[:-len(".weight")] for n in order if n.endswith(".weight") and (n[:-len(".weight")] + ".bias") in rel
               and len(dict(model.named_parameters())[pre + n].shape) == 2]
        lns = [n[:-len(".weight")] for n in order if n.endswith(".weight") and n[:-len(".weight")] not in lin]
        if len(lin) >= 1:
            o.h_w1, o.h_b1 = rel[lin[0] + ".weight"], rel[lin[0] + ".bias"]
        if len(lin) >= 2:
            o.h_w2, o.h_b2 = rel[lin[1] + ".weight"], rel[lin[1] + ".bias"]
        if len(lns) >= 1:
            o.h_ln1_g, o.h_ln1_b = rel[lns[0] + ".weight"], rel[lns[0] + ".bias"]
        if len(lns) >= 2:
            o.h_ln2_g, o.h_ln2_b = rel[lns[1] + ".weight"], rel[lns[1] + ".bias"]
    return o


def model_dims(config, in_channels, n_actions):
    d = N.ModelDims()
    d.in_channels, d.n_actions, d.embed_dim = int(in_channels), int(n_actions), 1024
    d.use_iqn = int(bool(config.use_iqn))
    d.n_basis, d.iqn_layers = config.iqn_n_basis_elements, config.iqn_quantile_model_layers
    d.iqn_width = config.iqn_quantile_model_feature_dim
    d.n_tau, d.n_tau_next = config.iqn_n_current_state_quantile_samples, config.iqn_n_next_state_quantile_samples
    d.use_layer_norm = int(bool(config.use_layer_norm))
    if config.use_ids:
        d.n_heads, d.head_layers, d.head_width = (config.ids_n_q_heads, config.ids_n_q_head_model_layers,
                                                  config.ids_q_head_feature_dim)
        d.theil_coef = config.ids_ensemble_variation_coef
    elif config.use_dqn:
        d.n_heads, d.head_layers, d.head_width = 1, config.dqn_n_model_layers, config.dqn_n_model_feature_dim
        d.theil_coef = 0.0
    d.has_target = int(bool(config.use_target_network))
    d.double_q = int(bool(config.use_double_q_learning))
    d.propagate_grad = int((config.ids_allow_distributional_gradients and config.use_ids) or not config.use_ids)
    d.huber_k, d.dist_loss_weight = config.iqn_huber_loss_kappa, config.distributional_loss_weight
    d.q_loss_weight = config.q_loss_weight
    from prism_amd.agents.squish_functions import SQUISH_IDS
    d.squish_fn = SQUISH_IDS.get(str(config.loss_squish_fn_id), 0)       # (unknown ids parse to no squish, model_factory.py:16-23)
    return d


class _Actions(torch.Tensor):
    """What ``HipAgent.forward`` returns when it ran from its hipGraph: a device tensor like the reference's
    (agent.py:41), whose ``.cpu()`` -- what every collector / evaluator calls next (experience_collector.py:127) -- hands
    out the copy the graph's own device-to-host node already made into pinned memory: one event wait instead of a second
    copy and a device synchronisation."""

    @staticmethod
    def wrap(dev, host, event):
        t = dev.as_subclass(_Actions)
        t._host, t._event = host, event
        return t

    def cpu(self, *a, **k):
        host = getattr(self, "_host", None)
        if host is None:
            return torch.Tensor.cpu(self.as_subclass(torch.Tensor), *a, **k)
        self._event.synchronize()
        return host.clone()

    def numpy(self, *a, **k):
        return self.cpu().numpy()

    def tolist(self):
        return self.cpu().tolist()

    def item(self):
        return self.cpu().item()


class HipAgent:
    def __init__(self, model, action_selector, eval_action_selector, target_model, config, in_channels,
                 n_actions, process_group=None):
        if not str(config.device).startswith("cuda"):
            raise N.NativeLibraryError("HipAgent needs a GPU device; prism_amd has no CPU fallback")
        N.lib()
        self.device = torch.device(config.device)
        self.config = config
        self.model, self.target_model = model, target_model
        self.action_selector, self.eval_action_selector = action_selector, eval_action_selector
        self.max_grad_norm = config.max_grad_norm
        self.use_cuda_graph = False
        self.n_updates = 0
        self._is_eval = False
        self._static_batch = None
        self._static_distribution_loss = None
        self._static_q_loss = None
        self._static_total_loss = None

        self.flat = _flatten_into(model, self.device)
        self.flat_target = _flatten_into(target_model, self.device) if target_model is not None else None
        self.grads = torch.zeros_like(self.flat)
        self.optimizer = HipAdam(list(model.named_parameters()), self.flat, config.learning_rate,
                                 (config.adam_beta1, config.adam_beta2), config.adam_epsilon)
        self.dims = model_dims(config, in_channels, n_actions)
        self.off = _offsets(model)
        self.tau_rng = getattr(config, "tau_rng", "philox")
        self.overlap_writeback = bool(getattr(config, "overlap_writeback", True))
        self.seed = int(config.seed)
        self._draw_offset = 0      # tau counters consumed by update() (host-issued offsets)
        self._fused_tau = 0        # ... and by fused steps (device counter, mirrored: one counter space for both paths)
        self._act_draws = 0        # acting draws: a Philox stream of their own (key 3), counted separately
        self._act_raw = None
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                         and getattr(config, "data_parallel", False)):
            self.world = torch.distributed.get_world_size(process_group)
        self._B = None
        # the all-reduce captured INSIDE the step graph: opt-in (config.collective_in_graph / PRISM_COLLECTIVE_IN_GRAPH=1)
        # until a multi-GPU run of it is on record; the default is two graphs around an eagerly launched collective
        self.collective_in_graph = bool(getattr(config, "collective_in_graph",
                                                os.environ.get("PRISM_COLLECTIVE_IN_GRAPH", "0") == "1"))
        # the fused tail's grid barrier needs the launch resident at once, which the library can only prove for a GPU this
        # process has to itself: PRISM_SHARED_GPU=1 (several learners per GPU, a normal MinAtar set-up) switches it off
        self.fuse_tail = bool(getattr(config, "fuse_tail", os.environ.get("PRISM_SHARED_GPU", "0") != "1"))
        self._status = pdist.StatusWords()        # pinned host mirror of the sticky status bits: polled every step
        # Agent.forward replayed from one hipGraph per (number of observations, selector): H2D of the observations, embed,
        # forward tiles, selector kernel, D2H of the actions (config.act_graph = False: the eager launches)
        self.act_graph = bool(getattr(config, "act_graph", True))
        self._param_epoch, self._act_packed_at = 0, None
        # (the reference's async evaluator loads weights straight into agent.model, async_agent_evaluator.py:29)
        self.model.register_load_state_dict_post_hook(lambda module, incompatible: self._params_replaced())
        self._capture_error = None
        # the step's one exchange: "rccl" (torch.distributed all_reduce, the default) or "direct" (two shots over peer-mapped
        # buffers, prism_amd/dist.py DirectAllReduce)
        self.collective = str(getattr(config, "collective", "rccl"))
        self._direct = None
        if self.world > 1 and self.collective == "direct":
            self._direct = pdist.DirectAllReduce(self.grads, self.pg, status=self._status)
            if not self._direct.use_flags:
                self.collective_in_graph = False          # host-side barriers cannot be captured
        self.model.train()

    def _allreduce(self):
        if self._direct is not None:
            return self._direct.allreduce(self.grads)
        return pdist.allreduce_grads(self.grads, self.pg)

    # ------------------------------------------------------------------ descriptor
    def _prepare(self, B):
        L = N.lib()
        rc = L.prism_learner_supported(ctypes.byref(self.dims), B)
        if rc != N.PRISM_OK:
            from prism_amd.factory.model_factory import UnsupportedConfig
            raise UnsupportedConfig(
                f"prism_amd HIP learner does not cover this configuration at batch {B} "
                f"(use_iqn={self.dims.use_iqn}, n_heads={self.dims.n_heads}, layer_norm={self.dims.use_layer_norm}, "
                f"T={self.dims.n_tau}); there is no eager fallback")
        dev = self.device
        ws_bytes = L.prism_learner_workspace_bytes(ctypes.byref(self.dims), B)
        self.workspace = torch.zeros((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
        self.out_dl = torch.zeros(B, device=dev)
        self.out_ql = torch.zeros(B, device=dev)
        self.out_td = torch.zeros(B, device=dev)
        self.scalars = torch.zeros(8, device=dev)
        maxT = max(self.dims.n_tau, self.dims.n_tau_next)
        self.tau_out = torch.zeros(3, maxT * B, device=dev)
        self.ones_w = None
        d = N.LearnerDesc()
        d.dims, d.off, d.batch = self.dims, self.off, B
        d.gemm_mode = N.GEMM_MODES[str(getattr(self.config, "gemm_mode", "auto"))]      # "fp32" | "bf16x3" | "auto"
        d.params, d.grads = self.flat.data_ptr(), self.grads.data_ptr()
        d.target_params = self.flat_target.data_ptr() if self.flat_target is not None else None
        d.adam_m, d.adam_v = self.optimizer.exp_avg.data_ptr(), self.optimizer.exp_avg_sq.data_ptr()
        d.adam_step = self.optimizer.step_t.data_ptr()
        d.tau_out = self.tau_out.data_ptr()
        d.out_dist_loss, d.out_q_loss = self.out_dl.data_ptr(), self.out_ql.data_ptr()
        d.out_td, d.out_scalars = self.out_td.data_ptr(), self.scalars.data_ptr()
        d.workspace, d.workspace_bytes = self.workspace.data_ptr(), ws_bytes
        d.host_status = self._status.data_ptr()
        if self._direct is not None:      # a collective that gives up poisons THIS workspace's status word (no update applied)
            self._direct._desc.poison = self.workspace.data_ptr() + 4 * N.WS_STATUS_WORD
        self.rng_counters = torch.zeros(3, dtype=torch.int64, device=dev)     # {PER draws, tau draws, acting draws}
        self._act_graphs, self._act_ptrs, self._act_dev_draws, self._act_packed_at = {}, {}, 0, None
        self._desc, self._B = d, B
        self._graphs = {}

    def _set_hyper(self):
        g, h = self.optimizer.param_groups[0], self._desc.hyper
        h.lr, h.beta1, h.beta2, h.eps = g["lr"], g["betas"][0], g["betas"][1], g["eps"]
        h.max_grad_norm, h.grad_scale = self.max_grad_norm, 1.0 / self.world

    # ------------------------------------------------------------------ reference API
    def update(self, batch, per_weights=1, taus=None):
        """One TD update (agent.py:43-79).  ``taus``: optional explicit quantile samples in the
        reference's draw order (parity tests); otherwise drawn in-kernel (Philox) or with
        ``torch.rand`` when ``config.tau_rng == 'torch'``."""
        self.train()
        obs = batch["observation"]
        B = int(obs.shape[0])
        if self._B != B:
            self._prepare(B)
        d = self._desc
        nobs, rew = batch["next"]["observation"], batch["next"]["reward"]
        act = batch["action"]
        if act.dim() == 2 and act.shape[-1] != 1:
            act = act.argmax(dim=-1)
        nt = batch["nonterminal"]
        if nt.dtype not in (torch.bool, torch.uint8):      # the kernels read one byte per sample
            nt = nt != 0
        keep = [obs.float().contiguous(), nobs.float().contiguous(), rew.float().contiguous(),
                nt.contiguous(), batch["gamma"].float().contiguous(),
                act.long().contiguous()]
        d.obs, d.next_obs, d.reward, d.nonterminal, d.gamma, d.action = [t.data_ptr() for t in keep]
        if torch.is_tensor(per_weights):
            w = per_weights.to(self.device, torch.float32).contiguous()
            keep.append(w)
            d.per_weights = w.data_ptr()
        else:
            if float(per_weights) != 1.0:
                w = torch.full((B,), float(per_weights), device=self.device)
                keep.append(w)
                d.per_weights = w.data_ptr()
            else:
                d.per_weights = None
        if taus is None and self.tau_rng == "torch" and self.dims.use_iqn:
            T, Tn = self.dims.n_tau, self.dims.n_tau_next
            taus = [torch.rand([T * B, 1], device=self.device)]
            if not self.dims.has_target or self.dims.double_q:
                taus.append(torch.rand([Tn * B, 1], device=self.device))
            if self.dims.has_target:
                taus.append(torch.rand([Tn * B, 1], device=self.device))
        d.tau_cur = d.tau_next_online = d.tau_next_target = None
        if taus is not None and len(taus) > 0 and self.dims.use_iqn:
            ts = [t.to(self.device, torch.float32).reshape(-1).contiguous() for t in taus]
            keep.extend(ts)
            it = iter(ts)
            d.tau_cur = next(it).data_ptr()
            if not self.dims.has_target or self.dims.double_q:
                d.tau_next_online = next(it).data_ptr()
            if self.dims.has_target:
                d.tau_next_target = next(it).data_ptr()
        d.seed = self.seed
        d.offset = self._draw_offset + self._fused_tau
        d.rng_counters, d.embed_done, d.fused_replay, d.fuse_tail = None, 0, None, 0
        self._draw_offset += 3 * max(self.dims.n_tau, self.dims.n_tau_next) * B
        self._set_hyper()
        L = N.lib()
        with torch.cuda.device(self.device):
            N.check(L.prism_learner_fwd_bwd(ctypes.byref(d), N.current_stream_handle()), "prism_learner_fwd_bwd")
            if self.world > 1:
                self._allreduce()
            N.check(L.prism_learner_clip_adam(ctypes.byref(d), N.current_stream_handle()), "prism_learner_clip_adam")
        self._keep = keep
        self._static_total_loss = self.scalars[0]
        self._static_distribution_loss = self.out_dl if self.dims.use_iqn else None
        self._static_q_loss = self.out_ql if self.dims.n_heads > 0 else None
        self.n_updates += 1
        return self.out_td

    # ------------------------------------------------------------------ fused hot path
    def _bind_fused(self, buf):
        """Point the descriptor at the buffer's static batch / IS weights (stable device addresses)."""
        B = int(buf._index.shape[0])
        if self._B != B:
            self._prepare(B)
        d = self._desc
        d.obs, d.next_obs = buf._obs.data_ptr(), buf._next_obs.data_ptr()
        d.reward, d.nonterminal = buf._reward.data_ptr(), buf._nonterminal.data_ptr()
        d.gamma, d.action = buf._gamma.data_ptr(), buf._action.data_ptr()
        d.per_weights = buf._weight.data_ptr() if buf.use_per else None
        d.tau_cur = d.tau_next_online = d.tau_next_target = None
        d.seed = self.seed
        self._set_hyper()
        return d

    def _launch_fused(self, buf, d, part="all"):
        L, st = N.lib(), N.current_stream_handle
        rp = ctypes.byref(buf._desc)
        smp = buf.buffer._sampler
        if buf.use_per and self.overlap_writeback:
            d.fused_replay = ctypes.cast(ctypes.pointer(buf._desc), ctypes.c_void_p)
            d.fused_index, d.fused_alpha, d.fused_eps = buf._index.data_ptr(), smp._alpha, smp._eps
        else:
            d.fused_replay = None
        # one GPU, whole step in one go: the gradient reduction rides in prism_step_back's launch (grid barrier)
        d.fuse_tail = int(part == "all" and self.world == 1 and self.fuse_tail)
        if part in ("all", "front"):
            d.embed_done = 1
            N.check(L.prism_step_front(ctypes.byref(d), rp, buf._size, None, buf.seed, buf._draws,
                                       buf.buffer._sampler._beta, N.ptr(buf._index), N.ptr(buf._weight), st()),
                    "prism_step_front")
            N.check(L.prism_learner_fwd_bwd(ctypes.byref(d), st()), "prism_learner_fwd_bwd")
            d.embed_done = 0
        if part == "all" and self.world > 1:
            self._allreduce()
        if part in ("all", "back"):
            N.check(L.prism_step_back(ctypes.byref(d), rp, N.ptr(buf._index), smp._alpha, smp._eps, st()),
                    "prism_step_back")

    def step_fused(self, buf, eager=False, use_graph=True):
        """Sample + update + priority writeback as four launches on one GPU, five around an all-reduce (prism_step_front, fwd_bwd,
        prism_step_back); replayed from a hipGraph when the replay is full (its size is baked into
        the captured launches) and the RNG counters live on the device."""
        if self.tau_rng != "philox" or buf.mass_rng != "philox":
            raise RuntimeError("fused step draws its random numbers in-kernel (per_mass_rng/tau_rng = 'philox')")
        if buf._index is None or buf._index.shape[0] != buf.buffer._batch_size:
            buf._alloc_batch(buf.buffer._batch_size)
        self.poll_status()
        d = self._bind_fused(buf)
        d.rng_counters = self.rng_counters.data_ptr()
        d.offset = self._draw_offset               # device counter + what update() has consumed: never the same draw twice
        graphable = use_graph and not eager and buf._size == buf.capacity
        with torch.cuda.device(self.device):
            if not graphable:
                self._launch_fused(buf, d)
            else:
                # a captured graph bakes every by-value launch argument: hyper-parameters, seeds, beta / alpha, offsets
                h, smp = d.hyper, buf.buffer._sampler
                key = (id(buf), buf._size, self.world, h.lr, h.beta1, h.beta2, h.eps, h.max_grad_norm, h.grad_scale,
                       self.seed, buf.seed, smp._beta, smp._alpha, smp._eps, self._draw_offset, buf._draws,
                       self.overlap_writeback)
                g = self._graphs.get(key)
                if g is None:
                    self._graphs.clear()                        # arguments changed: older captures are stale
                    self._launch_fused(buf, d)                  # first full-buffer step runs eagerly
                    self._graphs[key] = "warm"
                elif g == "warm":
                    torch.cuda.current_stream().synchronize()
                    self._graphs[key] = self._capture(buf, d)   # capture, then replay once = this step
                else:
                    self._replay(g, buf, d)
        self._fused_tau += 3 * max(self.dims.n_tau, self.dims.n_tau_next) * self._B
        buf._fused_draws += self._B
        self._static_total_loss = self.scalars[0]
        self._static_distribution_loss = self.out_dl if self.dims.use_iqn else None
        self._static_q_loss = self.out_ql if self.dims.n_heads > 0 else None
        self.n_updates += 1
        return self.out_td

    def _capture(self, buf, d):
        if self.world == 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._launch_fused(buf, d)
            g.replay()
            return (g,)
        if self.collective_in_graph:
            # data parallel, one graph: front + forward/backward, the RCCL all-reduce of the flat gradient, clip + Adam
            # + writeback -- one launch per step, no host round trip around the collective.  The collective must have run
            # once eagerly (communicator set up outside the capture): the warm step before this one did that.  Backends
            # whose all-reduce cannot be captured (gloo: host staging) raise here; the step then falls back to two
            # graphs around an eagerly launched collective, for good.
            g, err = torch.cuda.CUDAGraph(), None
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._launch_fused(buf, d)
            except Exception as e:          # noqa: BLE001 -- any capture failure selects the split form
                err = repr(e)
                torch.cuda.synchronize()
            # every rank must run the same form: agree on the outcome (MIN over ranks of "my capture worked")
            ok = torch.tensor([0 if err else 1], device=self.device, dtype=torch.int32)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=self.pg)
            if int(ok.item()) == 1:
                g.replay()
                return (g,)
            self.collective_in_graph = False
            self._capture_error = err or "capture failed on another rank"
            import warnings
            warnings.warn(f"prism_amd: the all-reduce could not be captured into the step graph ({self._capture_error}); "
                          "running two graphs around an eager collective")
            # nothing of the step has run yet (a capture only records): fall through
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            self._launch_fused(buf, d, "front")
        g1.replay()
        self._allreduce()
        with torch.cuda.graph(g2):
            self._launch_fused(buf, d, "back")
        g2.replay()
        return (g1, g2)

    def _replay(self, g, buf, d):
        if len(g) == 1:
            g[0].replay()
        else:
            g[0].replay()
            self._allreduce()
            g[1].replay()

    # ------------------------------------------------------------------ acting
    @torch.no_grad()
    def act_estimates(self, obs, taus=None):
        """``(q_estimates, return_distribution)`` as ``CompositeModel.forward(x, for_action=True)`` hands them to the
        action selector (composite_model.py:51-70): ``(n, A, heads)`` and ``(n_quantile_samples_per_action, n, A)``,
        computed by ``prism_act_forward`` on the learner's flat parameters (forward tiles for the IQN rows and the
        two-layer heads, one workgroup per observation for the single-Linear DQN head).  ``taus``: explicit quantile
        samples ``[T*n, 1]`` in the reference's order (parity tests)."""
        from prism_amd.agents.modules import _as_tensor
        obs = _as_tensor(obs, self.device).contiguous()
        dm = self.dims
        q_tiles = dm.n_heads > 0 and dm.head_layers == 2
        q_rows = dm.n_heads > 0 and dm.head_layers == 1
        if self._B is None:
            self._prepare(int(self.config.batch_size))
        n, B, A = int(obs.shape[0]), self._B, dm.n_actions
        cap = (B // 16) * 16 if q_tiles else B
        if cap < 1:
            raise ValueError(f"acting through the Q-head tiles needs a learner batch of at least 16 (batch {B})")
        if n > cap:          # more observations than the learner's workspace holds at once: in pieces
            parts = [self.act_estimates(obs[i:i + cap], None if taus is None else
                                        taus.view(-1, n)[:, i:i + cap].reshape(-1, 1)) for i in range(0, n, cap)]
            q = torch.cat([p[0] for p in parts], dim=0) if parts[0][0] is not None else None
            dist = torch.cat([p[1] for p in parts], dim=1) if parts[0][1] is not None else None
            self._act_raw = None          # (describes the last piece only: nobody may select from it)
            return q, dist
        T = int(self.model.distribution_model.n_quantile_samples_per_action) if dm.use_iqn else 0
        n_pad = (n + 15) // 16 * 16
        z = torch.empty(((n * T + 15) // 16 * 16, A), device=self.device) if dm.use_iqn else None
        qb = torch.empty((dm.n_heads, n_pad, A), device=self.device) if (q_tiles or q_rows) else None
        if taus is None and dm.use_iqn and self.tau_rng == "torch":
            taus = torch.rand([T * n, 1], device=self.device)
        tau = None if taus is None else taus.to(self.device, torch.float32).reshape(-1).contiguous()
        with torch.cuda.device(self.device):
            N.check(N.lib().prism_act_forward(ctypes.byref(self._desc), N.ptr(obs), n, T, N.ptr(tau) if tau is not None else None,
                                              self.seed, self._act_draws, N.ptr(z) if z is not None else None,
                                              N.ptr(qb) if qb is not None else None, N.current_stream_handle()),
                    "prism_act_forward")
        self._act_draws += T * n
        self._act_raw = (z, qb, n, n_pad, T)
        dist = z[:n * T].view(n, T, A).permute(1, 0, 2) if z is not None else None
        if qb is not None:
            q = qb[:, :n].permute(1, 2, 0)
        else:
            # models without Q heads (composite_model.py:66-68); the native selectors below take this mean from z themselves
            q = dist.mean(dim=0).unsqueeze(-1)
        return q, dist

    @torch.no_grad()
    def forward(self, obs):
        """Agent.forward (agent.py:31-41).  Deterministic information-directed sampling (``prism_ids_select``), greedy and
        epsilon-greedy selection (``prism_greedy_select``; the coin and the random actions come from the selector's host
        generator exactly as in action_selectors.py:35-45) run natively on the estimate buffers; only sampled IDS
        (``ids_use_random_samples``: ``torch.multinomial`` on torch's generator) reads them with the selector's torch code."""
        from prism_amd.agents.action_selectors import EGreedyActionSelector, GreedyActionSelector, IDSActionSelector
        from prism_amd.agents.modules import _as_tensor
        sel = self.eval_action_selector if self._is_eval else self.action_selector
        obs_in = obs
        n, A = int(np.shape(obs)[0]), self.dims.n_actions
        if type(sel) is EGreedyActionSelector:
            if sel.rng.uniform(0, 1) < sel.epsilon.update(n):       # one coin for the whole call (action_selectors.py:38-41)
                # the reference has ALREADY run the model by the time it tosses the coin (agent.py:33 -> :36): the forward's
                # quantile samples are drawn whether or not the estimates are used.  The estimates are not needed here, the
                # draws are: consume them, so that every later greedy action sees the stream it would see in the reference
                if self.dims.use_iqn:
                    T = int(self.model.distribution_model.n_quantile_samples_per_action)
                    if self.tau_rng == "torch":
                        torch.rand([T * n, 1], device=self.device)
                    else:
                        self._act_draws += T * n
                return torch.as_tensor(sel.rng.randint(A, size=(n,)), dtype=torch.long).to(self.device)
            sel = sel.greedy
        if self._B is None:
            self._prepare(int(self.config.batch_size))
        cap = (self._B // 16) * 16 if (self.dims.n_heads > 0 and self.dims.head_layers == 2) else self._B
        if n <= cap and self.act_graph and self.tau_rng == "philox":
            out = self._forward_graph(obs_in, n, sel)
            if out is not None:
                return out
        obs = _as_tensor(obs, self.device).contiguous()
        if n > cap:          # in pieces: every piece is selected from its own estimate buffers
            return torch.cat([self._forward_piece(obs[i:i + cap], sel) for i in range(0, n, cap)])
        return self._forward_piece(obs, sel)

    def _forward_graph(self, obs_in, n, sel):
        """One ``Agent.forward`` as ONE hipGraph launch: [observations host -> device] -> prism_act_forward -> selector kernel
        -> [actions device -> pinned host].  Keyed by everything a capture bakes: the number of observations, where they come
        from (the caller's device tensor by address -- the reference's collector reuses one inference buffer,
        experience_collector.py:77-78 -- or the pinned staging block for host arrays), the selector and its constants.  The
        quantile draws come from the device counter ``rng_counters[2]`` (include/prism_hip.h), which mirrors ``_act_draws``.
        Returns None for selectors that have no native kernel (the caller runs the eager path)."""
        from prism_amd.agents.action_selectors import GreedyActionSelector, IDSActionSelector
        dm = self.dims
        A = dm.n_actions
        shape = (n, 10, 10, dm.in_channels)
        q_tiles = dm.n_heads > 0 and dm.head_layers == 2
        q_rows = dm.n_heads > 0 and dm.head_layers == 1
        if type(sel) is IDSActionSelector:
            from prism_amd.agents.squish_functions import unsquish_id
            usq = unsquish_id(sel.unsquish_function)
            if sel.random_sample or usq is None or not (dm.use_iqn and (q_tiles or q_rows)):
                return None
            skey = ("ids", float(sel.lmbda), float(sel.epsilon), float(sel.ids_rho_lower_bound), usq)
        elif type(sel) is GreedyActionSelector:
            skey = ("greedy",)
        else:
            return None
        if torch.is_tensor(obs_in) and obs_in.is_cuda and not (obs_in.dtype == torch.float32 and obs_in.is_contiguous()
                                                               and obs_in.numel() == n * 100 * dm.in_channels):
            obs_in = obs_in.float().contiguous().view(shape)          # (a device tensor in another layout: one eager copy)
        on_host = not (torch.is_tensor(obs_in) and obs_in.is_cuda)
        mode = "host" if on_host else "ptr"
        if mode == "ptr":
            # a device tensor is read in place while its address repeats (the collector's one inference buffer); a caller
            # that hands over a fresh tensor every time gets one device-to-device copy into a staging block instead of a
            # capture per address
            seen = self._act_ptrs.setdefault((n, skey), set())
            seen.add(obs_in.data_ptr())
            if len(seen) > 3:
                mode = "dev"
        # the packed weight copies in the workspace follow the parameters: rebuilt by the first acting call after every update
        current = self._act_packed_at == self._param_version()
        key = (n, skey, mode, obs_in.data_ptr() if mode == "ptr" else None, self.seed, self._B, current)
        st = self._act_graphs.get(key)
        if st is None:
            if len(self._act_graphs) > 32:
                self._act_graphs.clear()
            T = int(self.model.distribution_model.n_quantile_samples_per_action) if dm.use_iqn else 0
            n_pad = (n + 15) // 16 * 16
            dev = self.device
            st = dict(T=T, n_pad=n_pad, calls=0, g=None,
                      pin_in=[torch.zeros(shape, dtype=torch.float32).pin_memory() for _ in range(4)] if on_host else None,
                      obs=obs_in if mode == "ptr" else (None if on_host else torch.empty(shape, dtype=torch.float32, device=dev)),
                      z=torch.empty(((n * T + 15) // 16 * 16, A), device=dev) if dm.use_iqn else None,
                      qb=torch.empty((dm.n_heads, n_pad, A), device=dev) if (q_tiles or q_rows) else None,
                      scores=torch.empty((n, A), device=dev) if skey[0] == "ids" else None,
                      act=[torch.empty(n, dtype=torch.int64, device=dev) for _ in range(4)],
                      pin_out=[torch.zeros(n, dtype=torch.int64).pin_memory() for _ in range(4)],
                      ev=[torch.cuda.Event() for _ in range(4)])
            if st["pin_in"] is not None:
                st["pin_np"] = [t.numpy() for t in st["pin_in"]]
            self._act_graphs[key] = st
        if self._act_dev_draws != self._act_draws:          # eager calls / an explore branch moved the host count on
            self.rng_counters[2] = self._act_draws
            self._act_dev_draws = self._act_draws
        k = st["calls"] & 3
        if on_host:
            if st["calls"] >= 4:
                st["ev"][k].synchronize()          # the launch that last read this staging slot (four calls ago) is through
            if torch.is_tensor(obs_in):
                st["pin_in"][k].copy_(obs_in.reshape(shape))
            else:
                st["pin_np"][k][...] = np.asarray(obs_in, dtype=np.float32).reshape(shape)
        elif mode == "dev":
            st["obs"].copy_(obs_in.view(shape))

        def launch(slot):
            # (host arrays: the embed kernel reads the pinned staging block in place -- 400 C bytes per observation over the
            # host link cost less than a copy node in front of it; the selector writes the actions to device AND pinned host)
            L, d, T = N.lib(), self._desc, st["T"]
            keep = d.rng_counters, d.act_flags
            d.rng_counters = self.rng_counters.data_ptr()
            d.act_flags = N.ACT_WEIGHTS_CURRENT if current else 0
            try:
                N.check(L.prism_act_forward(ctypes.byref(d), N.ptr(st["pin_in"][slot] if on_host else st["obs"]), n, T, None, self.seed, 0,
                                            N.ptr(st["z"]), N.ptr(st["qb"]), N.current_stream_handle()), "prism_act_forward")
            finally:
                d.rng_counters, d.act_flags = keep
            if skey[0] == "ids":
                N.check(L.prism_ids_select(N.ptr(st["z"]), N.ptr(st["qb"]), n, st["n_pad"], T, A, dm.n_heads, skey[1], skey[2],
                                           skey[3], skey[4], N.ptr(st["scores"]), None, N.ptr(st["act"][slot]), N.ptr(st["pin_out"][slot]),
                                           N.current_stream_handle()), "prism_ids_select")
            else:
                N.check(L.prism_greedy_select(N.ptr(st["z"]), N.ptr(st["qb"]), n, st["n_pad"], T, A, dm.n_heads,
                                              N.ptr(st["act"][slot]), None, N.ptr(st["pin_out"][slot]),
                                              N.current_stream_handle()), "prism_greedy_select")

        with torch.cuda.device(self.device):
            if st["g"] is None and st["calls"] == 0:
                launch(k)                                   # first call of this shape: eager (also warms the library up)
            else:
                if st["g"] is None:
                    # second call: capture one graph per output slot (a slot is a baked address); a capture only records,
                    # this call's work is the replay below
                    torch.cuda.current_stream().synchronize()
                    st["g"] = []
                    for slot in range(4):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g):
                            launch(slot)
                        st["g"].append(g)
                st["g"][k].replay()
            st["ev"][k].record()
        self._act_packed_at = self._param_version()
        st["calls"] += 1
        self._act_draws += st["T"] * n
        self._act_dev_draws = self._act_draws
        self._act_raw = (st["z"], st["qb"], n, st["n_pad"], st["T"])
        self._act_scores = st["scores"]
        return _Actions.wrap(st["act"][k], st["pin_out"][k], st["ev"][k])

    def _forward_piece(self, obs, sel):
        from prism_amd.agents.action_selectors import GreedyActionSelector, IDSActionSelector
        self._act_raw = None
        q, dist = self.act_estimates(obs)
        z, qb, n, n_pad, T = self._act_raw
        A = self.dims.n_actions
        action = torch.empty(n, dtype=torch.int64, device=self.device)
        from prism_amd.agents.squish_functions import unsquish_id
        usq = unsquish_id(sel.unsquish_function) if type(sel) is IDSActionSelector else None
        if (type(sel) is IDSActionSelector and not sel.random_sample and usq is not None
                and z is not None and qb is not None):
            scores = torch.empty((n, A), device=self.device)
            with torch.cuda.device(self.device):
                N.check(N.lib().prism_ids_select(N.ptr(z), N.ptr(qb), n, n_pad, T, A, self.dims.n_heads, float(sel.lmbda),
                                                 float(sel.epsilon), float(sel.ids_rho_lower_bound), usq, N.ptr(scores), None,
                                                 N.ptr(action), None, N.current_stream_handle()), "prism_ids_select")
            self._act_scores = scores
            return action
        if type(sel) is GreedyActionSelector:
            with torch.cuda.device(self.device):
                N.check(N.lib().prism_greedy_select(N.ptr(z), N.ptr(qb), n, n_pad, T, A, self.dims.n_heads, N.ptr(action),
                                                    None, None, N.current_stream_handle()), "prism_greedy_select")
            return action
        return sel.select_action(sel.generate_action_probs(dist, q))

    def _params_replaced(self):
        self._param_epoch += 1

    def _param_version(self):
        """Changes whenever the online parameters may have: every update, checkpoint load, deserialisation."""
        return (self.n_updates, self._param_epoch)

    def _raise_status(self, bits):
        self._status.clear()
        if self._B is not None:
            self.workspace.view(torch.int32)[N.WS_STATUS_WORD] = 0
        self._graphs = {}
        if bits & N.WS_STATUS_COLLECTIVE_TIMEOUT:
            raise pdist.CollectiveTimeout("prism_amd: the direct all-reduce gave up on a peer; the step that hit it applied no "
                                          "update and the replicas may have diverged -- restore a checkpoint")
        self.fuse_tail = False
        raise RuntimeError("prism_amd: a fused-tail grid barrier timed out (is the GPU shared with another process?); "
                           "the last steps are incomplete -- restore a checkpoint; fuse_tail is now off")

    def poll_status(self):
        """Raise if a kernel has raised a sticky status bit since the last look: an abandoned fused-tail grid barrier or a
        collective that gave up on a peer.  Reads the pinned host mirror of the bits (``prism_learner_desc.host_status``):
        no device synchronisation, so every ``step_fused`` / ``Learner.step`` calls it -- a failed step surfaces one step
        late at most, not at the next ``log()``."""
        bits = self._status.bits()
        if bits:
            self._raise_status(bits)

    def check_status(self):
        """``poll_status`` plus the device's own words (one D2H sync): the workspace status word and, data parallel with the
        direct collective, the sticky slot of the flag array.  An abandoned grid barrier only happens when the launch was not
        resident at once after all -- other processes' kernels on the same GPU -- and leaves the step that hit it half
        applied; the agent switches the fused tail off for what follows."""
        self.poll_status()
        if self._B is not None:
            bits = int(self.workspace.view(torch.int32)[N.WS_STATUS_WORD].item())
            if bits & (N.WS_STATUS_BARRIER_TIMEOUT | N.WS_STATUS_COLLECTIVE_TIMEOUT):
                self._raise_status(bits)
        if self._direct is not None:
            self._direct.check_status()

    def _target_changed(self):
        """The target parameters were written: its stream-packed copies in the workspace are stale (word 2 of the
        workspace, include/prism_hip.h)."""
        if self._B is not None:
            self.workspace.view(torch.int32)[2] = 0

    @torch.no_grad()
    def sync_target_model(self):
        with torch.cuda.device(self.device):
            N.check(N.lib().prism_sync_target(N.ptr(self.flat_target), N.ptr(self.flat), self.flat.numel(),
                                              N.current_stream_handle()), "prism_sync_target")
            self._target_changed()

    def set_static_batch(self, batch):
        self._static_batch = batch

    def get_static_batch(self):
        return self._static_batch

    def eval(self):
        self.model.eval()
        self._is_eval = True

    def train(self):
        self.model.train()
        self._is_eval = False

    def serialize_model(self):
        return self.flat.tolist() if not list(self.model.buffers()) else \
            [x for v in self.model.state_dict().values() for x in v.flatten().tolist()]

    def deserialize_model(self, values):
        self.flat.copy_(torch.as_tensor(values, dtype=torch.float32))
        self._param_epoch += 1

    def save(self, directory):
        """File layout of agent.py:179-203 (reference checkpoints interchange)."""
        self.check_status()
        path = os.path.join(directory, "agent")
        os.makedirs(path, exist_ok=True)
        torch.save(self.model.state_dict(), os.path.join(path, "model.pt"))
        torch.save(self.optimizer.state_dict(), os.path.join(path, "optimizer.pt"))
        if self.target_model is not None:
            torch.save(self.target_model.state_dict(), os.path.join(path, "target_model.pt"))
        state = {"action_selector": self.action_selector, "n_updates": self.n_updates,
                 "eval_action_selector": self.eval_action_selector, "max_grad_norm": self.max_grad_norm,
                 "use_cuda_graph": self.use_cuda_graph}
        with open(os.path.join(path, "state.pkl"), "wb") as f:
            ref_pickle.dump(state, f)          # class paths as the reference names them: it can load this file

    def load(self, directory):
        path = os.path.join(directory, "agent")
        # load_state_dict copies INTO the existing parameters, i.e. into the flat buffer views
        # (tensor-only files: weights_only refuses anything that would run code on load)
        self.model.load_state_dict(torch.load(os.path.join(path, "model.pt"), map_location=self.device, weights_only=True))
        self.optimizer.load_state_dict(torch.load(os.path.join(path, "optimizer.pt"), map_location=self.device,
                                                  weights_only=True))
        if self.target_model is not None:
            self.target_model.load_state_dict(torch.load(os.path.join(path, "target_model.pt"),
                                                         map_location=self.device, weights_only=True))
        with open(os.path.join(path, "state.pkl"), "rb") as f:
            state = ref_pickle.load(f)         # reference-written files name prism.agents.action_selectors.*
        self._graphs = {}                      # captured graphs bake the hyper-parameters restored here
        self._param_epoch += 1
        self._target_changed()
        self.action_selector = state["action_selector"]
        self.eval_action_selector = state["eval_action_selector"]
        self.max_grad_norm = state["max_grad_norm"]
        self.n_updates = state["n_updates"]
        self.train()

    @torch.no_grad()
    def log(self, logger):
        self.check_status()
        logger.log_data(data=float(self._static_total_loss), group_name="Report/Losses", var_name="Total Loss")
        if self._static_distribution_loss is not None:
            logger.log_data(data=float(self._static_distribution_loss.mean()), group_name="Report/Losses",
                            var_name="Distribution Loss")
        if self._static_q_loss is not None:
            logger.log_data(data=float(self._static_q_loss.mean()), group_name="Report/Losses", var_name="Q Loss")
        if getattr(logger, "holdout_data", None) is not None:
            idx = np.random.randint(0, logger.holdout_data["observation"].shape[0])
            obs = logger.holdout_data["observation"][idx]
            if obs.shape[0] != 1:
                obs = obs.unsqueeze(0)
            q, dist = self.model(obs)
            self.action_selector.log(logger, dist, q)
        self.model.log(logger)
