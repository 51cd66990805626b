"""HBM-resident replay ring with prioritized sum/min trees — duck-types the reference's
``TimestepBuffer`` (``/root/reference/prism/experience/timestep_buffer.py:10-77``) and the torchrl
buffer it wraps (``prism/factory/exp_buffer_factory.py:22-33``).

Data layout (all device memory, allocated once, sized for 288 GB of HBM3E):
    obs, succ_obs  fp32 [capacity, O]     reward fp32 [capacity]      action int32 [capacity]
    flags uint8 [capacity]                link/back int32 [capacity]
    tree fp32 [2 * tree_capacity][2]  ({sum, min} per node; sum_tree / min_tree are views)
Python ``Timestep`` objects are consumed at ``extend()`` and not kept.  ``sample()`` runs two
kernels (tree descent + IS weights; n-step walk + row gather) and returns the reference's static
batch layout without any host loop or H2D copy.
"""
import ctypes
import os
import pickle
import weakref

import numpy as np
import torch

from prism_amd import _native as N


class Batch(dict):
    """Nested dict of device tensors with the bits of TensorDict the learner touches
    (``learner.py:71`` ``.clone()``, ``agent.py:105`` ``.device``)."""

    def __init__(self, d=None, batch_size=None, device=None):
        super().__init__(d or {})
        self.batch_size, self.device = batch_size, device

    def clone(self):
        out = Batch({}, self.batch_size, self.device)
        for k, v in self.items():
            out[k] = v.clone()
        return out


class _SamplerShim:
    """``buffer.buffer._sampler`` — the learner writes ``_beta`` every step (learner.py:107)."""

    def __init__(self, alpha, beta, eps=1e-8):
        self._alpha, self._beta, self._eps = float(alpha), float(beta), float(eps)


class _WriterShim:
    def __init__(self):
        self._cursor = 0


class _RingShim:
    """``buffer.buffer`` — the attribute chain the reference reaches through."""

    def __init__(self, owner, batch_size, alpha, beta):
        self._owner = owner
        self._batch_size = batch_size
        self._sampler = _SamplerShim(alpha, beta)
        self._writer = _WriterShim()
        self._storage = owner          # len(buffer.buffer._storage)

    def __len__(self):
        return len(self._owner)


class HipReplayBuffer:
    STAGE_ROWS = 1024

    def __init__(self, capacity, batch_size, device="cuda:0", frame_stack=1, n_step=3, gamma=0.99,
                 use_per=True, alpha=0.5, beta=0.5, mass_rng="philox", seed=123, strict=False):
        if frame_stack != 1:
            raise NotImplementedError("prism_amd replay: frame_stack_size > 1 is not implemented")
        if not str(device).startswith("cuda"):
            raise N.NativeLibraryError("HipReplayBuffer needs a GPU device; there is no CPU fallback")
        if not 1 <= n_step <= N.PRISM_MAX_NSTEP:
            raise ValueError("n_step out of range")
        N.lib()
        self.device = torch.device(device)
        self.capacity = int(capacity)
        self.tree_capacity = 1
        while self.tree_capacity <= self.capacity:
            self.tree_capacity <<= 1
        self.frame_stack, self.n_step, self.gamma = frame_stack, int(n_step), float(gamma)
        self.gammas = [gamma ** i for i in range(n_step + 1)]
        self.use_per = bool(use_per)
        self.mass_rng, self.seed, self.strict = mass_rng, int(seed), bool(strict)
        self.buffer = _RingShim(self, batch_size, alpha, beta)
        self._size = 0
        self._draws = 0            # Philox counters consumed by sample() (host-issued offsets)
        self._fused_draws = 0      # ... and by fused steps (device counter; mirrored here so the two never overlap)
        self._obs_shape = None
        self._desc = None
        self._batch = None
        self._pending = {}        # successor Timestep.id -> (slot, id) of the stored predecessor
        self._slot_id = np.full(self.capacity, -1, np.int64)
        self._n_staged = 0
        self._index = None

    # ------------------------------------------------------------------ allocation
    def _allocate(self, obs_shape):
        dev, cap = self.device, self.capacity
        self._obs_shape = tuple(int(s) for s in obs_shape)
        O = int(np.prod(self._obs_shape))
        self.obs_elems = O
        self.obs = torch.zeros(cap, O, dtype=torch.float32, device=dev)
        self.succ_obs = torch.zeros(cap, O, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(cap, dtype=torch.float32, device=dev)
        self.action = torch.zeros(cap, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(cap, dtype=torch.uint8, device=dev)
        self.link = torch.full((cap,), -1, dtype=torch.int32, device=dev)
        self.back = torch.full((cap,), -1, dtype=torch.int32, device=dev)
        if self.use_per:
            # {sum, min} of a node side by side: one 16-byte load yields both children of a node
            self.tree = torch.zeros(2 * self.tree_capacity, 2, dtype=torch.float32, device=dev)
            self.sum_tree, self.min_tree = self.tree[:, 0], self.tree[:, 1]      # strided views
        else:
            self.tree = self.sum_tree = self.min_tree = None
        self.per_state = torch.zeros(4, dtype=torch.float32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        d = N.ReplayDesc()
        d.capacity, d.tree_capacity, d.obs_elems, d.n_step = cap, self.tree_capacity, O, self.n_step
        for name in ("obs", "succ_obs", "reward", "action", "flags", "link", "back", "tree",
                     "per_state", "status"):
            t = getattr(self, name)
            setattr(d, name, t.data_ptr() if t is not None else None)
        for i, g in enumerate(self.gammas):
            d.gammas[i] = g
        self._desc = d
        with torch.cuda.device(dev):
            N.check(N.lib().prism_replay_init(ctypes.byref(d), N.current_stream_handle()), "prism_replay_init")
        # staging (pinned host -> device) for extend()
        S = self.STAGE_ROWS
        pin = dict(pin_memory=True)
        # two staging sets: extend() fills one while the H2D copies + insert kernels of the other are in
        # flight; a set is only waited for (its event) when it comes round again
        # The five per-row scalars live side by side in ONE pinned block (and one device block): a flush is three
        # host-to-device copies -- that block whole (17 bytes a row), the used observation rows, the used successor rows --
        # instead of seven.  extend() writes through NumPy views of the pinned memory (a torch scalar store costs 2 us).
        def _stage(device=None):
            kw = dict(device=device) if device is not None else pin
            small = torch.zeros(17 * S, dtype=torch.uint8, **kw)      # slots | reward | action | prev (4 bytes each) | flags (1)
            w = lambda k: small[4 * S * k:4 * S * (k + 1)]
            st = dict(small=small, slots=w(0).view(torch.int32), reward=w(1).view(torch.float32),
                      action=w(2).view(torch.int32), prev=w(3).view(torch.int32), flags=small[16 * S:],
                      obs=torch.zeros(S, O, **kw), succ=torch.zeros(S, O, **kw))
            return st
        self._stages = [_stage(), _stage()]
        self._stage_np = [{k: v.numpy() for k, v in st.items()} for st in self._stages]
        self._stage_dev = [_stage(dev), _stage(dev)]
        self._stage_ev = [None, None]
        self._cur_stage = 0
        self._h, self._d, self._hn = self._stages[0], self._stage_dev[0], self._stage_np[0]
        self._alloc_batch(self.buffer._batch_size)

    def _alloc_batch(self, B):
        dev, fs = self.device, self.frame_stack
        shp = (B, fs) + self._obs_shape
        batch = Batch({
            "observation": torch.zeros(shp, dtype=torch.float32, device=dev),
            "next": Batch({"observation": torch.zeros(shp, dtype=torch.float32, device=dev),
                           "reward": torch.zeros(B, 1, dtype=torch.float32, device=dev)}, B, dev),
            "nonterminal": torch.zeros(B, 1, dtype=torch.bool, device=dev),
            "gamma": torch.ones(B, 1, dtype=torch.float32, device=dev),
            "action": torch.zeros(B, 1, dtype=torch.long, device=dev)}, B, dev)
        self.set_static_batch(batch)
        self._index = torch.zeros(B, dtype=torch.int64, device=dev)
        self._weight = torch.ones(B, dtype=torch.float32, device=dev)
        self._mass = torch.zeros(B, dtype=torch.float32, device=dev)
        self._q2 = torch.zeros(2, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ reference API
    def __len__(self):
        return self._size

    def extend(self, timestep):
        """TimestepBuffer.extend (timestep_buffer.py:32-33): one completed timestep."""
        if self._desc is None:
            self._allocate(tuple(timestep.obs.shape))
        if self._n_staged == self.STAGE_ROWS:
            self.flush()
        w = self.buffer._writer
        s = w._cursor
        w._cursor = (s + 1) % self.capacity
        self._size = min(self._size + 1, self.capacity)
        self._slot_id[s] = timestep.id

        prev_slot = -1
        rec = self._pending.pop(timestep.id, None)
        if rec is not None and self._slot_id[rec[0]] == rec[1]:
            prev_slot = rec[0]
        nxt = timestep.next
        node = None
        if nxt is not None:
            node = nxt() if isinstance(nxt, weakref.ReferenceType) else nxt
        flags = (N.FLAG_DONE if timestep.done else 0) | (N.FLAG_TRUNC if timestep.truncated else 0) | \
                (N.FLAG_HAS_NEXT if node is not None else 0)
        if node is not None and isinstance(nxt, weakref.ReferenceType) and not timestep.truncated:
            self._pending[node.id] = (s, timestep.id)

        i, h = self._n_staged, self._hn
        h["slots"][i] = s
        h["obs"][i] = np.asarray(timestep.obs, dtype=np.float32).reshape(-1)
        if node is not None:
            h["succ"][i] = np.asarray(node.obs, dtype=np.float32).reshape(-1)
        else:
            h["succ"][i] = 0.0
        h["reward"][i] = timestep.reward
        h["action"][i] = timestep.action
        h["flags"][i] = flags
        h["prev"][i] = prev_slot
        self._n_staged += 1
        return s

    def flush(self):
        """Push staged rows into the HBM ring (one H2D batch + two small kernels)."""
        n = self._n_staged
        if n == 0:
            return
        self._d["small"].copy_(self._h["small"], non_blocking=True)
        self._d["obs"][:n].copy_(self._h["obs"][:n], non_blocking=True)
        self._d["succ"][:n].copy_(self._h["succ"][:n], non_blocking=True)
        d, smp = self._d, self.buffer._sampler
        with torch.cuda.device(self.device):
            N.check(N.lib().prism_replay_insert(
                ctypes.byref(self._desc), n, N.ptr(d["slots"]), N.ptr(d["obs"]), N.ptr(d["succ"]),
                N.ptr(d["reward"]), N.ptr(d["action"]), N.ptr(d["flags"]), N.ptr(d["prev"]),
                smp._alpha, smp._eps, N.current_stream_handle()), "prism_replay_insert")
            ev = torch.cuda.Event()
            ev.record()
        # the other staging set takes the next rows; it was submitted a whole batch ago
        self._stage_ev[self._cur_stage] = ev
        self._cur_stage ^= 1
        self._h, self._d = self._stages[self._cur_stage], self._stage_dev[self._cur_stage]
        self._hn = self._stage_np[self._cur_stage]
        if self._stage_ev[self._cur_stage] is not None:
            self._stage_ev[self._cur_stage].synchronize()
            self._stage_ev[self._cur_stage] = None
        self._n_staged = 0

    @torch.no_grad()
    def sample(self, batch_size=None, return_info=False):
        if self._size == 0 and self._n_staged == 0:
            raise RuntimeError("Cannot sample from an empty storage.")
        self.flush()
        B = self.buffer._batch_size if batch_size is None else int(batch_size)
        if self._index is None or self._index.shape[0] != B:
            self._alloc_batch(B)
        L, dsc, st = N.lib(), ctypes.byref(self._desc), N.current_stream_handle
        with torch.cuda.device(self.device):
            if self.use_per:
                mass = None
                if self.mass_rng == "numpy":
                    N.check(L.prism_per_query(dsc, self._size, N.ptr(self._q2), st()), "prism_per_query")
                    p_sum, p_min = self._q2.tolist()
                    if p_sum <= 0 or p_min <= 0:
                        raise RuntimeError("non-positive p_sum / p_min")
                    m = np.random.uniform(0.0, p_sum, size=B).astype(np.float32)
                    self._mass.copy_(torch.from_numpy(m))
                    mass = self._mass
                N.check(L.prism_per_sample(dsc, self._size, B, N.ptr(mass), self.seed, self._draws + self._fused_draws,
                                           self.buffer._sampler._beta, N.ptr(self._index), N.ptr(self._weight),
                                           st()), "prism_per_sample")
            else:
                N.check(L.prism_uniform_sample(self._size, B, self.seed, self._draws + self._fused_draws,
                                               N.ptr(self._index), st()),
                        "prism_uniform_sample")
            self._draws += B
            N.check(L.prism_replay_gather(dsc, N.ptr(self._index), B, N.ptr(self._obs), N.ptr(self._next_obs),
                                          N.ptr(self._reward), N.ptr(self._nonterminal), N.ptr(self._gamma),
                                          N.ptr(self._action), st()), "prism_replay_gather")
        if self.strict:
            self.check_status()
        if return_info:
            info = {"index": self._index}
            if self.use_per:
                info["_weight"] = self._weight
            return self._batch, info
        return self._batch

    @torch.no_grad()
    def gather(self, indices):
        """The static batch for given slots: n-step return + collate (timestep_buffer.py:79-238) without sampling."""
        self.flush()
        idx = torch.as_tensor(indices).to(self.device, torch.int64).reshape(-1).contiguous()
        B = int(idx.numel())
        if self._index is None or self._index.shape[0] != B:
            self._alloc_batch(B)
        self._index.copy_(idx)
        with torch.cuda.device(self.device):
            N.check(N.lib().prism_replay_gather(ctypes.byref(self._desc), N.ptr(self._index), B, N.ptr(self._obs),
                                                N.ptr(self._next_obs), N.ptr(self._reward), N.ptr(self._nonterminal),
                                                N.ptr(self._gamma), N.ptr(self._action), N.current_stream_handle()),
                    "prism_replay_gather")
        return self._batch

    def check_status(self):
        """Raise what torchrl would have raised at sample time (costs one D2H sync)."""
        bits = int(self.status.item())
        if bits & (N.STATUS_NONPOSITIVE_PSUM | N.STATUS_NONPOSITIVE_PMIN):
            raise RuntimeError("non-positive p_sum / p_min in the priority trees")

    def update_priority(self, indices, priorities, take_abs=False):
        if not self.use_per:
            return
        idx = torch.as_tensor(indices).to(self.device, torch.int64).reshape(-1).contiguous()
        pr = torch.as_tensor(priorities).detach().to(self.device, torch.float32).reshape(-1)
        if pr.numel() == 1 and idx.numel() > 1:
            pr = pr.expand(idx.numel())
        pr = pr.contiguous()
        smp = self.buffer._sampler
        with torch.cuda.device(self.device):
            N.check(N.lib().prism_per_update(ctypes.byref(self._desc), N.ptr(idx), N.ptr(pr), idx.numel(),
                                             smp._alpha, smp._eps, int(take_abs), N.current_stream_handle()),
                    "prism_per_update")

    def set_static_batch(self, batch):
        self._batch = batch
        self._obs = batch["observation"]
        self._next_obs = batch["next"]["observation"]
        self._reward = batch["next"]["reward"]
        self._nonterminal = batch["nonterminal"]
        self._gamma = batch["gamma"]
        self._action = batch["action"]

    def get_static_batch(self):
        return self._batch

    def empty(self):
        self._size = 0
        self._n_staged = 0
        self.buffer._writer._cursor = 0
        self._pending.clear()
        self._slot_id[:] = -1
        if self._desc is not None:
            with torch.cuda.device(self.device):
                N.check(N.lib().prism_replay_init(ctypes.byref(self._desc), N.current_stream_handle()),
                        "prism_replay_init")

    # ------------------------------------------------------------------ bulk fill (bench / restore)
    def load_arrays(self, obs, succ_obs, reward, action, flags, link, priorities=None, n_sampleable=None):
        """Fill the first n slots from device/host arrays and rebuild the trees from the given leaf
        values (already (p+eps)**alpha).  Used for synthetic pre-fill and restore.  ``n_sampleable`` < n: only the
        first rows are stored items (sampled, counted by len()); the rest only serve as link targets."""
        n = int(obs.shape[0])
        ns = n if n_sampleable is None else int(n_sampleable)
        if self._desc is None:
            self._allocate(tuple(obs.shape[1:]))
        O = self.obs_elems
        self.obs[:n].copy_(torch.as_tensor(obs).reshape(n, O))
        self.succ_obs[:n].copy_(torch.as_tensor(succ_obs).reshape(n, O))
        self.reward[:n].copy_(torch.as_tensor(reward))
        self.action[:n].copy_(torch.as_tensor(action))
        self.flags[:n].copy_(torch.as_tensor(flags))
        lk = torch.as_tensor(link).to(self.device, torch.int32)
        self.link[:n].copy_(lk)
        self.back.fill_(-1)
        valid = lk >= 0
        self.back[lk[valid].long()] = torch.arange(n, device=self.device, dtype=torch.int32)[valid]
        self._size = ns
        self.buffer._writer._cursor = ns % self.capacity
        self._slot_id[:n] = np.arange(n)
        if self.use_per:
            tc = self.tree_capacity
            p = torch.ones(ns, device=self.device) if priorities is None else torch.as_tensor(priorities)[:ns]
            self.sum_tree[tc:tc + ns].copy_(p)
            self.min_tree[tc:tc + ns].copy_(p)
            with torch.cuda.device(self.device):
                N.check(N.lib().prism_per_rebuild(ctypes.byref(self._desc), N.current_stream_handle()),
                        "prism_per_rebuild")

    def save(self, path):
        """``TimestepBuffer.save`` (timestep_buffer.py:259-296): ``experience_buffer/timesteps.pkl`` in the reference's
        format (ref_format.py).  The reference also writes torchrl's sampler / writer dumps there -- third-party formats
        that cannot be pinned; priorities, running maximum and the ring cursor go to ``hip_sampler.pt``."""
        from prism_amd.experience import ref_format
        self.flush()
        d = os.path.join(path, "experience_buffer")
        os.makedirs(d, exist_ok=True)
        n = self._size
        flat = ref_format.serialize_ring(self.obs[:n].cpu().numpy(), self.succ_obs[:n].cpu().numpy(),
                                         self.reward[:n].cpu().numpy(), self.action[:n].cpu().numpy(),
                                         self.flags[:n].cpu().numpy(), self.link[:n].cpu().numpy(),
                                         self.back[:n].cpu().numpy(), self._slot_id[:n], self._obs_shape)
        with open(os.path.join(d, "timesteps.pkl"), "wb") as f:
            pickle.dump(flat, f)
        state = dict(size=n, cursor=self.buffer._writer._cursor, per_state=self.per_state.cpu())
        if self.use_per:
            tc = self.tree_capacity
            state["leaves"] = self.sum_tree[tc:tc + n].cpu()
        torch.save(state, os.path.join(d, "hip_sampler.pt"))

    def load(self, path):
        """``TimestepBuffer.load`` (timestep_buffer.py:298-318) of a file written here or by the reference; without
        ``hip_sampler.pt`` (a reference-written directory) every slot starts at the default priority."""
        from prism_amd.experience import ref_format
        d = os.path.join(path, "experience_buffer")
        from prism_amd.util import ref_pickle
        with open(os.path.join(d, "timesteps.pkl"), "rb") as f:
            flat = ref_pickle.load_plain(f)        # a flat list of numbers and booleans: no class may be named
        r = ref_format.ring_from_timesteps(flat)
        n, n_all = int(r["n_kept"]), min(int(r["obs"].shape[0]), self.capacity)
        if n > self.capacity:
            raise ValueError(f"the file holds {n} complete timesteps, this buffer only {self.capacity} "
                             "(experience_replay_capacity): load it into a buffer at least as large")
        sp = os.path.join(d, "hip_sampler.pt")
        st = torch.load(sp, weights_only=True) if os.path.exists(sp) else None
        leaves = None
        if st is not None and "leaves" in st and int(st["size"]) == n:
            leaves = st["leaves"]
        elif self.use_per:
            smp = self.buffer._sampler
            leaves = torch.full((n,), float(np.float32(np.float32(1.0 + smp._eps) ** np.float32(smp._alpha))))
        link = np.where(r["link"][:n_all] < n_all, r["link"][:n_all], -1)
        self.load_arrays(r["obs"][:n_all], r["succ_obs"][:n_all], r["reward"][:n_all], r["action"][:n_all], r["flags"][:n_all],
                         link, leaves, n_sampleable=n)
        self._slot_id[:n_all] = r["ids"][:n_all]
        if st is not None:
            self.per_state.copy_(st["per_state"])
        self.buffer._writer._cursor = int(st["cursor"]) if st is not None else n % self.capacity
