"""ORACLE — test infrastructure only (see ``oracle/per_oracle.c`` header; PARITY UNPINNED for the
segment tree, pinned for n-step/collate).

ctypes wrapper over ``oracle/_build/libprism_oracle.so`` plus a slot-indexed replay model that
mirrors what the reference keeps as linked ``Timestep`` objects
(/root/reference/prism/experience/timestep.py:12-28, timestep_buffer.py:32-33,198-238).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libprism_oracle.so")

FLAG_DONE, FLAG_TRUNC, FLAG_HAS_NEXT = 1, 2, 4


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "per_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
        L.oracle_tree_create.restype = vp
        L.oracle_tree_create.argtypes = [i64, ctypes.c_int]
        L.oracle_tree_destroy.argtypes = [vp]
        L.oracle_tree_capacity.restype = i64
        L.oracle_tree_capacity.argtypes = [vp]
        L.oracle_tree_values.restype = ctypes.POINTER(ctypes.c_float)
        L.oracle_tree_values.argtypes = [vp]
        L.oracle_tree_update.argtypes = [vp, i64, f32]
        L.oracle_tree_update_batch.argtypes = [vp, vp, vp, i64]
        L.oracle_tree_get.restype = f32
        L.oracle_tree_get.argtypes = [vp, i64]
        L.oracle_tree_query.restype = f32
        L.oracle_tree_query.argtypes = [vp, i64, i64]
        L.oracle_tree_scan_lower_bound.restype = i64
        L.oracle_tree_scan_lower_bound.argtypes = [vp, f32]
        L.oracle_per_sample.argtypes = [vp, vp, i64, vp, i64, f32, vp, vp, vp]
        L.oracle_per_sample_variant.argtypes = [vp, vp, i64, vp, i64, f32, ctypes.c_int, ctypes.c_int, vp, vp, vp]
        L.oracle_default_priority.restype = f32
        L.oracle_default_priority.argtypes = [f32, f32, f32]
        L.oracle_per_update.restype = f32
        L.oracle_per_update.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32]
        L.oracle_nstep_gather.argtypes = [vp, vp, vp, vp, vp, vp, i64, ctypes.c_int32, vp, vp, i64,
                                          vp, vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class SegmentTree:
    def __init__(self, size, is_min):
        self._h = lib().oracle_tree_create(int(size), int(bool(is_min)))
        self.size = int(size)
        self.capacity = lib().oracle_tree_capacity(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_tree_destroy(self._h)
            self._h = None

    def values(self):
        """Live numpy view of all 2*capacity nodes."""
        ptr = lib().oracle_tree_values(self._h)
        return np.ctypeslib.as_array(ptr, shape=(2 * self.capacity,))

    def update(self, index, value):
        index = np.ascontiguousarray(np.atleast_1d(index), dtype=np.int64)
        value = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float32), index.shape))
        lib().oracle_tree_update_batch(self._h, _p(index), _p(value), index.size)

    def get(self, index):
        return lib().oracle_tree_get(self._h, int(index))

    def query(self, l, r):
        return lib().oracle_tree_query(self._h, int(l), int(r))

    def scan_lower_bound(self, value):
        return lib().oracle_tree_scan_lower_bound(self._h, float(np.float32(value)))


class PrioritizedSamplerOracle:
    """Restates torchrl's PrioritizedSampler as driven by the reference (see C header)."""

    def __init__(self, max_capacity, alpha, beta, eps=1e-8):
        self.alpha, self.beta, self.eps = float(alpha), float(beta), float(eps)
        self.sum_tree = SegmentTree(max_capacity, False)
        self.min_tree = SegmentTree(max_capacity, True)
        self.max_priority = 1.0

    @property
    def default_priority(self):
        # (max + eps) ** alpha, evaluated in fp32 like update_priority's pow (see C header)
        return lib().oracle_default_priority(self.max_priority, self.alpha, self.eps)

    def add(self, index):
        p = np.float32(self.default_priority)
        self.sum_tree.update(index, p)
        self.min_tree.update(index, p)

    def draw_mass(self, length, batch_size, rng=np.random):
        p_sum = self.sum_tree.query(0, length)
        return rng.uniform(0.0, p_sum, size=batch_size).astype(np.float32)

    def sample(self, length, mass):
        mass = np.ascontiguousarray(mass, dtype=np.float32)
        n = mass.size
        idx = np.empty(n, np.int64)
        w = np.empty(n, np.float32)
        ps = np.empty(2, np.float32)
        lib().oracle_per_sample(self.sum_tree._h, self.min_tree._h, int(length), _p(mass), n,
                                self.beta, _p(idx), _p(w), _p(ps))
        if not (ps[0] > 0) or not (ps[1] > 0):
            raise RuntimeError("non-positive p_sum / p_min")
        return idx, w, float(ps[0]), float(ps[1])

    def sample_variant(self, length, mass, weight_form=0, query_full=False):
        """``sample`` under the torchrl version variants per_oracle.c documents (np.power weights, whole-capacity query)."""
        mass = np.ascontiguousarray(mass, dtype=np.float32)
        n = mass.size
        idx, w, ps = np.empty(n, np.int64), np.empty(n, np.float32), np.empty(2, np.float32)
        lib().oracle_per_sample_variant(self.sum_tree._h, self.min_tree._h, int(length), _p(mass), n, self.beta,
                                        int(weight_form), int(bool(query_full)), _p(idx), _p(w), _p(ps))
        return idx, w, float(ps[0]), float(ps[1])

    def update_priority(self, index, priority):
        index = np.ascontiguousarray(index, dtype=np.int64)
        priority = np.ascontiguousarray(priority, dtype=np.float32)
        self.max_priority = float(lib().oracle_per_update(
            self.sum_tree._h, self.min_tree._h, _p(index), _p(priority), index.size,
            self.alpha, self.eps, self.max_priority))


class ReplayOracle:
    """Slot-indexed SoA ring + PER + n-step collate.  Mirrors TimestepBuffer.extend / sample /
    update_priority (timestep_buffer.py:32-54) over arrays instead of linked Python objects."""

    def __init__(self, capacity, obs_elems, n_step, gamma, alpha=0.5, beta=0.5, use_per=True):
        self.capacity, self.obs_elems, self.n_step = int(capacity), int(obs_elems), int(n_step)
        self.gammas = np.array([gamma ** i for i in range(n_step + 1)], dtype=np.float64)
        self.obs = np.zeros((capacity, obs_elems), np.float32)
        self.succ_obs = np.zeros((capacity, obs_elems), np.float32)
        self.reward = np.zeros(capacity, np.float32)
        self.action = np.zeros(capacity, np.int32)
        self.flags = np.zeros(capacity, np.uint8)
        self.link = np.full(capacity, -1, np.int32)
        self.cursor = 0
        self.length = 0
        self.sampler = PrioritizedSamplerOracle(capacity, alpha, beta) if use_per else None

    def insert(self, obs, succ_obs, reward, action, done, truncated, has_next, prev_slot=-1):
        """Round-robin write (torchrl RoundRobinWriter); returns the slot."""
        s = self.cursor
        self.cursor = (self.cursor + 1) % self.capacity
        self.length = min(self.length + 1, self.capacity)
        # any slot that linked to the row being overwritten loses its successor
        self.link[self.link == s] = -1
        self.obs[s] = np.asarray(obs, np.float32).reshape(-1)
        self.succ_obs[s] = 0.0 if succ_obs is None else np.asarray(succ_obs, np.float32).reshape(-1)
        self.reward[s] = reward
        self.action[s] = action
        self.flags[s] = (FLAG_DONE if done else 0) | (FLAG_TRUNC if truncated else 0) | \
                        (FLAG_HAS_NEXT if has_next else 0)
        self.link[s] = -1
        if prev_slot >= 0:
            self.link[prev_slot] = s
        if self.sampler is not None:
            self.sampler.add(s)
        return s

    def gather(self, index):
        index = np.ascontiguousarray(index, dtype=np.int64)
        n, O = index.size, self.obs_elems
        out = dict(obs=np.empty((n, O), np.float32), next_obs=np.empty((n, O), np.float32),
                   reward=np.empty(n, np.float32), nonterminal=np.empty(n, np.uint8),
                   gamma=np.empty(n, np.float32), action=np.empty(n, np.int64),
                   needs_n_step=np.empty(n, np.uint8))
        lib().oracle_nstep_gather(_p(self.obs), _p(self.succ_obs), _p(self.reward), _p(self.action),
                                  _p(self.flags), _p(self.link), O, self.n_step, _p(self.gammas),
                                  _p(index), n, _p(out["obs"]), _p(out["next_obs"]), _p(out["reward"]),
                                  _p(out["nonterminal"]), _p(out["gamma"]), _p(out["action"]),
                                  _p(out["needs_n_step"]))
        return out
