"""CPU oracle for the replay half: n-step/collate pinned to the reference's golden chain; the
segment tree (PARITY UNPINNED, see oracle/per_oracle.c) checked for its own invariants."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import per_ref
from tests import helpers as H


def _load_chain():
    return np.load(os.path.join(H.GOLDEN, "nstep_chain.npz"))


def replay_from_chain(g):
    N = int(g["N"])
    O = int(np.prod(g["obs"].shape[1:]))
    rp = per_ref.ReplayOracle(capacity=N + 5, obs_elems=O, n_step=int(g["n_step"]), gamma=float(g["gamma"]))
    for i in range(N):
        rp.insert(g["obs"][i], g["succ_obs"][i], g["reward"][i], g["action"][i], bool(g["done"][i]),
                  bool(g["truncated"][i]), bool(g["has_next"][i]))
    rp.link[:N] = g["link"]
    return rp


def test_nstep_collate_matches_reference():
    g = _load_chain()
    rp = replay_from_chain(g)
    N = int(g["N"])
    out = rp.gather(np.arange(N))
    np.testing.assert_array_equal(out["reward"], g["exp_batch_reward"].reshape(-1))
    np.testing.assert_array_equal(out["reward"], g["exp_n_step_return"].astype(np.float32))
    np.testing.assert_array_equal(out["gamma"], g["exp_batch_gamma"].reshape(-1))
    np.testing.assert_array_equal(out["nonterminal"].astype(bool), g["exp_batch_nonterminal"].reshape(-1))
    np.testing.assert_array_equal(out["nonterminal"].astype(bool), ~g["exp_n_step_done"])
    np.testing.assert_array_equal(out["needs_n_step"].astype(bool), g["exp_needs_n_step"])
    np.testing.assert_array_equal(out["action"], g["exp_batch_action"].reshape(-1))
    np.testing.assert_array_equal(out["obs"].reshape(g["exp_batch_obs"].shape), g["exp_batch_obs"])
    np.testing.assert_array_equal(out["next_obs"].reshape(g["exp_batch_next_obs"].shape),
                                  g["exp_batch_next_obs"])
    assert g["exp_needs_n_step"].sum() > 0 and (~g["exp_batch_nonterminal"]).sum() > 0


def test_tree_capacity_rule():
    assert per_ref.SegmentTree(100_000, False).capacity == 131_072
    assert per_ref.SegmentTree(131_072, False).capacity == 262_144      # strictly greater
    assert per_ref.SegmentTree(1, False).capacity == 2


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 300), st.integers(0, 2 ** 31 - 1))
def test_sum_tree_invariants(size, seed):
    rng = np.random.default_rng(seed)
    t = per_ref.SegmentTree(size, False)
    m = per_ref.SegmentTree(size, True)
    idx = rng.integers(0, size, size=3 * size)
    val = rng.random(3 * size).astype(np.float32) + 1e-3
    t.update(idx, val)
    m.update(idx, val)
    v, cap = t.values(), t.capacity
    # every parent is exactly fl32(left + right)
    par = np.arange(1, cap)
    np.testing.assert_array_equal(v[par], v[2 * par] + v[2 * par + 1])
    mv = m.values()
    np.testing.assert_array_equal(mv[par], np.minimum(mv[2 * par], mv[2 * par + 1]))
    # duplicates: last occurrence wins
    last = {int(i): float(x) for i, x in zip(idx, val)}
    for i, x in last.items():
        assert t.get(i) == np.float32(x)
    assert t.query(0, size) == v[1]
    # scan_lower_bound is monotone in mass and lands on a written leaf
    masses = np.sort(rng.uniform(0, v[1], 64).astype(np.float32))
    found = [t.scan_lower_bound(x) for x in masses]
    assert all(a <= b for a, b in zip(found, found[1:]))
    assert t.scan_lower_bound(np.float32(v[1]) * 2 + 1) == size


def test_partial_range_query_order():
    t = per_ref.SegmentTree(100, False)
    vals = (np.arange(37) * 0.1 + 0.01).astype(np.float32)
    t.update(np.arange(37), vals)
    # bottom-up half-open walk restated in numpy
    v, cap = t.values(), t.capacity
    l, r, ret = cap, 37 | cap, np.float32(0)
    while l < r:
        if l & 1:
            ret = np.float32(ret + v[l]); l += 1
        if r & 1:
            r -= 1; ret = np.float32(ret + v[r])
        l >>= 1; r >>= 1
    assert t.query(0, 37) == ret


def test_sampler_semantics():
    s = per_ref.PrioritizedSamplerOracle(1000, 0.5, 0.5)
    with pytest.raises(RuntimeError):
        s.sample(10, np.zeros(4, np.float32))            # empty trees: p_sum == 0
    for i in range(700):
        s.add(i)
    assert s.sum_tree.get(0) == np.float32((1.0 + 1e-8) ** 0.5)
    s.update_priority(np.array([3, 3, 5]), np.array([2.0, 0.5, 9.0], np.float32))
    assert s.max_priority == 9.0
    assert s.sum_tree.get(3) == np.float32(np.float32(0.5 + 1e-8) ** np.float32(0.5))
    mass = s.draw_mass(700, 256, np.random.RandomState(1))
    idx, w, p_sum, p_min = s.sample(700, mass)
    assert idx.max() <= 699 and idx.min() >= 0
    assert w.max() <= 1.0 + 1e-6 and abs(p_min - s.sum_tree.get(3)) < 1e-7
    s.add(700)
    assert abs(s.sum_tree.get(700) - (9.0 + 1e-8) ** 0.5) < 1e-6


def test_torchrl_version_variants_of_the_sampler():
    """The reference pins no torchrl version.  Two things changed in torchrl's sampler over time (oracle/per_oracle.c,
    oracle_per_sample_variant); this states what each changes, so the assumption is tested instead of silent:
      * IS weights through ``np.power`` (older) instead of ``torch.pow`` (newer, what the oracle and the device evaluate):
        indices identical, weights within one unit in the last place;
      * p_sum / p_min over ``query(0, max_capacity)`` instead of ``query(0, len)``: identical in every output once the
        storage is full (the benched state); while it fills only the fp32 rounding of p_sum may differ -- never an index
        for the same masses."""
    rng = np.random.default_rng(11)
    cap = 5000
    for alpha, beta in ((0.5, 0.5), (0.6, 0.4), (1.0, 1.0)):
        smp = per_ref.PrioritizedSamplerOracle(cap, alpha, beta)
        for length in (cap, 3123):
            smp.sum_tree.values()[:] = 0.0
            smp.min_tree.values()[:] = np.finfo(np.float32).max
            idx = np.arange(length)
            smp.update_priority(idx, np.abs(rng.standard_normal(length)).astype(np.float32) * 3.0)
            mass = smp.draw_mass(length, 4096, rng)
            i0, w0, ps0, pm0 = smp.sample(length, mass)
            i00, w00, _, _ = smp.sample_variant(length, mass, 0, False)
            np.testing.assert_array_equal(i0, i00)
            np.testing.assert_array_equal(w0, w00)
            i1, w1, ps1, pm1 = smp.sample_variant(length, mass, 1, False)           # np.power weights
            np.testing.assert_array_equal(i0, i1)
            ulp = np.spacing(np.maximum(np.abs(w0), np.abs(w1)).astype(np.float32))
            assert np.all(np.abs(w0.astype(np.float64) - w1) <= ulp), "np.power and torch.pow weights differ by more than 1 ulp"
            if alpha == 1.0 and beta == 1.0:
                np.testing.assert_array_equal(w0, 1.0 / (smp.sum_tree.values()[i0 + smp.sum_tree.capacity] / np.float32(pm0)))
            i2, w2, ps2, pm2 = smp.sample_variant(length, mass, 0, True)            # whole-capacity query
            np.testing.assert_array_equal(i0, i2)
            assert pm2 == pm0                                                       # min is order-free
            if length == cap:
                assert ps2 == ps0
                np.testing.assert_array_equal(w0, w2)
            else:
                assert abs(ps2 - ps0) <= 4 * np.spacing(np.float32(ps0))
