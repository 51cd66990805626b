"""GPU parity: HBM replay ring + PER trees (through the C ABI) vs the CPU oracle.
Bit-exact bar: sampled indices, tree nodes, IS weights (alpha = beta = 0.5 uses correctly
rounded sqrt/div on both sides), n-step returns, gathered rows."""
import os
import weakref

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return "cuda:0"


def _mk_buffer(dev, capacity, B, O_shape=(10, 10, 4), **kw):
    from prism_amd.experience import HipReplayBuffer
    return HipReplayBuffer(capacity, B, device=dev, **kw)


def _chain_timesteps():
    """Rebuild linked Timestep objects from the golden chain, in insertion order."""
    from prism_amd.experience import Timestep
    g = np.load(os.path.join(H.GOLDEN, "nstep_chain.npz"))
    N = int(g["N"])
    ts = [Timestep(id=i) for i in range(N)]
    keep = []
    for i, t in enumerate(ts):
        t.obs = torch.from_numpy(g["obs"][i])
        t.reward, t.action = float(g["reward"][i]), int(g["action"][i])
        t.done, t.truncated = bool(g["done"][i]), bool(g["truncated"][i])
        if t.truncated:
            t.next = Timestep(id=10_000 + i, obs=torch.from_numpy(g["succ_obs"][i]))
        elif g["has_next"][i]:
            if g["link"][i] >= 0:
                t.next = weakref.ref(ts[int(g["link"][i])])
            else:
                node = Timestep(id=20_000 + i, obs=torch.from_numpy(g["succ_obs"][i]))
                keep.append(node)
                t.next = weakref.ref(node)
    return g, ts, keep


def test_extend_then_gather_matches_reference_chain(dev):
    g, ts, keep = _chain_timesteps()
    N = len(ts)
    buf = _mk_buffer(dev, N + 9, N, n_step=int(g["n_step"]), gamma=float(g["gamma"]), use_per=True)
    for t in ts:
        buf.extend(t)
    batch, info = buf.sample(return_info=True)       # flushes; sampled rows are random
    # gather every slot in order through the C ABI
    import ctypes
    from prism_amd import _native as Nn
    idx = torch.arange(N, device=dev, dtype=torch.int64)
    with torch.cuda.device(dev):
        Nn.check(Nn.lib().prism_replay_gather(ctypes.byref(buf._desc), Nn.ptr(idx), N, Nn.ptr(buf._obs),
                                              Nn.ptr(buf._next_obs), Nn.ptr(buf._reward), Nn.ptr(buf._nonterminal),
                                              Nn.ptr(buf._gamma), Nn.ptr(buf._action), Nn.current_stream_handle()),
                 "gather")
    torch.cuda.synchronize()
    b = buf.get_static_batch()
    np.testing.assert_array_equal(b["next"]["reward"].cpu().numpy(), g["exp_batch_reward"])
    np.testing.assert_array_equal(b["gamma"].cpu().numpy(), g["exp_batch_gamma"])
    np.testing.assert_array_equal(b["nonterminal"].cpu().numpy(), g["exp_batch_nonterminal"])
    np.testing.assert_array_equal(b["action"].cpu().numpy(), g["exp_batch_action"])
    np.testing.assert_array_equal(b["observation"].cpu().numpy(), g["exp_batch_obs"])
    np.testing.assert_array_equal(b["next"]["observation"].cpu().numpy(), g["exp_batch_next_obs"])
    # links built through the pending-successor map equal the reference's object graph
    np.testing.assert_array_equal(buf.link[:N].cpu().numpy(), g["link"])
    assert batch["observation"].shape == (N, 1, 10, 10, 4) and info["index"].dtype == torch.int64
    assert int(info["index"].max()) < N and float(info["_weight"].max()) <= 1.0


def _random_ring(rng, n, O=400):
    obs = (rng.random((n, O)) < 0.1).astype(np.float32)
    succ = (rng.random((n, O)) < 0.1).astype(np.float32)
    reward = rng.standard_normal(n).astype(np.float32)
    action = rng.integers(0, 6, n).astype(np.int32)
    done = rng.random(n) < 0.05
    trunc = (~done) & (rng.random(n) < 0.03)
    has_next = ~done | trunc
    link = np.where(~done & ~trunc & (rng.random(n) < 0.97), (np.arange(n) + 8) % n, -1).astype(np.int32)
    link[-8:] = -1
    flags = (done * 1 + trunc * 2 + has_next * 4).astype(np.uint8)
    return obs, succ, reward, action, flags, link


@pytest.mark.parametrize("capacity,n,B", [(1000, 700, 256), (1000, 1000, 512), (100_000, 100_000, 256)])
def test_per_sample_update_bit_exact(dev, capacity, n, B):
    from oracle import per_ref
    rng = np.random.default_rng(capacity + n)
    obs, succ, reward, action, flags, link = _random_ring(rng, n)
    prio = (np.abs(rng.standard_normal(n)).astype(np.float32) + np.float32(1e-8)) ** np.float32(0.5)
    # oracle
    orc = per_ref.ReplayOracle(capacity, 400, 3, 0.99)
    orc.obs[:n], orc.succ_obs[:n], orc.reward[:n], orc.action[:n] = obs, succ, reward, action
    orc.flags[:n], orc.link[:n], orc.length = flags, link, n
    orc.sampler.sum_tree.update(np.arange(n), prio)
    orc.sampler.min_tree.update(np.arange(n), prio)
    # device
    buf = _mk_buffer(dev, capacity, B, n_step=3, gamma=0.99, use_per=True)
    buf.load_arrays(torch.from_numpy(obs.reshape(n, 10, 10, 4)), torch.from_numpy(succ), torch.from_numpy(reward),
                    torch.from_numpy(action), torch.from_numpy(flags), torch.from_numpy(link),
                    torch.from_numpy(prio))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(buf.sum_tree.cpu().numpy(), orc.sampler.sum_tree.values())
    np.testing.assert_array_equal(buf.min_tree.cpu().numpy()[1:], orc.sampler.min_tree.values()[1:])

    import ctypes
    from prism_amd import _native as Nn
    for rnd in range(4):
        mass = orc.sampler.draw_mass(n, B, np.random.RandomState(rnd))
        if rnd == 3:
            mass[:4] = [0.0, np.float32(orc.sampler.sum_tree.values()[1]), np.float32(3e38), mass[5]]
        idx_o, w_o, psum_o, pmin_o = orc.sampler.sample(n, mass)
        m_d = torch.from_numpy(mass).to(dev)
        with torch.cuda.device(dev):
            Nn.check(Nn.lib().prism_per_sample(ctypes.byref(buf._desc), n, B, Nn.ptr(m_d), 0, 0, 0.5,
                                               Nn.ptr(buf._index), Nn.ptr(buf._weight),
                                               Nn.current_stream_handle()), "sample")
        torch.cuda.synchronize()
        np.testing.assert_array_equal(buf._index.cpu().numpy(), idx_o)          # bit-exact indices
        np.testing.assert_array_equal(buf._weight.cpu().numpy(), w_o)
        ps = buf.per_state.cpu().numpy()
        assert ps[1] == np.float32(psum_o) and ps[2] == np.float32(pmin_o)
        # gather of the sampled rows
        out = orc.gather(idx_o)
        with torch.cuda.device(dev):
            Nn.check(Nn.lib().prism_replay_gather(ctypes.byref(buf._desc), Nn.ptr(buf._index), B, Nn.ptr(buf._obs),
                                                  Nn.ptr(buf._next_obs), Nn.ptr(buf._reward),
                                                  Nn.ptr(buf._nonterminal), Nn.ptr(buf._gamma), Nn.ptr(buf._action),
                                                  Nn.current_stream_handle()), "gather")
        torch.cuda.synchronize()
        b = buf.get_static_batch()
        np.testing.assert_array_equal(b["next"]["reward"].cpu().numpy().ravel(), out["reward"])
        np.testing.assert_array_equal(b["gamma"].cpu().numpy().ravel(), out["gamma"])
        np.testing.assert_array_equal(b["nonterminal"].cpu().numpy().ravel(), out["nonterminal"].astype(bool))
        np.testing.assert_array_equal(b["action"].cpu().numpy().ravel(), out["action"])
        np.testing.assert_array_equal(b["observation"].cpu().numpy().reshape(B, -1), out["obs"])
        np.testing.assert_array_equal(b["next"]["observation"].cpu().numpy().reshape(B, -1), out["next_obs"])
        # priority writeback with duplicates (sampling is with replacement)
        td = rng.standard_normal(B).astype(np.float32)
        orc.sampler.update_priority(idx_o, np.abs(td))
        buf.update_priority(buf._index, torch.from_numpy(td).to(dev), take_abs=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(buf.sum_tree.cpu().numpy(), orc.sampler.sum_tree.values())
        np.testing.assert_array_equal(buf.min_tree.cpu().numpy()[1:], orc.sampler.min_tree.values()[1:])
        assert buf.per_state.cpu().numpy()[0] == np.float32(orc.sampler.max_priority)


def test_insert_default_priority_and_overwrite(dev):
    """Round-robin overwrite past capacity; new rows get (max+eps)**alpha; trees stay consistent."""
    from oracle import per_ref
    from prism_amd.experience import Timestep
    cap, B = 50, 16
    buf = _mk_buffer(dev, cap, B, n_step=3, gamma=0.99, use_per=True)
    orc = per_ref.ReplayOracle(cap, 400, 3, 0.99)
    rng = np.random.default_rng(3)
    cur = Timestep(id=0, obs=torch.from_numpy((rng.random((10, 10, 4)) < 0.1).astype(np.float32)))
    prev_slot = -1
    for i in range(130):
        nxt = Timestep(id=i + 1, obs=torch.from_numpy((rng.random((10, 10, 4)) < 0.1).astype(np.float32)))
        cur.reward, cur.action = float(np.float32(rng.standard_normal())), int(rng.integers(0, 6))
        cur.done, cur.truncated = bool(i % 11 == 10), False
        if not cur.done:
            cur.next = weakref.ref(nxt)
        s = buf.extend(cur)
        so = orc.insert(cur.obs.numpy(), nxt.obs.numpy() if not cur.done else None, cur.reward, cur.action,
                        cur.done, False, not cur.done, prev_slot)
        assert s == so
        prev_slot = -1 if cur.done else s
        if i % 37 == 36:      # interleave a sample + writeback so max_priority moves
            buf.flush()
            idx = torch.from_numpy(rng.integers(0, len(buf), B)).to(dev)
            td = torch.from_numpy((rng.random(B) * 3).astype(np.float32)).to(dev)
            buf.update_priority(idx, td)
            orc.sampler.update_priority(idx.cpu().numpy(), td.cpu().numpy())
        cur = nxt
    buf.flush()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(buf.sum_tree.cpu().numpy(), orc.sampler.sum_tree.values())
    np.testing.assert_array_equal(buf.link.cpu().numpy(), orc.link)
    np.testing.assert_array_equal(buf.flags.cpu().numpy(), orc.flags)
    np.testing.assert_array_equal(buf.obs.cpu().numpy(), orc.obs)
    np.testing.assert_array_equal(buf.succ_obs.cpu().numpy(), orc.succ_obs)
    out = orc.gather(np.arange(cap))
    import ctypes
    from prism_amd import _native as Nn
    idx = torch.arange(cap, device=dev)
    buf._alloc_batch(cap)
    with torch.cuda.device(dev):
        Nn.check(Nn.lib().prism_replay_gather(ctypes.byref(buf._desc), Nn.ptr(idx), cap, Nn.ptr(buf._obs),
                                              Nn.ptr(buf._next_obs), Nn.ptr(buf._reward), Nn.ptr(buf._nonterminal),
                                              Nn.ptr(buf._gamma), Nn.ptr(buf._action), Nn.current_stream_handle()),
                 "gather")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(buf._reward.cpu().numpy().ravel(), out["reward"])
    np.testing.assert_array_equal(buf._gamma.cpu().numpy().ravel(), out["gamma"])
    np.testing.assert_array_equal(buf._next_obs.cpu().numpy().reshape(cap, -1), out["next_obs"])


def test_philox_sampling_follows_priorities(dev):
    n, B = 4096, 4096
    buf = _mk_buffer(dev, n, B, use_per=True)
    rng = np.random.default_rng(0)
    obs, succ, reward, action, flags, link = _random_ring(rng, n)
    prio = np.ones(n, np.float32)
    prio[:n // 2] = 3.0
    buf.load_arrays(torch.from_numpy(obs.reshape(n, 10, 10, 4)), torch.from_numpy(succ), torch.from_numpy(reward),
                    torch.from_numpy(action), torch.from_numpy(flags), torch.from_numpy(link), torch.from_numpy(prio))
    counts = 0
    for _ in range(8):
        _, info = buf.sample(return_info=True)
        idx = info["index"].cpu().numpy()
        assert idx.min() >= 0 and idx.max() < n
        counts += (idx < n // 2).sum()
    frac = counts / (8 * B)
    assert abs(frac - 0.75) < 0.02
    w = info["_weight"].cpu().numpy()
    assert set(np.unique(w)).issubset({np.float32(1.0), np.float32(1.0) / np.sqrt(np.float32(3.0))})
    buf.check_status()


def test_uniform_replay_and_empty_errors(dev):
    buf = _mk_buffer(dev, 100, 8, use_per=False)
    with pytest.raises(RuntimeError):
        buf.sample()
    rng = np.random.default_rng(0)
    obs, succ, reward, action, flags, link = _random_ring(rng, 60)
    buf.load_arrays(torch.from_numpy(obs.reshape(60, 10, 10, 4)), torch.from_numpy(succ), torch.from_numpy(reward),
                    torch.from_numpy(action), torch.from_numpy(flags), torch.from_numpy(link))
    b, info = buf.sample(return_info=True)
    assert "_weight" not in info and int(info["index"].max()) < 60
    # all-zero priorities: torchrl raises at sample time; here the sticky status word does
    pbuf = _mk_buffer(dev, 100, 8, use_per=True, strict=True)
    pbuf.load_arrays(torch.from_numpy(obs.reshape(60, 10, 10, 4)), torch.from_numpy(succ), torch.from_numpy(reward),
                     torch.from_numpy(action), torch.from_numpy(flags), torch.from_numpy(link),
                     torch.zeros(60))
    with pytest.raises(RuntimeError):
        pbuf.sample()


@pytest.mark.parametrize("cap,n_env,total,seed", [(257, 4, 1500, 0), (64, 3, 700, 1), (1000, 8, 2500, 2)])
def test_interleaved_env_chains_wrap_the_ring(dev, cap, n_env, total, seed):
    """Several collectors' episodes interleaved round-robin, flushed in random batch sizes, wrapping a
    small ring many times: link/back/flags and the priority tree must equal the oracle's one-row-at-a-time
    insert (the device links a whole batch in three parallel phases, or row by row when a batch wraps
    over its own predecessors)."""
    from oracle import per_ref
    from prism_amd.experience import Timestep
    buf = _mk_buffer(dev, cap, 16, n_step=3, gamma=0.99, use_per=True)
    orc = per_ref.ReplayOracle(cap, 400, 3, 0.99)
    rng = np.random.default_rng(seed)
    new_obs = lambda: torch.from_numpy((rng.random((10, 10, 4)) < 0.1).astype(np.float32))
    ids = iter(range(10 ** 9))
    cur = [Timestep(id=next(ids), obs=new_obs()) for _ in range(n_env)]
    prev_slot = [-1] * n_env
    next_flush = int(rng.integers(1, min(cap, 200)))
    for i in range(total):
        e = i % n_env
        t = cur[e]
        nxt = Timestep(id=next(ids), obs=new_obs())
        t.reward, t.action = float(np.float32(rng.standard_normal())), int(rng.integers(0, 6))
        t.done, t.truncated = bool(rng.random() < 0.07), False
        if not t.done:
            t.next = weakref.ref(nxt)
        s = buf.extend(t)
        so = orc.insert(t.obs.numpy(), nxt.obs.numpy() if not t.done else None, t.reward, t.action, t.done, False,
                        not t.done, prev_slot[e])
        assert s == so
        prev_slot[e] = -1 if t.done else s
        cur[e] = nxt
        if buf._n_staged >= next_flush:
            buf.flush()
            next_flush = int(rng.integers(1, min(cap, 200)))
    buf.flush()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(buf.link.cpu().numpy(), orc.link)
    np.testing.assert_array_equal(buf.flags.cpu().numpy(), orc.flags)
    np.testing.assert_array_equal(buf.sum_tree.cpu().numpy(), orc.sampler.sum_tree.values())


@pytest.mark.parametrize("capacity,B", [(1000, 256), (5000, 512), (70_000, 64)])
def test_tree_invariants_and_monotone_sampling(dev, capacity, B):
    """Size-independent properties of the device trees after many writebacks with heavy duplication:
    every parent is fl32(left + right) / min(left, right) of its children, the root is the total mass,
    duplicate indices resolve to the LAST occurrence, and the sampled index is monotone in the mass."""
    import ctypes
    from prism_amd import _native as Nn
    buf = _mk_buffer(dev, capacity, B, use_per=True)
    rng = np.random.default_rng(capacity)
    O = 400
    buf.load_arrays(np.zeros((capacity, 10, 10, 4), np.float32), np.zeros((capacity, O), np.float32),
                    np.zeros(capacity, np.float32), np.zeros(capacity, np.int32), np.zeros(capacity, np.uint8),
                    np.full(capacity, -1, np.int32), priorities=np.full(capacity, 1.0, np.float32))
    last = {}
    for it in range(20):
        # a third of the batch hits only 8 distinct slots: many duplicates per call
        idx = np.where(rng.random(B) < 0.33, rng.integers(0, 8, B), rng.integers(0, capacity, B)).astype(np.int64)
        td = (rng.random(B) * 5).astype(np.float32)
        buf.update_priority(torch.from_numpy(idx).to(dev), torch.from_numpy(td).to(dev))
        for i, t in zip(idx, td):
            last[int(i)] = t
    torch.cuda.synchronize()
    tree = buf.tree.cpu().numpy()
    cap2 = buf.tree_capacity
    s, m = tree[:, 0], tree[:, 1]
    par = np.arange(1, cap2)
    np.testing.assert_array_equal(s[par], s[2 * par] + s[2 * par + 1])
    np.testing.assert_array_equal(m[par], np.minimum(m[2 * par], m[2 * par + 1]))
    for i, t in last.items():                                   # last occurrence wins, (|td| + eps) ** alpha
        want = np.float32(np.sqrt(np.float32(t) + np.float32(1e-8)))
        assert s[cap2 + i] == want and m[cap2 + i] == want, i
    # monotone: ascending masses -> non-decreasing indices, all inside [0, size)
    mass = np.sort(rng.random(B).astype(np.float32)) * np.float32(s[1])
    mt = torch.from_numpy(mass).to(dev)
    with torch.cuda.device(dev):
        Nn.check(Nn.lib().prism_per_sample(ctypes.byref(buf._desc), capacity, B, Nn.ptr(mt), 0, 0, 0.5,
                                           Nn.ptr(buf._index), Nn.ptr(buf._weight), Nn.current_stream_handle()), "sample")
    torch.cuda.synchronize()
    got = buf._index.cpu().numpy()
    assert (np.diff(got) >= 0).all() and got.min() >= 0 and got.max() < capacity
    assert abs(float(s[1]) - float(np.sum(s[cap2:cap2 + capacity], dtype=np.float64))) < 1e-3 * float(s[1])


def test_full_size_shard_properties(dev):
    """BASELINE configs[4] shard size (1.25 M slots, cap2 = 2^21): a full learner-sized writeback
    (B = 512, duplicates included) keeps every parent equal to op(children), checked on the device."""
    from prism_amd.synthetic import fill_replay
    capacity, B = 1_250_000, 512
    buf = _mk_buffer(dev, capacity, B, use_per=True)
    fill_replay(buf, capacity, seed=1)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    for _ in range(8):
        idx = torch.randint(0, capacity, (B,), device=dev, generator=g)
        idx[:64] = idx[64:128]                                   # duplicates
        td = torch.rand(B, device=dev, generator=g) * 3
        buf.update_priority(idx, td)
    batch, info = buf.sample(return_info=True)
    torch.cuda.synchronize()
    t, cap2 = buf.tree, buf.tree_capacity
    kids = t[2:2 * cap2].view(cap2 - 1, 2, 2)                    # node p -> children 2p, 2p+1
    assert torch.equal(t[1:cap2, 0], kids[:, 0, 0] + kids[:, 1, 0])
    assert torch.equal(t[1:cap2, 1], torch.minimum(kids[:, 0, 1], kids[:, 1, 1]))
    idx = info["index"]
    assert int(idx.min()) >= 0 and int(idx.max()) < capacity
    w = info["_weight"]
    assert float(w.max()) <= 1.0 + 1e-6 and float(w.min()) > 0.0
