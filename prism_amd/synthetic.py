"""Synthetic MinAtar-shaped replay contents for benchmarks and smoke tests:
binary observations ~ Bernoulli(0.1) stored fp32 (reference layout), per-step rewards ~ N(0,1),
episode ends so that ~5 % of 3-step bootstraps are terminal, actions ~ U{0..A-1}, initial
priorities |N(0,1)|**alpha + 1e-8.  Transitions form linked chains (slot i -> i + n_streams) as
interleaved env streams would (experience_collector.py:94-120), so the gather kernel really walks
n-step links."""
import numpy as np
import torch


def fill_replay(buffer, n, obs_shape=(10, 10, 4), n_actions=6, seed=0, n_streams=8, p_done=0.017, alpha=0.5,
                chunk=65536):
    dev = buffer.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    O = int(np.prod(obs_shape))
    if buffer._desc is None:
        buffer._allocate(tuple(obs_shape))
    idx = torch.arange(n, device=dev)
    done = torch.rand(n, device=dev, generator=gen) < p_done
    link = idx + n_streams
    link = torch.where((link < n) & ~done, link, torch.full_like(link, -1))
    has_next = ~done
    flags = (done.to(torch.uint8) * 1 + has_next.to(torch.uint8) * 4)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        buffer.obs[s:e] = (torch.rand(e - s, O, device=dev, generator=gen) < 0.1).float()
    # successor observation = the linked row's observation (random where the chain is still open)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        lk = link[s:e]
        rnd = (torch.rand(e - s, O, device=dev, generator=gen) < 0.1).float()
        src = buffer.obs[lk.clamp(min=0)]
        buffer.succ_obs[s:e] = torch.where((lk >= 0).unsqueeze(1), src, rnd)
    buffer.reward[:n] = torch.randn(n, device=dev, generator=gen)
    buffer.action[:n] = torch.randint(0, n_actions, (n,), device=dev, generator=gen, dtype=torch.int32)
    buffer.flags[:n] = flags
    buffer.link[:n] = link.to(torch.int32)
    buffer.back.fill_(-1)
    valid = link >= 0
    buffer.back[link[valid]] = idx[valid].to(torch.int32)
    buffer._size = n
    buffer.buffer._writer._cursor = n % buffer.capacity
    buffer._slot_id[:n] = np.arange(n)
    if buffer.use_per:
        import ctypes
        from prism_amd import _native as N
        tc = buffer.tree_capacity
        p = torch.randn(n, device=dev, generator=gen).abs().pow(alpha) + 1e-8
        buffer.sum_tree[tc:tc + n] = p
        buffer.min_tree[tc:tc + n] = p
        buffer.per_state[0] = 1.0
        with torch.cuda.device(dev):
            N.check(N.lib().prism_per_rebuild(ctypes.byref(buffer._desc), N.current_stream_handle()),
                    "prism_per_rebuild")
    torch.cuda.synchronize(dev)
    return buffer
