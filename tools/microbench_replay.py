"""Micro-timings of the replay kernels (GPU box only)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prism_amd import _native as N
from prism_amd.experience import HipReplayBuffer
from prism_amd.synthetic import fill_replay

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for cap in (1000, 100_000, 1_250_000):
    for B in (256, 512):
        buf = HipReplayBuffer(cap, B, device="cuda:0")
        fill_replay(buf, cap, seed=0)
        buf.sample(return_info=True)
        idx = buf._index.clone(); td = torch.rand(B, device="cuda:0")
        L = N.lib(); d = ctypes.byref(buf._desc); st = N.current_stream_handle()
        t_upd = timeit(lambda: L.prism_per_update(d, N.ptr(idx), N.ptr(td), B, 0.5, 1e-8, 1, st))
        t_smp = timeit(lambda: L.prism_per_sample(d, cap, B, None, 1, 0, 0.5, N.ptr(buf._index), N.ptr(buf._weight), st))
        t_gat = timeit(lambda: L.prism_replay_gather(d, N.ptr(idx), B, N.ptr(buf._obs), N.ptr(buf._next_obs), N.ptr(buf._reward), N.ptr(buf._nonterminal), N.ptr(buf._gamma), N.ptr(buf._action), st))
        print(f"cap={cap:8d} B={B}: update {t_upd:6.1f} us  sample {t_smp:6.1f} us  gather {t_gat:6.1f} us (back-to-back launches, incl. launch gap)")

# ---- producer seam: extend() + flush() (timesteps/s into the HBM ring, links and priorities included)
import weakref
import numpy as np
from prism_amd.experience import Timestep
for cap, n in ((100_000, 30_000),):
    buf = HipReplayBuffer(cap, 256, device="cuda:0")
    rng = np.random.default_rng(0)
    obs = [torch.from_numpy((rng.random((10, 10, 4)) < 0.1).astype(np.float32)) for _ in range(64)]
    steps = [Timestep(id=i, obs=obs[i % 64]) for i in range(n + 1)]
    for i in range(n):
        t = steps[i]
        t.reward, t.action, t.done, t.truncated = 0.5, i % 6, (i % 200 == 199), False
        if not t.done:
            t.next = weakref.ref(steps[i + 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        buf.extend(steps[i])
    buf.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L = N.lib(); d = ctypes.byref(buf._desc); st = N.current_stream_handle()
    dd = buf._stage_dev[0]
    t_ins = timeit(lambda: L.prism_replay_insert(d, 1024, N.ptr(dd["slots"]), N.ptr(dd["obs"]), N.ptr(dd["succ"]), N.ptr(dd["reward"]),
                                                 N.ptr(dd["action"]), N.ptr(dd["flags"]), N.ptr(dd["prev"]), 0.5, 1e-8, st), n=50)
    print(f"extend+flush: {n / dt:9.0f} timesteps/s end to end (Python staging included); "
          f"prism_replay_insert of 1024 rows: {t_ins:6.1f} us")
