"""GPU: data-parallel learner path.  Two ranks share the one GPU of the test box (gloo carries the
gradient all-reduce through host memory; on a real node the backend is RCCL): replicas must stay
bit-identical while sampling different shards, and a 2-rank run on identical data must equal the
single-replica run."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, same_data, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from prism_amd import dist as pdist
    from prism_amd.agents import hip_agent
    from prism_amd.config import baseline_config
    from prism_amd.learner import Learner
    from prism_amd.synthetic import fill_replay

    def host_allreduce(flat, group=None):          # gloo over host memory (no RCCL with two ranks on one device)
        h = flat.cpu()
        dist.all_reduce(h)
        flat.copy_(h)
        return 1.0 / world
    hip_agent.pdist.allreduce_grads = host_allreduce

    cfg = baseline_config(2, device="cuda:0", batch_size=32, experience_replay_capacity=2048)
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6, process_group=dist.group.WORLD)
    buf, ag = ln.experience_buffer, ln.agent
    assert ag.world == world
    if not same_data:
        _, buf.seed, ag.seed = pdist.rank_seeds(cfg.seed, rank)
    fill_replay(buf, 2048, seed=0 if same_data else rank)
    sums = []
    for step in range(5):
        ln.step()
        torch.cuda.synchronize()
        assert pdist.assert_replicas_identical(ag.flat.cpu())
        sums.append(buf._index.cpu().numpy().copy())
    q.put((rank, ag.flat.cpu().numpy(), np.stack(sums), any(isinstance(g, tuple) and len(g) == 2 for g in ag._graphs.values())))
    dist.barrier()
    dist.destroy_process_group()


def _run(same_data):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, same_data, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240), q.get(timeout=240)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(600)
def test_two_ranks_stay_identical_on_different_shards():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    (r0, p0, i0, g0), (r1, p1, i1, g1) = _run(same_data=False)
    np.testing.assert_array_equal(p0, p1)             # replicas bit-identical
    assert not np.array_equal(i0, i1)                 # ... while sampling different transitions
    assert g0 and g1                                  # the split hipGraph path really ran


@pytest.mark.timeout(600)
def test_two_ranks_on_identical_data_equal_single_replica():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from prism_amd.config import baseline_config
    from prism_amd.learner import Learner
    from prism_amd.synthetic import fill_replay
    (r0, p0, i0, _), (r1, p1, i1, _) = _run(same_data=True)
    np.testing.assert_array_equal(i0, i1)
    cfg = baseline_config(2, device="cuda:0", batch_size=32, experience_replay_capacity=2048)
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
    fill_replay(ln.experience_buffer, 2048, seed=0)
    for step in range(5):
        ln.step()
    torch.cuda.synchronize()
    # (g + g) * 0.5 == g exactly; only the clip norm is summed in a different order
    np.testing.assert_allclose(p0, ln.agent.flat.cpu().numpy(), rtol=0, atol=1e-6)


# ---- the direct all-reduce (prism_direct_reduce_scatter / prism_direct_all_gather over peer-mapped buffers)
def _direct_worker(rank, world, port, q, mode=None):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from prism_amd import _native as N
    from prism_amd import dist as pdist
    N_MAX = N.MAX_PEERS
    ok = True
    for n in (201_430, 1_544_210, 7, 4096):                 # configs[2] / configs[3] parameter counts, a tail-only and an even size
        g = torch.Generator(device="cuda:0").manual_seed(100 + rank)
        flat = torch.randn(n, device="cuda:0", generator=g)
        want = flat.cpu()
        dist.all_reduce(want)                                # the oracle: gloo over host memory
        ar = pdist.DirectAllReduce(flat, use_flags=mode, wait_seconds=5.0)
        if mode is None:
            assert ar.use_flags is False and not ar.paced    # two ranks on ONE device: host-side barriers, no device spinning
        try:
            scale = ar.allreduce()
            torch.cuda.synchronize()
            ar.check_status()
        except pdist.CollectiveTimeout:
            # (mode True only: the two processes' one-wave kernels did not run side by side on the shared device)
            q.put((rank, "timeout"))
            dist.barrier()
            dist.destroy_process_group()
            return
        ok = ok and scale == 1.0 / world and torch.equal(flat.cpu(), want)      # (two addends: the sum is order-free, bit for bit)
        # a second round on the same mapping (what a training loop does)
        flat.copy_(torch.full((n,), float(rank + 1), device="cuda:0"))
        ar.allreduce()
        torch.cuda.synchronize()
        ok = ok and bool((flat == float(sum(range(1, world + 1)))).all())
        if mode is not None:
            # the protocol's own words: two all-reduces done, every peer's last announcement is phase 3 * 1 + 3, no time-out;
            # the flag array is the library's uncached allocation, the peer's is mapped through its raw IPC handle
            f = ar.read_flags()
            ok = ok and f[N_MAX + 1] == 2 and all(f[s] == 6 for s in range(world)) and f[N_MAX] == 0
            ok = ok and ar._desc.flags[rank] == ar._flags_ptr and ar._desc.flags[1 - rank] not in (None, 0, ar._flags_ptr)
        dist.barrier()
        ar.close()
        del ar
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def _run_direct(mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240), q.get(timeout=240)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(600)
def test_direct_allreduce_equals_the_collective():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    assert all(ok is True for _, ok in _run_direct(None))


@pytest.mark.timeout(600)
def test_direct_allreduce_device_flags_paced_by_the_host():
    """World = 2 through the DEVICE-FLAG kernels: two processes, two different uncached flag allocations (each maps the
    other's through its IPC handle), announce and wait launched apart with the host's barrier between them -- the ranks share
    the one device of this box, so a wait must never spin on a peer that cannot run.  Flag addressing (who stores into
    whose slot), the phase arithmetic 3 E + k and the epoch counter are exactly those of the step's path."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    assert all(ok is True for _, ok in _run_direct("paced"))


@pytest.mark.timeout(600)
def test_direct_allreduce_device_flags_spinning_between_two_processes():
    """The step's real form (announce + bounded spin in one launch, use_flags = 1) between two processes.  On one shared
    device it only completes when the two processes' one-wave kernels are scheduled side by side, which HIP does not promise:
    a time-out (5 s bound, sticky slot, CollectiveTimeout on the host) is reported as a skip, wrong sums never are."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    res = _run_direct(True)
    if any(ok == "timeout" for _, ok in res):
        pytest.skip("kernels of two processes did not overlap on the shared device: the bounded wait gave up, as designed")
    assert all(ok is True for _, ok in res)


def _direct_learner_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from prism_amd import dist as pdist
    from prism_amd.config import baseline_config
    from prism_amd.learner import Learner
    from prism_amd.synthetic import fill_replay
    cfg = baseline_config(2, device="cuda:0", batch_size=32, experience_replay_capacity=2048, collective="direct")
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6, process_group=dist.group.WORLD)
    buf, ag = ln.experience_buffer, ln.agent
    assert ag._direct is not None and ag.world == world
    _, buf.seed, ag.seed = pdist.rank_seeds(cfg.seed, rank)
    fill_replay(buf, 2048, seed=rank)
    for step in range(5):
        ln.step()
        torch.cuda.synchronize()
        assert pdist.assert_replicas_identical(ag.flat.cpu())
    q.put((rank, ag.flat.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_learner_steps_over_the_direct_allreduce():
    """The data-parallel step with collective = "direct": replicas bit-identical after every step, and equal to the run
    whose gradients travel through the process group's all_reduce (two addends: same bits)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_learner_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240), q.get(timeout=240)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])
    (_, p0, _, _), _ = _run(same_data=False)               # the same two shards through the host all_reduce
    np.testing.assert_array_equal(res[0][1], p0)


def test_direct_allreduce_flag_kernels_single_rank_and_graph_replay():
    """The device-flag form needs one device per rank, which this box does not have; what CAN run here is its one-rank
    degenerate case (signal to / wait on the own flag array): the phase counter lives on the device and advances per
    all-reduce, also when the calls are replayed from a captured hipGraph."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import ctypes
    from prism_amd import _native as N
    dev = "cuda:0"
    flat = torch.randn(10_007, device=dev)
    want = flat.clone()
    fp = ctypes.c_void_p(0)
    N.check(N.lib().prism_direct_flags_alloc(ctypes.byref(fp), None), "prism_direct_flags_alloc")      # uncached device words
    d = N.DirectDesc()
    d.world, d.rank, d.n = 1, 0, flat.numel()
    d.bufs[0], d.flags[0] = flat.data_ptr(), fp.value

    def read_flags():
        out = (ctypes.c_uint32 * N.DIRECT_FLAG_WORDS)()
        N.check(N.lib().prism_direct_flags_read(fp.value, out, N.current_stream_handle()), "prism_direct_flags_read")
        return list(out)

    def allreduce():
        N.check(N.lib().prism_direct_reduce_scatter(ctypes.byref(d), 1, N.current_stream_handle()), "reduce_scatter")
        N.check(N.lib().prism_direct_all_gather(ctypes.byref(d), 1, N.current_stream_handle()), "all_gather")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        allreduce()
        allreduce()
        f = read_flags()
        assert f[N.MAX_PEERS + 1] == 2 and f[0] == 3 * 1 + 3 and f[N.MAX_PEERS] == 0
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            allreduce()
        for _ in range(3):
            g.replay()
        f = read_flags()
    assert f[N.MAX_PEERS + 1] == 5 and f[0] == 3 * 4 + 3 and f[N.MAX_PEERS] == 0
    assert torch.equal(flat, want)
    N.check(N.lib().prism_direct_flags_free(fp.value), "prism_direct_flags_free")


def test_direct_allreduce_timeout_poisons_the_step_and_the_host_sees_it():
    """A peer that never arrives: the bounded wait (0.2 s here) gives up, sets the sticky slot, ORs
    PRISM_WS_STATUS_COLLECTIVE_TIMEOUT into the learner's status word and the pinned host word; the reduce / gather kernels
    and clip + Adam behind it then leave gradients and parameters alone, and ``poll_status`` / ``save`` / ``log`` raise."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import ctypes
    from prism_amd import _native as N
    from prism_amd import dist as pdist
    dev = "cuda:0"
    flat = torch.randn(4096, device=dev)
    peer = torch.full((4096,), 7.0, device=dev)           # a "peer" buffer whose owner never announces anything
    want = flat.clone()
    fp, fq = ctypes.c_void_p(0), ctypes.c_void_p(0)
    N.check(N.lib().prism_direct_flags_alloc(ctypes.byref(fp), None), "alloc")
    N.check(N.lib().prism_direct_flags_alloc(ctypes.byref(fq), None), "alloc")
    status, poison = pdist.StatusWords(), torch.zeros(1, dtype=torch.int32, device=dev)
    d = N.DirectDesc()
    d.world, d.rank, d.n = 2, 0, flat.numel()
    d.bufs[0], d.bufs[1], d.flags[0], d.flags[1] = flat.data_ptr(), peer.data_ptr(), fp.value, fq.value
    d.poison, d.host_status, d.wait_seconds = poison.data_ptr(), status.data_ptr(), 0.2
    N.check(N.lib().prism_direct_reduce_scatter(ctypes.byref(d), 1, N.current_stream_handle()), "reduce_scatter")
    N.check(N.lib().prism_direct_all_gather(ctypes.byref(d), 1, N.current_stream_handle()), "all_gather")
    torch.cuda.synchronize()
    assert int(poison.item()) == N.WS_STATUS_COLLECTIVE_TIMEOUT
    assert status.bits() == N.WS_STATUS_COLLECTIVE_TIMEOUT
    assert torch.equal(flat, want)                        # no partial sum was written
    out = (ctypes.c_uint32 * N.DIRECT_FLAG_WORDS)()
    N.check(N.lib().prism_direct_flags_read(fp.value, out, N.current_stream_handle()), "read")
    assert out[N.MAX_PEERS] == 1
    assert list((ctypes.c_uint32 * N.DIRECT_FLAG_WORDS).from_buffer_copy(bytes(out)))[1] == 0      # the peer never announced
    for p in (fp, fq):
        N.check(N.lib().prism_direct_flags_free(p.value), "free")

    # the learner's side: with the bit in its workspace, a data-parallel clip + Adam applies nothing, and the host raises
    from prism_amd.config import baseline_config
    from prism_amd.learner import Learner
    from prism_amd.synthetic import fill_replay
    cfg = baseline_config(2, device=dev, batch_size=32, experience_replay_capacity=2048)
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
    fill_replay(ln.experience_buffer, 2048, seed=0)
    ag = ln.agent
    ln.step(eager=True)
    torch.cuda.synchronize()
    ag.world = 2                                          # (as if data parallel: grad_scale = 1/2 arms the poison check)
    before, step0 = ag.flat.clone(), int(ag.optimizer.step_t.item())
    ag.workspace.view(torch.int32)[N.WS_STATUS_WORD] = N.WS_STATUS_COLLECTIVE_TIMEOUT
    ag._status._np[1] = 1
    ag._allreduce = lambda: 0.5
    with pytest.raises(pdist.CollectiveTimeout):
        ln.step(eager=True)                               # the poll at the top of the step
    ag.workspace.view(torch.int32)[N.WS_STATUS_WORD] = N.WS_STATUS_COLLECTIVE_TIMEOUT
    ag._launch_fused(ln.experience_buffer, ag._bind_fused(ln.experience_buffer))
    torch.cuda.synchronize()
    assert torch.equal(ag.flat, before) and int(ag.optimizer.step_t.item()) == step0
    with pytest.raises(pdist.CollectiveTimeout):
        ag.save("/tmp/prism_never_written")
