"""Data-parallel helpers (host logic; no device code).

Scheme: one process per GPU; replay capacity split across ranks, every rank samples
its own shard; parameters and Adam state replicated (same init seed); per step ONE all-reduce (sum)
of the flat gradient, scaled by 1/world inside the clip+Adam kernel, which then runs redundantly on
every rank so replicas stay bit-identical.  torch.distributed backend "nccl" is RCCL on ROCm.
"""
import torch
import torch.distributed as dist


def shard_capacity(total_capacity, world):
    """Per-rank ring capacity when `total_capacity` transitions are split across `world` GPUs."""
    return (int(total_capacity) + world - 1) // world


def rank_seeds(seed, rank):
    """(init_seed, replay_seed, tau_seed): the init seed is shared so replicas start identical, the
    sampling streams are decorrelated across ranks."""
    return int(seed), int(seed) + 7919 * int(rank), int(seed) + 104729 * int(rank)


def allreduce_grads(flat_grads, group=None):
    """Sum the flat gradient over ranks in place; returns the scale (1/world) the optimizer kernel
    must apply before the global-norm clip."""
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def assert_replicas_identical(flat_params, group=None):
    """Debug check: every rank holds bit-identical parameters (cheap: compares a checksum pair)."""
    world = dist.get_world_size(group)
    if world == 1:
        return True
    v = torch.stack([flat_params.double().sum(), flat_params.double().abs().sum()])
    lo, hi = v.clone(), v.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


class StatusWords:
    """Four uint32 words of pinned, device-mapped host memory (``prism_learner_desc.host_status``): the kernels store 1
    into word k when they raise sticky status bit k of the learner workspace, so the host can look for an abandoned grid
    barrier (bit 0) or a collective that gave up on a peer (bit 1) on EVERY step without synchronising with the device."""

    def __init__(self):
        self.t = torch.zeros(4, dtype=torch.int32).pin_memory()
        self._np = self.t.numpy()

    def data_ptr(self):
        return self.t.data_ptr()

    def bits(self):
        w = self._np
        return int(w[0] != 0) | (int(w[1] != 0) << 1) | (int(w[2] != 0) << 2) | (int(w[3] != 0) << 3)

    def clear(self):
        self._np[:] = 0


class CollectiveTimeout(RuntimeError):
    pass


class DirectAllReduce:
    """The step's gradient all-reduce without RCCL: two shots over peer-mapped buffers (``prism_direct_reduce_scatter`` /
    ``prism_direct_all_gather``, include/prism_hip.h).  Every rank pulls its 1/world slice of the
    flat gradient from all peers at once and sums it in rank order, then pulls the other slices from their owners: all
    replicas hold bit-identical sums.  ``config.collective = "direct"`` selects it; RCCL stays the default and the yardstick it is tested against.

    Memory types of every word that crosses devices:
      * gradient buffers -- ordinary (coarse-grained) allocations of torch's caching allocator, mapped into the peers with
        torch's CUDA-IPC tensor sharing (``hipIpcGetMemHandle`` / ``hipIpcOpenMemHandle`` underneath; needs
        ``HSA_ENABLE_IPC_MODE_LEGACY=0`` on this stack).  Peers only read them from kernels that start after the producing
        kernel has ended (direct.hip, "Memory types").
      * flag words -- UNCACHED device memory the LIBRARY allocates (``prism_direct_flags_alloc``:
        ``hipExtMallocWithFlags(hipDeviceMallocUncached)``) and exports as a raw IPC handle: peers store into them while
        the owner's kernel polls, which coarse-grained memory does not support.
    Peer access between the devices is enabled explicitly (``prism_direct_enable_peer``), the handles travel through
    ``all_gather_object`` of whatever process group is there.  Ranks on distinct devices synchronise with device flags on
    the stream (no host involvement, capturable); ranks that share a device (tests on a one-GPU box) must not spin on the
    device they share: the three barriers are then host-side (stream synchronize + group barrier) or the host paces
    announce / wait launches apart (``use_flags="paced"``).

    A flag wait that is not through after ``wait_seconds`` poisons the step (no clip + Adam is applied, direct.hip) and
    raises ``CollectiveTimeout`` at the next ``allreduce()`` / ``poll()`` -- one step late at most, without a device sync."""

    def __init__(self, flat_grads, group=None, use_flags=None, poison_ptr=None, status=None, wait_seconds=None):
        import ctypes
        import os
        import socket
        import torch.multiprocessing.reductions as red
        from prism_amd import _native as N
        self._N, self._ctypes = N, ctypes
        self.group, self.world, self.rank = group, dist.get_world_size(group), dist.get_rank(group)
        if self.world > N.MAX_PEERS:
            raise ValueError(f"direct all-reduce covers up to {N.MAX_PEERS} ranks of one node")
        self.flat = flat_grads
        dev = flat_grads.device
        self.status = status if status is not None else StatusWords()
        self._opened, self._flags_ptr = [], None
        with torch.cuda.device(dev):
            fp, handle = ctypes.c_void_p(0), ctypes.create_string_buffer(N.IPC_HANDLE_BYTES)
            N.check(N.lib().prism_direct_flags_alloc(ctypes.byref(fp), handle), "prism_direct_flags_alloc")
            self._flags_ptr = fp.value
        props = torch.cuda.get_device_properties(dev)
        where = (socket.gethostname(), getattr(props, "pci_bus_id", None), getattr(props, "uuid", None) and str(props.uuid),
                 dev.index)
        mine = (red.reduce_tensor(flat_grads.detach()), bytes(handle.raw), where)
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        if len({w[0] for _, _, w in everyone}) != 1:
            raise RuntimeError("direct all-reduce: all ranks must be on one node")
        distinct = len({w[1:] for _, _, w in everyone}) == self.world
        self.use_flags = bool(distinct if use_flags is None else use_flags)
        # "paced": device flags, but announce and wait are separate launches with a host barrier between them (ranks that
        # share a device; the flag words and phase numbers are exercised exactly as on the step's path)
        self.paced = use_flags == "paced"
        if self.paced:
            self.use_flags = False
        # the peers' devices as THIS process numbers them (a rank may see the devices in another order, or only its own)
        local = {}
        for i in range(torch.cuda.device_count()):
            pr = torch.cuda.get_device_properties(i)
            local[(getattr(pr, "pci_bus_id", None), getattr(pr, "uuid", None) and str(pr.uuid))] = i
        self._peers = []          # keeps the mapped tensors alive
        d = N.DirectDesc()
        d.world, d.rank, d.n = self.world, self.rank, flat_grads.numel()
        d.poison = poison_ptr
        d.host_status = self.status.data_ptr()
        d.wait_seconds = float(wait_seconds if wait_seconds is not None else os.environ.get("PRISM_DIRECT_TIMEOUT_S", 0.0))
        with torch.cuda.device(dev):
            for s, (rb, rh, w) in enumerate(everyone):
                if s == self.rank:
                    tb, fptr = flat_grads, self._flags_ptr
                else:
                    peer_dev = local.get(tuple(w[1:3]))
                    if peer_dev is not None and peer_dev != dev.index:
                        # another device this process can see: peer access is switched on explicitly before anything of
                        # it is mapped (a device it cannot see -- one visible device per rank -- is left to the lazy
                        # enabling of hipIpcOpenMemHandle)
                        N.check(N.lib().prism_direct_enable_peer(int(peer_dev)), "prism_direct_enable_peer")
                    tb = rb[0](*rb[1])
                    fp = ctypes.c_void_p(0)
                    N.check(N.lib().prism_direct_flags_open(rh, ctypes.byref(fp)), "prism_direct_flags_open")
                    fptr = fp.value
                    self._opened.append(fptr)
                self._peers.append(tb)
                d.bufs[s], d.flags[s] = tb.data_ptr(), fptr
        self._desc = d
        dist.barrier(group=group)          # every rank has mapped everybody before the first use

    def close(self):
        N = self._N
        for p in self._opened:
            N.lib().prism_direct_flags_close(p)
        self._opened = []
        if self._flags_ptr:
            N.lib().prism_direct_flags_free(self._flags_ptr)
            self._flags_ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # noqa: BLE001 -- interpreter shutdown
            pass

    def _host_barrier(self):
        torch.cuda.current_stream().synchronize()
        dist.barrier(group=self.group)

    def poll(self):
        """Raise if a flag wait of this rank has given up (reads pinned host memory: no device sync)."""
        if self.status.bits() & self._N.WS_STATUS_COLLECTIVE_TIMEOUT:
            raise CollectiveTimeout("direct all-reduce: a peer did not arrive in time; the step that hit it applied no update "
                                    "(replicas may have diverged: restore a checkpoint)")

    def allreduce(self, flat=None):
        """Sum ``self.flat`` over the ranks in place; returns 1 / world (the scale the optimizer kernel applies)."""
        N, ct = self._N, self._ctypes
        assert flat is None or flat.data_ptr() == self.flat.data_ptr(), "the all-reduce is bound to the buffer it was built for"
        self.poll()
        flg = int(self.use_flags)
        L, st = N.lib(), N.current_stream_handle

        def sync(phase):
            if flg:
                return
            if self.paced:
                N.check(L.prism_direct_phase(ct.byref(self._desc), phase, 1, st()), "prism_direct_phase")
            self._host_barrier()
            if self.paced:      # every peer has announced: the wait finds its flags set and never spins
                N.check(L.prism_direct_phase(ct.byref(self._desc), phase, 2, st()), "prism_direct_phase")

        with torch.cuda.device(self.flat.device):
            sync(1)
            N.check(L.prism_direct_reduce_scatter(ct.byref(self._desc), flg, st()), "prism_direct_reduce_scatter")
            sync(2)
            N.check(L.prism_direct_all_gather(ct.byref(self._desc), flg, st()), "prism_direct_all_gather")
            sync(3)
        return 1.0 / self.world

    def read_flags(self):
        """The own flag array as a list of ints (one device-to-host copy; diagnostics and tests)."""
        N, ct = self._N, self._ctypes
        out = (ct.c_uint32 * N.DIRECT_FLAG_WORDS)()
        with torch.cuda.device(self.flat.device):
            N.check(N.lib().prism_direct_flags_read(self._flags_ptr, out, N.current_stream_handle()), "prism_direct_flags_read")
        return list(out)

    def check_status(self):
        self.poll()
        if self.read_flags()[self._N.MAX_PEERS] != 0:
            raise CollectiveTimeout("direct all-reduce: a peer did not arrive in time")
