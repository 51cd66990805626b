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


class DirectAllReduce:
    """The step's gradient all-reduce without RCCL: two shots over peer-mapped buffers (``prism_direct_reduce_scatter`` /
    ``prism_direct_all_gather``, include/prism_hip.h).  Every rank pulls its 1/world slice of the
    flat gradient from all peers at once and sums it in rank order, then pulls the other slices from their owners: all
    replicas hold bit-identical sums.  ``config.collective = "direct"`` selects it; RCCL stays the default and the yardstick it is tested against.

    The peers' buffers are mapped with torch's CUDA-IPC tensor sharing (``hipIpcGetMemHandle`` / ``hipIpcOpenMemHandle``
    underneath; needs ``HSA_ENABLE_IPC_MODE_LEGACY=0`` on this stack), the handles travel through ``all_gather_object`` of
    whatever process group is there.  Ranks on distinct devices synchronise with device flags on the stream (no host
    involvement, capturable); ranks that share a device (tests on a one-GPU box) must not spin on the device they share:
    the three barriers are then host-side (stream synchronize + group barrier)."""

    def __init__(self, flat_grads, group=None, use_flags=None):
        import ctypes
        import socket
        import torch.multiprocessing.reductions as red
        from prism_amd import _native as N
        self._N, self._ctypes = N, ctypes
        self.group, self.world, self.rank = group, dist.get_world_size(group), dist.get_rank(group)
        if self.world > N.MAX_PEERS:
            raise ValueError(f"direct all-reduce covers up to {N.MAX_PEERS} ranks of one node")
        self.flat = flat_grads
        dev = flat_grads.device
        self.flags = torch.zeros(N.MAX_PEERS + 2, dtype=torch.int32, device=dev)
        props = torch.cuda.get_device_properties(dev)
        where = (socket.gethostname(), getattr(props, "pci_bus_id", None), getattr(props, "uuid", None) and str(props.uuid),
                 dev.index)
        mine = (red.reduce_tensor(flat_grads.detach()), red.reduce_tensor(self.flags), where)
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        if len({w[0] for _, _, w in everyone}) != 1:
            raise RuntimeError("direct all-reduce: all ranks must be on one node")
        distinct = len({w[1:] for _, _, w in everyone}) == self.world
        self.use_flags = bool(distinct if use_flags is None else use_flags)
        if self.use_flags and not distinct:
            raise RuntimeError("device-flag synchronisation needs one device per rank")
        self._peers = []          # keeps the mapped tensors alive
        d = N.DirectDesc()
        d.world, d.rank, d.n = self.world, self.rank, flat_grads.numel()
        for s, (rb, rf, _) in enumerate(everyone):
            if s == self.rank:
                tb, tf = flat_grads, self.flags
            else:
                tb, tf = rb[0](*rb[1]), rf[0](*rf[1])
                if tb.device != dev:
                    tb[:1].to(dev)        # (first peer-to-peer copy: the runtime enables peer access between the two devices)
            self._peers.append((tb, tf))
            d.bufs[s], d.flags[s] = tb.data_ptr(), tf.data_ptr()
        self._desc = d
        dist.barrier(group=group)          # every rank has mapped everybody before the first use

    def _host_barrier(self):
        torch.cuda.current_stream().synchronize()
        dist.barrier(group=self.group)

    def allreduce(self, flat=None):
        """Sum ``self.flat`` over the ranks in place; returns 1 / world (the scale the optimizer kernel applies)."""
        N, ct = self._N, self._ctypes
        assert flat is None or flat.data_ptr() == self.flat.data_ptr(), "the all-reduce is bound to the buffer it was built for"
        flg = int(self.use_flags)
        with torch.cuda.device(self.flat.device):
            if not flg:
                self._host_barrier()
            N.check(N.lib().prism_direct_reduce_scatter(ct.byref(self._desc), flg, N.current_stream_handle()),
                    "prism_direct_reduce_scatter")
            if not flg:
                self._host_barrier()
            N.check(N.lib().prism_direct_all_gather(ct.byref(self._desc), flg, N.current_stream_handle()),
                    "prism_direct_all_gather")
            if not flg:
                self._host_barrier()
        return 1.0 / self.world

    def check_status(self):
        if int(self.flags[self._N.MAX_PEERS].item()) != 0:
            raise RuntimeError("direct all-reduce: a peer did not arrive within 2 s")
