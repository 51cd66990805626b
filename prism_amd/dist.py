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
