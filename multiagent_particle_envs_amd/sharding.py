"""Multi-GPU: worlds shard by batch index, one process per GPU, no collective on the step path.

World b of a global batch lives on rank b // (B_global / world_size) (contiguous slices).  The only
communication is bookkeeping around the timed region: a barrier on each side and a MAX-reduce of the
elapsed time (RCCL when the process group is nccl, gloo in the CPU tests).  RNG streams are indexed
by global world number (`world.world_offset`), so G shards reproduce one big batch bit-for-bit.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world_size):
    """(offset, count) of rank's contiguous slice; counts differ by at most one."""
    base, rem = divmod(int(global_batch), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def barrier(device=None):
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def reduce_max(value, device="cpu"):
    """MAX over ranks of a Python float (elapsed seconds)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def whole_job_throughput(local_units, local_seconds, device="cpu"):
    """Units all ranks processed / slowest rank's time (the bench contract's `value`)."""
    return reduce_sum(local_units, device) / reduce_max(local_seconds, device)
