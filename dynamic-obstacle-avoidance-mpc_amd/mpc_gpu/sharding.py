"""Multi-GPU sharding of independent MPC instances: contiguous batch slices per rank, no data-path
collective; the only exchange is an all-gather of the per-scenario costs (RCCL over xGMI when the backend is "nccl",
gloo on CPU in the tests).  One process per GPU, `torch.distributed` is plumbing only."""
import torch


def shard_slice(total, rank, world):
    """Contiguous slice [lo, hi) of `total` instances owned by `rank` (C4: rank r gets [r*32768, (r+1)*32768))."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_costs(cost, world=None, group=None, async_op=False, out=None):
    """All-gather equally sized per-rank cost vectors into one [world * n] tensor (rank-major).
    Returns (tensor, work-handle-or-None)."""
    import torch.distributed as dist
    world = world or dist.get_world_size(group)
    if out is None:
        out = torch.empty(world * cost.numel(), dtype=cost.dtype, device=cost.device)
    work = dist.all_gather_into_tensor(out, cost.contiguous(), group=group, async_op=async_op)
    return out, work


def gather_costs_ragged(cost, sizes, group=None):
    """All-gather for unequal shard sizes (total not divisible by world): pads to the largest shard."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    m = max(sizes)
    buf = torch.zeros(m, dtype=cost.dtype, device=cost.device)
    buf[: cost.numel()] = cost
    out = torch.empty(world * m, dtype=cost.dtype, device=cost.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(world)])
