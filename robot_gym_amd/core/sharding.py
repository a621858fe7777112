"""Multi-GPU sharding of the robot batch: robots are independent, so rank r owns a
contiguous slab and there is NO collective on the data path.  The only exchange is the optional
all-gather of the [B/G, 60] action slab (RCCL over xGMI; `nccl` backend is RCCL on ROCm) when the
gym side wants one concatenated action array on every rank."""
import torch


def shard_bounds(total, rank, world):
    """Contiguous slab [lo, hi) of `total` robots for `rank` of `world`; sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_actions(local_action, group=None, out=None):
    """local_action: [b, 60] on this rank (equal b on all ranks).  Returns [world*b, 60]."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty(world * local_action.shape[0], local_action.shape[1], dtype=local_action.dtype, device=local_action.device)
    if local_action.is_cuda:
        dist.all_gather_into_tensor(out, local_action.contiguous(), group=group)
    else:  # gloo (CPU tests) has no all_gather_into_tensor for every build: use the list form
        chunks = list(out.chunk(world, dim=0))
        dist.all_gather(chunks, local_action.contiguous(), group=group)
    return out
