"""Multi-GPU sharding of the robot batch: robots are independent, so rank r owns a
contiguous slab and there is NO collective on the data path.  The only exchange is the optional
all-gather of the [B/G, 60] action slab (RCCL over xGMI; `nccl` backend is RCCL on ROCm) when the
gym side wants one concatenated action array on every rank.

Two schedules for that exchange (0.98 MB per rank at 4096 robots, 8 ranks):
  "ring"    -- `all_gather_into_tensor`: whatever RCCL picks for an all-gather, on this message size a ring: every slab
               travels G - 1 hops, each hop bound by ONE xGMI link;
  "direct"  -- every rank sends its slab straight to each of its G - 1 peers in one grouped batch of point-to-point
               operations (RCCL runs a `batch_isend_irecv` as one ncclGroup): on an MI355X node every pair of GPUs has its own
               xGMI link (7 links x ~153 GB/s per GPU), so all transfers run at once -- one hop, ~6.4 us of wire time per link
               (SURVEY.md section 5); the step is then launch / latency bound, where the ring pays seven serial hops.
Neither has been timed on a multi-GPU node from this repo (none was reachable); bench.py reports both whenever it runs with
more than one rank, so the first such run decides.
"""
import torch


def shard_bounds(total, rank, world):
    """Contiguous slab [lo, hi) of `total` robots for `rank` of `world`; sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GatherBuffers:
    """Persistent staging for all_gather_actions with uneven shards (`total` given): the padded local slab and the
    padded [world * rows, width] receive array, allocated once so that a timed loop does not allocate per call."""

    def __init__(self, total, world, width, dtype, device):
        self.total, self.world, self.width = int(total), int(world), int(width)
        self.rows = max(hi - lo for lo, hi in (shard_bounds(total, r, world) for r in range(world)))
        self.local = torch.zeros(self.rows, width, dtype=dtype, device=device)
        self.padded = torch.empty(world * self.rows, width, dtype=dtype, device=device)


def _global_rank(dist, group, r):
    """P2POp peers are GLOBAL ranks; `r` is a rank inside `group` (None = the default group, where the two coincide)."""
    return r if group is None else dist.get_global_rank(group, r)


def all_gather_actions(local_action, group=None, out=None, schedule="ring", total=None, buffers=None):
    """local_action: [b, 60] on this rank.  Returns the concatenated [sum b, 60] array on every rank.
    Equal b on all ranks unless `total` is given: then rank r holds shard_bounds(total, r, world) robots (slabs that
    differ by one row travel padded to the largest one; `buffers`: a GatherBuffers to stage them in without allocating).
    schedule: "ring" | "direct" (module docstring).  `group`: any process group; ranks below are ranks inside it."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if schedule not in ("ring", "direct"):
        raise ValueError("schedule must be 'ring' or 'direct'")
    width = local_action.shape[1]
    if total is None:
        rows, bounds = local_action.shape[0], None
    else:
        bounds = [shard_bounds(total, r, world) for r in range(world)]
        rows = max(hi - lo for lo, hi in bounds)
        if local_action.shape[0] != bounds[rank][1] - bounds[rank][0]:
            raise ValueError(f"rank {rank} holds {local_action.shape[0]} robots, its shard of {total} is {bounds[rank][1] - bounds[rank][0]}")
        if buffers is not None and (buffers.total != total or buffers.world != world or buffers.width != width):
            raise ValueError("GatherBuffers were sized for another (total, world, width)")
    local = local_action.contiguous()
    if bounds is not None and local.shape[0] < rows:   # pad the short slabs
        if buffers is not None:
            buffers.local[:local.shape[0]].copy_(local)
            local = buffers.local
        else:
            local = torch.cat([local, local.new_zeros(rows - local.shape[0], width)])
    if out is not None and bounds is None:
        padded = out
    elif buffers is not None and bounds is not None:
        padded = buffers.padded
    else:
        padded = torch.empty(world * rows, width, dtype=local.dtype, device=local.device)
    chunks = list(padded.chunk(world, dim=0))
    if schedule == "direct" and world > 1:
        chunks[rank].copy_(local)
        ops = []
        for step in range(1, world):   # peers in rotated order: rank r starts with r + 1, so no link is everybody's first target
            dst, src = (rank + step) % world, (rank - step) % world
            ops.append(dist.P2POp(dist.isend, local, _global_rank(dist, group, dst), group))
            ops.append(dist.P2POp(dist.irecv, chunks[src], _global_rank(dist, group, src), group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    elif local.is_cuda:
        dist.all_gather_into_tensor(padded, local, group=group)
    else:  # gloo (CPU tests) has no all_gather_into_tensor for every build: use the list form
        dist.all_gather(chunks, local, group=group)
    if bounds is None:
        return padded
    full = out if out is not None else torch.empty(total, width, dtype=local.dtype, device=local.device)
    for r, (lo, hi) in enumerate(bounds):
        full[lo:hi].copy_(chunks[r][:hi - lo])
    return full
