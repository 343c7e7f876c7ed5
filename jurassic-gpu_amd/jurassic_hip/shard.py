"""Ray sharding across the GPUs of one node.

Rays are independent, so rank r of W owns the contiguous range
[r*N/W, (r+1)*N/W) and the only exchange is the final gather of obs.rad
(and optionally tau) to rank 0 -- one RCCL gather over xGMI, each peer sending
straight to the root.
"""
import torch
import torch.distributed as dist


def ray_range(rank, world, n):
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def gather_rows(local, counts, dst=0, group=None):
    """Gather row blocks of unequal length to `dst`; returns the concatenation
    on dst, None elsewhere.  `counts[r]` = rows owned by rank r."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    width = local.shape[1:]
    nmax = max(counts)
    pad = local
    if local.shape[0] < nmax:            # dist.gather needs equal shapes
        pad = torch.zeros((nmax,) + tuple(width), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
