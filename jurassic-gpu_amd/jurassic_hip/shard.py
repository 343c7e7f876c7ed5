"""Ray sharding across the GPUs of one node.

Rays are independent, so rank r of W owns the contiguous range
[r*N/W, (r+1)*N/W) of ONE global ray set and the only exchange is the final
gather of obs.rad (and optionally tau) to rank 0: every peer sends its block
straight to the root (one xGMI link each on RCCL), the root receives each
block in place -- no padding, no ring.  The reference has no counterpart: its
device loop hands the same package to every GPU (GPUdrivers.cu:344-358).

bench.py and tests/test_distributed_gloo.py both go through these functions.
"""
import torch
import torch.distributed as dist


def ray_range(rank, world, n):
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def ray_counts(world, n):
    return [ray_range(r, world, n)[1] - ray_range(r, world, n)[0] for r in range(world)]


def balanced_ranges(ctl, atm, geom, world):
    """Contiguous ranges [(lo, hi)] * world of a ray set with equal estimated LINE-OF-SIGHT POINTS instead of equal ray
    counts (a tangent-height scan in its natural order has 393 .. 122 points per ray, so equal
    counts leave the rank with the high tangent altitudes idle half of the time).  Host arithmetic of the library
    (jur_balance_rays: no GPU, no model); for ray sets in random order -- bench.py's -- ray_range is balanced already."""
    from . import lib
    b = lib.balance_rays(ctl, atm, geom, world)
    return [(b[k], b[k + 1]) for k in range(world)]


def gather_rows(local, counts, dst=0, group=None, out=None):
    """Gather row blocks of unequal length to `dst`.

    `counts[r]` = rows owned by rank r (its block is `local`, shape (counts[r], ...)).
    On dst returns the concatenation in rank order -- written into `out` when given
    (shape (sum(counts), ...), reused from call to call) -- and None elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if len(counts) != world or local.shape[0] != counts[rank]:
        raise ValueError("gather_rows: counts %r do not describe this rank's block of %d rows" % (counts, local.shape[0]))
    if world == 1:
        if out is None:
            return local
        out.copy_(local)
        return out
    if rank != dst:
        if counts[rank]:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), dst, group)]):
                req.wait()
        return None
    total = sum(counts)
    if out is None:
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    elif out.shape[0] != total or out.shape[1:] != local.shape[1:]:
        raise ValueError("gather_rows: out has shape %r, need (%d, ...)" % (tuple(out.shape), total))
    ops, lo = [], 0
    for r, c in enumerate(counts):
        if r == dst:
            out[lo:lo + c].copy_(local)
        elif c:
            ops.append(dist.P2POp(dist.irecv, out[lo:lo + c], r, group))
        lo += c
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out
