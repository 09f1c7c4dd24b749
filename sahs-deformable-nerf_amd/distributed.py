"""Ray sharding across GPUs (one process per GPU, torch.distributed: backend "nccl" = RCCL on ROCm).

Rays are independent: rank r of N renders the contiguous block
[r*R/N, (r+1)*R/N) of the flattened ray list with replicated weights and per-frame conditioning,
nothing is exchanged while rendering, and ONE all-gather of the per-ray outputs (36 floats/ray:
the reference's 8-tuple) assembles the frame on every rank.  On a fully connected xGMI node the
all-gather moves 4.7 MB per rank per 512x512 frame -- microseconds against the render.
The functions are backend-agnostic (the CPU tests run them over gloo).
"""
import os

import torch
import torch.distributed as dist

OUT_COLUMNS = 36   # rgb_c 15, disp_c, acc_c, rgb_f 15, disp_f, acc_f, w_bg, depth_f


def shard_bounds(num_rays, world, rank):
    """Contiguous, balanced, order-preserving: concatenating shards 0..world-1 restores the ray order."""
    return rank * num_rays // world, (rank + 1) * num_rays // world


def pack_outputs(outs):
    """8-tuple (flat shapes) -> (n, 36) rows."""
    cols = [o if o.dim() == 2 else o[:, None] for o in outs]
    return torch.cat(cols, dim=1)


def unpack_outputs(rows):
    """(n, 36) rows -> the reference's 8-tuple (flat shapes)."""
    r = rows
    return (r[:, 0:15], r[:, 15], r[:, 16], r[:, 17:32], r[:, 32], r[:, 33], r[:, 34], r[:, 35])


def all_gather_rows(rows, num_rays, group=None):
    """Gather per-rank row blocks (shard_bounds order) into the full (num_rays, C) tensor on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return rows
    sizes = [shard_bounds(num_rays, world, r)[1] - shard_bounds(num_rays, world, r)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        full = torch.empty(num_rays, rows.shape[1], dtype=rows.dtype, device=rows.device)
        dist.all_gather_into_tensor(full, rows.contiguous(), group=group)
        return full
    mx = max(sizes)   # ragged: pad to the largest shard, gather, trim
    pad = torch.zeros(mx, rows.shape[1], dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    full = torch.empty(world * mx, rows.shape[1], dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(full, pad, group=group)
    return torch.cat([full[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def all_gather_rows_inplace(full, num_rays, group=None):
    """``full`` is the (num_rays, C) frame buffer of which THIS rank has filled rows [lo, hi) = shard_bounds(...): one all-gather
    completes it on every rank.  Equal shards: the send buffer is the rank's own slice of the receive buffer (the in-place form NCCL /
    RCCL document -- sendbuff == recvbuff + rank * sendcount -- and gloo accepts), so no staging copy exists; ragged shards go through
    the padded gather of all_gather_rows."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return full
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(num_rays, world, rank)
    if num_rays % world == 0 and full.is_contiguous():
        # SAHS_ALLGATHER_OUT_OF_PLACE=1: send from a copy of the slice instead (an escape hatch should a collective library ever reject
        # the aliasing; costs one copy of R/N x 36 floats)
        send = full[lo:hi].clone() if os.environ.get("SAHS_ALLGATHER_OUT_OF_PLACE") else full[lo:hi]
        dist.all_gather_into_tensor(full, send, group=group)
        return full
    full.copy_(all_gather_rows(full[lo:hi].contiguous(), num_rays, group))
    return full


def render_rows_sharded(render_block, num_rays, device, group=None, columns=OUT_COLUMNS):
    """The ray-sharded frame: allocate the (num_rays, columns) frame buffer, let ``render_block(lo, hi, rows)`` fill
    THIS rank's contiguous ray block [lo, hi) in place (``rows`` = the buffer's row slice; draws must be keyed by global ray index so
    the frame does not depend on the number of ranks), then one all-gather.  Without an initialised process group: one block, no
    collective.  train_utils.run_one_iter_of_nerf(_shard=...) and bench.py's N > 1 step are this function."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(num_rays, world, rank)
    full = torch.empty(num_rays, columns, dtype=torch.float32, device=device)
    if hi > lo:
        render_block(lo, hi, full[lo:hi])
    return all_gather_rows_inplace(full, num_rays, group)


def render_sharded(render_fn, num_rays, group=None):
    """render_fn(lo, hi) -> 8-tuple for rays [lo, hi); returns the full-frame 8-tuple on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(num_rays, world, rank)
    rows = pack_outputs(render_fn(lo, hi))
    if world > 1:
        rows = all_gather_rows(rows, num_rays, group)
    return unpack_outputs(rows)


def all_reduce_gradients(params, group=None, average=True):
    """Data-parallel training over rays: ONE all-reduce of all parameter gradients as a single flat
    bucket (2,775,633 fp32 = 11.1 MB for AudioFaceModel; on xGMI a ring moves 2*(N-1)/N of that per link, ~0.13 ms at 8 GPUs),
    issued after backward -- the HIP backward produces every gradient at once (one flat buffer), so there is nothing to
    overlap bucket by bucket.  Parameters without a gradient contribute zeros."""
    params = [p for p in params if p.requires_grad]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 or not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= world
    o = 0
    for p in params:
        n = p.numel()
        g = flat[o:o + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        o += n


def shard_batch(num_rays, group=None):
    """This rank's slice of a training batch's ray list (every rank draws the same batch from a shared seed)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(num_rays, world, rank)
    return slice(lo, hi)
