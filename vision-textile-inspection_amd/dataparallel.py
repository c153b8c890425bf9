"""Frame-level data parallelism: one process per GPU, torch.distributed (backend "nccl" is RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no parallelism at all; frames are independent, so the only
exchange steps are a root->peers scatter of uint8 frames and a peers->root gather of
fixed-capacity detections and bit-packed masks.  No collective runs inside the
model.  A one-root scatter uses 7 distinct point-to-point xGMI links, so it is link-parallel.
"""
import torch
import torch.distributed as dist


def shard_range(total, world_size, rank):
    """Contiguous, balanced [lo, hi) of `total` frames for `rank`."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def scatter_frames(frames, per_rank, shape_tail, device, src=0, group=None):
    """Root holds uint8 [world*per_rank, *shape_tail]; every rank returns its [per_rank, *shape_tail] shard."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = torch.empty((per_rank, *shape_tail), dtype=torch.uint8, device=device)
    if rank == src:
        if frames.shape[0] != world * per_rank:
            raise ValueError("scatter_frames: root batch must be world_size * per_rank frames")
        chunks = [c.contiguous() for c in frames.to(device).chunk(world, 0)]
        dist.scatter(out, chunks, src=src, group=group)
    else:
        dist.scatter(out, None, src=src, group=group)
    return out


def gather_tensor(t, dst=0, group=None):
    """Gather equally-shaped tensors to `dst`; returns the concatenation there, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.gather(t.contiguous(), parts, dst=dst, group=group)
        return torch.cat(parts, 0)
    dist.gather(t.contiguous(), None, dst=dst, group=group)
    return None


def gather_detections(out, dst=0, group=None):
    """Gather one predict_into() output set: dets f32 [b,max_det,6+nm], counts i32 [b], xyxy f32
    [b,max_det,4] and the fixed-capacity bit-packed masks + per-rank offsets.  Root gets a dict of
    concatenated tensors (mask slots of rank r start at r*capacity), others None."""
    keys = ("dets", "counts", "xyxy", "masks", "offsets")
    got = {k: gather_tensor(out[k], dst, group) for k in keys}
    return got if dist.get_rank(group) == dst else None


def max_over_ranks(value, device, group=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
