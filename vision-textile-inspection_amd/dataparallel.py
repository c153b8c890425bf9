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


_recv = {}      # (key, shape, dtype, device, world) -> the root's receive buffer [world * n, ...], allocated once


def gather_tensor(t, dst=0, group=None, key=None):
    """Gather equally-shaped tensors to `dst`; returns the concatenation there ([world * n, ...]), None elsewhere.
    With `key` the root receives into a buffer that is allocated once per (key, shape) and reused by every later call (the
    per-step gathers of bench.py: no allocation and no concatenation copy on the root; the result is overwritten by the next
    call with the same key)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    t = t.contiguous()
    if rank == dst:
        if key is None:
            buf = torch.empty((world * t.shape[0], *t.shape[1:]), dtype=t.dtype, device=t.device)
        else:
            k = (key, tuple(t.shape), t.dtype, t.device, world)
            buf = _recv.get(k)
            if buf is None:
                buf = _recv[k] = torch.empty((world * t.shape[0], *t.shape[1:]), dtype=t.dtype, device=t.device)
        parts = list(buf.chunk(world, 0)) if t.shape[0] else [torch.empty_like(t) for _ in range(world)]   # views: the gather lands in place
        dist.gather(t, parts, dst=dst, group=group)
        return buf
    dist.gather(t, None, dst=dst, group=group)
    return None


COMPACT_KEYS = ("dets", "counts", "xyxy", "offsets", "stats", "envelope")


def gather_detections(out, dst=0, group=None, keys=None, reuse=None):
    """Gather what the consumer needs from one predict_into() output set: dets f32 [b,max_det,6+nm], counts i32 [b], xyxy f32
    [b,max_det,4], offsets i32 [b+1] and -- when the caller computed them (Engine.mask_stats_bits / envelope_bits: SURVEY 8
    row N1) -- stats i64 [capacity,5] and envelope i32 [b,W0]: a few MB per rank instead of the mask buffer (capacity x H x W/8
    bytes: 210 MB per rank at 64 frames of 640x640, most of it dead slots).  Root gets a dict of concatenated tensors, others
    None.  Masks themselves travel through gather_live_masks when a consumer really wants the bitmaps."""
    keys = [k for k in (keys or COMPACT_KEYS) if k in out and out[k] is not None]
    # reuse = a caller-chosen tag (e.g. the double-buffer slot): receive buffers are then kept per (tag, key) instead of allocated per call
    got = {k: gather_tensor(out[k], dst, group, key=None if reuse is None else (reuse, k)) for k in keys}
    return got if dist.get_rank(group) == dst else None


def gather_live_masks(masks, offsets, dst=0, group=None):
    """Variable-length gather of the LIVE mask slots only: rank r sends masks[:offsets[-1]] (its instances), not its whole
    fixed-capacity buffer.  Sizes are exchanged first (one small all_gather, one host read per rank), then every peer sends
    exactly its live slots point to point -- 7 distinct xGMI links into the root.  Root returns (masks [sum_live, ...],
    live i64 [world]) with rank r's slots at [live[:r].sum(), live[:r+1].sum()); others None."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_live = offsets[-1:].to(torch.int64).clamp(min=0, max=masks.shape[0])     # never trust an uninitialised offsets buffer
    sizes = [torch.zeros_like(n_live) for _ in range(world)]
    dist.all_gather(sizes, n_live, group=group)
    live = torch.cat(sizes).cpu()
    if rank == dst:
        total = int(live.sum())
        buf = torch.empty((total, *masks.shape[1:]), dtype=masks.dtype, device=masks.device)
        reqs, at = [], 0
        for r in range(world):
            n = int(live[r])
            if r == dst:
                buf[at:at + n].copy_(masks[:n])
            elif n:
                reqs.append(dist.irecv(buf[at:at + n], src=r, group=group))
            at += n
        for q in reqs:
            q.wait()
        return buf, live
    n = int(live[rank])
    if n:
        dist.send(masks[:n].contiguous(), dst=dst, group=group)
    return None


def max_over_ranks(value, device, group=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
