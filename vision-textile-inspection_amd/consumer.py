"""Device-side mirror of measurement.py's mask post-processing.

Function names and argument meaning follow the reference so its call sites translate 1:1:
    get_instance_mask_as_bitmap(result, idx, h, w)        measurement.py:70-86
    combine_masks(mask_list, h, w)                         measurement.py:160-168
    fabric_lower_envelope(fabric_mask)                     measurement.py:170-185
    stitch_moments(mask, box)                              measurement.py:302-323
plus batched forms that keep everything on the GPU (one launch per reduction instead of the
reference's per-instance Python loops).
"""
import torch


def _engine_of(result):
    eng = getattr(result, "_engine", None)
    if eng is None:
        raise RuntimeError("result carries no engine handle; use consumer.attach(result, engine)")
    return eng


def attach(result, engine):
    result._engine = engine
    return result


def instance_bitmaps(engine, result, h, w):
    """All instances at once: (bitmaps u8 [N,h,w] on device, nonzero i32 [N]).  A4 batched."""
    if result.masks is None:
        dev = engine.device
        return torch.empty((0, h, w), dtype=torch.uint8, device=dev), torch.empty((0,), dtype=torch.int32, device=dev)
    return engine.mask_to_frame(result.masks.data_u8.contiguous(), h, w)


def get_instance_mask_as_bitmap(engine, result, idx, h, w):
    """measurement.py:70-86: bitmap of instance `idx` at frame size, or None when empty/missing."""
    try:
        bm, nz = engine.mask_to_frame(result.masks.data_u8[idx:idx + 1].contiguous(), h, w)
        return bm[0] if int(nz[0].item()) > 0 else None
    except Exception:
        return None


def combine_and_envelope(engine, bitmaps, select):
    """measurement.py:160-185 fused: OR of bitmaps[select] and its per-column lower envelope.
    -> (union u8 [h,w], envelope i32 [w]); (None, None) for an empty selection."""
    if len(select) == 0:
        return None, None
    return engine.union_envelope(bitmaps, select)


def stitch_moments(engine, bitmaps):
    """measurement.py:302-318 batched: i64 [N,5] = m00, m10, m01, min_col, max_col per bitmap."""
    return engine.mask_stats(bitmaps)


def stitch_meta_from_stats(stats_row, box):
    """measurement.py:303-323: (cx, cy, px_width, left_px, right_px) from one stats row with the
    reference's fall-backs to the (int) box when the mask is empty."""
    m00, m10, m01, mn, mx = (int(v) for v in stats_row)
    x1, y1, x2, y2 = box
    if m00 > 0:
        return float(m10 / m00), float(m01 / m00), float(mx - mn), float(mn), float(mx)
    return float((x1 + x2) / 2), float((y1 + y2) / 2), float(x2 - x1), float(x1), float(x2)
