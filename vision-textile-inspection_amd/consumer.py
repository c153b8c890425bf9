"""Device-side mirror of measurement.py's mask post-processing.

Function names and argument meaning follow the reference so its call sites translate 1:1:
    get_instance_mask_as_bitmap(result, idx, h, w)        measurement.py:70-86
    combine_masks(mask_list, h, w)                         measurement.py:160-168
    fabric_lower_envelope(fabric_mask)                     measurement.py:170-185
    stitch_moments(mask, box)                              measurement.py:302-323
plus batched forms that keep everything on the GPU (one launch per reduction instead of the
reference's per-instance Python loops).
"""
import numpy as np
import torch

from . import engine as _engine_mod


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


# ---- measurement geometry ---------------------------------------------------
def rodrigues(rvec):
    """cv2.Rodrigues(rvec)[0] (measurement.py:139): R = cos(th) I + (1 - cos(th)) r r^T + sin(th) [r]x, th = |rvec|."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    th = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if th < 2.220446049250313e-16:
        return np.eye(3)
    c, s = np.cos(th), np.sin(th)
    k = r / th
    rrt = np.outer(k, k)
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=np.float64)
    return c * np.eye(3) + (1 - c) * rrt + s * kx


def compute_camera_plane(R, t):
    """measurement.py:44-48."""
    n_c = np.asarray(R, dtype=np.float64)[:, 2].astype(np.float64)
    return n_c, -float(n_c.dot(np.asarray(t, dtype=np.float64)))


def pixels_to_world(uv, K, dist, R, t):
    """Batched pixel_to_world_using_camera_plane (measurement.py:50-65): uv [n,2] (device tensor or array) ->
    (xyz f64 [n,3] device tensor in world metres, valid i32 [n]; valid == 0 where the reference returns None)."""
    if not isinstance(uv, torch.Tensor):
        uv = torch.as_tensor(np.asarray(uv, dtype=np.float64))
    if not uv.is_cuda:
        uv = uv.cuda()
    return _engine_mod.pixels_to_world(uv.reshape(-1, 2), K, dist, R, t)


def pixel_to_world_using_camera_plane(u, v, K, dist, R, t, n_c=None, d_c=None):
    """measurement.py:50-65, one point (n_c / d_c are recomputed from R, t on the device; accepted for signature parity)."""
    try:
        xyz, valid = pixels_to_world(np.array([[float(u), float(v)]]), K, dist, R, t)
        return xyz[0].cpu().numpy() if int(valid[0].item()) else None
    except Exception:
        return None


def kmeans_1d_two_clusters(values, max_iters=10):
    """measurement.py:88-113 on the device (one frame): -> (labels int array, (c0, c1))."""
    vals = np.asarray(values, dtype=np.float64).ravel()
    n = vals.size
    v = torch.zeros((1, max(n, 1)), dtype=torch.float64)
    v[0, :n] = torch.from_numpy(vals)
    labels, centers = _engine_mod.kmeans1d2(v.cuda(), torch.tensor([n], dtype=torch.int32).cuda(), max_iters)
    c = centers[0].cpu().numpy()
    return labels[0, :n].cpu().numpy().astype(int), (float(c[0]), float(c[1]))
