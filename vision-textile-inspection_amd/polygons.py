"""`Results.masks.xy`: one polygon per instance, in ORIGINAL frame pixels (host side, off the hot path).

The reference reads `r.masks.xy[idx]` only as a fall-back when `masks.data` cannot be used
(Utils/check_model.py:185-188, Utils/check_stitch_distance.py:115: `cv2.fillPoly` of the polygon).  Ultralytics builds it with
`ops.masks2segments(masks.data)` -- `cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)`, the contour with the most points
("largest" strategy) -- followed by `ops.scale_coords` back to the frame.  cv2 is not available here, so the outer-border following
(Suzuki & Abe 1985, the algorithm behind findContours) and the CHAIN_APPROX_SIMPLE compression are restated below: 8-connected
foreground, outer borders only, traversal starting at the top-most left-most pixel of a component and leaving it downwards
(OpenCV's orientation for outer borders), vertices kept where the 8-direction chain code changes.  PARITY UNPINNED against OpenCV
itself; pinned by closed-form cases in tests/test_polygons.py (rectangles, single pixels, diagonals, holes, several blobs).
"""
import numpy as np

# 8-neighbourhood in counter-clockwise order on the screen (y grows downwards): E, NE, N, NW, W, SW, S, SE
_DY = (0, -1, -1, -1, 0, 1, 1, 1)
_DX = (1, 1, 0, -1, -1, -1, 0, 1)


def _trace_outer(img, sy, sx):
    """Border following from the start pixel (sy, sx), whose west neighbour is background.  `img` is zero padded by one
    pixel.  Returns the closed chain of border pixels [(y, x), ...] (no repeated end point), counter-clockwise on the screen."""
    # first neighbour met when turning CLOCKWISE from west (W, NW, N, NE, E, SE, S, SW)
    d0 = None
    for k in range(8):
        d = (4 - k) % 8
        if img[sy + _DY[d], sx + _DX[d]]:
            d0 = d
            break
    if d0 is None:
        return [(sy, sx)]                                   # an isolated pixel
    chain = [(sy, sx)]
    py, px = sy + _DY[d0], sx + _DX[d0]                      # "previous" point (i2, j2)
    cy, cx = sy, sx                                          # current point (i3, j3)
    first_prev = (py, px)
    while True:
        # search counter-clockwise around the current point, starting just after the direction of the previous point
        dprev = next(d for d in range(8) if (cy + _DY[d], cx + _DX[d]) == (py, px))
        for k in range(1, 9):
            d = (dprev + k) % 8
            ny, nx = cy + _DY[d], cx + _DX[d]
            if img[ny, nx]:
                break
        if (ny, nx) == (sy, sx) and (cy, cx) == first_prev:
            break                                            # back at the start, about to repeat the first step
        py, px, cy, cx = cy, cx, ny, nx
        chain.append((cy, cx))
        if len(chain) > 4 * img.size:                        # cannot happen on a finite image; keeps a bug from spinning
            raise RuntimeError("border following did not terminate")
    if len(chain) > 1 and chain[-1] == chain[0]:
        chain.pop()
    return chain


def _approx_simple(chain):
    """CHAIN_APPROX_SIMPLE: keep the points where the direction of the 8-connected chain changes."""
    n = len(chain)
    if n <= 2:
        return chain
    keep = []
    for i in range(n):
        y0, x0 = chain[i - 1]
        y1, x1 = chain[i]
        y2, x2 = chain[(i + 1) % n]
        if (y1 - y0, x1 - x0) != (y2 - y1, x2 - x1):
            keep.append(chain[i])
    return keep or [chain[0]]


def _label(mask):
    """8-connected components: (labels int32 [H,W], count).  scipy when present, otherwise a small two-pass union-find."""
    try:
        from scipy import ndimage
        lab, n = ndimage.label(mask, structure=np.ones((3, 3), np.uint8))
        return lab.astype(np.int32), int(n)
    except Exception:                                        # pragma: no cover - scipy ships in this image
        H, W = mask.shape
        lab = np.zeros((H, W), np.int32)
        parent = [0]

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a
        for y in range(H):
            for x in range(W):
                if not mask[y, x]:
                    continue
                nb = [lab[yy, xx] for yy, xx in ((y, x - 1), (y - 1, x - 1), (y - 1, x), (y - 1, x + 1))
                      if yy >= 0 and 0 <= xx < W and lab[yy, xx]]
                if not nb:
                    parent.append(len(parent))
                    lab[y, x] = len(parent) - 1
                else:
                    r = min(find(a) for a in nb)
                    lab[y, x] = r
                    for a in nb:
                        parent[find(a)] = r
        roots = {}
        for y in range(H):
            for x in range(W):
                if lab[y, x]:
                    lab[y, x] = roots.setdefault(find(lab[y, x]), len(roots) + 1)
        return lab, len(roots)


def find_external_contours(mask):
    """cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)[0] restated: list of int32 [n,2] (x, y) arrays, one per
    8-connected component, in raster order of their top-most left-most pixels."""
    m = (np.asarray(mask) != 0)
    if not m.any():
        return []
    lab, n = _label(m)
    img = np.zeros((m.shape[0] + 2, m.shape[1] + 2), np.uint8)
    out = []
    # first pixel of every component in raster order
    flat = lab.ravel()
    first = np.full(n + 1, -1, np.int64)
    idx = np.flatnonzero(flat)
    # np.unique returns the first occurrence index of each label in the (sorted-by-position) idx list
    labels, pos = np.unique(flat[idx], return_index=True)
    first[labels] = idx[pos]
    for l in labels[np.argsort(first[labels])]:
        img[1:-1, 1:-1] = (lab == l)
        sy, sx = divmod(int(first[l]), m.shape[1])
        chain = _approx_simple(_trace_outer(img, sy + 1, sx + 1))
        out.append(np.array([(x - 1, y - 1) for y, x in chain], dtype=np.int32).reshape(-1, 2))
    return out


def masks2segments(masks_u8, strategy="largest"):
    """Ultralytics ops.masks2segments: one float32 [n,2] (x, y) polygon per mask, [0,2] for an empty mask.
    strategy "largest": the contour with the most points (8.0.x default); "concat": all contours concatenated."""
    segs = []
    for m in np.asarray(masks_u8):
        c = find_external_contours(m)
        if not c:
            segs.append(np.zeros((0, 2), np.float32))
        elif strategy == "concat":
            segs.append(np.concatenate(c).astype(np.float32))
        else:
            segs.append(c[int(np.argmax([len(x) for x in c]))].astype(np.float32))
    return segs


def scale_coords(img1_shape, coords, img0_shape):
    """Ultralytics ops.scale_coords(img1_shape, coords, img0_shape, normalize=False): letterboxed (img1) -> frame (img0) pixels."""
    coords = np.array(coords, dtype=np.float32, copy=True)
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = ((img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2)
    if len(coords):
        coords[:, 0] -= np.float32(pad[0])
        coords[:, 1] -= np.float32(pad[1])
        coords /= np.float32(gain)
        coords[:, 0] = coords[:, 0].clip(0, img0_shape[1])
        coords[:, 1] = coords[:, 1].clip(0, img0_shape[0])
    return coords
