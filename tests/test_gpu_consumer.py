"""-m gpu parity of consumer.hip through the C ABI: reductions straight from bit-packed masks and the
measurement geometry (row N3) against oracle/consumer.py, oracle/geometry.py (numpy float64 restatements of
measurement.py:44-113,160-185,302-323).  Integer results bit-exact; float64 geometry within 1e-12 relative (numpy's dot
products may associate differently than the kernel's left-to-right sums)."""
import numpy as np
import pytest
import torch

from gpu_util import need_gpu, synth_pred
from oracle import consumer as oc
from oracle import geometry as og
from test_oracle_geometry import load_calib

pytestmark = pytest.mark.gpu


def _pack(masks_u8):
    """u8 0/1 [n,H,W] -> LSB-first bits [n,H,W/8] (the VTI_PACK_BITS layout)."""
    return np.packbits(masks_u8.astype(bool), axis=-1, bitorder="little")


def _blobs(rng, n, H, W):
    m = np.zeros((n, H, W), np.uint8)
    for i in range(n - 1):                                   # the last mask stays empty
        y, x = rng.integers(0, H - 40), rng.integers(0, W - 40)
        m[i, y:y + rng.integers(3, 150), x:x + rng.integers(3, 200)] = 1
        m[i] &= (rng.uniform(size=(H, W)) > 0.3).astype(np.uint8)
    return m


@pytest.mark.parametrize("H,W,H0,W0", [(736, 960, 960, 1280), (640, 640, 640, 640), (640, 640, 480, 360), (320, 352, 1000, 333)])
def test_mask_stats_bits_equals_resize_then_moments(H, W, H0, W0):
    need_gpu()
    import vti_amd
    eng = vti_amd.Engine("n", 2, H=H, W=W, max_batch=1)
    rng = np.random.default_rng(H0)
    masks = _blobs(rng, 6, H, W)
    bits = torch.from_numpy(_pack(masks)).cuda()
    stats = eng.mask_stats_bits(bits, H0, W0).cpu().numpy()
    for i in range(len(masks)):
        ref = oc.resize_nearest(masks[i], W0, H0) > 0
        ys, xs = np.nonzero(ref)
        exp = [len(xs), xs.sum(), ys.sum(), xs.min() if len(xs) else -1, xs.max() if len(xs) else -1]
        assert stats[i].tolist() == exp, (i, stats[i].tolist(), exp)


def test_bits_reductions_on_real_pipeline_output():
    """End to end on what vti_masks writes: stats and envelopes from the bit-packed masks == the round-1 kernels on the
    unpacked, frame-resized bitmaps == the numpy restatement of measurement.py."""
    need_gpu()
    import vti_amd
    H, W, H0, W0 = 736, 960, 960, 1280
    eng = vti_amd.Engine("n", 2, H=H, W=W, max_batch=2, dtype="fp16")
    eng.load_weights(vti_amd.random_weights(eng, 1), 0)
    rng = np.random.default_rng(8)
    B = 2
    pred = synth_pred(rng, B, 2, 32, eng.num_anchors, H=H, W=W, n_inst=14)
    proto = torch.from_numpy(rng.standard_normal((B, H // 4, W // 4, 32)).astype(np.float32)).half().cuda()
    dets, counts = eng.nms(torch.from_numpy(pred).cuda(), 0.25, 0.7, 300)
    bits, offsets = eng.masks(dets, counts, proto, "logit", "bits")
    u8, _ = eng.masks(dets, counts, proto, "logit", "u8")
    stats = eng.mask_stats_bits(bits, H0, W0)
    bm, _ = eng.mask_to_frame(u8, H0, W0)
    assert torch.equal(stats, eng.mask_stats(bm))
    off = offsets.cpu().tolist()
    cls = dets[..., 5].cpu().numpy()
    for c in (1, 0, -1):
        env = eng.envelope_bits(bits, offsets, dets, c, H0, W0).cpu().numpy()
        for b in range(B):
            n = off[b + 1] - off[b]
            sel = [off[b] + i for i in range(n) if c < 0 or int(cls[b, i]) == c]
            if sel:
                ref = oc.lower_envelope(oc.combine_masks([bm[s].cpu().numpy() for s in sel], H0, W0))
            else:
                ref = np.full((W0,), -1)
            assert np.array_equal(env[b], ref), (c, b, int((env[b] != ref).sum()))
            assert (env[b] >= 0).any() == bool(sel)


def test_pixels_to_world_on_the_reference_calibration():
    need_gpu()
    import vti_amd
    K, dist, R, t = load_calib()
    assert np.allclose(vti_amd.consumer.rodrigues([-0.8631369244225452, -0.3919482615538663, -1.3591256137314185]), R, atol=1e-15)
    n_c, d_c = og.compute_camera_plane(R, t)
    n2, d2 = vti_amd.consumer.compute_camera_plane(R, t)
    assert np.array_equal(n_c, n2) and d_c == d2
    rng = np.random.default_rng(0)
    uv = np.concatenate([rng.uniform([0, 0], [1280, 960], (500, 2)), [[636.148901113533, 422.3901781816556], [0, 0], [1279, 959]]])
    xyz, valid = vti_amd.consumer.pixels_to_world(uv, K, dist, R, t)
    xyz, valid = xyz.cpu().numpy(), valid.cpu().numpy()
    for i, (u, v) in enumerate(uv):
        ref = og.pixel_to_world_using_camera_plane(u, v, K, dist, R, t, n_c, d_c)
        assert (ref is not None) == bool(valid[i])
        if ref is not None:
            assert np.abs(xyz[i] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (i, xyz[i], ref)
    # the reference's scalar entry point, and its "None" branch: a ray parallel to the plane
    one = vti_amd.consumer.pixel_to_world_using_camera_plane(700.0, 500.0, K, dist, R, t, n_c, d_c)
    assert np.abs(one - og.pixel_to_world_using_camera_plane(700.0, 500.0, K, dist, R, t, n_c, d_c)).max() < 1e-13
    Rp = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])       # plane normal = (0,-1,0)... third column; ray through the principal point is parallel
    assert vti_amd.consumer.pixel_to_world_using_camera_plane(K[0, 2], K[1, 2], K, np.zeros(5), Rp, np.array([0.0, 0.1, 0.5])) is None
    # stitch width in mm as measurement.py:356-359 computes it
    a, _ = vti_amd.consumer.pixels_to_world(np.array([[600.0, 500.0], [640.0, 500.0]]), K, dist, R, t)
    w_mm = float(torch.linalg.norm(a[1] - a[0]).item()) * 1000.0
    pa = og.pixel_to_world_using_camera_plane(600.0, 500.0, K, dist, R, t, n_c, d_c)
    pb = og.pixel_to_world_using_camera_plane(640.0, 500.0, K, dist, R, t, n_c, d_c)
    assert abs(w_mm - float(np.linalg.norm(pb - pa)) * 1000.0) < 1e-9 and 0.5 < w_mm < 50


def test_kmeans1d2_matches_the_reference_loop():
    need_gpu()
    import vti_amd
    rng = np.random.default_rng(3)
    rows = [np.array([400.0, 402.5, 399.0, 520.0, 523.0, 518.5, 521.0]), np.array([5.0, 5.0, 5.0]), np.array([7.0]), np.array([]),
            np.array([1.0, 2.0]), rng.uniform(300, 760, 200),
            np.concatenate([rng.normal(420, 6, 150), rng.normal(600, 9, 141)]),          # > 128 per cluster: numpy's pairwise halves
            rng.normal(500, 1e-3, 600), np.round(rng.uniform(300, 700, 37))]
    max_n = max(len(r) for r in rows)
    vals = np.zeros((len(rows), max_n))
    for i, r in enumerate(rows):
        vals[i, :len(r)] = r
    counts = torch.tensor([len(r) for r in rows], dtype=torch.int32).cuda()
    labels, centers = vti_amd.engine.kmeans1d2(torch.from_numpy(vals).cuda(), counts)
    labels, centers = labels.cpu().numpy(), centers.cpu().numpy()
    for i, r in enumerate(rows):
        if len(r) == 0:
            assert np.isnan(centers[i]).all() and (labels[i] == 0).all()
            continue
        ref_l, (c0, c1) = og.kmeans_1d_two_clusters(r)
        assert labels[i, :len(r)].tolist() == ref_l.tolist(), i
        assert (labels[i, len(r):] == 0).all()
        assert centers[i, 0] == c0 and centers[i, 1] == c1, (i, centers[i], c0, c1)           # bit-exact: same summation order
    l1, c = vti_amd.consumer.kmeans_1d_two_clusters(rows[0])
    assert l1.tolist() == [0, 0, 0, 1, 1, 1, 1] and c == og.kmeans_1d_two_clusters(rows[0])[1]
