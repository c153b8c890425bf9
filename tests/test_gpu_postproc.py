"""Parity of the pre/post-processing HIP kernels (post.hip) with the CPU oracle, through the C ABI.

Bit-exact for the integer/index work (letterbox bytes, NMS kept set + order + row contents,
scale_boxes, nearest resize, OR, envelope, moments); mask bitmaps at IoU >= 0.999 per instance
(the coeff x proto dot product is summed in a different order than torch's sgemm, so a pixel whose
logit is ~1e-7 from the threshold may flip)."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import mask_iou, need_gpu, synth_pred
from oracle import consumer as oc
from oracle.letterbox import letterbox
from oracle.postproc import non_max_suppression, process_mask, scale_boxes

pytestmark = pytest.mark.gpu
HAND = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hand_cases.json")))


def _engine(nc=80, H=640, W=640, B=4, dtype="fp16"):
    import vti_amd
    from gpu_util import engine_and_oracle
    return engine_and_oracle("n", nc, H, W, B, dtype)[0]


def _check_nms(eng, pred, conf, iou, max_det, nc, agnostic=False):
    dets, counts = eng.nms(torch.from_numpy(pred).cuda(), conf, iou, max_det, agnostic)
    torch.cuda.synchronize()
    ref = non_max_suppression(pred, conf, iou, max_det, nc=nc, agnostic=agnostic)
    dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
    for b, r in enumerate(ref):
        assert counts[b] == len(r), (b, counts[b], len(r))
        assert np.array_equal(dets[b, :len(r)], r), f"frame {b}: rows differ"
        assert (dets[b, len(r):] == 0).all()
    return dets, counts


def test_nms_planted_instances_exact():
    need_gpu()
    eng = _engine()
    rng = np.random.default_rng(1)
    pred = synth_pred(rng, 4, 80, 32, 8400)
    _, counts = _check_nms(eng, pred, 0.25, 0.7, 300, 80)
    assert counts.tolist() == [50, 50, 50, 50]           # every planted instance survives, duplicates do not
    _check_nms(eng, pred, 0.20, 0.25, 200, 80)           # the reference's thresholds (config.py:71-73)
    _check_nms(eng, pred, 0.25, 0.7, 7, 80)              # max_det truncation
    _check_nms(eng, pred, 0.25, 0.7, 300, 80, agnostic=True)


def test_nms_edge_cases_exact():
    need_gpu()
    eng = _engine()
    rng = np.random.default_rng(2)
    # (a) no candidate at all / one frame empty, one not
    pred = synth_pred(rng, 2, 80, 32, 8400, n_inst=3)
    pred[0, 4:84] = 0.01
    _, counts = _check_nms(eng, pred, 0.25, 0.7, 300, 80)
    assert counts[0] == 0
    # (b) every anchor is a candidate (8400 > 8192: global-memory sort path), random overlapping boxes
    pred = synth_pred(rng, 2, 80, 32, 8400, n_inst=0, bg=0.9)
    pred[:, 4:84] = np.maximum(pred[:, 4:84], 0.3)
    _check_nms(eng, pred, 0.25, 0.5, 300, 80)
    # (c) tied scores: order falls back to anchor index (stable sort)
    pred = synth_pred(rng, 1, 80, 32, 8400, n_inst=20, dup=4)
    pred[:, 4:84] = np.round(pred[:, 4:84] * 4) / 4
    _check_nms(eng, pred, 0.25, 0.7, 300, 80)
    # (d) boxes that only touch / zero-area boxes (NaN IoU is "not suppressed")
    pred = synth_pred(rng, 1, 80, 32, 8400, n_inst=10, dup=2)
    pred[0, 2:4, :50] = 0
    pred[0, 4, :50] = 0.6
    _check_nms(eng, pred, 0.25, 0.7, 300, 80)


def test_nms_hand_case_and_two_classes():
    need_gpu()
    h = HAND["nms"]
    eng2 = _engine(nc=2, H=64, W=64, B=2)     # A = 64+16+4 = 84 anchors
    A = eng2.num_anchors
    pred = np.zeros((1, 4 + 2 + 32, A), np.float32)
    b = np.asarray(h["boxes_xyxy"], np.float32)
    n = len(b)
    pred[0, 0, :n], pred[0, 1, :n] = (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2
    pred[0, 2, :n], pred[0, 3, :n] = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    for a, (s, c) in enumerate(zip(h["scores"], h["classes"])):
        pred[0, 4 + c, a] = s
    dets, counts = _check_nms(eng2, pred, h["conf"], h["iou"], 300, 2)
    assert counts[0] == 4 and np.allclose(dets[0, :4, 4], np.asarray(h["scores"], np.float32)[h["keep_class_aware"]])
    dets, counts = _check_nms(eng2, pred, h["conf"], h["iou"], 300, 2, agnostic=True)
    assert counts[0] == 3


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
@pytest.mark.parametrize("mode", ["logit", "sigmoid"])
def test_masks_vs_process_mask(dtype, mode):
    """50 instances per frame at 640x640 (BASELINE config 5's post-processing stress)."""
    need_gpu()
    eng = _engine(B=4, dtype=dtype)
    rng = np.random.default_rng(3)
    B = 2
    pred = synth_pred(rng, B, 80, 32, 8400)
    proto = rng.standard_normal((B, 160, 160, 32)).astype(np.float32)
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    proto_d = torch.from_numpy(proto).to(tdt).cuda()
    dets, counts = eng.nms(torch.from_numpy(pred).cuda(), 0.25, 0.7, 300)
    masks, offsets = eng.masks(dets, counts, proto_d, mode, "u8")
    bits, offsets2 = eng.masks(dets, counts, proto_d, mode, "bits")
    torch.cuda.synchronize()
    import vti_amd
    assert torch.equal(vti_amd.unpack_bits(bits, 640), masks) and torch.equal(offsets, offsets2)   # packings agree
    off = offsets.cpu().numpy()
    assert off.tolist() == [0, 50, 100]
    worst, flips = 1.0, 0
    for b in range(B):
        d = dets[b, :counts[b]].cpu().numpy()
        ref = process_mask(proto_d[b].float().cpu().permute(2, 0, 1), d[:, 6:], d[:, :4], (640, 640), mode).numpy()
        got = masks[off[b]:off[b + 1]].cpu().numpy()
        assert set(np.unique(got)) <= {0, 1}
        for i in range(len(d)):
            worst = min(worst, mask_iou(got[i], ref[i]))
            flips += int((got[i] != (ref[i] > 0)).sum())
        assert ref.sum() > 0
    assert worst >= 0.999, worst
    assert flips <= 20, flips          # stray threshold-tie pixels over 100 masks x 409600 px


def test_masks_hand_case_and_capacity():
    need_gpu()
    h = HAND["process_mask"]
    eng = _engine(nc=2, H=64, W=64, B=2, dtype="fp32")
    proto = torch.zeros((1, 16, 16, 32))
    proto[..., 1] = 1.0
    dets = torch.zeros((1, 10, 38))
    dets[0, 0, :4] = torch.tensor(h["box_xyxy"], dtype=torch.float32)
    dets[0, 0, 4], dets[0, 0, 6 + 1] = 0.9, 1.0
    counts = torch.tensor([1], dtype=torch.int32)
    for mode, key in (("logit", "logit_rows"), ("sigmoid", "sigmoid_rows")):
        m, off = eng.masks(dets.cuda(), counts.cuda(), proto.cuda(), mode, "u8")
        lo, hi = h[key]
        exp = np.zeros((64, 64), np.uint8)
        exp[lo:hi, lo:hi] = 1
        assert np.array_equal(m[0].cpu().numpy(), exp), mode
    # capacity smaller than the detection count: extra instances are dropped, offsets still report them
    counts = torch.tensor([3], dtype=torch.int32)
    m, off = eng.masks(dets.cuda(), counts.cuda(), proto.cuda(), "logit", "u8", capacity=2)
    assert m.shape[0] == 2 and off.cpu().tolist() == [0, 3]
    m, off = eng.masks(dets.cuda(), torch.tensor([0], dtype=torch.int32).cuda(), proto.cuda(), "logit", "u8")
    assert m.shape[0] == 0 and off.cpu().tolist() == [0, 0]


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_masks_other_coefficient_count(dtype):
    """nm = 16 (a model trained with fewer prototypes): the generic-nm instantiation of the tile kernel, and the fp16 engine
    WITHOUT the split-coefficient table (it exists for nm = 32 only)."""
    need_gpu()
    import vti_amd
    eng = vti_amd.Engine("n", 3, H=320, W=320, max_batch=2, dtype=dtype, nm=16)
    eng.load_weights(vti_amd.random_weights(eng, seed=2), 0)
    rng = np.random.default_rng(5)
    B = 2
    pred = synth_pred(rng, B, 3, 16, eng.num_anchors, H=320, W=320, n_inst=9)
    proto = rng.standard_normal((B, 80, 80, 16)).astype(np.float32)
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    proto_d = torch.from_numpy(proto).to(tdt).cuda()
    dets, counts = eng.nms(torch.from_numpy(pred).cuda(), 0.25, 0.7, 300)
    masks, offsets = eng.masks(dets, counts, proto_d, "logit", "u8")
    bits, _ = eng.masks(dets, counts, proto_d, "logit", "bits")
    torch.cuda.synchronize()
    assert torch.equal(vti_amd.unpack_bits(bits, 320), masks)
    off = offsets.cpu().numpy()
    worst = 1.0
    for b in range(B):
        d = dets[b, :counts[b]].cpu().numpy()
        ref = process_mask(proto_d[b].float().cpu().permute(2, 0, 1), d[:, 6:], d[:, :4], (320, 320), "logit").numpy()
        got = masks[off[b]:off[b + 1]].cpu().numpy()
        assert len(d) == 9 and ref.sum() > 0
        for i in range(len(d)):
            worst = min(worst, mask_iou(got[i], ref[i]))
    assert worst >= 0.999, worst


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
@pytest.mark.parametrize("packing", ["bits", "u8"])
def test_masks_dirty_oversized_buffer(dtype, packing):
    """vti_masks into a caller buffer that is larger than the instance count and full of garbage (include/vti.h): the live
    slots come out exactly as into a fresh exact-size buffer (every byte no tile item covers is zeroed), the slots beyond the
    instance count are not touched.  Ragged 736x960 geometry: a partial last tile column and row."""
    need_gpu()
    eng = _engine(nc=2, H=736, W=960, B=2, dtype=dtype)
    rng = np.random.default_rng(17)
    B, A = 2, eng.num_anchors
    pred = synth_pred(rng, B, 2, 32, A, H=736, W=960, n_inst=12)
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    proto_d = torch.from_numpy(rng.standard_normal((B, 184, 240, 32)).astype(np.float32)).to(tdt).cuda()
    dets, counts = eng.nms(torch.from_numpy(pred).cuda(), 0.25, 0.7, 300)
    ref, off = eng.masks(dets, counts, proto_d, "logit", packing)
    total = int(off[-1])
    assert total >= 4
    wb = 960 if packing == "u8" else 960 // 8
    buf = torch.full((total + 5, 736, wb), 0xAB, dtype=torch.uint8, device="cuda")
    got, off2 = eng.masks(dets, counts, proto_d, "logit", packing, capacity=total + 5, masks=buf)
    torch.cuda.synchronize()
    assert torch.equal(off, off2) and got.data_ptr() == buf.data_ptr()
    assert torch.equal(got[:total], ref)
    assert bool((got[total:] == 0xAB).all())
    assert int(ref.sum()) > 0


@pytest.mark.parametrize("H0,W0,imgsz", [(960, 1280, 960), (480, 640, 640), (333, 517, 640), (1080, 1920, 640), (640, 640, 640),
                                           (1280, 1280, 640), (960, 1280, 640)])
def test_letterbox_bit_exact(H0, W0, imgsz):
    """(1280, 1280, 640) and (960, 1280, 640) are exact 2x downscales: OpenCV's INTER_LINEAR -> INTER_AREA substitution."""
    need_gpu()
    import vti_amd
    H, W = vti_amd.letterbox_shape(H0, W0, imgsz)
    eng = vti_amd.Engine("n", 2, H=H, W=W, max_batch=2)      # letterbox needs no weights
    fr = np.random.default_rng(H0).integers(0, 256, (2, H0, W0, 3), dtype=np.uint8)
    out = eng.letterbox(torch.from_numpy(fr).cuda()).cpu().numpy()
    for b in range(2):
        ref, g = letterbox(fr[b], imgsz)
        assert ref.shape == (H, W, 3)
        assert np.array_equal(out[b], ref), f"{(out[b] != ref).sum()} bytes differ"


def test_scale_boxes_exact():
    need_gpu()
    eng = _engine(nc=2, H=736, W=960, B=2)
    rng = np.random.default_rng(4)
    dets = np.zeros((2, 16, 38), np.float32)
    dets[..., :4] = rng.uniform(-20, 1000, (2, 16, 4))
    counts = np.array([16, 5], np.int32)
    got = eng.scale_boxes(torch.from_numpy(dets).cuda(), torch.from_numpy(counts).cuda(), 960, 1280).cpu().numpy()
    for b in range(2):
        ref = scale_boxes((736, 960), dets[b, :counts[b], :4], (960, 1280))
        assert np.array_equal(got[b, :counts[b]], ref)
        assert (got[b, counts[b]:] == 0).all()


def test_consumer_reductions_exact():
    """measurement.py:70-86,160-185,300-330 on device vs the numpy restatement, incl. the hand case."""
    need_gpu()
    eng = _engine(nc=2, H=736, W=960, B=2)
    rng = np.random.default_rng(5)
    n, H, W, H0, W0 = 7, 736, 960, 960, 1280
    masks = np.zeros((n, H, W), np.uint8)
    for i in range(n - 1):                       # blobs; the last mask stays empty
        y, x = rng.integers(50, 600), rng.integers(50, 800)
        masks[i, y:y + rng.integers(5, 120), x:x + rng.integers(5, 150)] = 1
        masks[i] &= (rng.uniform(size=(H, W)) > 0.2).astype(np.uint8)
    md = torch.from_numpy(masks).cuda()
    bm, nz = eng.mask_to_frame(md, H0, W0)
    ref = np.stack([oc.resize_nearest(m, W0, H0) > 0 for m in masks]).astype(np.uint8)
    assert np.array_equal(bm.cpu().numpy(), ref) and nz.cpu().tolist() == ref.reshape(n, -1).sum(1).tolist()
    assert nz[-1].item() == 0                     # -> the reference returns None for this instance
    sel = [0, 2, 3, 6]
    uni, env = eng.union_envelope(bm, sel)
    runi = oc.combine_masks([ref[i] for i in sel], H0, W0)
    assert np.array_equal(uni.cpu().numpy(), runi) and np.array_equal(env.cpu().numpy(), oc.lower_envelope(runi))
    stats = eng.mask_stats(bm).cpu().numpy()
    for i in range(n):
        ys, xs = np.nonzero(ref[i])
        exp = [len(xs), xs.sum(), ys.sum(), xs.min() if len(xs) else -1, xs.max() if len(xs) else -1]
        assert stats[i].tolist() == exp
    c = HAND["consumer"]
    m = torch.tensor(c["mask"], dtype=torch.uint8).cuda()[None]
    bm, _ = eng.mask_to_frame(m, 4, 6)
    _, env = eng.union_envelope(bm, [0])
    st = eng.mask_stats(bm).cpu().numpy()[0]
    assert env.cpu().tolist() == c["envelope"] and st.tolist() == [c["m00"], c["m10"], c["m01"], c["min_col"], c["max_col"]]


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
@pytest.mark.parametrize("H,W", [(224, 288), (640, 640)])
def test_masks_crowded_tiles_rounds_and_ragged_width(dtype, H, W):
    """The grouped tile kernel's corner cases in one frame set: 90 large overlapping boxes (more than 64 instances per frame: two
    list rounds; more than 16 per tile: several MFMA groups), a frame with 3 and a frame with none; W = 288 has a half tile column
    and 36-byte bit rows (4-byte aligned only); one instance carries coefficients beyond fp16 range (the power-of-two rescue of the
    split) and one has a zero-area box.  Both packings, both mask modes, against process_mask."""
    need_gpu()
    import vti_amd
    eng = _engine(nc=2, H=H, W=W, B=3, dtype=dtype)
    rng = np.random.default_rng(23)
    B, max_det = 3, 128
    dets = np.zeros((B, max_det, 38), np.float32)
    n = [90, 3, 0]
    for b in range(B):
        for i in range(n[b]):
            w, h = rng.uniform(0.3, 0.8) * W, rng.uniform(0.3, 0.8) * H
            x1, y1 = rng.uniform(0, W - w), rng.uniform(0, H - h)
            dets[b, i, :4] = [x1, y1, x1 + w, y1 + h]
            dets[b, i, 4], dets[b, i, 5] = 0.9 - 0.001 * i, i % 2
            dets[b, i, 6:] = rng.standard_normal(32)
    dets[0, 5, 6:] *= 1.0e5                                  # |c| >= 3e4: outside the half range of the split
    dets[0, 7, :4] = [40.0, 40.0, 40.0, 40.0]                # empty crop box: an all-zero mask
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    proto = torch.from_numpy(rng.standard_normal((B, H // 4, W // 4, 32)).astype(np.float32)).to(tdt).cuda()
    dd, cc = torch.from_numpy(dets).cuda(), torch.tensor(n, dtype=torch.int32).cuda()
    for mode in ("logit", "sigmoid"):
        masks, off = eng.masks(dd, cc, proto, mode, "u8")
        bits, off2 = eng.masks(dd, cc, proto, mode, "bits")
        torch.cuda.synchronize()
        assert off.cpu().tolist() == [0, 90, 93, 93] and torch.equal(off, off2)
        assert torch.equal(vti_amd.unpack_bits(bits, W), masks)
        worst = 1.0
        for b in range(2):
            d = dets[b, :n[b]]
            ref = process_mask(proto[b].float().cpu().permute(2, 0, 1), d[:, 6:], d[:, :4], (H, W), mode).numpy()
            got = masks[int(off[b]):int(off[b + 1])].cpu().numpy()
            for i in range(n[b]):
                if ref[i].sum() == 0:
                    assert got[i].sum() == 0, (b, i)
                else:
                    worst = min(worst, mask_iou(got[i], ref[i]))
        assert worst >= 0.999, (mode, worst)
        assert int(masks[7].sum()) == 0
