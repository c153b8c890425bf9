"""End-to-end predict() through the YOLO/Results mirror of the reference protocol
(measurement.py:145,208-211,242-245; Utils/check_model.py:170-209) against the oracle pipeline."""
import numpy as np
import pytest
import torch

from gpu_util import frames_u8, mask_iou, need_gpu
from oracle import consumer as oc
from oracle.letterbox import letterbox
from oracle.model import OracleModel
from oracle.postproc import non_max_suppression, process_mask, scale_boxes

pytestmark = pytest.mark.gpu


def _oracle_predict(blob, frame, imgsz, conf, iou, max_det, nc, mode):
    lb, g = letterbox(frame, imgsz)
    om = OracleModel(blob, g["H"], g["W"], mode)
    pred, proto = om.forward_u8(lb[None], swap_rb=True)
    det = non_max_suppression(pred.numpy(), conf, iou, max_det, nc=nc)[0]
    if len(det) == 0:
        return det, None, None
    masks = process_mask(proto[0], det[:, 6:], det[:, :4], (g["H"], g["W"]), "logit").numpy()
    return det, masks, scale_boxes((g["H"], g["W"]), det[:, :4], frame.shape[:2])


def _calibrated_model(vti_amd, nc, dtype, frame, imgsz, conf, target=400):
    """Random nets with the stock class prior detect nothing; shift the class bias so that about
    `target` anchors clear `conf` (done with the GPU path only)."""
    m0 = vti_amd.YOLO(None, scale="n", nc=nc, seed=1, cls_bias=0.0, dtype=dtype, max_batch=2)
    H, W = vti_amd.letterbox_shape(*frame.shape[:2], imgsz)
    eng = m0._engine(H, W, 1)
    x = torch.from_numpy(frame[None]).cuda()
    inp = x if frame.shape[:2] == (H, W) else eng.letterbox(x)
    pred, _ = eng.forward(inp)
    p = pred[0, 4:4 + nc].amax(0).clamp(1e-6, 1 - 1e-6)
    logit = torch.log(p / (1 - p))
    top = torch.topk(logit, target + 1).values
    kth = 0.5 * (top[-2].item() + top[-1].item())      # between two anchors, so that none sits exactly ON the threshold
    bias = float(np.log(conf / (1 - conf)) - kth)
    return vti_amd.YOLO(None, scale="n", nc=nc, seed=1, cls_bias=bias, dtype=dtype, max_batch=2)


@pytest.mark.parametrize("dtype", ["fp32", "h2", "fp16"])
def test_predict_reference_call_shape(dtype):
    """The reference's own call: 1280x960 frame, conf 0.20, iou 0.25, max_det 200, imgsz 960, nc=2
    (measurement.py:208-210, config.py:69-73)."""
    need_gpu()
    import vti_amd
    frame = frames_u8(1, 960, 1280, seed=21)[0]
    model = _calibrated_model(vti_amd, 2, dtype, frame, 960, 0.20)
    results = model.predict(frame, verbose=False, conf=0.20, iou=0.25, max_det=200, imgsz=960)
    r = results[0]
    assert len(results) == 1 and r.boxes is not None
    det, omasks, oxyxy = _oracle_predict(model._blob, frame, 960, 0.20, 0.25, 200, 2, "fp32" if dtype == "h2" else dtype)
    n = len(r.boxes)
    assert n > 0 and r.masks is not None and r.masks.data.shape == (n, 736, 960) and r.masks.data.dtype == torch.float32
    cls, xyxy, conf = r.boxes.cls.cpu().numpy(), r.boxes.xyxy.cpu().numpy(), r.boxes.conf.cpu().numpy()
    assert (np.diff(conf) <= 0).all() and xyxy.min() >= 0 and xyxy[:, [0, 2]].max() <= 1280 and xyxy[:, [1, 3]].max() <= 960
    if dtype in ("fp32", "h2"):     # the two engines that meet the north-star gate: same kept set, boxes, masks
        assert n == len(det) and np.array_equal(cls, det[:, 5])
        assert np.abs(xyxy - oxyxy).max() < (5e-3 if dtype == "fp32" else 2e-2) and np.abs(xyxy - oxyxy).max() / 1280 < 1e-3
        assert np.abs(conf - det[:, 4]).max() < 1e-4
        got = r.masks.data.cpu().numpy()
        assert min(mask_iou(got[i], omasks[i]) for i in range(n)) >= 0.999
    else:   # fp16 storage vs fp16-emulating oracle: near-tie NMS decisions may differ, so match by box
        matched = 0
        for i in range(len(det)):
            j = np.abs(xyxy - oxyxy[i]).max(1).argmin()
            if np.abs(xyxy[j] - oxyxy[i]).max() < 4.0 and cls[j] == det[i, 5]:
                matched += 1
        assert matched >= 0.9 * len(det), (matched, len(det), n)
    # consumer side, exactly as measurement.py:249-330 walks the results
    keep, ib = oc.roi_keep(xyxy, 960, 1280)
    eng = model._engine(736, 960, 1)
    bm, nz = vti_amd.consumer.instance_bitmaps(eng, r, 960, 1280)
    assert bm.shape == (n, 960, 1280)
    first = vti_amd.consumer.get_instance_mask_as_bitmap(eng, r, 0, 960, 1280)
    ref0 = oc.instance_bitmap(r.masks.data[0].cpu().numpy(), 960, 1280)
    assert (first is None) == (ref0 is None) and (first is None or np.array_equal(first.cpu().numpy(), ref0))


def test_fp16_engine_drift_against_the_fp32_oracle():
    """The BENCHMARKED dtype against the fp32 CPU oracle, end to end (boxes AND masks), with the tolerance it actually meets written
    down.  north_star asks IoU >= 0.999 / |d box| < 1e-3: the fp32 engine meets that (test above and bench.py's fp32_engine line);
    fp16 storage of weights and of 76 layers of activations drifts by a few fp16 ulps per layer, which on these seeded random nets
    (no trained margin between classes) means (bounds = about twice what is measured): boxes within 2.8 px (measured 1.0-1.4),
    >= 97 % of the oracle's instances kept with the same class, |d conf| < 2e-2 (measured 7-9e-3), mean mask IoU >= 0.99 over the
    matched ones (measured 0.995-0.997), worst one >= 0.82 (measured 0.91-0.93).  That is NOT the north-star gate -- the h2 engine
    (test below) and the fp32 engine meet it; plain fp16 is kept as the fastest, stated-drift option."""
    need_gpu()
    import vti_amd
    from oracle import parity as op
    fr = frames_u8(2, 640, 640, seed=23)
    model = _calibrated_model(vti_amd, 80, "fp16", fr[0], 640, 0.25, target=60)
    eng = model._engine(640, 640, 2)
    got = op.engine_predict(eng, torch.from_numpy(fr).cuda(), 0.25, 0.7, 300)
    want = op.oracle_predict(model._blob, fr, 80, 0.25, 0.7, 300, mode="fp32")
    res = op.compare(got, want, 640, 640)
    assert res["n_instances"] >= 20, res
    assert res["n_matched"] >= 0.97 * res["n_instances"] and abs(res["n_engine"] - res["n_instances"]) <= 0.05 * res["n_instances"], res
    assert res["box_px_max"] < 2.8 and res["conf_abs_max"] < 2e-2, res
    assert res["mask_iou_mean"] >= 0.99 and res["mask_iou_min"] >= 0.82, res
    # ... and the same pipeline on the exact-f32 engine meets the north-star gate itself
    m32 = vti_amd.YOLO(model._blob, dtype="fp32", max_batch=2)
    got32 = op.engine_predict(m32._engine(640, 640, 2), torch.from_numpy(fr).cuda(), 0.25, 0.7, 300)
    r32 = op.compare(got32, want, 640, 640)
    assert r32["meets_north_star"] and r32["mask_iou_min"] >= 0.999 and r32["box_norm_max"] < 1e-3 and r32["kept_set_equal"], r32


def test_h2_engine_meets_the_north_star_gate():
    """The benchmarked dtype (h2: split-fp16 pairs on the fp16 matrix pipe) against the fp32 CPU oracle, end to end, on 8 frames
    of the bench's size, through the same entry points the bench times (vti_forward_scored / vti_nms_scored / vti_masks): the
    north-star GATE itself -- identical kept set and order, |d box| < 1e-3 normalised, mask IoU >= 0.999 for EVERY instance."""
    need_gpu()
    import vti_amd
    from oracle import parity as op
    fr = frames_u8(8, 640, 640, seed=29)
    model = _calibrated_model(vti_amd, 80, "h2", fr[0], 640, 0.25, target=60)
    eng = model._engine(640, 640, 8)
    got = op.engine_predict(eng, torch.from_numpy(fr).cuda(), 0.25, 0.7, 300)
    want = op.oracle_predict(model._blob, fr, 80, 0.25, 0.7, 300, mode="fp32")
    res = op.compare(got, want, 640, 640)
    assert res["n_instances"] >= 100, res
    assert res["kept_set_equal"] and res["n_engine"] == res["n_instances"] == res["n_matched"], res
    assert res["box_norm_max"] < 1e-3 and res["box_px_max"] < 0.05, res
    assert res["mask_iou_min"] >= 0.999, res
    assert res["meets_north_star"], res


def test_predict_batch_640_and_empty_results():
    need_gpu()
    import vti_amd
    fr = frames_u8(3, 640, 640, seed=22)
    model = _calibrated_model(vti_amd, 80, "fp16", fr[0], 640, 0.25)
    res = model(fr, conf=0.25, iou=0.7)                       # __call__ alias, batched ndarray
    assert len(res) == 3 and all(len(r.boxes) > 0 for r in res)
    one = model.predict(source=fr[1])                          # keyword `source=` as check_model.py:331
    assert torch.equal(one[0].boxes.data, res[1].boxes.data)
    assert torch.equal(one[0].masks.data_u8, res[1].masks.data_u8)
    assert model.names[0] == "class0" and len(model.names) == 80
    # masks.xy (Utils/check_model.py:185): one polygon per instance inside the frame, every vertex a pixel of the mask
    polys = one[0].masks.xy
    m0 = one[0].masks.data_u8.cpu().numpy()
    assert len(polys) == len(one[0].boxes)
    for poly, m in list(zip(polys, m0))[:5]:
        assert poly.dtype == np.float32 and poly.shape[1] == 2 and (len(poly) > 0) == bool(m.any())
        assert poly[:, 0].min() >= 0 and poly[:, 0].max() <= 640 and all(m[int(y), int(x)] == 1 for x, y in poly)
    quiet = model.predict(fr[0], conf=0.999999)
    assert len(quiet[0].boxes) == 0 and quiet[0].masks is None and quiet[0].boxes.xyxy.shape == (0, 4)


def test_drop_empty_masks_flag_and_output_reuse():
    """Newer Ultralytics releases drop instances whose thresholded mask is empty (`keep = masks.sum((-2, -1)) > 0` at the end of
    the segmentation predictor; the reference reads the result at measurement.py:208-211 either way); `drop_empty_masks=True`
    reproduces that: exactly the rows with a non-empty mask survive, in order.  Also: predict() reuses one cached output set across
    calls, so Results of an earlier call must not change when a later call overwrites it."""
    need_gpu()
    import vti_amd
    fr = frames_u8(2, 640, 640, seed=41)
    base = _calibrated_model(vti_amd, 80, "h2", fr[0], 640, 0.25, target=300)
    plain = base.predict(fr, conf=0.25, iou=0.7)
    snap = [(r.boxes.data.clone(), r.masks.data_u8.clone()) for r in plain]
    drop = vti_amd.YOLO(base._blob, dtype="h2", max_batch=2, drop_empty_masks=True).predict(fr, conf=0.25, iou=0.7)
    again = base.predict(fr[::-1].copy(), conf=0.25, iou=0.7)              # overwrites the cached output set of `base`
    n_empty = 0
    for r, (bx, mk), d in zip(plain, snap, drop):
        assert torch.equal(r.boxes.data, bx) and torch.equal(r.masks.data_u8, mk)          # earlier Results are copies
        keep = mk.flatten(1).any(1).bool()
        n_empty += int((~keep).sum())
        assert len(d.boxes) == int(keep.sum())
        assert torch.equal(d.boxes.data, bx[keep]) and torch.equal(d.masks.data_u8, mk[keep])
    assert torch.equal(again[1].boxes.data, snap[0][0])
    print("empty masks dropped:", n_empty)


def test_predict_errors_raise_not_abort():
    need_gpu()
    import vti_amd
    model = vti_amd.YOLO(None, scale="n", nc=2, dtype="fp16", max_batch=1)
    with pytest.raises(ValueError):
        model.predict(np.zeros((64, 64), np.uint8))
    with pytest.raises(ValueError):
        model.predict(np.zeros((64, 64, 3), np.float32))
    with pytest.raises(ValueError):
        model.predict()
    with pytest.raises((vti_amd.VtiError, ValueError)):
        vti_amd.YOLO(b"not a container")


def test_predict_pipeline_is_graph_capturable():
    """The whole pipeline (incl. the forward's side-stream fork/join lanes and the mask memset) captures
    into a HIP graph and replays with bit-identical outputs -- no hidden sync, allocation or host read."""
    need_gpu()
    import vti_amd
    B = 2
    eng = vti_amd.Engine("n", 80, H=640, W=640, max_batch=B, dtype="fp16")
    eng.load_weights(vti_amd.random_weights(eng, 1, cls_bias=-6.0), 0)
    x = torch.from_numpy(frames_u8(B, 640, 640, seed=31)).cuda()
    out = eng.alloc_outputs(B, 300, B * 64, "bits")

    def run():
        eng.forward(x, True, pred=out["pred"], proto=out["proto"])
        eng.nms(out["pred"], 0.25, 0.7, 300, dets=out["dets"], counts=out["counts"])
        eng.masks(out["dets"], out["counts"], out["proto"], "logit", "bits", capacity=B * 64, masks=out["masks"], offsets=out["offsets"])
        eng.scale_boxes(out["dets"], out["counts"], 640, 640, xyxy=out["xyxy"])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in out.items()}
    assert int(ref["counts"].sum()) > 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        run()
    for k in ("pred", "proto", "dets", "counts", "masks", "xyxy"):
        out[k].zero_()
    g.replay()
    torch.cuda.synchronize()
    for k in ("pred", "proto", "dets", "counts", "masks", "offsets", "xyxy"):
        assert torch.equal(out[k], ref[k]), k
