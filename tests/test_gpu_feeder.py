"""SURVEY section 8 row N4, device half: the pinned-ring host -> HBM frame feed (vti_amd.FrameFeeder) in front of the predict path.
The reference hands `model.predict` one pageable host frame at a time (main.py:188 -> measurement.py:205-210)."""
import numpy as np
import pytest
import torch

from gpu_util import frames_u8, need_gpu

pytestmark = pytest.mark.gpu

KEYS = ("dets", "counts", "xyxy", "offsets", "masks")


def test_overlapped_feed_is_bit_identical_to_the_synchronous_path():
    """Five host batches through a 3-slot ring -- copies of later batches in flight while earlier ones compute, slots reused --
    give exactly the outputs of copying each batch synchronously and predicting on it."""
    need_gpu()
    import vti_amd
    B, H, W, nb, max_det = 4, 640, 640, 5, 100
    eng = vti_amd.Engine("n", 80, H=H, W=W, max_batch=B, dtype="h2")
    eng.load_weights(vti_amd.random_weights(eng, seed=1, cls_bias=-6.5), 0)
    batches = [frames_u8(B, H, W, seed=50 + i) for i in range(nb)]
    ref = []
    for a in batches:                                    # synchronous: pageable -> device, then predict
        o = eng.alloc_outputs(B, max_det, B * max_det, "bits")
        eng.predict_into(torch.from_numpy(a).cuda(), o, 0.25, 0.7, max_det)
        torch.cuda.synchronize()
        ref.append({k: o[k].clone() for k in KEYS})
    assert sum(int(r["counts"].sum()) for r in ref) > 0
    feeder = vti_amd.FrameFeeder(B, H, W, depth=3)
    outs = [eng.alloc_outputs(B, max_det, B * max_det, "bits") for _ in range(nb)]
    slots = [feeder.put(batches[0]), feeder.put(batches[1])]          # two copies in flight before the first predict
    for i in range(nb):
        if i + 2 < nb:
            slots.append(feeder.put(batches[i + 2]))     # reuses a slot: waits (by event) for the predict that last read it
        feeder.predict_into(eng, slots[i], outs[i], conf=0.25, iou=0.7, max_det=max_det)
    torch.cuda.synchronize()
    for i in range(nb):
        n = int(ref[i]["offsets"][-1])
        for k in KEYS:
            a, b = (outs[i][k][:n], ref[i][k][:n]) if k == "masks" else (outs[i][k], ref[i][k])
            if k in ("dets", "xyxy"):                    # rows beyond counts[b] are unspecified
                for f in range(B):
                    c = int(ref[i]["counts"][f])
                    assert torch.equal(a[f, :c], b[f, :c]), (i, k, f)
            else:
                assert torch.equal(a, b), (i, k)


def test_host_view_and_partial_batches():
    need_gpu()
    import vti_amd
    feeder = vti_amd.FrameFeeder(3, 64, 96, depth=2)
    s = feeder.next_slot()
    v = feeder.host_view(s)
    assert v.shape == (3, 64, 96, 3) and v.dtype == np.uint8
    v[:2] = frames_u8(2, 64, 96, seed=1)
    feeder.submit(s, 2)
    x = feeder.frames(s)
    torch.cuda.synchronize()
    assert x.shape == (2, 64, 96, 3) and np.array_equal(x.cpu().numpy(), v[:2])
    feeder.release(s)
    with pytest.raises(ValueError):
        feeder.put(np.zeros((4, 64, 96, 3), np.uint8))          # more frames than the slot holds
    with pytest.raises(ValueError):
        feeder.submit(s, 0)
