"""U5 -> U6 hand-over (vti_forward_scored / vti_nms_scored): the class towers write (best class score, first class that has it)
per anchor from their epilogue and the NMS candidate filter reads those 8 bytes instead of nc scores per anchor
(Ultralytics non_max_suppression: `xc = prediction[:, 4:mi].amax(1) > conf_thres`, then `conf, j = cls.max(1, keepdim=True)`).
Bit-exact against the stored rows and against the unscored pair of entry points; a plan whose towers do not write pred
(VTI_NO_PRED_SCATTER=1) derives the pairs from pred and must give the same."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from gpu_util import frames_u8, need_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype,nc", [("h2", 80), ("fp16", 80), ("fp32", 80), ("fp16", 2), ("h2", 2)])
def test_pairs_and_detections_match_the_unscored_path(dtype, nc):
    need_gpu()
    import vti_amd
    B, S = 3, 320
    eng = vti_amd.Engine("n", nc, H=S, W=S, max_batch=B, dtype=dtype)
    eng.load_weights(vti_amd.random_weights(eng, seed=5, gain=1.4), 0)
    x = torch.from_numpy(frames_u8(B, S, S, seed=9)).cuda()
    best = eng.alloc_best(B)
    best.fill_(float("nan"))
    pred, proto = eng.forward(x, True, best=best)
    pred0, proto0 = eng.forward(x, True)
    torch.cuda.synchronize()
    assert torch.equal(pred, pred0) and torch.equal(proto, proto0)
    cls = pred[:, 4:4 + nc, :]                       # [B, nc, A]
    top, arg = cls.max(dim=1)                         # torch returns the FIRST maximal index on CUDA and CPU alike for distinct values ...
    first = (cls == top[:, None, :]).float().argmax(dim=1)       # ... this is the first one whatever the ties
    assert torch.equal(best[..., 0], top)
    assert torch.equal(best[..., 1], first.float())
    # a threshold that keeps a workable number of candidates per frame
    conf = float(torch.quantile(top.flatten().float(), 1.0 - 200.0 / top.shape[1]))
    d0, c0 = eng.nms(pred, conf, 0.6, 100)
    d1, c1 = eng.nms(pred, conf, 0.6, 100, best=best)
    torch.cuda.synchronize()
    assert int(c0.min()) > 3
    assert torch.equal(c0, c1) and torch.equal(d0, d1)


def test_pairs_from_a_plan_without_fused_class_towers():
    need_gpu()
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import vti_amd
eng = vti_amd.Engine("n", 80, H=256, W=320, max_batch=2, dtype="fp16")
eng.load_weights(vti_amd.random_weights(eng, 1), 0)
x = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (2, 256, 320, 3), dtype=np.uint8)).cuda()
best = eng.alloc_best(2)
pred, _ = eng.forward(x, True, best=best)
torch.cuda.synchronize()
np.save(sys.argv[1], np.concatenate([pred.transpose(1, 2).contiguous().cpu().numpy().ravel(), best.cpu().numpy().ravel()]))   # pred as [B, A, no]
print("launches", eng.num_launches)
''' % ROOT
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for tag, flag in (("a", None), ("b", "1")):
            env = dict(os.environ)
            if flag:
                env["VTI_NO_PRED_SCATTER"] = flag
            path = os.path.join(td, tag + ".npy")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append((np.load(path), int(r.stdout.split()[1])))
    assert outs[1][1] > outs[0][1]                      # the unfused plan has the decode kernel's extra work
    a, b = outs[0][0], outs[1][0]
    n_pred = 2 * (4 + 80 + 32) * (8 * 10 * (16 + 4 + 1))
    # class scores go through another kernel in the unfused plan (same arithmetic); the pairs must describe each plan's own rows
    for arr in (a, b):
        pred = arr[:n_pred].reshape(2, -1, 116)          # anchor-major memory
        best = arr[n_pred:].reshape(2, -1, 2)
        cls = pred[:, :, 4:84]
        assert np.array_equal(best[..., 0], cls.max(-1)) and np.array_equal(best[..., 1], cls.argmax(-1).astype(np.float32))
