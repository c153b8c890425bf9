"""BASELINE config 5 on the GPU: YOLOv8m-seg, nc=80, 1280x1280, bs=16, fp16 + post-processing at 50 instances per frame
(A = 33600 anchors, 320x320 prototypes, 1280x1280 masks).  Reference call site: measurement.py:208-210.

At this size the oracle is used on ONE frame (a 418-GFLOP forward on the host) and the batch is covered by
size-independent properties: duplicated frames are bit-identical wherever they sit, a frame's outputs do not depend on the
engine's max_batch (the plans differ), everything is finite.  NMS rows are bit-exact against the oracle, incl. Ultralytics'
max_nms = 30000 cut, which only a 1280x1280 input (33600 anchors) can reach; masks at IoU >= 0.999."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from gpu_util import frames_u8, mask_iou, need_gpu, synth_pred
from oracle.postproc import non_max_suppression, process_mask

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, B5 = 1280, 16


@pytest.fixture(scope="module")
def m_engine():
    need_gpu()
    import vti_amd
    eng = vti_amd.Engine("m", 80, H=S, W=S, max_batch=B5, dtype="fp16")
    blob = vti_amd.random_weights(eng, seed=1, gain=1.5)
    eng.load_weights(blob, 0)
    return eng, blob


def test_config5_forward_layerwise_one_frame(m_engine):
    """One 1280x1280 frame through the bs=16 plan of the m model against the fp16-emulating oracle, layer by layer
    (same bound as test_gpu_forward: 1e-2 of the layer's max, a few fp16 ulps of drift over 96 convs)."""
    import vti_amd
    from oracle.model import OracleModel
    eng, blob = m_engine
    assert eng.fused_params == 27268704 and eng.num_anchors == 33600
    fr = frames_u8(1, S, S, seed=41)
    pred, proto = eng.forward(torch.from_numpy(fr).cuda(), swap_rb=True)
    torch.cuda.synchronize()
    om = OracleModel(blob, S, S, mode="fp16")
    opred, oproto = om.forward_u8(fr, swap_rb=True, record=True)
    table = eng.conv_table()
    checked = 0
    for i, t in enumerate(table):
        try:
            got = eng.debug_conv_output(i, 1).cpu()
        except vti_amd.VtiError:
            continue            # fused into its consumer's kernel: checked through that consumer
        ref = om.taps[t["name"]]
        err = (got - ref).abs().max().item()
        assert err <= 1e-2 * max(ref.abs().max().item(), 1.0), f"{t['name']}: max|d|={err:.3e} ref max={ref.abs().max():.3e}"
        checked += 1
    assert checked >= len(table) - 29
    assert pred.shape == (1, 116, 33600) and proto.shape == (1, 320, 320, 32)
    assert torch.isfinite(pred).all() and torch.isfinite(proto.float()).all()
    pe = (proto.float().cpu().permute(0, 3, 1, 2) - oproto).abs().max().item()
    assert pe <= 1e-2 * max(oproto.abs().max().item(), 1.0), pe


def test_config5_h2_engine_one_frame_and_batch_property():
    """The same configuration on the h2 engine (the dtype that meets the north-star tolerance): one 1280x1280 frame through the
    bs=16 plan against the plain fp32 oracle, layer by layer, at the h2 bound (4e-5 of the layer's max); and frames of a bs=16
    batch are bit-identical to what the max_batch=2 plan computes for them."""
    need_gpu()
    import vti_amd
    from oracle.model import OracleModel
    eng = vti_amd.Engine("m", 80, H=S, W=S, max_batch=B5, dtype="h2")
    blob = vti_amd.random_weights(eng, seed=1, gain=1.5)
    eng.load_weights(blob, 0)
    base = frames_u8(2, S, S, seed=41)
    fr = np.concatenate([base] * 8, 0)
    pred, proto = eng.forward(torch.from_numpy(fr).cuda(), swap_rb=True)
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all() and torch.isfinite(proto).all() and proto.dtype == torch.float32
    for r in range(1, 8):
        assert torch.equal(pred[:2], pred[2 * r:2 * r + 2]) and torch.equal(proto[:2], proto[2 * r:2 * r + 2])
    om = OracleModel(blob, S, S, mode="fp32")
    opred, oproto = om.forward_u8(base[:1], swap_rb=True, record=True)
    checked = 0
    for i, t in enumerate(eng.conv_table()):
        try:
            got = eng.debug_conv_output(i, 1).cpu()
        except vti_amd.VtiError:
            continue
        ref = om.taps[t["name"]]
        err = (got - ref).abs().max().item()
        assert err <= 4e-5 * max(ref.abs().max().item(), 1.0), f"{t['name']}: max|d|={err:.3e} ref max={ref.abs().max():.3e}"
        checked += 1
    assert checked >= 60
    pe = (proto[:1].cpu().permute(0, 3, 1, 2) - oproto).abs().max().item()
    assert pe <= 4e-5 * max(oproto.abs().max().item(), 1.0), pe
    small = vti_amd.Engine("m", 80, H=S, W=S, max_batch=2, dtype="h2")
    small.load_weights(blob, 0)
    p2, q2 = small.forward(torch.from_numpy(base).cuda(), swap_rb=True)
    torch.cuda.synchronize()
    assert torch.equal(p2, pred[:2]) and torch.equal(q2, proto[:2])


def test_config5_batch16_properties(m_engine):
    """bs=16: 4 distinct frames x 4 copies in a shuffled order -> copies bit-identical; and bit-identical to what a
    max_batch=2 engine (different tile plans) computes for the same frames."""
    import vti_amd
    eng, blob = m_engine
    base = frames_u8(4, S, S, seed=42)
    fr = np.concatenate([base] * 4, 0)
    perm = np.random.default_rng(1).permutation(B5)
    pred, proto = eng.forward(torch.from_numpy(fr[perm]).cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all() and torch.isfinite(proto.float()).all()
    inv = np.argsort(perm)
    pred, proto = pred[inv], proto[inv]
    for r in range(1, 4):
        assert torch.equal(pred[:4], pred[4 * r:4 * r + 4]) and torch.equal(proto[:4], proto[4 * r:4 * r + 4])
    small = vti_amd.Engine("m", 80, H=S, W=S, max_batch=2, dtype="fp16")
    small.load_weights(blob, 0)
    p2, q2 = small.forward(torch.from_numpy(base[1:3]).cuda())
    torch.cuda.synchronize()
    assert torch.equal(p2, pred[1:3]) and torch.equal(q2, proto[1:3])


def test_config5_nms_and_masks_50_instances(m_engine):
    """50 planted instances x 5 duplicates per frame on A = 33600 anchors; masks at 1280x1280 from 320x320 prototypes."""
    import vti_amd
    eng, _ = m_engine
    rng = np.random.default_rng(43)
    Bp = 2
    pred = synth_pred(rng, Bp, 80, 32, 33600, H=S, W=S)
    pd = torch.from_numpy(pred).cuda()
    dets, counts = eng.nms(pd, 0.25, 0.7, 300)
    torch.cuda.synchronize()
    ref = non_max_suppression(pred, 0.25, 0.7, 300, nc=80)
    assert counts.cpu().tolist() == [50, 50]
    for b in range(Bp):
        assert np.array_equal(dets[b, :50].cpu().numpy(), ref[b])
    proto = torch.from_numpy(rng.standard_normal((Bp, 320, 320, 32)).astype(np.float32)).half().cuda()
    masks, offsets = eng.masks(dets, counts, proto, "logit", "u8")
    bits, _ = eng.masks(dets, counts, proto, "logit", "bits")
    torch.cuda.synchronize()
    assert torch.equal(vti_amd.unpack_bits(bits, S), masks)
    off = offsets.cpu().tolist()
    assert off == [0, 50, 100]
    worst = 1.0
    for b in range(Bp):
        want = process_mask(proto[b].float().cpu().permute(2, 0, 1), ref[b][:, 6:], ref[b][:, :4], (S, S), "logit").numpy()
        got = masks[off[b]:off[b + 1]].cpu().numpy()
        assert want.sum() > 0
        for i in range(50):
            worst = min(worst, mask_iou(got[i], want[i]))
    assert worst >= 0.999, worst


def test_nms_max_nms_cut_above_30000_candidates(m_engine):
    """Every one of the 33600 anchors clears conf: Ultralytics keeps the 30000 best-scoring ones (max_nms) before the
    greedy pass.  max_det is set above 30000 so that rows past the cut WOULD show up without it."""
    eng, _ = m_engine
    rng = np.random.default_rng(44)
    pred = synth_pred(rng, 1, 80, 32, 33600, H=S, W=S, n_inst=0, bg=0.9)
    pred[:, 4:84] = np.maximum(pred[:, 4:84], 0.3)
    assert int((pred[0, 4:84].max(0) > 0.25).sum()) == 33600
    max_det = 32000
    dets, counts = eng.nms(torch.from_numpy(pred).cuda(), 0.25, 0.5, max_det)
    torch.cuda.synchronize()
    ref = non_max_suppression(pred, 0.25, 0.5, max_det, nc=80)[0]
    n = int(counts[0])
    assert n == len(ref) and 1000 < n <= 30000
    got = dets[0, :n].cpu().numpy()
    assert np.array_equal(got, ref)
    # the weakest kept score is one of the 30000 best: nothing from beyond the cut got in
    best = np.sort(pred[0, 4:84].max(0))[::-1]
    assert got[:, 4].min() >= best[29999]


def test_persistent_kernels_fall_back_for_huge_tensors():
    """Tensors of >= 2 GiB cannot go through the persistent kernels (one buffer resource, bit 31 = out of range): the
    plan keeps such convs on the per-tile kernel and does not fold the Upsample.  Forced here by lowering the limit
    (VTI_PK_LIMIT_BYTES) in a child process; outputs must agree with the normal plan's."""
    need_gpu()
    code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r)
import vti_amd
eng = vti_amd.Engine("n", 80, H=320, W=320, max_batch=2, dtype="fp16")
eng.load_weights(vti_amd.random_weights(eng, 1), 0)
x = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (2, 320, 320, 3), dtype=np.uint8)).cuda()
pred, proto = eng.forward(x)
torch.cuda.synchronize()
npk = sum(1 for t in eng.conv_table() if t["persistent"])
np.save(sys.argv[1], np.concatenate([pred.cpu().numpy().ravel(), proto.float().cpu().numpy().ravel()]))
print("persistent", npk, "launches", eng.num_launches)
''' % ROOT
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for tag, lim in (("a", None), ("b", "1000000")):     # 1 MB: every activation of a 320x320 bs=2 n-model above 40x40 exceeds it
            env = dict(os.environ)
            if lim:
                env["VTI_PK_LIMIT_BYTES"] = lim
            path = os.path.join(td, tag + ".npy")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append((np.load(path), r.stdout))
    n_a = int(outs[0][1].split()[1]); n_b = int(outs[1][1].split()[1])
    l_a = int(outs[0][1].split()[3]); l_b = int(outs[1][1].split()[3])
    assert n_b < n_a, (outs[0][1], outs[1][1])             # fewer persistent launches ...
    assert l_b > l_a                                        # ... and the Upsample ops are back
    # same network, other kernels: the per-tile / unfused kernels sum some K chunks in another order than the fused ones (the C2f tail
    # of bneck_pk takes y2 from registers in the accumulator's channel order), so equal up to fp16 rounding drift, not bit for bit
    a, b2 = outs[0][0], outs[1][0]
    assert np.isfinite(a).all() and np.isfinite(b2).all()
    assert np.abs(a - b2).max() <= 1e-2 * np.abs(a).max(), np.abs(a - b2).max()
