"""Parity of the whole HIP forward pass (vti_forward) against the CPU oracle, layer by layer.

Tolerances (north_star: mask IoU >= 0.999, |d box| < 1e-3):
  fp32 engine vs fp32 oracle : every conv output within 2e-5 relative; class scores 1e-4;
      boxes < 1e-3 NORMALISED (|d|/max(H,W)) and < 5e-3 letterboxed px -- the px figure is the fp32
      accumulation-order floor at stride 32, shown by the fp64 cross-check below.
  fp16 engine vs fp16-emulating oracle (rounds where the engine rounds): every conv output within
      6e-3 relative (measured 2.8e-3: a few fp16 ulps of drift over 76 layers); scores 2e-2.
  h2 engine (split-fp16 pairs) vs the fp32 oracle: every conv output within 4e-5 relative (measured 7e-6), boxes < 2e-2 px."""
import numpy as np
import pytest
import torch

from gpu_util import engine_and_oracle, frames_u8, need_gpu

pytestmark = pytest.mark.gpu

CONFIGS = [
    # scale, nc, H, W, B, dtype, per-layer rel tol
    ("n", 80, 640, 640, 1, "fp32", 2e-5),      # BASELINE config geometry, exact-f32 MFMA
    ("n", 80, 640, 640, 2, "fp16", 6e-3),      # BASELINE config (fp16); measured 2.8e-3 of the layer's max
    ("n", 2, 736, 960, 1, "fp16", 6e-3),       # the reference's real input: ragged 92x120 / 46x60 / 23x30 maps
    ("m", 80, 320, 320, 1, "fp16", 6e-3),      # m-scale channel counts (48..576), 96 convs
    ("s", 80, 256, 256, 2, "fp32", 2e-5),
    # h2 (split-fp16 pairs, ~22 significant bits, hardware-rate SiLU) against the plain fp32 oracle
    ("n", 80, 640, 640, 2, "h2", 4e-5),
    ("n", 2, 736, 960, 1, "h2", 4e-5),
    ("m", 80, 320, 320, 1, "h2", 4e-5),
]


@pytest.mark.parametrize("scale,nc,H,W,B,dtype,tol", CONFIGS, ids=lambda v: str(v))
def test_layerwise_parity(scale, nc, H, W, B, dtype, tol):
    need_gpu()
    # seeded random nets sit on a knife edge between dying and exploding activations; the deeper
    # m-scale net needs a smaller He gain than n/s to stay inside fp16 range
    eng, om, _ = engine_and_oracle(scale, nc, H, W, B, dtype, gain=1.5 if scale == "m" else 1.7)
    fr = frames_u8(B, H, W, seed=3)
    pred, proto = eng.forward(torch.from_numpy(fr).cuda(), swap_rb=True)
    torch.cuda.synchronize()
    opred, oproto = om.forward_u8(fr, swap_rb=True, record=True)
    import vti_amd
    checked = 0
    table = eng.conv_table()
    for i, t in enumerate(table):
        if i + 1 < len(table) and table[i + 1]["fused"]:
            # this 3x3's output feeds a 1x1 fused into its epilogue and never reaches memory;
            # it is verified through that 1x1's output (next row)
            with pytest.raises(vti_amd.VtiError):
                eng.debug_conv_output(i, B)
            continue
        try:
            got = eng.debug_conv_output(i, B).cpu()
        except vti_amd.VtiError:
            # class / coefficient towers whose fused 1x1 writes straight into pred, and (fp16 engine) box towers whose fused
            # 1x1 stage also does DFL + dist2bbox: checked via pred below
            # ... and proto.upsample when the plan folded it into proto.cv2 (four 2x2 convs on the low-resolution map): checked
            # through proto.cv3's output
            # ... and the second 3x3 of a bottleneck whose C2f's closing 1x1 runs in the same kernel (y2 stays in registers)
            assert (t["fused"] and (".cv3." in t["name"] or ".cv4." in t["name"] or ".m." in t["name"] or (dtype != "fp32" and ".cv2." in t["name"]))) or \
                   t["name"] == "model.22.proto.upsample", t["name"]
            continue
        checked += 1
        ref = om.taps[t["name"]]
        assert got.shape == ref.shape and torch.isfinite(ref).all(), t["name"]
        err = (got - ref).abs().max().item()
        assert err <= tol * max(ref.abs().max().item(), 1.0), f"{t['name']}: max|d|={err:.3e} ref max={ref.abs().max():.3e}"
    assert checked >= len(table) - 28      # n-scale fp16: stem + layer 1, 10 fused 3x3 mids, 9 tower outputs that live in pred only
    assert pred.shape == opred.shape and torch.isfinite(pred).all()
    if scale != "n":
        return      # m/s random nets carry |logit| ~ 100: only the per-layer bound above is meaningful there
    e = (pred.cpu() - opred).abs()
    # scores: |d sigmoid| <= |d logit| / 4 and |d logit| <= tol * max|logit| (per-layer bound above)
    logit_max = max(om.taps[f"model.22.cv3.{l}.2"].abs().max().item() for l in range(3))
    exact = dtype in ("fp32", "h2")
    cls_tol = max(1e-4 if exact else 2e-2, tol * logit_max)
    box_px, mc_tol = ((5e-3 if dtype == "fp32" else 2e-2), 1e-3) if exact else (4.0, 0.01 * opred[:, 4 + nc:].abs().max().item() + 0.2)
    assert e[:, 4:4 + nc].max() < cls_tol
    assert e[:, :4].max() < box_px and e[:, :4].max() / max(H, W) < (1e-3 if exact else 1e-2)
    assert e[:, 4 + nc:].max() < mc_tol
    pe = (proto.float().cpu().permute(0, 3, 1, 2) - oproto).abs().max().item()
    assert pe < (1e-3 if exact else 0.1)


def test_fp32_box_error_is_the_fp32_floor():
    """|d box| in px of the fp32 engine vs an fp64 evaluation of the same net is no larger than the fp32
    CPU reference's own error vs fp64 (x3 slack): the 1e-3 px gate is met to the precision fp32 has."""
    need_gpu()
    from oracle.model import OracleModel
    eng, om, blob = engine_and_oracle("n", 80, 640, 640, 1, "fp32")
    fr = frames_u8(1, 640, 640, seed=3)
    pred, _ = eng.forward(torch.from_numpy(fr).cuda())
    o32, _ = om.forward_u8(fr)
    o64, _ = OracleModel(blob, 640, 640, mode="fp64").forward_u8(fr)
    e_gpu = (pred.cpu().double() - o64)[:, :4].abs().max().item()
    e_cpu = (o32.double() - o64)[:, :4].abs().max().item()
    assert e_gpu < 5e-3 and e_gpu < 3 * e_cpu + 1e-4, (e_gpu, e_cpu)


def test_batch_invariance_and_flags():
    """Frames are independent: a frame's outputs do not depend on its batch slot or neighbours
    (bit-exact), and swap_rb=False equals pre-flipping the channels."""
    need_gpu()
    eng, _, _ = engine_and_oracle("n", 80, 640, 640, 2, "fp16")
    fr = frames_u8(2, 640, 640, seed=5)
    x = torch.from_numpy(fr).cuda()
    p2, q2 = eng.forward(x)
    p1, q1 = eng.forward(x[1:2].contiguous())
    assert torch.equal(p2[1], p1[0]) and torch.equal(q2[1], q1[0])
    # swap_rb=False == pre-flipping the channels.  Equal in exact arithmetic; the stem's banded weights put the flipped channels at
    # other K positions of the MFMA, so the fp32 sums may round differently (a few fp16 ulps after 76 layers): close, not bit-equal,
    # and each path matches the oracle run with the same flag
    pa, _ = eng.forward(x, swap_rb=False)
    pb, _ = eng.forward(x.flip(-1).contiguous(), swap_rb=True)
    assert (pa[:, 4:84] - pb[:, 4:84]).abs().max() < 2e-2 and (pa[:, :4] - pb[:, :4]).abs().max() < 4.0
    _, om, _ = engine_and_oracle("n", 80, 640, 640, 2, "fp16")
    oa, _ = om.forward_u8(fr, swap_rb=False)
    assert (pa[:, 4:84].cpu() - oa[:, 4:84]).abs().max() < 2e-2
    assert not torch.allclose(pa[:, 4:84], p2[:, 4:84], atol=1e-3)         # and the flag does something
    # the same invariance on the h2 engine (22-bit storage), where "close" is tight enough to pin the stem's two banded weight
    # packings (pack_stem_toeplitz: both channel orders) against each other AND against the oracle run with the same flag:
    # a misplaced tap or channel costs far more than 2e-2 px / 1e-4 in the scores
    eng2, om2, _ = engine_and_oracle("n", 80, 640, 640, 2, "h2")
    ha, qa = eng2.forward(x, swap_rb=False)
    hb, qb = eng2.forward(x.flip(-1).contiguous(), swap_rb=True)
    assert (ha[:, 4:84] - hb[:, 4:84]).abs().max() < 1e-4 and (ha[:, :4] - hb[:, :4]).abs().max() < 2e-2 and (qa - qb).abs().max() < 1e-3
    oa2, op2 = om2.forward_u8(fr, swap_rb=False)
    assert (ha[:, 4:84].cpu() - oa2[:, 4:84]).abs().max() < 1e-4 and (ha[:, :4].cpu() - oa2[:, :4]).abs().max() < 2e-2
    assert (qa.cpu().permute(0, 3, 1, 2) - op2).abs().max() < 1e-3


@pytest.mark.parametrize("dtype", ["h2", "fp16"])
def test_full_size_batch_properties(dtype):
    """BASELINE config size (bs=64, 640x640; the benchmarked h2 engine and the fp16 engine): checked through size-independent properties --
    duplicated frames give bit-identical outputs wherever they sit in the batch, and the first frames
    still match the oracle."""
    need_gpu()
    import vti_amd
    B = 64
    eng = vti_amd.Engine("n", 80, H=640, W=640, max_batch=B, dtype=dtype)
    blob = vti_amd.random_weights(eng, seed=1)
    eng.load_weights(blob, 0)
    base = frames_u8(8, 640, 640, seed=9)
    fr = np.concatenate([base] * 8, 0)
    perm = np.random.default_rng(0).permutation(B)
    pred, proto = eng.forward(torch.from_numpy(fr[perm]).cuda())
    torch.cuda.synchronize()
    inv = np.argsort(perm)
    pred, proto = pred[inv], proto[inv]
    for r in range(1, 8):
        assert torch.equal(pred[:8], pred[8 * r:8 * r + 8]) and torch.equal(proto[:8], proto[8 * r:8 * r + 8])
    from oracle.model import OracleModel
    opred, _ = OracleModel(blob, 640, 640, "fp32" if dtype == "h2" else "fp16").forward_u8(base[:2])
    assert (pred[:2, 4:84].cpu() - opred[:, 4:84]).abs().max() < (1e-4 if dtype == "h2" else 2e-2)
    if dtype == "h2":
        assert (pred[:2, :4].cpu() - opred[:, :4]).abs().max() < 2e-2


@pytest.mark.parametrize("dtype", ["h2", "fp16"])
def test_batch_and_geometry_invariance_at_full_size(dtype):
    """BASELINE size (64 frames of 640x640): every frame's pred/proto must be BIT-identical to what the same engine code
    produces for that frame in a 3-frame batch -- the plans differ (tile shapes, persistent tile chains, workgroups per
    layer are chosen per max_batch), the arithmetic per output element must not.  Size-independent property: no oracle."""
    need_gpu()
    import vti_amd
    fr = frames_u8(64, 640, 640, seed=11)
    big = vti_amd.Engine("n", 80, H=640, W=640, max_batch=64, dtype=dtype)
    blob = vti_amd.random_weights(big, 1, cls_bias=-6.0)
    big.load_weights(blob, 0)
    small = vti_amd.Engine("n", 80, H=640, W=640, max_batch=3, dtype=dtype)
    small.load_weights(blob, 0)
    x = torch.from_numpy(fr).cuda()
    pred, proto = big.forward(x, swap_rb=True)
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all() and torch.isfinite(proto.float()).all()
    for lo in (0, 30, 61):
        p3, q3 = small.forward(x[lo:lo + 3].contiguous(), swap_rb=True)
        torch.cuda.synchronize()
        assert torch.equal(p3, pred[lo:lo + 3]), f"pred differs for frames {lo}..{lo + 2}: max|d|={(p3 - pred[lo:lo + 3]).abs().max().item():.3e}"
        assert torch.equal(q3, proto[lo:lo + 3]), f"proto differs for frames {lo}..{lo + 2}"
