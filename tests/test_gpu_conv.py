"""Kernel-level parity of the MFMA conv family (conv.hip) through vti_debug_conv2d.

Exact-integer cases: operands are small integers so every product and partial sum is exactly
representable -- the HIP result must equal the CPU convolution BIT FOR BIT, for every kernel
variant (1x1, 3x3 s1, 3x3 s2, ConvTranspose 2x2, stem), wave split, register tile, ragged
spatial size, channel remainder and channel-slice offset.  Operands are asymmetric random ints,
so a transposed fragment or swapped operand cannot pass."""
import numpy as np
import pytest
import torch

from gpu_util import need_gpu, ref_conv

pytestmark = pytest.mark.gpu


def _to_dev(a, dtype):
    """host float array -> device tensor in the engine's storage type (h2: split-fp16 pairs carried as f32 bits)."""
    import vti_amd
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype == "h2":
        return vti_amd.h2_encode(t).cuda()
    return t.to(torch.float16 if dtype == "fp16" else torch.float32).cuda()


def _to_host(t, dtype, out_f32=False):
    import vti_amd
    return vti_amd.h2_decode(t.cpu()) if (dtype == "h2" and not out_f32) else t.float().cpu()


def _ints(rng, shape, lo, hi):
    return rng.integers(lo, hi + 1, shape).astype(np.float32)


CASES = [
    # k, s, kind, c1, c2, H, W, forced (wn, nrep), extras
    (1, 1, 1, 32, 32, 16, 20, (0, 0), {}),
    (1, 1, 1, 48, 32, 12, 20, (1, 2), {}),                      # channel remainder chunk (48 = 32 + 16)
    (1, 1, 1, 128, 256, 8, 20, (4, 4), {}),                     # waves split along Cout, grid.y = 1
    (1, 1, 1, 64, 256, 8, 20, (4, 2), {}),                      # grid.y = 2
    (1, 1, 1, 64, 2, 7, 9, (0, 0), {"out_f32": True}),          # nc=2 head: scalar-store path, ragged
    (3, 1, 1, 16, 16, 20, 16, (1, 1), {}),                      # half-empty K chunk
    (3, 1, 1, 64, 80, 16, 20, (1, 5), {}),                      # NREP 5 (cls tower)
    (3, 1, 1, 80, 80, 10, 20, (1, 5), {}),                      # 80 = 2.5 chunks
    (3, 1, 1, 32, 32, 23, 30, (0, 0), {}),                      # ragged tiles (736x960 feature map)
    (3, 1, 1, 128, 128, 20, 20, (2, 2), {}),
    (3, 1, 1, 64, 64, 9, 11, (2, 2), {"res": True}),            # bottleneck residual
    (3, 2, 1, 32, 64, 40, 40, (2, 2), {}),
    (3, 2, 1, 16, 32, 32, 32, (2, 1), {}),
    (3, 2, 1, 128, 256, 21, 19, (0, 0), {}),                    # odd input, stride 2
    (2, 2, 2, 64, 64, 10, 12, (0, 0), {}),                      # ConvTranspose2d(2,2): scattered stores
    (2, 2, 2, 192, 192, 6, 5, (0, 0), {}),                      # m-scale proto upsample
    (1, 1, 1, 32, 32, 16, 20, (1, 2), {"in_coff": 16, "in_ld": 64, "out_coff": 32, "out_ld": 96}),   # concat slices
    (3, 1, 1, 32, 48, 12, 12, (1, 3), {"in_coff": 32, "in_ld": 64}),
    # persistent LDS-DMA kernel (conv_pk.hip): W >= 20
    (3, 1, 1, 64, 64, 40, 40, (1, 4), {"res": True}),            # two stationary weight chunks + residual
    (3, 1, 1, 32, 32, 44, 60, (0, 0), {"in_coff": 32, "in_ld": 64, "out_coff": 32, "out_ld": 96}),   # slices, ragged rows
    (3, 1, 1, 96, 64, 16, 40, (1, 4), {}),                       # three chunks: streamed weights
    (3, 1, 1, 32, 32, 80, 80, (1, 2), {"pk_wgs": 8}),            # 5 tiles per workgroup, one chunk
    (3, 1, 1, 64, 64, 80, 40, (1, 4), {"pk_wgs": 8, "res": True}),   # tile chain with two chunks per tile
    (3, 1, 1, 80, 80, 48, 40, (1, 5), {"pk_wgs": 3}),            # odd chunk count across tile seams, no XCD ranges
    (3, 1, 1, 128, 128, 40, 40, (2, 2), {"pk_wgs": 8}),          # waves split along Cout, two n-groups
    (3, 1, 1, 64, 2, 24, 20, (0, 0), {"out_f32": True}),         # generic epilogue from the persistent kernel
]


@pytest.mark.parametrize("dtype", ["fp16", "fp32", "h2"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"k{c[0]}s{c[1]}kind{c[2]}_{c[3]}to{c[4]}_{c[5]}x{c[6]}_wn{c[7][0]}n{c[7][1]}")
def test_conv_exact_integers(case, dtype, monkeypatch):
    need_gpu()
    import vti_amd
    k, s, kind, c1, c2, H, W, (wn, nrep), ex = case
    if "pk_wgs" in ex:
        monkeypatch.setenv("VTI_PK_MAX_WGS", str(ex["pk_wgs"]))
    rng = np.random.default_rng(hash((k, s, kind, c1, c2, H, W)) % (2 ** 32))
    B = 2
    in_ld, in_coff = ex.get("in_ld", c1), ex.get("in_coff", 0)
    x_full = _ints(rng, (B, H, W, in_ld), -2, 2)
    x = x_full[..., in_coff:in_coff + c1]
    wshape = (c1, c2, k, k) if kind == 2 else (c2, c1, k, k)
    w = _ints(rng, wshape, -1, 1)
    b = _ints(rng, (c2,), -3, 3)
    xd = _to_dev(x_full, dtype)
    Ho, Wo = (2 * H, 2 * W) if kind == 2 else ((H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1)
    res = resd = None
    if ex.get("res"):
        res = _ints(rng, (B, Ho, Wo, c2), -4, 4)
        resd = _to_dev(res, dtype)
    out, _, cfg = vti_amd.debug_conv2d(xd, w, b, k, s, kind, dtype, res=resd, in_coff=in_coff, c1=c1,
                                       out_coff=ex.get("out_coff", 0), out_ld=ex.get("out_ld"),
                                       out_f32=ex.get("out_f32", False), waves_n=wn, nrep=nrep)
    torch.cuda.synchronize()
    ref = ref_conv(x, w, b, k, s, kind, dtype, res=res, act=False)
    oc = ex.get("out_coff", 0)
    got = _to_host(out, dtype, ex.get("out_f32", False))
    assert torch.equal(got[..., oc:oc + c2], ref), f"cfg={cfg} max|d|={(got[..., oc:oc + c2] - ref).abs().max()}"
    if k == 3 and s == 1 and W >= 20:
        assert cfg["pk"], cfg       # these shapes must exercise the persistent kernel
    if oc:      # neighbours of the written channel slice stay untouched
        assert (got[..., :oc] == 0).all() and (got[..., oc + c2:] == 0).all()


@pytest.mark.parametrize("dtype,tol", [("fp16", 2e-3), ("fp32", 2e-6), ("h2", 4e-6)])
def test_conv_silu_random(dtype, tol):
    """Random operands + fused bias/SiLU/residual epilogue: within rounding of the CPU op."""
    need_gpu()
    import vti_amd
    rng = np.random.default_rng(7)
    B, H, W, c1, c2 = 2, 24, 20, 64, 64
    x = rng.standard_normal((B, H, W, c1)).astype(np.float32)
    w = (rng.standard_normal((c2, c1, 3, 3)) / np.sqrt(9 * c1)).astype(np.float32)
    b = rng.standard_normal(c2).astype(np.float32) * 0.1
    res = rng.standard_normal((B, H, W, c2)).astype(np.float32)
    out, _, _ = vti_amd.debug_conv2d(_to_dev(x, dtype), w, b, 3, 1, 0, dtype, res=_to_dev(res, dtype))
    ref = ref_conv(x, w, b, 3, 1, 0, dtype, res=res)
    if dtype == "fp16":
        ref = ref.half().float()
    err = (_to_host(out, dtype) - ref).abs().max().item()
    assert err < tol * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("shape", [(16, 16, 40, 40, 0, 0), (32, 32, 40, 60, 0, 0), (64, 64, 44, 40, 0, 0), (64, 80, 40, 40, 0, 0),
                                   (80, 80, 24, 40, 0, 0), (128, 80, 40, 40, 2, 3), (128, 128, 20, 20, 0, 0), (64, 64, 80, 80, 1, 4)],
                         ids=lambda s: f"{s[0]}to{s[1]}_{s[2]}x{s[3]}_wn{s[4]}n{s[5]}")
@pytest.mark.parametrize("wgs", [0, 8])
def test_conv3x3_silu_persistent(shape, wgs, monkeypatch):
    """3x3/s1 Conv-BN-SiLU without residual through the persistent kernel (every register tile / wave split it is
    planned with): one tile per workgroup and chains of tiles, ragged rows, channel-group remainders."""
    need_gpu()
    import vti_amd
    c1, c2, H, W, wn, nrep = shape
    if wgs:
        monkeypatch.setenv("VTI_PK_MAX_WGS", str(wgs))
    rng = np.random.default_rng(c1 * 1000 + c2)
    B = 3
    x = rng.standard_normal((B, H, W, c1)).astype(np.float32)
    w = (rng.standard_normal((c2, c1, 3, 3)) / np.sqrt(9 * c1)).astype(np.float32)
    b = rng.standard_normal(c2).astype(np.float32) * 0.5
    out, _, cfg = vti_amd.debug_conv2d(torch.from_numpy(x).half().cuda(), w, b, 3, 1, 0, "fp16", waves_n=wn, nrep=nrep)
    assert cfg["pk"], cfg
    ref = ref_conv(x, w, b, 3, 1, 0, "fp16").half().float()
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), (err, cfg)


@pytest.mark.parametrize("dtype,tol", [("fp16", 1e-3), ("fp32", 1e-6), ("h2", 2e-6)])
@pytest.mark.parametrize("swap", [False, True])
def test_stem_conv_u8(dtype, tol, swap):
    """model.0: u8 HWC3 frame -> /255 -> 3x3 s2 conv (+ channel flip), ragged size."""
    need_gpu()
    import vti_amd
    rng = np.random.default_rng(11)
    B, H, W, c2 = 2, 46, 62, 16
    x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    w = (rng.standard_normal((c2, 3, 3, 3)) / np.sqrt(27)).astype(np.float32)
    b = rng.standard_normal(c2).astype(np.float32) * 0.1
    out, _, _ = vti_amd.debug_conv2d(torch.from_numpy(x).cuda(), w, b, 3, 2, 0, dtype, c1=3, swap_rb=swap)
    xin = (x[..., ::-1] if swap else x).astype(np.float32) / np.float32(255)
    ref = ref_conv(xin, w, b, 3, 2, 0, dtype)
    if dtype == "fp16":
        ref = ref.half().float()
    err = (_to_host(out, dtype) - ref).abs().max().item()
    assert out.shape == (B, 23, 31, c2) and err < tol * max(1.0, ref.abs().max().item()), err
