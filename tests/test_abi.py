"""C-ABI checks that need no GPU: libvti.so loads, exports every symbol include/vti.h declares,
builds the plan on the host, agrees with the oracle's independent table, and reports errors as
status codes (never aborts)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle.spec import Spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(lib_built):
    vti_amd = lib_built
    hdr = open(os.path.join(ROOT, "include", "vti.h")).read()
    declared = set(re.findall(r"\b(vti_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = vti_amd.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in vti.h but not exported by libvti.so"
    assert declared == set(vti_amd.SIGNATURES), declared ^ set(vti_amd.SIGNATURES)


@pytest.mark.parametrize("scale,nc,H,W", [("n", 80, 640, 640), ("n", 2, 736, 960), ("m", 80, 1280, 1280), ("s", 80, 320, 320)])
def test_plan_matches_oracle_table(lib_built, scale, nc, H, W):
    eng = lib_built.Engine(scale, nc, H=H, W=W, max_batch=2)
    sp = Spec(scale, nc, H=H, W=W)
    got = [(t["name"], t["c1"], t["c2"], t["k"], t["s"], t["kind"], t["h_in"], t["w_in"], t["h_out"], t["w_out"], t["macs"])
           for t in eng.conv_table()]
    ref = [(r.name, r.c1, r.c2, r.k, r.s, r.kind, r.h_in, r.w_in, r.h_out, r.w_out, r.macs) for r in sp.rows]
    assert got == ref
    assert eng.fused_params == sp.fused_params and eng.macs_per_frame == sp.macs and eng.num_anchors == sp.num_anchors
    assert eng.workspace_bytes > 0
    for t in eng.conv_table():      # every conv got a launch geometry that fits the kernel's register tile
        th, tw = t["tile"]
        if t["persistent"] and tw == 80:        # conv1_pk: th = M-waves of 80 consecutive pixels each
            assert 1 <= th <= 4 and th * t["waves_n"] <= 4 and 1 <= t["nrep"] <= 5 and t["lds"] <= 160 * 1024
            continue
        if t["persistent"]:     # conv_pk.hip: 20-wide tiles, 4 rows per M-wave, <= 4 compute waves, up to the whole 160 KiB of LDS
            assert tw == 20 and th % 4 == 0 and (th // 4) * t["waves_n"] <= 4 and 1 <= t["nrep"] <= 5 and t["lds"] <= 160 * 1024
            continue
        assert th > 0 and tw > 0 and th * tw <= (4 // t["waves_n"]) * 80 and 1 <= t["nrep"] <= 5 and t["lds"] <= 80 * 1024


def test_published_counts_through_the_abi(lib_built):
    assert lib_built.Engine("n", 80).fused_params == 3404320
    assert lib_built.Engine("m", 80).fused_params == 27268704
    assert lib_built.Engine("n", 2).fused_params == 3258454


@pytest.mark.parametrize("kw", [dict(scale="q"), dict(H=650), dict(nc=0), dict(max_batch=0), dict(reg_max=8)])
def test_create_rejects_bad_descriptions(lib_built, kw):
    args = dict(scale="n", nc=80, H=640, W=640, max_batch=1)
    args.update(kw)
    with pytest.raises(lib_built.VtiError) as ei:
        lib_built.Engine(**args)
    assert ei.value.code == -1 and "vti_create" in str(ei.value)


def test_weight_container_roundtrip_and_mismatch(lib_built):
    vti_amd = lib_built
    eng = vti_amd.Engine("n", 2, H=64, W=64, max_batch=1)
    blob = vti_amd.random_weights(eng, seed=5)
    meta, table, tensors = vti_amd.unpack_container(blob)
    assert meta == dict(scale="n", nc=2, nm=32, reg_max=16) and len(table) == 76
    assert vti_amd.pack_container("n", 2, 32, 16, eng.conv_table(), tensors) == blob
    from oracle.blob import read_blob      # the oracle's independent reader agrees
    ometa, oconvs = read_blob(blob)
    assert ometa == meta and list(oconvs) == [t["name"] for t in table]
    for t in table:
        assert np.array_equal(oconvs[t["name"]][5], tensors[t["name"]][0])
    # wrong model / truncated / garbage containers -> VTI_ERR_WEIGHTS before any GPU call
    L = vti_amd.lib()
    other = vti_amd.Engine("n", 80, H=64, W=64, max_batch=1)
    for bad in (blob[:-4], b"nope" + blob[4:], blob + b"\0\0\0\0"):
        buf = (C.c_char * len(bad)).from_buffer_copy(bad)
        assert L.vti_load_weights(eng._ctx, buf, len(bad), 0) == -3
        assert b"weights" in L.vti_last_error(eng._ctx)
    buf = (C.c_char * len(blob)).from_buffer_copy(blob)
    assert L.vti_load_weights(other._ctx, buf, len(blob), 0) == -3


def test_calls_before_setup_fail_with_status(lib_built):
    vti_amd = lib_built
    L = vti_amd.lib()
    eng = vti_amd.Engine("n", 2, H=64, W=64, max_batch=1)
    one = C.c_void_p(16)   # never dereferenced: state checks come first
    assert L.vti_forward(eng._ctx, one, 1, 1, one, one, None) == -2       # weights not loaded
    assert L.vti_forward(eng._ctx, one, 5, 1, one, one, None) == -1       # B > max_batch
    assert L.vti_nms(eng._ctx, one, 1, 0.25, 0.7, 10, 0, one, one, None) == -2
    assert L.vti_set_workspace(eng._ctx, C.c_void_p(12345), 1 << 30) == -1   # misaligned
    assert L.vti_set_workspace(eng._ctx, C.c_void_p(4096), 16) == -5          # too small
    assert L.vti_conv_at(eng._ctx, 999, C.byref(vti_amd._lib.VtiConvInfo())) == -1


def test_letterbox_shape_matches_oracle():
    from oracle.letterbox import letterbox_geometry
    from vti_amd import letterbox_shape
    for (h0, w0, imgsz) in [(960, 1280, 960), (640, 640, 640), (480, 640, 640), (1080, 1920, 640), (333, 517, 640), (720, 1280, 1280)]:
        g = letterbox_geometry(h0, w0, imgsz)
        assert letterbox_shape(h0, w0, imgsz) == (g["H"], g["W"])
    assert letterbox_shape(960, 1280, 960) == (736, 960)      # the reference's real input
    assert letterbox_shape(960, 1280, 950) == (736, 960)      # Ultralytics check_imgsz: imgsz rounds UP to a multiple of the stride
    assert letterbox_shape(600, 600, 610) == (640, 640)


def test_letterbox_2x_downscale_is_the_box_mean():
    """OpenCV swaps INTER_LINEAR for INTER_AREA at an exact 2x downscale: rounded mean of each 2x2 block."""
    from oracle.letterbox import letterbox
    img = np.zeros((4, 4, 3), np.uint8)
    img[0, 0] = (1, 2, 3); img[0, 1] = (2, 2, 3); img[1, 0] = (1, 3, 255); img[1, 1] = (2, 2, 254)
    img[2:, 2:] = 200
    out, g = letterbox(img, 2, auto=False)
    assert (g["new_h"], g["new_w"], g["top"], g["left"]) == (2, 2, 0, 0)
    assert out[0, 0].tolist() == [2, 2, 129] and out[1, 1].tolist() == [200, 200, 200] and out[0, 1].tolist() == [0, 0, 0]


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vision-textile-inspection_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"
