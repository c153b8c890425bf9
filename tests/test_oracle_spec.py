"""Oracle pinning (CPU): the restated layer table reproduces Ultralytics' published
model.info() parameter counts to the unit (SURVEY.md section 8c golden (1)) and the forward has the
documented output shapes."""
import numpy as np
import pytest
import torch

from oracle.spec import Spec


@pytest.mark.parametrize("scale,nc,fused,unfused", [
    ("n", 80, 3404320, 3409968),     # yolov8n-seg
    ("s", 80, 11810560, None),       # yolov8s-seg
    ("m", 80, 27268704, 27285968),   # yolov8m-seg
    ("n", 2, 3258454, None),         # the reference's 2-class model (config.py:69-70)
])
def test_published_param_counts(scale, nc, fused, unfused):
    s = Spec(scale, nc)
    assert s.fused_params == fused
    if unfused is not None:
        assert s.unfused_params == unfused


def test_macs_and_shapes():
    s = Spec("n", 80)
    assert len(s.rows) == 76 and s.num_anchors == 8400 and s.proto_hw == (160, 160)
    assert abs(s.macs / 1e9 - 6.001) < 5e-4           # SURVEY section 8 layer table
    s = Spec("n", 2, H=736, W=960)                    # the reference's real input (imgsz=960 on 1280x960)
    assert s.num_anchors == 14490 and s.proto_hw == (184, 240) and abs(s.macs / 1e9 - 9.782) < 5e-4
    s = Spec("m", 80, H=1280, W=1280)
    assert len(s.rows) == 96 and s.num_anchors == 33600 and abs(s.macs / 1e9 - 209.074) < 5e-4


def test_survey_appendix_rows():
    """Spot-check rows of SURVEY.md Appendix A."""
    rows = {r.name: r for r in Spec("n", 80).rows}
    r = rows["model.22.proto.cv2"]
    assert (r.c1, r.c2, r.k, r.s, r.h_out) == (64, 64, 3, 1, 160) and r.macs == 943718400
    r = rows["model.22.cv3.0.1"]
    assert (r.c1, r.c2, r.k) == (80, 80, 3) and r.unfused_params == 57760
    r = rows["model.22.proto.upsample"]
    assert r.kind == 2 and r.unfused_params == 16448 and r.macs == 104857600
    r = rows["model.9.cv2"]
    assert (r.c1, r.c2) == (512, 256)


def test_oracle_forward_shapes(lib_built):
    vti_amd = lib_built
    from oracle.model import OracleModel
    eng = vti_amd.Engine("n", 2, H=64, W=96, max_batch=1)     # host-only plan: no GPU touched
    blob = vti_amd.random_weights(eng, seed=3)
    om = OracleModel(blob, 64, 96, "fp32")
    frames = np.random.default_rng(0).integers(0, 256, (2, 64, 96, 3), dtype=np.uint8)
    pred, proto = om.forward_u8(frames)
    A = 8 * 12 + 4 * 6 + 2 * 3
    assert pred.shape == (2, 4 + 2 + 32, A) and proto.shape == (2, 32, 16, 24)
    assert torch.isfinite(pred).all() and torch.isfinite(proto).all()
    assert (pred[:, 4:6] >= 0).all() and (pred[:, 4:6] <= 1).all()
    # fp16-emulation mode stays close to fp32
    p16, _ = OracleModel(blob, 64, 96, "fp16").forward_u8(frames)
    assert (p16[:, 4:6] - pred[:, 4:6]).abs().max() < 2e-2
