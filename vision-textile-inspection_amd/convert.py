"""Ultralytics YOLOv8-seg checkpoint -> VTIW1 fused-weight container.

What `YOLO("single_needle_model.pt")` unpickles in the reference (measurement.py:145, config.py:67) is a
module tree whose state dict uses the names below; this module folds every BatchNorm into its conv
(`w' = w * gamma / sqrt(var + eps)`, `b' = beta - mean * gamma / sqrt(var + eps)`, eps = 1e-3 -- what
Ultralytics' `fuse_conv_and_bn` does before inference) and writes the flat container `vti_load_weights`
consumes.  The state-dict half (`convert_state_dict`) needs only torch/numpy and is what the tests cover;
`convert_checkpoint` additionally needs the `ultralytics` package to unpickle the .pt and therefore runs
wherever the reference runs (it is not importable on the build/GPU boxes of this repo, and no .pt ships
with the reference: .MISSING_LARGE_BLOBS).

State-dict names per conv-table row `name` (Engine.conv_table()):
  kind 0  Conv-BN-SiLU      name.conv.weight, name.bn.{weight,bias,running_mean,running_var}
  kind 1  plain Conv2d      name.weight, name.bias            (model.22.cv2|cv3|cv4.<level>.2)
  kind 2  ConvTranspose2d   name.weight [IOHW], name.bias     (model.22.proto.upsample)
`model.22.dfl.conv.weight` (the frozen arange(16)) is implied by reg_max and not stored.

    python -m vti_amd.convert best_Model.pt best_Model.vtiw        # on a machine with ultralytics installed
"""
import sys

import numpy as np

from .weights import pack_container

BN_EPS = 1e-3        # ultralytics.nn.modules.conv.Conv: nn.BatchNorm2d(c2, eps=0.001, momentum=0.03)


def _np(t):
    return t.detach().cpu().float().numpy() if hasattr(t, "detach") else np.asarray(t, dtype=np.float32)


def convert_state_dict(state_dict, engine):
    """Fold BN and pack `state_dict` (name -> tensor/ndarray) for `engine`'s conv table -> VTIW1 bytes.
    Raises KeyError/ValueError with the offending tensor name if the checkpoint does not match the plan
    (scale, nc, nm), so a wrong model never loads silently."""
    tensors = {}
    for t in engine.conv_table():
        name, c1, c2, k, kind = t["name"], t["c1"], t["c2"], t["k"], t["kind"]
        if kind == 0:
            w = _np(state_dict[name + ".conv.weight"]).astype(np.float64)
            g = _np(state_dict[name + ".bn.weight"]).astype(np.float64)
            beta = _np(state_dict[name + ".bn.bias"]).astype(np.float64)
            mean = _np(state_dict[name + ".bn.running_mean"]).astype(np.float64)
            var = _np(state_dict[name + ".bn.running_var"]).astype(np.float64)
            s = g / np.sqrt(var + BN_EPS)
            w = w * s[:, None, None, None]
            b = beta - mean * s
            shape = (c2, c1, k, k)
        else:
            w = _np(state_dict[name + ".weight"]).astype(np.float64)
            b = _np(state_dict[name + ".bias"]).astype(np.float64)
            shape = (c1, c2, k, k) if kind == 2 else (c2, c1, k, k)
        if tuple(w.shape) != shape or b.shape != (c2,):
            raise ValueError(f"{name}: checkpoint tensor {tuple(w.shape)} does not match the plan {shape} "
                             f"(scale/nc/nm of the engine must be the checkpoint's)")
        tensors[name] = (w.astype(np.float32), b.astype(np.float32))
    return pack_container(engine.scale, engine.nc, engine.nm, engine.reg_max, engine.conv_table(), tensors)


def convert_checkpoint(pt_path, out_path=None, imgsz=640):
    """Unpickle an Ultralytics segmentation checkpoint and write its VTIW1 container.  Needs `ultralytics`."""
    try:
        from ultralytics import YOLO       # noqa: WPS433  (only where the reference's own stack is installed)
    except ImportError as e:               # pragma: no cover - not installable here (no network)
        raise RuntimeError("convert_checkpoint needs the `ultralytics` package to unpickle the .pt; run it on the "
                           "reference's machine, or pass a state dict to convert_state_dict") from e
    from . import Engine
    model = YOLO(pt_path).model            # pragma: no cover
    seg = model.model[-1]
    scale = str(model.yaml.get("scale", "n"))
    nc, nm = int(seg.nc), int(seg.nm)
    eng = Engine(scale, nc, H=imgsz, W=imgsz, max_batch=1, nm=nm)
    blob = convert_state_dict(model.state_dict(), eng)
    if out_path:
        with open(out_path, "wb") as f:
            f.write(blob)
    return blob, dict(scale=scale, nc=nc, nm=nm, names=dict(getattr(model, "names", {})))


if __name__ == "__main__":                 # pragma: no cover
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    _, meta = convert_checkpoint(sys.argv[1], sys.argv[2])
    print(f"wrote {sys.argv[2]}: {meta}")
