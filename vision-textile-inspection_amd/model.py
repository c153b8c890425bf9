"""Drop-in mirror of the `ultralytics.YOLO` protocol as the reference consumes it.

    model = YOLO(path)                                         measurement.py:145
    results = model.predict(rgb, verbose=False, conf=..., iou=..., max_det=..., imgsz=960)
                                                               measurement.py:208-210
    r = results[0]; r.boxes.cls / .xyxy / .conf; len(r.boxes); r.masks.data[idx]; model.names
                                   measurement.py:74-75,242-245; Utils/check_model.py:170-209,341

Same names, argument meaning and error behaviour (any failure raises; the caller's try/except
at measurement.py:207-216 turns it into an error dict).  Everything numeric runs in libvti.so.
"""
import math
import os

import numpy as np
import torch

from .engine import Engine, unpack_bits
from .weights import random_weights, unpack_container


def letterbox_shape(H0, W0, imgsz, auto=True, stride=32):
    """Ultralytics LetterBox output size."""
    new_shape = (imgsz, imgsz) if isinstance(imgsz, int) else tuple(imgsz)
    r = min(new_shape[0] / H0, new_shape[1] / W0)
    new_w, new_h = int(round(W0 * r)), int(round(H0 * r))
    dw, dh = new_shape[1] - new_w, new_shape[0] - new_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw, dh = dw / 2, dh / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_h + top + bottom, new_w + left + right


class Boxes:
    """Results.boxes: xyxy in ORIGINAL frame pixels, conf-descending."""

    def __init__(self, data, orig_shape):
        self.data = data                 # f32 [N,6]: x1,y1,x2,y2,conf,cls
        self.orig_shape = orig_shape

    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, 4]

    @property
    def cls(self):
        return self.data[:, 5]

    def __len__(self):
        return self.data.shape[0]


class Masks:
    """Results.masks: `.data` is f32 0/1 [N,H,W] at the LETTERBOXED size, as Ultralytics returns it."""

    def __init__(self, data_u8, orig_shape):
        self.data_u8 = data_u8           # u8 [N,H,W] device tensor (0/1)
        self.orig_shape = orig_shape
        self._f = None

    @property
    def data(self):
        if self._f is None:
            self._f = self.data_u8.float()
        return self._f

    @property
    def xy(self):
        raise NotImplementedError("polygon masks (.xy) are not produced; use .data")

    def __len__(self):
        return self.data_u8.shape[0]


class Results:
    def __init__(self, orig_shape, names, boxes, masks, dets=None):
        self.orig_shape = orig_shape
        self.names = names
        self.boxes = boxes
        self.masks = masks               # None when there are no detections (as Ultralytics)
        self.dets = dets                 # raw rows incl. mask coefficients, letterboxed px

    def __len__(self):
        return len(self.boxes)


class YOLO:
    """`YOLO(path)` takes a VTIW1 container; `YOLO(None, scale=, nc=, seed=)` makes seeded random weights."""

    def __init__(self, model=None, *, scale="n", nc=80, seed=1, cls_bias=None, dtype="fp16", device=0,
                 names=None, mask_mode="logit", max_batch=64):
        self._blob = None
        if isinstance(model, (bytes, bytearray, memoryview)):
            self._blob = bytes(model)
        elif isinstance(model, (str, os.PathLike)):
            with open(model, "rb") as f:
                self._blob = f.read()
        elif model is not None:
            raise TypeError("model must be a path, bytes or None")
        if self._blob is not None:
            meta, _, _ = unpack_container(self._blob)
            scale, nc = meta["scale"], meta["nc"]
            self._nm, self._reg_max = meta["nm"], meta["reg_max"]
        else:
            self._nm, self._reg_max = 32, 16
        self.scale, self.nc, self.dtype, self.device = scale, nc, dtype, device
        self._seed, self._cls_bias = seed, cls_bias
        self.mask_mode = mask_mode
        self.max_batch = max_batch
        self.names = names if names is not None else {i: f"class{i}" for i in range(nc)}
        self._engines = {}

    def _engine(self, H, W, B):
        key = (H, W)
        eng = self._engines.get(key)
        if eng is None or eng.max_batch < B:
            eng = Engine(self.scale, self.nc, self._nm, self._reg_max, H, W, max(B, 1), self.dtype)
            if self._blob is None:
                self._blob = random_weights(eng, self._seed, self._cls_bias)
            eng.load_weights(self._blob, self.device)
            self._engines[key] = eng
        return eng

    def _to_device_batch(self, source):
        if isinstance(source, torch.Tensor):
            t = source
        elif isinstance(source, (list, tuple)):
            t = torch.from_numpy(np.stack([np.asarray(s) for s in source]))
        else:
            t = torch.from_numpy(np.ascontiguousarray(source))
        if t.dim() == 3:
            t = t.unsqueeze(0)
        if t.dim() != 4 or t.shape[-1] != 3 or t.dtype != torch.uint8:
            raise ValueError(f"source must be uint8 HxWx3 or BxHxWx3, got {t.dtype} {tuple(t.shape)}")
        dev = torch.device("cuda", self.device) if isinstance(self.device, int) else torch.device(self.device)
        return t.to(dev, non_blocking=True).contiguous()

    @torch.inference_mode()
    def predict(self, source=None, *, verbose=False, conf=0.25, iou=0.7, max_det=300, imgsz=640,
                agnostic_nms=False, swap_rb=True, **_ignored):
        """Returns list[Results], one per frame.  `swap_rb=True` keeps Ultralytics' channel flip of
        ndarray sources."""
        if source is None:
            raise ValueError("predict() needs a source")
        frames = self._to_device_batch(source)
        B, H0, W0, _ = frames.shape
        H, W = letterbox_shape(H0, W0, imgsz)
        eng = self._engine(H, W, B)
        inp = frames if (H0, W0) == (H, W) else eng.letterbox(frames)
        pred, proto = eng.forward(inp, swap_rb)
        dets, counts = eng.nms(pred, conf, iou, max_det, agnostic_nms)
        masks, offsets = eng.masks(dets, counts, proto, self.mask_mode, "u8")
        xyxy = eng.scale_boxes(dets, counts, H0, W0)
        cnt = counts.cpu().tolist()
        off = offsets.cpu().tolist()
        out = []
        for b in range(B):
            n = cnt[b]
            data = torch.cat((xyxy[b, :n], dets[b, :n, 4:6]), 1)
            m = Masks(masks[off[b]:off[b] + n], (H0, W0)) if n else None
            out.append(Results((H0, W0), self.names, Boxes(data, (H0, W0)), m, dets[b, :n]))
        return out

    __call__ = predict
