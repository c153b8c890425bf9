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
from .polygons import masks2segments, scale_coords
from .weights import random_weights, unpack_container


def letterbox_shape(H0, W0, imgsz, auto=True, stride=32):
    """Ultralytics LetterBox output size."""
    new_shape = (imgsz, imgsz) if isinstance(imgsz, int) else tuple(imgsz)
    new_shape = tuple(max(math.ceil(x / stride) * stride, stride) for x in new_shape)   # Ultralytics check_imgsz: round up to the stride
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
    """Results.masks: `.data` is f32 0/1 [N,H,W] at the LETTERBOXED size, as Ultralytics returns it.  The engine writes
    bit-packed masks (u8 [N,H,W/8]); they are expanded on the device the first time `.data` / `.data_u8` is read."""

    def __init__(self, bits, W, orig_shape):
        self.bits = bits                 # u8 [N,H,W/8] device tensor, LSB-first
        self._W = W
        self.orig_shape = orig_shape
        self._u8 = None
        self._f = None
        self._xy = None

    @property
    def data_u8(self):
        if self._u8 is None:
            self._u8 = unpack_bits(self.bits, self._W)
        return self._u8

    @property
    def data(self):
        if self._f is None:
            self._f = self.data_u8.float()
        return self._f

    @property
    def xy(self):
        """One float32 [n,2] (x, y) polygon per instance in ORIGINAL frame pixels: the largest outer contour of the mask
        (Ultralytics masks2segments + scale_coords; Utils/check_model.py:185-188 fills it).  Host side, computed on first use."""
        if self._xy is None:
            H, W = self.bits.shape[1], self._W
            segs = masks2segments(self.data_u8.cpu().numpy())
            self._xy = [scale_coords((H, W), s, self.orig_shape) for s in segs]
        return self._xy

    def __len__(self):
        return self.bits.shape[0]


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

    def __init__(self, model=None, *, scale="n", nc=80, seed=1, cls_bias=None, dtype="h2", device=0,
                 names=None, mask_mode="logit", max_batch=64, drop_empty_masks=False):
        """dtype: "h2" (default; split-fp16 storage on the fp16 matrix pipe -- results within the reference tolerance of the fp32 CPU
        path: mask IoU >= 0.999, |d box| < 1e-3), "fp32" (exact-f32 MFMA, slower) or "fp16" (fastest; boxes drift ~1 px and masks to
        IoU ~0.92-0.99 on untrained margins).  drop_empty_masks: newer Ultralytics releases drop instances whose thresholded mask is
        empty (`keep = masks.sum((-2, -1)) > 0`); older ones (and the default here) return them."""
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
        self.drop_empty_masks = drop_empty_masks
        self.max_batch = max_batch
        self._outs = {}
        self.names = names if names is not None else {i: f"class{i}" for i in range(nc)}
        self._engines = {}

    def _engine(self, H, W, B, max_det=300):
        key = (H, W)
        eng = self._engines.get(key)
        # one vti_masks call handles max_batch * 512 instances (VTI_MASK_SLOTS_PER_FRAME): a predict() with a larger max_det
        # gets an engine with proportionally more batch slots, so every detection still gets its mask
        need = max(B, -(-B * max_det // 512), 1)
        if eng is None or eng.max_batch < need:
            eng = Engine(self.scale, self.nc, self._nm, self._reg_max, H, W, need, self.dtype)
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
        eng = self._engine(H, W, B, max_det)
        # the whole pipeline is enqueued without a host read in between (vti_predict: letterbox -> net -> NMS -> bit-packed
        # masks for up to B * max_det instances -> scale_boxes); the counts are read once, at the end, to cut the Results.
        # The output set (~1 GB for 64 frames x 300 slots of 640x640 bit masks) is allocated once per (engine, B, max_det) and
        # reused by later calls; what a Results object keeps are COPIES of its own rows (a few KB .. MB per frame).
        key = (id(eng), B, max_det)
        o = self._outs.get(key)
        if o is None:
            self._outs.clear()              # one cached set at a time: a new shape replaces the old one
            o = self._outs[key] = eng.alloc_outputs(B, max_det, B * max_det, "bits", frames.device)
        eng.predict_into(frames, o, conf, iou, max_det, agnostic_nms, swap_rb, self.mask_mode, "bits")
        dets, xyxy, masks = o["dets"], o["xyxy"], o["masks"]
        nonempty = None
        if self.drop_empty_masks:           # m00 of every live slot straight from the bit-packed masks (vti_mask_stats_bits)
            nonempty = eng.mask_stats_bits(masks, H, W, offsets=o["offsets"])[:, 0] > 0
        cnt = o["counts"].cpu().tolist()
        off = o["offsets"].cpu().tolist()
        out = []
        for b in range(B):
            n = cnt[b]
            data = torch.cat((xyxy[b, :n], dets[b, :n, 4:6]), 1)
            mb, db = masks[off[b]:off[b] + n], dets[b, :n]
            if nonempty is not None and n:
                keep = nonempty[off[b]:off[b] + n]
                data, mb, db = data[keep], mb[keep], db[keep]
                n = int(data.shape[0])
            else:
                mb, db = mb.clone(), db.clone()
            m = Masks(mb, W, (H0, W0)) if n else None
            r = Results((H0, W0), self.names, Boxes(data, (H0, W0)), m, db)
            r._engine = eng
            out.append(r)
        return out

    __call__ = predict
