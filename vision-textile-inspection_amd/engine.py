"""Thin host wrapper around one libvti context: owns the torch-ROCm tensors the C ABI writes into.

PyTorch is plumbing here (device memory + the current HIP stream); all arithmetic runs in
libvti.so's HIP kernels.  Replaces the predictor object Ultralytics builds lazily on the first
`model.predict(...)` (reference: measurement.py:208-210).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import VtiConvInfo, VtiDesc, check, lib

# "h2": split-fp16 storage (every element an fp16 (hi, lo) pair, all products on the fp16 matrix pipe): the dtype whose results
# meet the reference tolerance (mask IoU >= 0.999, |d box| < 1e-3) at a multiple of the fp32 engine's rate -- include/vti.h
DTYPES = {"fp16": _lib.VTI_F16, "f16": _lib.VTI_F16, "half": _lib.VTI_F16,
          "fp32": _lib.VTI_F32, "f32": _lib.VTI_F32, "float": _lib.VTI_F32,
          "h2": _lib.VTI_H2, "fp16x2": _lib.VTI_H2, "split": _lib.VTI_H2}
_DTYPE_NAME = {_lib.VTI_F16: "fp16", _lib.VTI_F32: "fp32", _lib.VTI_H2: "h2"}
MASK_MODES = {"logit": _lib.VTI_MASK_LOGIT, "sigmoid": _lib.VTI_MASK_SIGMOID}
PACKINGS = {"u8": _lib.VTI_PACK_U8, "bits": _lib.VTI_PACK_BITS}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    """One (model description, input size, max batch, dtype) context."""

    def __init__(self, scale="n", nc=80, nm=32, reg_max=16, H=640, W=640, max_batch=1, dtype="fp16"):
        self.scale, self.nc, self.nm, self.reg_max = scale, nc, nm, reg_max
        self.H, self.W, self.max_batch = H, W, max_batch
        self.dtype = _DTYPE_NAME[DTYPES[dtype]]
        self._ctx = C.c_void_p(0)
        desc = VtiDesc(scale.encode()[:1], nc, nm, reg_max, H, W, max_batch, DTYPES[dtype])
        rc = lib().vti_create(C.byref(desc), C.byref(self._ctx))
        if rc != 0:
            raise _lib.VtiError(rc, lib().vti_last_error(None).decode())
        self.device = None
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_ctx", None) and self._ctx.value:
                lib().vti_destroy(self._ctx)
                self._ctx = C.c_void_p(0)
        except Exception:
            pass

    # ---- host-only plan introspection ------------------------------------------------
    def conv_table(self):
        out = []
        info = VtiConvInfo()
        for i in range(lib().vti_num_convs(self._ctx)):
            check(self._ctx, lib().vti_conv_at(self._ctx, i, C.byref(info)))
            out.append(dict(name=info.name.decode(), c1=info.c1, c2=info.c2, k=info.k, s=info.s, kind=info.kind,
                            h_in=info.h_in, w_in=info.w_in, h_out=info.h_out, w_out=info.w_out, macs=info.macs,
                            tile=(info.tile_h, info.tile_w), waves_n=info.waves_n, nrep=info.nrep, lds=info.lds_bytes,
                            fused=bool(info.fused), persistent=bool(info.persistent)))
        return out

    @property
    def num_anchors(self):
        return lib().vti_num_anchors(self._ctx)

    @property
    def fused_params(self):
        return lib().vti_fused_params(self._ctx)

    @property
    def macs_per_frame(self):
        return lib().vti_macs_per_frame(self._ctx)

    @property
    def workspace_bytes(self):
        return lib().vti_workspace_bytes(self._ctx)

    @property
    def num_launches(self):
        return lib().vti_num_launches(self._ctx)

    @property
    def no(self):
        return 4 + self.nc + self.nm

    @property
    def torch_dtype(self):
        return torch.float16 if self.dtype == "fp16" else torch.float32

    # ---- device setup ------------------------------------------------------------------
    def load_weights(self, blob, device=0):
        """blob: bytes of a VTIW1 container.  Uploads to `device` and allocates the workspace."""
        if not torch.cuda.is_available():
            raise RuntimeError("vti_amd needs a ROCm GPU: torch.cuda.is_available() is False (no CPU fallback)")
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        torch.cuda.set_device(dev)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        check(self._ctx, lib().vti_load_weights(self._ctx, buf, len(blob), dev.index or 0))
        self.device = dev
        self._ws = torch.empty(self.workspace_bytes + 256, dtype=torch.uint8, device=dev)
        base = self._ws.data_ptr()
        aligned = (base + 255) & ~255
        check(self._ctx, lib().vti_set_workspace(self._ctx, C.c_void_p(aligned), self.workspace_bytes))
        return self

    # ---- stages ------------------------------------------------------------------------
    def letterbox(self, frames):
        """frames: u8 [B,H0,W0,3] on device -> u8 [B,H,W,3]."""
        B, H0, W0, _ = frames.shape
        out = torch.empty((B, self.H, self.W, 3), dtype=torch.uint8, device=frames.device)
        check(self._ctx, lib().vti_letterbox(self._ctx, _ptr(frames), B, H0, W0, _ptr(out), _stream()))
        return out

    def alloc_pred(self, B, device):
        """pred in Ultralytics' logical shape [B, 4+nc+nm, A] over the library's ANCHOR-MAJOR memory [B, A, 4+nc+nm]
        (each anchor's box, scores and coefficients are one contiguous row: that is how the head towers and NMS touch it)."""
        return torch.empty((B, self.num_anchors, self.no), dtype=torch.float32, device=device).transpose(1, 2)

    @staticmethod
    def _anchor_major(pred):
        """-> a tensor with pred's values whose memory is [B, A, no] contiguous (no copy for tensors from alloc_pred)."""
        pa = pred.transpose(1, 2)
        return pa if pa.is_contiguous() else pa.contiguous()

    def forward(self, inp, swap_rb=True, pred=None, proto=None, best=None):
        """inp: u8 [B,H,W,3] letterboxed -> pred f32 [B,4+nc+nm,A] (a view of anchor-major memory, see alloc_pred),
        proto T [B,H/4,W/4,nm] (NHWC).  best: optional f32 [B,A,2] that receives (best class score, its class) per anchor
        for nms(best=...) -- vti_forward_scored."""
        self._check_input(inp, (self.H, self.W))
        B = inp.shape[0]
        if pred is None:
            pred = self.alloc_pred(B, inp.device)
        elif not pred.transpose(1, 2).is_contiguous():
            raise ValueError("pred must come from Engine.alloc_pred / alloc_outputs (anchor-major memory)")
        if proto is None:
            proto = torch.empty((B, self.H // 4, self.W // 4, self.nm), dtype=self.torch_dtype, device=inp.device)
        if best is None:
            check(self._ctx, lib().vti_forward(self._ctx, _ptr(inp), B, int(bool(swap_rb)), _ptr(pred), _ptr(proto), _stream()))
        else:
            self._check_best(best, B)
            check(self._ctx, lib().vti_forward_scored(self._ctx, _ptr(inp), B, int(bool(swap_rb)), _ptr(pred), _ptr(proto), _ptr(best), _stream()))
        return pred, proto

    def alloc_best(self, B, device=None):
        return torch.empty((B, self.num_anchors, 2), dtype=torch.float32, device=device or self.device)

    def _check_best(self, best, B):
        if best.dtype != torch.float32 or tuple(best.shape) != (B, self.num_anchors, 2) or not best.is_contiguous():
            raise ValueError("best must be a contiguous f32 [B, A, 2] tensor (Engine.alloc_best)")

    def nms(self, pred, conf=0.25, iou=0.7, max_det=300, agnostic=False, dets=None, counts=None, best=None):
        """pred: [B, 4+nc+nm, A] (any layout; tensors from forward()/alloc_pred are used in place).  best: the pairs forward(best=...)
        wrote for THIS pred (vti_nms_scored: the candidate filter reads them instead of the class scores)."""
        pred = self._anchor_major(pred)
        B = pred.shape[0]
        if dets is None:
            dets = torch.empty((B, max_det, 6 + self.nm), dtype=torch.float32, device=pred.device)
        if counts is None:
            counts = torch.empty((B,), dtype=torch.int32, device=pred.device)
        if best is None:
            check(self._ctx, lib().vti_nms(self._ctx, _ptr(pred), B, float(conf), float(iou), int(max_det), int(bool(agnostic)),
                                           _ptr(dets), _ptr(counts), _stream()))
        else:
            self._check_best(best, B)
            check(self._ctx, lib().vti_nms_scored(self._ctx, _ptr(pred), _ptr(best), B, float(conf), float(iou), int(max_det),
                                                  int(bool(agnostic)), _ptr(dets), _ptr(counts), _stream()))
        return dets, counts

    def masks(self, dets, counts, proto, mode="logit", packing="u8", capacity=None, masks=None, offsets=None):
        """-> (masks u8 [capacity,H,W] or [capacity,H,W/8], offsets i32 [B+1]).  With capacity=None the
        detection counts are read back first (one small D2H sync) to size the output exactly."""
        B, max_det = dets.shape[0], dets.shape[1]
        if capacity is None:
            capacity = int(counts.clamp(0, max_det).sum().item())
        wb = self.W if packing == "u8" else self.W // 8
        if masks is None:
            masks = torch.empty((capacity, self.H, wb), dtype=torch.uint8, device=dets.device)
        if offsets is None:
            offsets = torch.empty((B + 1,), dtype=torch.int32, device=dets.device)
        check(self._ctx, lib().vti_masks(self._ctx, _ptr(dets), _ptr(counts), _ptr(proto), B, max_det, MASK_MODES[mode],
                                         PACKINGS[packing], _ptr(masks) if capacity else C.c_void_p(0), capacity,
                                         _ptr(offsets), _stream()))
        return masks, offsets

    def scale_boxes(self, dets, counts, H0, W0, xyxy=None):
        B, max_det = dets.shape[0], dets.shape[1]
        if xyxy is None:
            xyxy = torch.empty((B, max_det, 4), dtype=torch.float32, device=dets.device)
        check(self._ctx, lib().vti_scale_boxes(self._ctx, _ptr(dets), _ptr(counts), B, max_det, H0, W0, _ptr(xyxy), _stream()))
        return xyxy

    def alloc_outputs(self, B, max_det, capacity, packing="bits", device=None):
        """Preallocated output set for predict_into (the no-sync, graph-friendly form)."""
        dev = device or self.device
        wb = self.W if packing == "u8" else self.W // 8
        return dict(
            pred=self.alloc_pred(B, dev),
            proto=torch.empty((B, self.H // 4, self.W // 4, self.nm), dtype=self.torch_dtype, device=dev),
            dets=torch.empty((B, max_det, 6 + self.nm), dtype=torch.float32, device=dev),
            counts=torch.empty((B,), dtype=torch.int32, device=dev),
            masks=torch.empty((capacity, self.H, wb), dtype=torch.uint8, device=dev),
            offsets=torch.empty((B + 1,), dtype=torch.int32, device=dev),
            xyxy=torch.empty((B, max_det, 4), dtype=torch.float32, device=dev),
            input=torch.empty((B, self.H, self.W, 3), dtype=torch.uint8, device=dev),
            best=self.alloc_best(B, dev),
        )

    def predict_into(self, frames, out, conf=0.25, iou=0.7, max_det=300, agnostic=False, swap_rb=True,
                     mask_mode="logit", packing="bits"):
        """Whole pipeline (letterbox -> net -> NMS -> masks -> scale_boxes) on the current stream with
        no host synchronisation; `out` from alloc_outputs()."""
        B, H0, W0, _ = frames.shape
        capacity = out["masks"].shape[0]
        check(self._ctx, lib().vti_predict(
            self._ctx, _ptr(frames), B, H0, W0, int(bool(swap_rb)), float(conf), float(iou), int(max_det),
            int(bool(agnostic)), MASK_MODES[mask_mode], PACKINGS[packing], _ptr(out["input"]), _ptr(out["pred"]),
            _ptr(out["proto"]), _ptr(out["dets"]), _ptr(out["counts"]), _ptr(out["masks"]), capacity,
            _ptr(out["offsets"]), _ptr(out["xyxy"]), _stream()))
        return out

    # ---- consumer-side reductions (measurement.py:70-86,160-185,300-330) ----------------
    def mask_to_frame(self, masks_u8, H0, W0):
        n, H, W = masks_u8.shape
        bitmaps = torch.empty((n, H0, W0), dtype=torch.uint8, device=masks_u8.device)
        nonzero = torch.empty((n,), dtype=torch.int32, device=masks_u8.device)
        check(self._ctx, lib().vti_mask_to_frame(self._ctx, _ptr(masks_u8), n, H, W, H0, W0, _ptr(bitmaps), _ptr(nonzero), _stream()))
        return bitmaps, nonzero

    def union_envelope(self, bitmaps, select):
        n, H0, W0 = bitmaps.shape
        sel = torch.as_tensor(select, dtype=torch.int32, device=bitmaps.device)
        uni = torch.empty((H0, W0), dtype=torch.uint8, device=bitmaps.device)
        env = torch.empty((W0,), dtype=torch.int32, device=bitmaps.device)
        check(self._ctx, lib().vti_union_envelope(self._ctx, _ptr(bitmaps), _ptr(sel), sel.numel(), H0, W0, _ptr(uni), _ptr(env), _stream()))
        return uni, env

    def mask_stats(self, bitmaps):
        n, H0, W0 = bitmaps.shape
        stats = torch.empty((n, 5), dtype=torch.int64, device=bitmaps.device)
        check(self._ctx, lib().vti_mask_stats(self._ctx, _ptr(bitmaps), n, H0, W0, _ptr(stats), _stream()))
        return stats

    # ---- the same reductions straight from bit-packed masks (no frame-sized bitmaps) ------
    def mask_stats_bits(self, masks_bits, H0, W0, stats=None, offsets=None):
        """masks_bits u8 [n,H,W/8] (VTI_PACK_BITS) -> i64 [n,5] = m00, m10, m01, min_col, max_col of each instance's
        H0 x W0 nearest-resized bitmap (measurement.py:70-86,302-318).  `offsets` (i32 [B+1] from masks()): slots at and
        beyond offsets[B] of a fixed-capacity buffer are skipped and report the empty mask."""
        n, H, wb = masks_bits.shape
        if stats is None:
            stats = torch.empty((n, 5), dtype=torch.int64, device=masks_bits.device)
        n_live = C.c_void_p(offsets.data_ptr() + 4 * (offsets.numel() - 1)) if offsets is not None else C.c_void_p(0)
        check(self._ctx, lib().vti_mask_stats_bits(self._ctx, _ptr(masks_bits), n, n_live, H, wb * 8, H0, W0, _ptr(stats), _stream()))
        return stats

    def envelope_bits(self, masks_bits, offsets, dets, cls, H0, W0, envelope=None):
        """Per frame: lower envelope i32 [B,W0] of the union of its instances of class `cls` (< 0: all)
        (measurement.py:160-185 on the bitmaps of measurement.py:70-86)."""
        B, max_det = dets.shape[0], dets.shape[1]
        if envelope is None:
            envelope = torch.empty((B, W0), dtype=torch.int32, device=dets.device)
        check(self._ctx, lib().vti_envelope_bits(self._ctx, _ptr(masks_bits), _ptr(offsets), _ptr(dets), B, max_det,
                                                 masks_bits.shape[0], int(cls), H0, W0, _ptr(envelope), _stream()))
        return envelope

    # ---- test hook ---------------------------------------------------------------------
    def debug_conv_output(self, i, B):
        t = self.conv_table()[i]
        out = torch.empty((B, t["c2"], t["h_out"], t["w_out"]), dtype=torch.float32, device=self.device)
        check(self._ctx, lib().vti_debug_conv_output(self._ctx, i, B, _ptr(out), _stream()))
        return out

    def _check_input(self, t, hw):
        if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3 or tuple(t.shape[1:3]) != tuple(hw):
            raise ValueError(f"expected uint8 [B,{hw[0]},{hw[1]},3], got {t.dtype} {tuple(t.shape)}")
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("input must be a contiguous device tensor")
        if t.shape[0] > self.max_batch:
            raise ValueError(f"batch {t.shape[0]} exceeds max_batch {self.max_batch}")


def _f64(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())
    if a.size != n:
        raise ValueError(f"expected {n} float64 values, got {a.size}")
    return a


def pixels_to_world(uv, K, dist, R, t):
    """measurement.py:50-65 for n points at once.  uv f64 [n,2] device tensor; K (3x3), dist (5), R (3x3), t (3) host arrays.
    -> (xyz f64 [n,3] device, valid i32 [n])."""
    uv = uv.to(torch.float64).contiguous()
    n = uv.shape[0]
    xyz = torch.empty((n, 3), dtype=torch.float64, device=uv.device)
    valid = torch.empty((n,), dtype=torch.int32, device=uv.device)
    Kh, dh, Rh, th = _f64(K, 9), _f64(dist, 5), _f64(R, 9), _f64(t, 3)
    rc = lib().vti_pixels_to_world(None, _ptr(uv), n, Kh.ctypes.data_as(C.c_void_p), dh.ctypes.data_as(C.c_void_p),
                                   Rh.ctypes.data_as(C.c_void_p), th.ctypes.data_as(C.c_void_p), _ptr(xyz), _ptr(valid), _stream())
    check(None, rc)
    return xyz, valid


def kmeans1d2(values, counts, max_iters=10):
    """measurement.py:88-113 batched.  values f64 [B,max_n] device, counts i32 [B] -> (labels i32 [B,max_n], centers f64 [B,2])."""
    values = values.to(torch.float64).contiguous()
    B, max_n = values.shape
    labels = torch.empty((B, max_n), dtype=torch.int32, device=values.device)
    centers = torch.empty((B, 2), dtype=torch.float64, device=values.device)
    check(None, lib().vti_kmeans1d2(None, _ptr(values), _ptr(counts), B, max_n, int(max_iters), _ptr(labels), _ptr(centers), _stream()))
    return labels, centers


H2_SCALE = 16.0     # conv_dev.h: H2_SX


def h2_encode(t):
    """float tensor -> the h2 engine's storage: per element the fp16 pair (hi | lo << 16) of value * 16, returned as float32-typed
    bits (same shape).  Host-side helper for tests and the single-conv debug entry point; the engine itself never needs it."""
    s = t.float() * H2_SCALE
    hi = s.half()
    lo = (s - hi.float()).half()
    bits = hi.view(torch.int16).to(torch.int32) & 0xFFFF | (lo.view(torch.int16).to(torch.int32) << 16)
    return bits.view(torch.float32)


def h2_decode(t):
    """inverse of h2_encode (exact: hi + lo has at most 24 significant bits)."""
    bits = t.view(torch.int32)
    hi = (bits & 0xFFFF).to(torch.int16).view(torch.float16).float()
    lo = (bits >> 16).to(torch.int16).view(torch.float16).float()
    return (hi + lo) / H2_SCALE


def unpack_bits(bits, W):
    """u8 [...,W/8] LSB-first -> u8 [...,W] of 0/1 (host-side convenience for bit-packed masks)."""
    b = bits.unsqueeze(-1)
    sh = torch.arange(8, device=bits.device, dtype=torch.uint8)
    return ((b >> sh) & 1).reshape(*bits.shape[:-1], W)


def to_numpy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def debug_conv2d(x, w, b, k, s, kind=0, dtype="fp16", res=None, out=None, in_coff=0, c1=None, out_coff=0, out_ld=None,
                 out_f32=False, swap_rb=False, tile=(0, 0), waves_n=0, nrep=0, iters=1):
    """Run ONE conv of the engine's conv family (vti_debug_conv2d): unit tests / micro-benchmarks.
    x: device tensor NHWC [B,H,W,ld] of the engine dtype (or uint8 [B,H,W,3] for the stem);
    w: f32 OIHW (kind 2: IOHW) numpy/torch on host; returns (out NHWC, ms_per_launch, cfg)."""
    B, H, W, ld = x.shape
    w = np.ascontiguousarray(to_numpy(w), dtype=np.float32)
    b = np.ascontiguousarray(to_numpy(b), dtype=np.float32)
    c2 = w.shape[1] if kind == 2 else w.shape[0]
    if c1 is None:
        c1 = w.shape[0] if kind == 2 else w.shape[1]
    Ho, Wo = (2 * H, 2 * W) if kind == 2 else ((H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1)
    tdt = torch.float32 if (out_f32 or DTYPES[dtype] != _lib.VTI_F16) else torch.float16      # h2: 4-byte pairs, carried as f32 bits
    if out is None:
        out_ld = out_ld or (out_coff + c2)
        out = torch.zeros((B, Ho, Wo, out_ld), dtype=tdt, device=x.device)
    else:
        out_ld = out.shape[3]
    ms = C.c_float(0)
    cfg = (C.c_int32 * 5)()
    rc = lib().vti_debug_conv2d(DTYPES[dtype], _ptr(x), B, H, W, ld, in_coff, c1, w.ctypes.data_as(C.c_void_p),
                                b.ctypes.data_as(C.c_void_p), c2, k, s, kind, _ptr(res), res.shape[3] if res is not None else 0,
                                0, _ptr(out), out_ld, out_coff, int(out_f32), int(swap_rb), tile[0], tile[1], waves_n, nrep,
                                iters, C.byref(ms), cfg, _stream())
    if rc != 0:
        raise _lib.VtiError(rc, lib().vti_last_error(None).decode())
    return out, ms.value, dict(tile=(cfg[0], cfg[1]), waves_n=cfg[2], nrep=cfg[3], lds=abs(cfg[4]), pk=cfg[4] < 0)
