"""torch-CPU restatement of the YOLOv8-seg forward pass (SegmentationModel, fused BN).

TEST INFRASTRUCTURE -- PARITY UNPINNED (see oracle/__init__.py).
Follows SURVEY.md section 8 rows U2-U5 (Ultralytics 8.x nn/modules: Conv, C2f, Bottleneck,
SPPF, Segment/Detect, Proto, DFL, make_anchors, dist2bbox).  The reference reaches it
through measurement.py:208-210 / Utils/check_model.py:331-337.  All ops are the same
ATen CPU kernels Ultralytics' CPU path dispatches to (F.conv2d, F.max_pool2d,
F.conv_transpose2d, F.interpolate, F.silu, softmax).

mode="fp32": plain fp32 everywhere (the reference's CPU numerics).
mode="fp64": same fp32 weights and input, every op evaluated in double (measures the fp32 noise floor).
mode="fp16": emulates the GPU engine's fp16 storage -- weights and every stored
  activation are rounded to fp16 at exactly the points the engine rounds (after
  bias+SiLU(+residual) of each conv, after the deconv, input after /255); all
  accumulation and the head's final 1x1 convs + decode stay fp32.
"""
import torch
import torch.nn.functional as F

from .spec import Spec, KIND_DECONV
from .blob import read_blob


class OracleModel:
    def __init__(self, blob, H=640, W=640, mode="fp32"):
        meta, convs = read_blob(blob)
        self.meta = meta
        self.spec = Spec(meta["scale"], meta["nc"], meta["nm"], meta["reg_max"], H, W)
        names = [r.name for r in self.spec.rows]
        if names != list(convs.keys()):
            raise ValueError("container conv order does not match the oracle's table")
        self.mode = mode
        self.p = {}
        for r in self.spec.rows:
            c1, c2, k, s, kind, w, b = convs[r.name]
            assert (c1, c2, k, s, kind) == (r.c1, r.c2, r.k, r.s, r.kind), r.name
            w = torch.from_numpy(w.copy())
            b = torch.from_numpy(b.copy())
            if mode == "fp16":
                w = w.half().float()
            if mode == "fp64":
                w, b = w.double(), b.double()
            self.p[r.name] = (w, b, k, s, kind)
        self.taps = None  # optional dict name -> activation, filled when forward(record=True)

    # -- rounding point ---------------------------------------------------
    def q(self, t):
        return t.half().float() if self.mode == "fp16" else t

    # -- modules ----------------------------------------------------------
    def conv(self, x, name, res=None, out_fp32=False):
        w, b, k, s, kind = self.p[name]
        if kind == KIND_DECONV:
            y = F.conv_transpose2d(x, w, b, stride=s)
        else:
            y = F.conv2d(x, w, b, stride=s, padding=k // 2)
            if kind == 0:
                y = F.silu(y)
        if res is not None:
            y = res + y
        y = y if out_fp32 else self.q(y)
        if self.taps is not None:
            self.taps[name] = y
        return y

    def c2f(self, x, idx, n, shortcut):
        y = list(self.conv(x, f"model.{idx}.cv1").chunk(2, 1))
        for j in range(n):
            t = self.conv(y[-1], f"model.{idx}.m.{j}.cv1")
            y.append(self.conv(t, f"model.{idx}.m.{j}.cv2", res=y[-1] if shortcut else None))
        return self.conv(torch.cat(y, 1), f"model.{idx}.cv2")

    def sppf(self, x):
        y = [self.conv(x, "model.9.cv1")]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        return self.conv(torch.cat(y, 1), "model.9.cv2")

    # -- forward ----------------------------------------------------------
    @torch.inference_mode()
    def features(self, x):
        """x: f32 [B,3,H,W] already /255 (and fp16-rounded in fp16 mode) -> (P3,P4,P5)."""
        sp = self.spec
        x = self.conv(x, "model.0")
        x = self.conv(x, "model.1")
        x = self.c2f(x, 2, sp.reps[0], True)
        x = self.conv(x, "model.3")
        x4 = self.c2f(x, 4, sp.reps[1], True)
        x = self.conv(x4, "model.5")
        x6 = self.c2f(x, 6, sp.reps[2], True)
        x = self.conv(x6, "model.7")
        x = self.c2f(x, 8, sp.reps[3], True)
        x9 = self.sppf(x)
        x = torch.cat([F.interpolate(x9, scale_factor=2, mode="nearest"), x6], 1)
        x12 = self.c2f(x, 12, sp.neck_rep, False)
        x = torch.cat([F.interpolate(x12, scale_factor=2, mode="nearest"), x4], 1)
        p3 = self.c2f(x, 15, sp.neck_rep, False)
        x = torch.cat([self.conv(p3, "model.16"), x12], 1)
        p4 = self.c2f(x, 18, sp.neck_rep, False)
        x = torch.cat([self.conv(p4, "model.19"), x9], 1)
        p5 = self.c2f(x, 21, sp.neck_rep, False)
        return p3, p4, p5

    @torch.inference_mode()
    def head(self, feats):
        """-> pred f32 [B,4+nc+nm,A] (decoded, Ultralytics inference layout), proto f32 [B,nm,Hp,Wp]."""
        sp = self.spec
        B = feats[0].shape[0]
        p = self.conv(feats[0], "model.22.proto.cv1")
        p = self.conv(p, "model.22.proto.upsample")
        p = self.conv(p, "model.22.proto.cv2")
        proto = self.conv(p, "model.22.proto.cv3")
        box, cls, mc = [], [], []
        for lvl, f in enumerate(feats):
            for tower, dst in (("cv2", box), ("cv3", cls), ("cv4", mc)):
                t = self.conv(f, f"model.22.{tower}.{lvl}.0")
                t = self.conv(t, f"model.22.{tower}.{lvl}.1")
                t = self.conv(t, f"model.22.{tower}.{lvl}.2", out_fp32=True)
                dst.append(t.reshape(B, t.shape[1], -1))
        box, cls, mc = torch.cat(box, 2), torch.cat(cls, 2), torch.cat(mc, 2)
        # make_anchors(offset 0.5): cell centres in grid units + per-anchor stride
        pts, strides = [], []
        for _, hl, wl, st in sp.levels:
            sx = torch.arange(wl, dtype=torch.float32) + 0.5
            sy = torch.arange(hl, dtype=torch.float32) + 0.5
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            pts.append(torch.stack((xx, yy), -1).view(-1, 2))
            strides.append(torch.full((hl * wl, 1), float(st)))
        anchors = torch.cat(pts).transpose(0, 1)[None]       # [1,2,A]
        strides = torch.cat(strides).transpose(0, 1)[None]   # [1,1,A]
        # DFL: softmax over reg_max bins, expectation with arange weights
        A = box.shape[2]
        d = box.view(B, 4, sp.reg_max, A).transpose(2, 1).softmax(1)
        proj = torch.arange(sp.reg_max, dtype=torch.float32).view(1, sp.reg_max, 1, 1)
        dist = (d * proj).sum(1)                              # [B,4,A]  ltrb
        lt, rb = dist.chunk(2, 1)
        x1y1 = anchors - lt
        x2y2 = anchors + rb
        dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides
        pred = torch.cat((dbox, cls.sigmoid(), mc), 1)
        return pred, proto

    @torch.inference_mode()
    def forward_u8(self, frames_u8_nhwc, swap_rb=True, record=False):
        """frames: uint8 [B,H,W,3] already letterboxed to the model size.
        swap_rb=True reproduces Ultralytics' `im[..., ::-1]` on ndarray sources
        (SURVEY section 8 row A2 quirk)."""
        self.taps = {} if record else None
        x = torch.as_tensor(frames_u8_nhwc)
        if swap_rb:
            x = x.flip(-1)
        x = x.permute(0, 3, 1, 2).contiguous().float() / 255
        x = self.q(x)
        if self.mode == "fp64":
            x = x.double()
        return self.head(self.features(x))
