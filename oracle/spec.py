"""Independent restatement of the YOLOv8-seg module table (Ultralytics yolov8-seg.yaml).

Reference call sites: measurement.py:145 (YOLO(model_path)), measurement.py:208-210.
The table itself is third-party [U]; SURVEY.md section 8 U2-U5 + Appendix A is the
offline spec.  Pinned by the published fused/unfused parameter counts
(tests/test_oracle_spec.py).
"""
import math
from dataclasses import dataclass

SCALES = {  # depth, width, max_channels
    "n": (0.33, 0.25, 1024),
    "s": (0.33, 0.50, 1024),
    "m": (0.67, 0.75, 768),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.25, 512),
}

KIND_CONV_ACT = 0   # Conv2d(bias=False)+BN+SiLU, BN folded -> weight+bias, SiLU
KIND_CONV_LIN = 1   # plain Conv2d with bias, no activation (head `.2` rows)
KIND_DECONV = 2     # ConvTranspose2d(k=2,s=2,bias=True), no BN, no activation


@dataclass
class ConvRow:
    name: str
    c1: int
    c2: int
    k: int
    s: int
    kind: int
    h_in: int = 0
    w_in: int = 0
    h_out: int = 0
    w_out: int = 0

    @property
    def fused_params(self):
        return self.c1 * self.c2 * self.k * self.k + self.c2

    @property
    def unfused_params(self):
        n = self.c1 * self.c2 * self.k * self.k
        return n + (2 * self.c2 if self.kind == KIND_CONV_ACT else self.c2)

    @property
    def macs(self):
        if self.kind == KIND_DECONV:
            return self.h_in * self.w_in * self.c1 * self.c2 * self.k * self.k
        return self.h_out * self.w_out * self.c1 * self.c2 * self.k * self.k


def make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


class Spec:
    """Layer table for yolov8{scale}-seg with nc classes at input H x W."""

    def __init__(self, scale="n", nc=80, nm=32, reg_max=16, H=640, W=640):
        depth, width, maxc = SCALES[scale]
        self.scale, self.nc, self.nm, self.reg_max, self.H, self.W = scale, nc, nm, reg_max, H, W
        ch = lambda c: make_divisible(min(c, maxc) * width, 8)
        rep = lambda n: max(round(n * depth), 1)
        self.c = [ch(64), ch(128), ch(256), ch(512), ch(1024)]
        self.reps = [rep(3), rep(6), rep(6), rep(3)]
        self.neck_rep = rep(3)
        self.npr = ch(256)
        c = self.c
        self.c2_box = max(16, c[2] // 4, reg_max * 4)
        self.c3_cls = max(c[2], min(nc, 100))
        self.c4_mc = max(c[2] // 4, nm)
        rows = []

        def add(name, c1, c2, k, s, kind, hin, win):
            if kind == KIND_DECONV:
                ho, wo = hin * 2, win * 2
            else:
                ho, wo = (hin + 2 * (k // 2) - k) // s + 1, (win + 2 * (k // 2) - k) // s + 1
            rows.append(ConvRow(name, c1, c2, k, s, kind, hin, win, ho, wo))
            return ho, wo

        def c2f(idx, c1, c2, n, h, w):
            cc = c2 // 2
            add(f"model.{idx}.cv1", c1, 2 * cc, 1, 1, KIND_CONV_ACT, h, w)
            for j in range(n):
                add(f"model.{idx}.m.{j}.cv1", cc, cc, 3, 1, KIND_CONV_ACT, h, w)
                add(f"model.{idx}.m.{j}.cv2", cc, cc, 3, 1, KIND_CONV_ACT, h, w)
            add(f"model.{idx}.cv2", (2 + n) * cc, c2, 1, 1, KIND_CONV_ACT, h, w)

        h, w = H, W
        h, w = add("model.0", 3, c[0], 3, 2, KIND_CONV_ACT, h, w)
        h, w = add("model.1", c[0], c[1], 3, 2, KIND_CONV_ACT, h, w)
        c2f(2, c[1], c[1], self.reps[0], h, w)
        h, w = add("model.3", c[1], c[2], 3, 2, KIND_CONV_ACT, h, w)
        c2f(4, c[2], c[2], self.reps[1], h, w)
        h3, w3 = h, w
        h, w = add("model.5", c[2], c[3], 3, 2, KIND_CONV_ACT, h, w)
        c2f(6, c[3], c[3], self.reps[2], h, w)
        h4, w4 = h, w
        h, w = add("model.7", c[3], c[4], 3, 2, KIND_CONV_ACT, h, w)
        c2f(8, c[4], c[4], self.reps[3], h, w)
        h5, w5 = h, w
        add("model.9.cv1", c[4], c[4] // 2, 1, 1, KIND_CONV_ACT, h, w)
        add("model.9.cv2", c[4] * 2, c[4], 1, 1, KIND_CONV_ACT, h, w)
        c2f(12, c[4] + c[3], c[3], self.neck_rep, h4, w4)
        c2f(15, c[3] + c[2], c[2], self.neck_rep, h3, w3)
        add("model.16", c[2], c[2], 3, 2, KIND_CONV_ACT, h3, w3)
        c2f(18, c[2] + c[3], c[3], self.neck_rep, h4, w4)
        add("model.19", c[3], c[3], 3, 2, KIND_CONV_ACT, h4, w4)
        c2f(21, c[3] + c[4], c[4], self.neck_rep, h5, w5)
        self.levels = [(c[2], h3, w3, 8), (c[3], h4, w4, 16), (c[4], h5, w5, 32)]
        for lvl, (cl, hl, wl, _) in enumerate(self.levels):
            for tower, cmid, cout in (("cv2", self.c2_box, 4 * reg_max), ("cv3", self.c3_cls, nc),
                                      ("cv4", self.c4_mc, nm)):
                add(f"model.22.{tower}.{lvl}.0", cl, cmid, 3, 1, KIND_CONV_ACT, hl, wl)
                add(f"model.22.{tower}.{lvl}.1", cmid, cmid, 3, 1, KIND_CONV_ACT, hl, wl)
                add(f"model.22.{tower}.{lvl}.2", cmid, cout, 1, 1, KIND_CONV_LIN, hl, wl)
        add("model.22.proto.cv1", c[2], self.npr, 3, 1, KIND_CONV_ACT, h3, w3)
        add("model.22.proto.upsample", self.npr, self.npr, 2, 2, KIND_DECONV, h3, w3)
        add("model.22.proto.cv2", self.npr, self.npr, 3, 1, KIND_CONV_ACT, 2 * h3, 2 * w3)
        add("model.22.proto.cv3", self.npr, nm, 1, 1, KIND_CONV_ACT, 2 * h3, 2 * w3)
        self.rows = rows
        self.num_anchors = sum(hl * wl for _, hl, wl, _ in self.levels)
        self.proto_hw = (2 * h3, 2 * w3)
        self.no = 4 + nc + nm

    # Ultralytics model.info() counts the DFL conv's frozen arange(reg_max) weight too.
    @property
    def fused_params(self):
        return sum(r.fused_params for r in self.rows) + self.reg_max

    @property
    def unfused_params(self):
        return sum(r.unfused_params for r in self.rows) + self.reg_max

    @property
    def macs(self):
        return sum(r.macs for r in self.rows)


if __name__ == "__main__":
    for sc, nc, H, W in (("n", 80, 640, 640), ("s", 80, 640, 640), ("m", 80, 640, 640),
                         ("n", 2, 640, 640), ("n", 2, 736, 960), ("m", 80, 1280, 1280)):
        s = Spec(sc, nc, H=H, W=W)
        print(sc, nc, H, W, "convs", len(s.rows), "fused", s.fused_params, "unfused", s.unfused_params,
              "GMAC %.3f" % (s.macs / 1e9), "anchors", s.num_anchors, "proto", s.proto_hw)
