"""Shared helpers for the -m gpu parity tests (all call through the C ABI via vti_amd)."""
import functools

import numpy as np
import torch
import torch.nn.functional as F


def need_gpu():
    import pytest
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this container (run with gpurun)")


@functools.lru_cache(maxsize=8)
def engine_and_oracle(scale, nc, H, W, B, dtype, seed=1, cls_bias=None, gain=1.7):
    import vti_amd
    from oracle.model import OracleModel
    eng = vti_amd.Engine(scale, nc, H=H, W=W, max_batch=B, dtype=dtype)
    blob = vti_amd.random_weights(eng, seed=seed, cls_bias=cls_bias, gain=gain)
    eng.load_weights(blob, 0)
    # the h2 engine (split-fp16 pairs, ~22 bits) is held against the plain fp32 oracle
    return eng, OracleModel(blob, H, W, mode="fp32" if dtype == "h2" else dtype), blob


def frames_u8(B, H, W, seed=0):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, (B, H, W, 3), dtype=np.uint8)


def ref_conv(x_nhwc, w, b, k, s, kind, dtype, res=None, act=None):
    """torch-CPU reference of one engine conv on NHWC float input; fp16 mode rounds operands and result."""
    q = (lambda t: t.half().float()) if dtype == "fp16" else (lambda t: t)      # fp32 and h2: no operand rounding
    x = q(torch.as_tensor(x_nhwc).float()).permute(0, 3, 1, 2)
    w = q(torch.as_tensor(w).float())
    b = torch.as_tensor(b).float()
    if kind == 2:
        y = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=s).float()
    else:
        y = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=k // 2).float()
    if kind == 0 if act is None else act:
        y = F.silu(y)
    if res is not None:
        y = y + q(torch.as_tensor(res).float()).permute(0, 3, 1, 2)
    return y.permute(0, 2, 3, 1).contiguous()


def mask_iou(a, b):
    a = np.asarray(a) > 0
    b = np.asarray(b) > 0
    u = np.logical_or(a, b).sum()
    return 1.0 if u == 0 else np.logical_and(a, b).sum() / u


def synth_pred(rng, B, nc, nm, A, H=640, W=640, n_inst=50, dup=5, bg=0.05):
    """SURVEY section 8d post-processing stress: `n_inst` planted, well separated boxes per frame (conf in
    [0.5,0.9]) x `dup` jittered duplicates each (IoU > 0.7) + background scores below conf."""
    pred = np.zeros((B, 4 + nc + nm, A), np.float32)
    pred[:, 4:4 + nc] = rng.uniform(0, bg, (B, nc, A)).astype(np.float32)
    pred[:, 0] = rng.uniform(0, W, (B, A))
    pred[:, 1] = rng.uniform(0, H, (B, A))
    pred[:, 2:4] = rng.uniform(8, 80, (B, 2, A))
    pred[:, 4 + nc:] = rng.standard_normal((B, nm, A)).astype(np.float32)
    g = int(np.ceil(np.sqrt(n_inst)))
    for b in range(B):
        slots = rng.choice(A, n_inst * dup, replace=False)
        for i in range(n_inst):
            cx = (i % g + 0.5) * W / g + rng.uniform(-3, 3)
            cy = (i // g + 0.5) * H / g + rng.uniform(-3, 3)
            w, h = rng.uniform(0.45, 0.8, 2) * np.array([W / g, H / g])
            c = int(rng.integers(nc))
            for d in range(dup):
                a = slots[i * dup + d]
                pred[b, :4, a] = [cx + rng.normal(0, 0.5), cy + rng.normal(0, 0.5), w + rng.normal(0, 0.5), h + rng.normal(0, 0.5)]
                pred[b, 4:4 + nc, a] = rng.uniform(0, bg, nc)
                pred[b, 4 + c, a] = rng.uniform(0.5, 0.9)
    return pred
