"""VTIW1 fused-weight container: writer, reader, and the seeded random generator used by the
benchmarks (there is no network for checkpoints and the reference's .pt files are absent
blobs, .MISSING_LARGE_BLOBS:1-2).  Stands in for what `YOLO(model_path)` loads
(reference: measurement.py:145, config.py:67).

Format (little endian) -- see DESIGN.md:
  header 64 B: b"VTIW", u32 version=1, char scale[4], u32 nc, nm, reg_max, n_convs, pad
  per conv   : char name[48], u32 c1,c2,k,s,kind, pad[12]; f32 weight; f32 bias[c2]
               kind 0/1 weight is OIHW (BN folded), kind 2 (ConvTranspose2d) is IOHW.
"""
import math
import struct

import numpy as np

MAGIC = b"VTIW"
_HDR = struct.Struct("<4sI4sIIII36x")
_REC = struct.Struct("<48sIIIII12x")


def pack_container(scale, nc, nm, reg_max, table, tensors):
    """table: Engine.conv_table(); tensors: {name: (weight, bias)} float32 arrays."""
    parts = [_HDR.pack(MAGIC, 1, scale.encode().ljust(4, b"\0"), nc, nm, reg_max, len(table))]
    for t in table:
        w, b = tensors[t["name"]]
        w = np.ascontiguousarray(w, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        if w.size != t["c1"] * t["c2"] * t["k"] ** 2 or b.size != t["c2"]:
            raise ValueError(f"bad tensor size for {t['name']}")
        parts.append(_REC.pack(t["name"].encode().ljust(48, b"\0"), t["c1"], t["c2"], t["k"], t["s"], t["kind"]))
        parts.append(w.tobytes())
        parts.append(b.tobytes())
    return b"".join(parts)


def unpack_container(blob):
    mv = memoryview(blob)
    if len(mv) < _HDR.size:
        raise ValueError("not a VTIW1 container (too short)")
    magic, ver, scale, nc, nm, reg_max, n = _HDR.unpack_from(mv, 0)
    if magic != MAGIC or ver != 1:
        raise ValueError("not a VTIW1 container")
    off = _HDR.size
    tensors, table = {}, []
    for _ in range(n):
        name, c1, c2, k, s, kind = _REC.unpack_from(mv, off)
        off += _REC.size
        name = name.rstrip(b"\0").decode()
        nw = c1 * c2 * k * k
        w = np.frombuffer(mv, np.float32, nw, off).reshape((c1, c2, k, k) if kind == 2 else (c2, c1, k, k))
        off += 4 * nw
        b = np.frombuffer(mv, np.float32, c2, off)
        off += 4 * c2
        tensors[name] = (w, b)
        table.append(dict(name=name, c1=c1, c2=c2, k=k, s=s, kind=kind))
    return dict(scale=scale.rstrip(b"\0").decode(), nc=nc, nm=nm, reg_max=reg_max), table, tensors


def random_weights(engine, seed=1, cls_bias=None, gain=1.7):
    """Seeded He-scaled fused weights for `engine`'s conv table so activations stay O(1) through
    all layers in fp16.  Head biases follow Ultralytics' bias_init convention (box 1.0,
    cls log(5/nc/(640/stride)^2)) unless `cls_bias` overrides the class prior -- random nets
    with the stock prior emit no detections, so tests/benches raise it to get instances."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tensors = {}
    for t in engine.conv_table():
        c1, c2, k, kind, name = t["c1"], t["c2"], t["k"], t["kind"], t["name"]
        fan_in = c1 * k * k if kind != 2 else c1
        g = gain if kind == 0 else 1.0
        shape = (c1, c2, k, k) if kind == 2 else (c2, c1, k, k)
        w = rng.standard_normal(shape, dtype=np.float32) * np.float32(g / math.sqrt(fan_in))
        b = rng.standard_normal(c2, dtype=np.float32) * np.float32(0.1)
        if name.startswith("model.22.cv2.") and name.endswith(".2"):
            b[:] = 1.0
        if name.startswith("model.22.cv3.") and name.endswith(".2"):
            lvl = int(name.split(".")[3])
            stride = (8, 16, 32)[lvl]
            b[:] = math.log(5 / engine.nc / (640 / stride) ** 2) if cls_bias is None else cls_bias
        tensors[name] = (w, b)
    return pack_container(engine.scale, engine.nc, engine.nm, engine.reg_max, engine.conv_table(), tensors)
