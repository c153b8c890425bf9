"""Reader/writer for the VTIW1 fused-weight container (own flat format).

TEST INFRASTRUCTURE.  Stands in for `YOLO(model_path)` unpickling
(measurement.py:145) -- the reference's .pt files are absent blobs
(.MISSING_LARGE_BLOBS) and cannot be unpickled without ultralytics anyway.

Layout (little endian):
  header  64 B : magic b"VTIW", u32 version=1, char scale[4], u32 nc, nm, reg_max, n_convs, pad
  per conv     : char name[48], u32 c1, c2, k, s, kind, pad[3]   (80 B)
                 f32 weight  -- kind 0/1: [c2][c1][k][k] (OIHW, BN already folded)
                                kind 2  : [c1][c2][k][k] (torch ConvTranspose2d IOHW)
                 f32 bias[c2]
"""
import struct
import numpy as np

MAGIC = b"VTIW"
HDR = struct.Struct("<4sI4sIIII36x")
REC = struct.Struct("<48sIIIII12x")
assert HDR.size == 64 and REC.size == 80


def write_blob(scale, nc, nm, reg_max, convs):
    """convs: list of (name, c1, c2, k, s, kind, weight ndarray f32, bias ndarray f32)."""
    out = [HDR.pack(MAGIC, 1, scale.encode().ljust(4, b"\0"), nc, nm, reg_max, len(convs))]
    for name, c1, c2, k, s, kind, w, b in convs:
        w = np.ascontiguousarray(w, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        assert w.size == c1 * c2 * k * k and b.size == c2, name
        out.append(REC.pack(name.encode().ljust(48, b"\0"), c1, c2, k, s, kind))
        out.append(w.tobytes())
        out.append(b.tobytes())
    return b"".join(out)


def read_blob(blob):
    """-> (meta dict, {name: (c1,c2,k,s,kind, weight f32 ndarray, bias f32 ndarray)}) in file order."""
    mv = memoryview(blob)
    magic, ver, scale, nc, nm, reg_max, n = HDR.unpack_from(mv, 0)
    if magic != MAGIC or ver != 1:
        raise ValueError("not a VTIW1 container")
    off = HDR.size
    convs = {}
    for _ in range(n):
        name, c1, c2, k, s, kind = REC.unpack_from(mv, off)
        off += REC.size
        name = name.rstrip(b"\0").decode()
        nw = c1 * c2 * k * k
        w = np.frombuffer(mv, dtype=np.float32, count=nw, offset=off)
        off += 4 * nw
        b = np.frombuffer(mv, dtype=np.float32, count=c2, offset=off)
        off += 4 * c2
        shape = (c1, c2, k, k) if kind == 2 else (c2, c1, k, k)
        convs[name] = (c1, c2, k, s, kind, w.reshape(shape), b)
    if off != len(mv):
        raise ValueError("trailing bytes in VTIW1 container")
    meta = dict(scale=scale.rstrip(b"\0").decode(), nc=nc, nm=nm, reg_max=reg_max)
    return meta, convs
