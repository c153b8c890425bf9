"""ctypes binding of libvti.so (include/vti.h).  This is the stub INTEGRATION.md shows a
maintainer of the reference adding next to measurement.py.  There is no CPU fallback: if the
HIP library is missing the import fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VTI_LIB_VARIANT=stamps selects the diagnostic build (tools/ only); the product always loads libvti.so
_VARIANT = os.environ.get("VTI_LIB_VARIANT", "")       # developer builds: "stamps" (in-kernel timing), experiment variants (csrc/Makefile: variant)
LIB_PATH = os.path.join(_HERE, f"libvti_{_VARIANT}.so" if _VARIANT else "libvti.so")

VTI_F16, VTI_F32, VTI_H2 = 0, 1, 2
VTI_MASK_LOGIT, VTI_MASK_SIGMOID = 0, 1
VTI_PACK_U8, VTI_PACK_BITS = 0, 1


class VtiDesc(C.Structure):
    _fields_ = [("scale", C.c_char), ("nc", C.c_int32), ("nm", C.c_int32), ("reg_max", C.c_int32),
                ("H", C.c_int32), ("W", C.c_int32), ("max_batch", C.c_int32), ("dtype", C.c_int32)]


class VtiConvInfo(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("c1", C.c_int32), ("c2", C.c_int32), ("k", C.c_int32),
                ("s", C.c_int32), ("kind", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32),
                ("h_out", C.c_int32), ("w_out", C.c_int32), ("macs", C.c_int64),
                ("tile_h", C.c_int32), ("tile_w", C.c_int32), ("waves_n", C.c_int32), ("nrep", C.c_int32),
                ("lds_bytes", C.c_int32), ("fused", C.c_int32), ("persistent", C.c_int32)]


class VtiError(RuntimeError):
    """Raised for any non-zero vti_status (the reference catches every predict exception,
    measurement.py:207-216)."""

    def __init__(self, code, msg):
        super().__init__(f"libvti error {code}: {msg}")
        self.code = code


_P, _I32, _I64, _F, _D, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); every symbol include/vti.h declares
SIGNATURES = {
    "vti_create": (_I32, [C.POINTER(VtiDesc), C.POINTER(_P)]),
    "vti_destroy": (None, [_P]),
    "vti_last_error": (C.c_char_p, [_P]),
    "vti_num_convs": (_I32, [_P]),
    "vti_conv_at": (_I32, [_P, _I32, C.POINTER(VtiConvInfo)]),
    "vti_num_anchors": (_I32, [_P]),
    "vti_fused_params": (_I64, [_P]),
    "vti_macs_per_frame": (_I64, [_P]),
    "vti_workspace_bytes": (_I64, [_P]),
    "vti_num_launches": (_I32, [_P]),
    "vti_load_weights": (_I32, [_P, _P, _SZ, _I32]),
    "vti_set_workspace": (_I32, [_P, _P, _SZ]),
    "vti_letterbox": (_I32, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "vti_forward": (_I32, [_P, _P, _I32, _I32, _P, _P, _P]),
    "vti_nms": (_I32, [_P, _P, _I32, _F, _D, _I32, _I32, _P, _P, _P]),
    "vti_forward_scored": (_I32, [_P, _P, _I32, _I32, _P, _P, _P, _P]),
    "vti_nms_scored": (_I32, [_P, _P, _P, _I32, _F, _D, _I32, _I32, _P, _P, _P]),
    "vti_masks": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _I32, _P, _P]),
    "vti_scale_boxes": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "vti_predict": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _F, _D, _I32, _I32, _I32, _I32,
                           _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P]),
    "vti_mask_to_frame": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "vti_union_envelope": (_I32, [_P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "vti_mask_stats": (_I32, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "vti_mask_stats_bits": (_I32, [_P, _P, _I32, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "vti_envelope_bits": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "vti_pixels_to_world": (_I32, [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "vti_kmeans1d2": (_I32, [_P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "vti_debug_conv_output": (_I32, [_P, _I32, _I32, _P, _P]),
    "vti_debug_conv2d": (_I32, [_I32, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _I32, _I32, _I32,
                                _P, _I32, _I32, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32,
                                _P, _P, _P]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                              f"g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(ctx, rc):
    if rc != 0:
        msg = lib().vti_last_error(ctx)
        raise VtiError(rc, msg.decode() if msg else "")
