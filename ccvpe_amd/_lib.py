"""ctypes binding of libccvpe_hip.so (include/ccvpe.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libccvpe_hip.so")

OK = 0
VARIANT_ID = {"vigor": 0, "vigor_ori_prior": 1, "kitti": 2, "oxford": 3}


class Config(C.Structure):
    _fields_ = [("variant", C.c_int32), ("circular_padding", C.c_int32), ("ori_noise", C.c_float),
                ("device", C.c_int32), ("micro_batch", C.c_int32), ("reserved", C.c_int32 * 3)]


class Outputs(C.Structure):
    _fields_ = [("logits_flattened", C.c_void_p), ("heatmap", C.c_void_p), ("ori", C.c_void_p),
                ("matching_score", C.c_void_p * 6)]


class Pose(C.Structure):
    _fields_ = [("index", C.c_int32), ("prob", C.c_float), ("cos_v", C.c_float), ("sin_v", C.c_float),
                ("angle_deg", C.c_float)]


# every symbol include/ccvpe.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("ccvpe_last_error", C.c_char_p, []),
    ("ccvpe_version", C.c_char_p, []),
    ("ccvpe_create", C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    ("ccvpe_destroy", C.c_int, [C.c_void_p]),
    ("ccvpe_set_weight", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    ("ccvpe_skip_weight", C.c_int, [C.c_void_p, C.c_char_p]),
    ("ccvpe_finalize_weights", C.c_int, [C.c_void_p]),
    ("ccvpe_output_channels", C.c_int, [C.c_void_p, C.c_int32]),
    ("ccvpe_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    ("ccvpe_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                C.POINTER(Outputs), C.c_void_p]),
    ("ccvpe_postprocess", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    ("ccvpe_set_debug", C.c_int, [C.c_void_p, C.c_int32]),
    ("ccvpe_read_tap", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                 C.POINTER(C.c_int32 * 4)]),
    ("ccvpe_profile_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                        C.POINTER(Outputs), C.c_void_p]),
    ("ccvpe_profile_row", C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_size_t, C.POINTER(C.c_float),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]),
]

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library and bind every entry point; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m ccvpe_amd.build` (hipcc, gfx950). "
            "ccvpe_amd has no CPU or PyTorch fallback for the forward pass.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class CcvpeError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc < 0:
        msg = load().ccvpe_last_error()
        raise CcvpeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
