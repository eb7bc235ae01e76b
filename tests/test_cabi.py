"""The C-ABI library: builds for gfx950, loads, exports every symbol include/ccvpe.h declares, and
fails loudly without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest
import torch

from ccvpe_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "ccvpe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ccvpe_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_header_symbols(built_library):
    lib = C.CDLL(built_library)
    names = header_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ccvpe.h but not exported"
    assert sorted(n for n, _, _ in _lib.SYMBOLS) == names, "ccvpe_amd._lib binds exactly the header's functions"


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.Config) == 32
    assert C.sizeof(_lib.Outputs) == 9 * C.sizeof(C.c_void_p)
    assert C.sizeof(_lib.Pose) == 20


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_create_fails_loudly_without_gpu(built_library):
    lib = _lib.load()
    cfg = _lib.Config(variant=1, circular_padding=1, ori_noise=180.0, device=0, micro_batch=0)
    h = C.c_void_p()
    rc = lib.ccvpe_create(C.byref(cfg), C.byref(h))
    assert rc < 0
    assert b"no CPU fallback" in lib.ccvpe_last_error() or b"HIP" in lib.ccvpe_last_error()


def test_null_arguments_are_rejected(built_library):
    lib = _lib.load()
    assert lib.ccvpe_create(None, None) == -1
    assert lib.ccvpe_forward(None, None, 0, 0, None, 0, None, None) == -1
    assert lib.ccvpe_destroy(None) == 0
    assert lib.ccvpe_version().startswith(b"ccvpe-hip")


def test_micro_batch_bound_is_host_arithmetic():
    """Every kernel addresses tensors with 32-bit byte offsets; the library must refuse (not wrap) a micro-batch whose
    largest tensor reaches 2 GiB.  ccvpe_max_micro_batch is pure host code: the 6x-expanded 256x256x96 tensor of the
    aerial encoder (25.2 MB per sample) caps every variant at 85 samples per pass."""
    from ccvpe_amd import _lib
    lib = _lib.load()
    for variant, noise, gh, gw in [(0, 0.0, 320, 640), (1, 180.0, 320, 640), (1, 72.0, 320, 192), (2, 0.0, 256, 1024), (3, 0.0, 154, 231)]:
        cap = lib.ccvpe_max_micro_batch(variant, noise, gh, gw)
        assert cap == (2 ** 31 - 1) // (256 * 256 * 96 * 4), (variant, cap)
    assert lib.ccvpe_max_micro_batch(2, 0.0, 250, 1024) < 0          # 7 feature rows: not a KITTI-shaped ground image
    assert b"feature volume" in lib.ccvpe_last_error()
    assert lib.ccvpe_max_micro_batch(7, 0.0, 320, 640) < 0
