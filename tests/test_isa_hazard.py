"""gfx950 hazard lint (tools/repro_pk_mfma.hip): a packed fp32 VALU instruction whose LOW result takes the
HIGH half of src1 (VOP3P op_sel[1] = 1: v_pk_fma_f32 ... op_sel:[0,1,0], v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1])
returns wrong values in lanes 48-63 while ANOTHER wave on the SIMD executes a 16- or 8-bit-input MFMA.  The bf16x3
precision mode runs such MFMAs on one stream beside every other kernel of the path on the second stream, and any other
handle, torch GEMM or process sharing the GPU may run them beside an fp32 plan - so NO kernel of the library may contain that
encoding (round 3: conv_wino_kernel's hand-written column pass now crosses its operands through src0, op_sel:[1,0])."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from ccvpe_amd import build as B

VULNERABLE = re.compile(r"v_pk_(fma|mul|add)_f32\b.*\bop_sel:\[[01],1")
KERNEL = re.compile(r"^(_Z\w+):\s")
ALLOWED = ()   # no exemptions


def _asm(src):
    cmd = [B._hipcc(), *[f for f in B.FLAGS if f != "-fPIC"], *B.EXTRA_FLAGS.get(src, []), "--cuda-device-only", "-S", "-o", "-",
           os.path.join(B.CSRC, src)]
    return src, subprocess.run(cmd, capture_output=True, text=True, timeout=1200)


_CACHE = {}


def _all_asm():
    if "r" not in _CACHE:
        try:
            subprocess.run([B._hipcc(), "--version"], capture_output=True, timeout=60, check=True)
        except (OSError, subprocess.SubprocessError):
            pytest.skip("hipcc not available")
        with ThreadPoolExecutor(max_workers=4) as ex:
            _CACHE["r"] = list(ex.map(_asm, B.SOURCES))
    return _CACHE["r"]


def test_lint_pattern_matches_the_known_encodings():
    assert VULNERABLE.search("\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]")
    assert VULNERABLE.search("\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
    assert not VULNERABLE.search("\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]")
    assert not VULNERABLE.search("\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0] neg_lo:[0,1]")


def test_no_kernel_has_the_op_sel_hazard_encoding():
    results = _all_asm()
    offenders = []
    seen_wino = False
    for src, r in results:
        assert r.returncode == 0, f"{src}: {r.stderr[-2000:]}"
        kernel = "?"
        for line in r.stdout.splitlines():
            m = KERNEL.match(line)
            if m:
                kernel = m.group(1)
            if VULNERABLE.search(line):
                if any(a in kernel for a in ALLOWED):
                    seen_wino = True
                else:
                    offenders.append(f"{src}: {kernel}: {line.strip()}")
    assert not offenders, "packed fp32 op_sel[1]=1 encodings (wrong beside another wave's 16-/8-bit MFMAs):\n" + "\n".join(offenders[:20])
    assert not seen_wino


WIDE_STORE = re.compile(r"^\s*(buffer|global|flat|scratch)_store_dwordx[34]\s+v\[(\d+):(\d+)\]")
VDST = re.compile(r"^\s*v_\w+\s+v(?:\[(\d+):(\d+)\]|(\d+))")


def test_no_vector_write_lands_on_the_data_of_a_wide_store_in_the_next_slot():
    """Second gfx950 hazard met in round 2 (conv_wino4_kernel<8> epilogue): `buffer_store_dwordx4 v[88:91], ..., s68 offen`
    directly followed by `v_pk_add_f32 v[88:89], ...` stored the NEW v89 for lanes 12-15 of every 16 - the store had not read its
    data yet.  hipcc's hazard recogniser only pads this write-after-read when the store has no SGPR soffset.  No kernel may
    overwrite a register of a 12/16-byte store's data in the instruction slot right behind it (one independent instruction or
    an s_nop in between is what the recogniser itself considers enough for the soffset-less form)."""
    offenders = []
    for src, r in _all_asm():
        assert r.returncode == 0, f"{src}: {r.stderr[-2000:]}"
        kernel = "?"
        pending = None
        for line in r.stdout.splitlines():
            m = KERNEL.match(line)
            if m:
                kernel, pending = m.group(1), None
                continue
            t = line.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue            # comments, directives, labels: not an issue slot
            if pending is not None:
                lo, hi, text = pending
                w = VDST.match(line)
                if w and not t.startswith("v_cmp"):
                    a = int(w.group(1) if w.group(1) is not None else w.group(3))
                    b = int(w.group(2) if w.group(2) is not None else w.group(3))
                    if a <= hi and b >= lo:
                        offenders.append(f"{src}: {kernel}: {text}  ->  {t}")
                pending = None
            s = WIDE_STORE.match(line)
            if s:
                pending = (int(s.group(2)), int(s.group(3)), t)
    assert not offenders, "vector write onto the data registers of the preceding wide store:\n" + "\n".join(offenders[:30])


SCRATCH = re.compile(r"^\s*\.private_segment_fixed_size:\s+(\d+)")
NAME = re.compile(r"^\s*\.name:\s+(\S+)")


def test_no_kernel_uses_scratch():
    """Round 4: a kernel that touches scratch (register spills, a non-inlined call) starts ~5 us later than one that does not
    (tools/ubench_scratch.hip: 3.3 -> 7.9 us per launch of 256 workgroups, more with larger grids) - a fixed price on every launch of
    a batch-1 frame, and three of the Winograd kernels had paid it since round 2 for twelve hoisted index registers.  No kernel of
    the library may have a private segment."""
    offenders = []
    for src, r in _all_asm():
        assert r.returncode == 0, f"{src}: {r.stderr[-2000:]}"
        name = "?"
        for line in r.stdout.splitlines():
            m = NAME.match(line)
            if m:
                name = m.group(1)
            s = SCRATCH.match(line)
            if s and int(s.group(1)) > 0:
                offenders.append(f"{src}: {name}: {s.group(1)} bytes of scratch")
    assert not offenders, "\n".join(offenders[:20])
