"""Shared helpers for the golden fixtures under tests/golden/ (data only: expected values on a fixed
lattice + whole-tensor statistics, captured from the reference by oracle/make_golden.py)."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (variant, ctor kwargs, fov, batch, seed)
CONFIGS = {
    "vigor_prior180_circ": dict(variant="vigor_ori_prior", circular=True, ori_noise=180.0, fov=360.0, batch=1, seed=0),
    "vigor_prior72_fov108": dict(variant="vigor_ori_prior", circular=False, ori_noise=72.0, fov=108.0, batch=1, seed=0),
    "vigor_circ": dict(variant="vigor", circular=True, ori_noise=None, fov=360.0, batch=1, seed=0),
    "kitti": dict(variant="kitti", circular=False, ori_noise=None, fov=360.0, batch=1, seed=0),
    "oxford": dict(variant="oxford", circular=False, ori_noise=None, fov=360.0, batch=1, seed=0),
    "vigor_prior180_b2": dict(variant="vigor_ori_prior", circular=True, ori_noise=180.0, fov=360.0, batch=2, seed=3),
}

OUTPUT_NAMES = ["logits", "heatmap", "ori", "ms1", "ms2", "ms3", "ms4", "ms5", "ms6"]
FULL_LIMIT = 32768
LATTICE_TARGET = 16384


def lattice(numel: int) -> np.ndarray:
    """Deterministic sample positions in a flattened tensor: everything for small tensors, otherwise
    ~16k positions on an odd stride (odd => not aligned with the power-of-two image widths)."""
    if numel <= FULL_LIMIT:
        return np.arange(numel, dtype=np.int64)
    stride = (numel + LATTICE_TARGET - 1) // LATTICE_TARGET
    stride |= 1
    return np.arange(0, numel, stride, dtype=np.int64)


def summarize(name: str, a: np.ndarray) -> Dict[str, np.ndarray]:
    a = np.ascontiguousarray(a, dtype=np.float32)
    flat = a.reshape(-1)
    a64 = flat.astype(np.float64)
    return {
        f"{name}/shape": np.array(a.shape, dtype=np.int64),
        f"{name}/values": flat[lattice(flat.size)].copy(),
        f"{name}/stats": np.array([a64.sum(), np.abs(a64).sum(), np.sqrt((a64 * a64).sum()), np.abs(a64).max()]),
        f"{name}/argmax": np.array([int(np.argmax(flat))], dtype=np.int64),
    }


def compare(name: str, fx, got: np.ndarray, rtol: float) -> float:
    """Assert `got` matches the fixture entry `name`; returns the scale-relative max error.
    Error is measured relative to the tensor's max |value| (fixture stats[3])."""
    got = np.ascontiguousarray(got, dtype=np.float32)
    shape = tuple(int(s) for s in fx[f"{name}/shape"])
    assert tuple(got.shape) == shape, f"{name}: shape {got.shape} != fixture {shape}"
    flat = got.reshape(-1)
    exp = fx[f"{name}/values"]
    idx = lattice(flat.size)
    assert idx.size == exp.size
    scale = float(fx[f"{name}/stats"][3])
    err = float(np.abs(flat[idx].astype(np.float64) - exp.astype(np.float64)).max()) / max(scale, 1e-30)
    assert np.isfinite(flat).all(), f"{name}: non-finite values"
    assert err <= rtol, f"{name}: lattice error {err:.3g} > {rtol:g} (scale {scale:.3g})"
    g64 = flat.astype(np.float64)
    st = fx[f"{name}/stats"]
    n = flat.size
    # whole-tensor statistics catch errors off the lattice; tolerances scale with sqrt(n) rounding growth
    assert abs(np.abs(g64).sum() - st[1]) <= rtol * max(st[1], 1e-30), f"{name}: abs-sum {np.abs(g64).sum():.6g} vs {st[1]:.6g}"
    assert abs(np.sqrt((g64 * g64).sum()) - st[2]) <= rtol * max(st[2], 1e-30), f"{name}: l2 mismatch"
    assert abs(g64.sum() - st[0]) <= rtol * max(st[1], 1e-30), f"{name}: sum mismatch"
    return err


def load(name: str):
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    return np.load(path, allow_pickle=False)
