"""Deterministic synthetic parameters for the 818-key CCVPE state_dict.

There is no network for the ImageNet-pretrained EfficientNet weights or the authors' CCVPE
checkpoints, so parity fixtures, tests and bench.py all run on parameters from this generator
(SURVEY 8c).  Every tensor is drawn from `numpy.random.default_rng` seeded by
(seed, crc32(key)) so a single key can be regenerated in isolation and the values do not depend on
iteration order.  Scales are fan-in based with per-layer-type gains chosen so activations stay
O(1) through ~100 layers and the final logits span several units (the default torch init gives a
numerically uniform heatmap, which would make parity tests vacuous).
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

from . import spec

try:
    from ._bn_calib import BN_CALIB
except ImportError:  # before tests/calibrate_bn.py has been run
    BN_CALIB = {}


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def _gain_for(key: str) -> float:
    # swish(x) for x~N(0,1) has rms ~0.6; relu ~0.7.  Gains compensate so that the next layer
    # again sees ~unit-scale input.
    if key.endswith("_conv_stem.weight"):
        return 1.0
    if "_expand_conv" in key or "_conv_head" in key:
        return 1.0
    if "_depthwise_conv" in key:
        return 1.0
    if "_project_conv" in key:
        return 1.0
    if "_se_reduce" in key:
        return 1.0
    if "_se_expand" in key:
        return 1.5
    if key.startswith("grd_feature_to_descriptor"):
        return 1.3
    if key.startswith("sat_feature_to_descriptors"):
        return 1.3
    if key.startswith("deconv"):
        # the localisation deconvs (and deconv6_ori) eat an L2-normalised tensor (unit norm per pixel,
        # models.py:514,632): scale so their output is O(1) next to the encoder skip, otherwise the
        # matching path would be numerically invisible in the outputs.
        if "_ori" not in key or key.startswith("deconv6_ori"):
            return -0.7     # negative = flat std, not fan-in scaled
        return 1.0
    if key.startswith("conv1") and ".2." in key:
        return 4.0          # final 16->1 / 16->2 conv: spread the logits
    if key.startswith("conv") and ".0." in key:
        return 1.3
    if key.startswith("conv") and ".2." in key:
        return 1.45         # follows a ReLU
    return 1.0


def _fan_in(key: str, shape) -> int:
    if key.startswith("deconv"):
        # ConvTranspose2d weight [Cin, Cout, 2, 2]: each output pixel sees Cin inputs
        return shape[0]
    if len(shape) == 4:
        return shape[1] * shape[2] * shape[3]
    if len(shape) == 2:
        return shape[1]
    return 1


def generate_state_dict_numpy(variant: str, seed: int = 0) -> Dict[str, np.ndarray]:
    v = spec.VARIANTS[variant]
    out: Dict[str, np.ndarray] = {}
    for key, shape, dt in spec.state_dict_spec(v):
        r = _rng(seed, key)
        if dt == "i64":
            out[key] = np.array(0, dtype=np.int64)
            continue
        leaf = key.rsplit(".", 1)[1]
        is_bn = "._bn" in key
        if is_bn:
            cal = BN_CALIB.get(key.rsplit(".", 1)[0], 1.0)   # measured input variance of this BN layer
            if leaf == "weight":
                a = r.uniform(0.8, 1.2, size=shape)
            elif leaf == "bias":
                a = r.normal(0.0, 0.1, size=shape)
            elif leaf == "running_mean":
                a = r.normal(0.0, 0.1, size=shape) * np.sqrt(cal)
            else:  # running_var
                a = r.uniform(0.6, 1.4, size=shape) * cal
        elif leaf == "bias":
            a = r.normal(0.0, 0.05, size=shape)
        else:
            gain = _gain_for(key)
            std = -gain if gain < 0 else gain / np.sqrt(_fan_in(key, shape))
            a = r.normal(0.0, std, size=shape)
        out[key] = a.astype(np.float32)
    return out


def generate_state_dict(variant: str, seed: int = 0):
    """Same values as torch tensors (CPU)."""
    import torch
    return {k: (torch.from_numpy(np.ascontiguousarray(a)) if a.ndim else torch.tensor(int(a), dtype=torch.int64))
            for k, a in generate_state_dict_numpy(variant, seed).items()}


def generate_inputs(variant: str, batch: int, seed: int = 0, fov: float = 360.0):
    """Synthetic (grd, sat) float32 NCHW standard-normal inputs (ImageNet-normalised images are
    roughly zero-mean unit-scale).  `fov` < 360 applies the driver's width crop
    grd[..., :int(W*fov/360)] (train_VIGOR.py:272-273)."""
    v = spec.VARIANTS[variant]
    r = np.random.default_rng([seed, 12345])
    gh, gw = v.grd_hw
    grd = r.standard_normal((batch, 3, gh, gw), dtype=np.float32)
    sat = r.standard_normal((batch, 3) + spec.SAT_HW, dtype=np.float32)
    if fov < 360.0:
        grd = np.ascontiguousarray(grd[..., : int(gw * fov / 360)])
    return grd, sat
