"""Checkpoint ingestion: a reference `model.pt` -> the module's state dict -> packed device weights.

The reference saves `torch.save(CVM_model.state_dict(), path)` every epoch (train_VIGOR.py:229-231, train_KITTI.py:262-264,
train_OxfordRobotCar.py:177-179) and reloads it with `CVM_model.load_state_dict(torch.load(test_model_path))`
(train_VIGOR.py:252, train_KITTI.py:285, train_OxfordRobotCar.py:200).  `load_reference_checkpoint` does the same for a
`ccvpe_amd.models` module, with two differences a deployment wants:

* the file is opened with `weights_only=True` (nothing in it is executed), on the CPU, whatever device it was saved from;
* keys and shapes are checked against the variant's 818-entry layout (`spec.state_dict_spec`) before anything is copied, so a
  checkpoint of the wrong variant fails with the list of offending keys instead of a partial load.  `module.` prefixes
  (nn.DataParallel) and a `{"state_dict": ...}` wrapper are accepted.

Folding (BatchNorm into the convolutions), packing into the kernels' layouts and the upload happen at the next forward
(`ccvpe_finalize_weights`).  With a packed-weight cache directory (`weight_cache=` of the model constructor or the
CCVPE_WEIGHT_CACHE environment variable) the packed device weights are written once, keyed by the content hash of the state
dict, and later processes load them with `ccvpe_load_packed` instead of re-packing (ccvpe_amd/models.py `_cache_path`).
"""
from __future__ import annotations

import hashlib
from typing import Dict

import torch

from . import spec


class CheckpointError(ValueError):
    pass


def _unwrap(obj) -> Dict[str, torch.Tensor]:
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict) and not torch.is_tensor(obj["state_dict"]):
        obj = obj["state_dict"]
    if not isinstance(obj, dict) or not obj or not all(isinstance(k, str) and torch.is_tensor(v) for k, v in obj.items()):
        raise CheckpointError("not a state dict: expected a mapping of parameter names to tensors")
    if all(k.startswith("module.") for k in obj):
        obj = {k[len("module."):]: v for k, v in obj.items()}
    return dict(obj)


def validate_state_dict(variant: str, sd: Dict[str, torch.Tensor]) -> None:
    """Raise CheckpointError naming every missing / unexpected key and shape mismatch of `sd` for `variant`."""
    want = {k: shape for k, shape, _ in spec.state_dict_spec(spec.VARIANTS[variant])}
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want]
    shapes = [f"{k}: {tuple(sd[k].shape)} != {tuple(want[k])}" for k in want if k in sd and tuple(sd[k].shape) != tuple(want[k])]
    if missing or unexpected or shapes:
        def head(xs):
            return ", ".join(xs[:6]) + (f", ... (+{len(xs) - 6})" if len(xs) > 6 else "")
        parts = []
        if missing:
            parts.append(f"{len(missing)} missing: {head(missing)}")
        if unexpected:
            parts.append(f"{len(unexpected)} unexpected: {head(unexpected)}")
        if shapes:
            parts.append(f"{len(shapes)} shape mismatches: {head(shapes)}")
        raise CheckpointError(f"checkpoint does not match variant '{variant}' ({'; '.join(parts)})")


def load_reference_checkpoint(model, path: str) -> Dict[str, object]:
    """Load a reference checkpoint file into `model` (a ccvpe_amd.models.CVM_* module).  Returns a small summary
    (key count, parameter count, sha256 of the tensor bytes in key order) for logging / cache keys."""
    obj = torch.load(path, map_location="cpu", weights_only=True)
    sd = _unwrap(obj)
    variant = model._variant
    validate_state_dict(variant, sd)
    clean = {}
    digest = hashlib.sha256()
    n_params = 0
    for k, _, _ in spec.state_dict_spec(spec.VARIANTS[variant]):
        t = sd[k].detach()
        t = t.to(torch.int64) if k.endswith("num_batches_tracked") else t.to(torch.float32)
        t = t.contiguous()
        clean[k] = t
        digest.update(k.encode())
        digest.update(t.numpy().tobytes())
        n_params += t.numel()
    model.load_state_dict(clean, strict=True)
    return {"variant": variant, "keys": len(clean), "elements": n_params, "sha256": digest.hexdigest()}


def save_checkpoint(model, path: str) -> None:
    """Write the module's weights in the reference's format (`torch.save(model.state_dict(), path)`, CPU tensors)."""
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)
