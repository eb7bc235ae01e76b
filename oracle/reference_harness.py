"""Import the real reference (/root/reference) in the build container - TEST INFRASTRUCTURE ONLY.

Never shipped to or used on the GPU box (the reference does not exist there); every caller must
check `available()` first.  Recipe = SURVEY Appendix A:
  * models.py:1-13 imports torchvision / IPython / tensorboard, none of which the model classes
    use and none of which are installed -> empty stub modules are registered first;
  * every constructor calls EfficientNet.from_pretrained (models.py:55,99,353,397,...) which would
    fetch weights over the network (utils.py:747) -> rebound to the vendored no-fetch from_name.
    The original is never called.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
_mod = None


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models.py"))


def load():
    """Returns the reference `models` module (cached)."""
    global _mod
    if _mod is not None:
        return _mod
    if not available():
        raise RuntimeError("reference tree not present")
    import numpy as np
    import torch

    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from efficientnet_pytorch.model import EfficientNet  # vendored in the reference, torch only

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        for k, val in attrs.items():
            setattr(m, k, val)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    tv.models = stub("torchvision.models")
    tv.utils = stub("torchvision.utils", make_grid=None, save_image=None)
    ip = stub("IPython")
    ip.display = stub("IPython.display", Image=None)
    stub("torch.utils.tensorboard", SummaryWriter=None)

    EfficientNet.from_pretrained = classmethod(
        lambda cls, name, circular, **kw: cls.from_name(name, circular))

    # the repo root also has a drop-in module called `models`; make sure we get the reference's
    saved = sys.modules.pop("models", None)
    import importlib.util
    sp = importlib.util.spec_from_file_location("ccvpe_reference_models", os.path.join(REFERENCE_ROOT, "models.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    if saved is not None:
        sys.modules["models"] = saved
    # importing it reseeds the global RNGs (models.py:16-17); callers use explicit generators.
    _mod = mod
    return mod


def build(variant: str, state_dict, circular: bool = False, ori_noise=None):
    """Construct the reference nn.Module for `variant` on CPU in eval mode with `state_dict` loaded."""
    m = load()
    if variant == "vigor":
        net = m.CVM_VIGOR("cpu", circular)
    elif variant == "vigor_ori_prior":
        net = m.CVM_VIGOR_ori_prior("cpu", ori_noise, circular)
    elif variant == "kitti":
        net = m.CVM_KITTI("cpu")
    elif variant == "oxford":
        net = m.CVM_OxfordRobotCar("cpu")
    else:
        raise ValueError(variant)
    net.load_state_dict(state_dict, strict=True)
    return net.eval()
