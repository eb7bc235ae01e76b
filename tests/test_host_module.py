"""Host-side mirror of the reference `models` module: constructor signatures, state_dict contract,
loud failure off-GPU / outside eval mode."""
import pytest
import torch

import models as dropin
from ccvpe_amd import models, spec, weights


def test_dropin_module_exports_reference_names():
    for n in ("CVM_VIGOR", "CVM_VIGOR_ori_prior", "CVM_KITTI", "CVM_OxfordRobotCar"):
        assert getattr(dropin, n) is getattr(models, n)


@pytest.mark.parametrize("variant,ctor", [
    ("vigor", lambda: models.CVM_VIGOR("cpu", True)),
    ("vigor_ori_prior", lambda: models.CVM_VIGOR_ori_prior("cpu", 72.0, False)),
    ("kitti", lambda: models.CVM_KITTI("cpu")),
    ("oxford", lambda: models.CVM_OxfordRobotCar("cpu")),
])
def test_state_dict_contract(variant, ctor):
    m = ctor()
    want = spec.state_dict_spec(spec.VARIANTS[variant])
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in want]
    for k, shape, dt in want:
        assert tuple(sd[k].shape) == tuple(shape)
        assert sd[k].dtype == (torch.int64 if dt == "i64" else torch.float32)
    gen = weights.generate_state_dict(variant, 0)
    res = m.load_state_dict(gen, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    bad = dict(gen)
    bad.pop("conv3.0.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)


def test_forward_refuses_cpu_and_train_mode():
    m = models.CVM_OxfordRobotCar("cpu")
    g, s = torch.zeros(1, 3, 154, 231), torch.zeros(1, 3, 512, 512)
    with pytest.raises(RuntimeError, match="eval"):
        m(g, s)
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(g, s)
