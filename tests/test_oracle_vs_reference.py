"""Full-tensor oracle vs the real reference, in the build container only (skipped where
/root/reference does not exist, e.g. on the GPU box)."""
import pytest
import torch

from ccvpe_amd import weights
from oracle import ccvpe_oracle as orc
from oracle import reference_harness as rh

pytestmark = pytest.mark.skipif(not rh.available(), reason="reference tree not present")


@pytest.mark.parametrize("variant,circ,noise,fov", [
    ("vigor_ori_prior", True, 36.0, 360.0),      # all four model classes of the reference (models.py:49, 346, 655, 954)
    ("vigor", True, None, 360.0),
    ("kitti", False, None, 360.0),
    ("oxford", False, None, 360.0),
])
def test_full_tensor_agreement(variant, circ, noise, fov):
    torch.set_num_threads(8)
    sd = weights.generate_state_dict(variant, 5)
    grd, sat = weights.generate_inputs(variant, 1, 5, fov)
    grd, sat = torch.from_numpy(grd), torch.from_numpy(sat)
    net = rh.build(variant, sd, circ, noise)
    raw = {}
    hook = net.conv1_ori.register_forward_hook(lambda _m, _i, out: raw.__setitem__("ref", out.detach()))   # models.py:648-649, before F.normalize
    with torch.no_grad():
        ref = net(grd, sat)
    hook.remove()
    taps = {}
    got = orc.forward(variant, sd, grd, sat, circ, noise, taps)
    assert len(ref) == len(got) == 9
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a.shape == b.shape
        if i == 2:
            continue
        assert (a - b).abs().max().item() <= 2e-5 * a.abs().max().item()
    # orientation: the un-normalised map in full, and the unit field weighted by the un-normalised magnitude
    # (F.normalize is ill-conditioned where the raw vector is ~0)
    assert (raw["ref"] - taps["ori_level1"]).abs().max().item() <= 2e-5 * raw["ref"].abs().max().item()
    mag = raw["ref"].double().norm(dim=1, keepdim=True)
    err = ((ref[2].double() - got[2].double()).abs() * mag).max().item()
    assert err <= 2e-5 * mag.max().item()


def test_reference_state_dict_keys_match_spec():
    from ccvpe_amd import spec
    sd = weights.generate_state_dict("kitti", 0)
    net = rh.build("kitti", sd)
    assert list(net.state_dict().keys()) == [k for k, _, _ in spec.state_dict_spec(spec.VARIANTS["kitti"])]
