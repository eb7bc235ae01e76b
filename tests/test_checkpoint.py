"""Checkpoint ingestion: files in the reference's format (torch.save(model.state_dict())) load into the
drop-in module through the safe loader; wrong-variant / damaged files fail loudly before anything is copied."""
import pytest
import torch

from ccvpe_amd import checkpoint, models, weights


_SD = {}


def _sd(variant, seed):
    if (variant, seed) not in _SD:
        _SD[(variant, seed)] = weights.generate_state_dict(variant, seed)
    return {k: v.clone() for k, v in _SD[(variant, seed)].items()}


def _module(variant):
    return {"vigor": lambda: models.CVM_VIGOR("cpu", True), "kitti": lambda: models.CVM_KITTI("cpu"),
            "oxford": lambda: models.CVM_OxfordRobotCar("cpu"),
            "vigor_ori_prior": lambda: models.CVM_VIGOR_ori_prior("cpu", 180.0, True)}[variant]()


@pytest.mark.parametrize("variant", ["vigor_ori_prior", "kitti"])
@pytest.mark.parametrize("form", ["plain", "dataparallel", "wrapped", "fp64"])
def test_reference_format_round_trip(tmp_path, variant, form):
    sd = _sd(variant, 3)
    obj = dict(sd)
    if form == "dataparallel":
        obj = {"module." + k: v for k, v in sd.items()}
    elif form == "wrapped":
        obj = {"state_dict": dict(sd), "epoch": 7}
    elif form == "fp64":
        obj = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    path = tmp_path / "model.pt"
    torch.save(obj, path)
    m = _module(variant)
    info = checkpoint.load_reference_checkpoint(m, str(path))
    assert info["keys"] == 818 and info["variant"] == variant and len(info["sha256"]) == 64
    got = m.state_dict()
    assert list(got) == list(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    # and back out in the reference's format
    out = tmp_path / "out.pt"
    checkpoint.save_checkpoint(m, str(out))
    again = torch.load(out, map_location="cpu", weights_only=True)
    assert list(again) == list(sd) and all(torch.equal(again[k], sd[k]) for k in sd)


def test_wrong_variant_and_damaged_files_fail_before_loading(tmp_path):
    sd = _sd("kitti", 3)
    path = tmp_path / "kitti.pt"
    torch.save(sd, path)
    m = _module("vigor_ori_prior")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    with pytest.raises(checkpoint.CheckpointError, match="shape mismatches"):
        checkpoint.load_reference_checkpoint(m, str(path))
    assert all(torch.equal(before[k], v) for k, v in m.state_dict().items()), "a rejected file must not touch the module"
    sd2 = _sd("vigor_ori_prior", 3)
    sd2.pop("deconv3.weight")
    sd2["extra.key"] = torch.zeros(1)
    torch.save(sd2, path)
    with pytest.raises(checkpoint.CheckpointError, match="1 missing: deconv3.weight; 1 unexpected: extra.key"):
        checkpoint.load_reference_checkpoint(m, str(path))
    torch.save([1, 2, 3], path)
    with pytest.raises(checkpoint.CheckpointError, match="not a state dict"):
        checkpoint.load_reference_checkpoint(m, str(path))
