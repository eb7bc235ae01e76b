"""Direct oracle check of the batch-32 plans - the plans the headline times (Winograd F(4x4) 128-channel form, tail split, odd
split-K): BASELINE.json configs 2, 3 and 4 run at batch 32 on the HIP path, and samples 0, 13 and 31 of the batch are compared
full-tensor with the CPU oracle's forward of those three inputs (reference models.py:448-652 / 752-950 restated in
oracle/ccvpe_oracle.py, pinned to the reference by tests/test_oracle_vs_reference.py and the committed goldens).

Tolerance: 5e-4 of each tensor's max |value| (BASELINE.json north_star: 1e-3 relative fp32); the orientation output is checked
twice - the un-normalised 2-channel map (debug tap of a second, debug-mode run of the same batch) and the unit field weighted by
the oracle's un-normalised magnitude."""
import pytest
import torch

from ccvpe_amd import weights
from tests import golden_util as gu
from tests.test_parity_gpu import build_model, ori_weighted_error

pytestmark = pytest.mark.gpu

FULL_RTOL = 5e-4
SAMPLES = [0, 13, 31]


@pytest.mark.parametrize("name", ["vigor_prior180_circ", "vigor_prior72_fov108", "kitti"])
def test_batch32_samples_match_the_oracle(name):
    from oracle import ccvpe_oracle as orc   # checker only
    cfg = gu.CONFIGS[name]
    seed = 33
    sd = weights.generate_state_dict(cfg["variant"], seed)
    g, s = weights.generate_inputs(cfg["variant"], 32, seed, cfg["fov"])
    gt, st = torch.from_numpy(g), torch.from_numpy(s)
    taps = {}
    ref = orc.forward(cfg["variant"], sd, gt[SAMPLES], st[SAMPLES], cfg["circular"], cfg["ori_noise"], taps=taps)
    m = build_model(dict(cfg, seed=seed))
    outs = m(gt.cuda(), st.cuda())                       # the two-stream batch-32 plan bench.py times
    assert outs[0].shape[0] == 32
    idx = torch.tensor(SAMPLES)
    mag = taps["ori_level1"].pow(2).sum(dim=1, keepdim=True).sqrt()
    for i, (a, b) in enumerate(zip(ref, outs)):
        b = b.cpu()[idx]
        assert a.shape == b.shape, gu.OUTPUT_NAMES[i]
        if i == 2:
            err = ori_weighted_error(a, b, mag)
        else:
            err = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30)
        assert err <= FULL_RTOL, f"{gu.OUTPUT_NAMES[i]}: {err:.3g}"
    # the un-normalised orientation map of the same batch (debug plan: every tensor keeps its memory, one stream)
    md = build_model(dict(cfg, seed=seed))
    md.set_debug(True)
    outs_d = md(gt.cuda(), st.cuda())
    raw = md.read_tap("ori_level1_nchw")[idx]
    raw_ref = taps["ori_level1"]
    err = (raw_ref - raw).abs().max().item() / raw_ref.abs().max().item()
    assert err <= FULL_RTOL, f"ori (un-normalised): {err:.3g}"
    # debug and product plans share their launches' tuning entries: same bits
    for i, (a, b) in enumerate(zip(outs, outs_d)):
        assert torch.equal(a, b), gu.OUTPUT_NAMES[i]
