"""The CPU oracle against the golden vectors captured from the real reference (oracle/make_golden.py).

This pins the oracle on every box (the reference itself exists only in the build container)."""
import numpy as np
import pytest
import torch

from ccvpe_amd import weights
from oracle import ccvpe_oracle as orc
from tests import golden_util as gu

RTOL = 2e-5   # oracle and reference are both torch-CPU fp32; they agree to ~1e-6 of each tensor's scale


@pytest.mark.parametrize("name", list(gu.CONFIGS))
def test_oracle_matches_reference_golden(name):
    cfg = gu.CONFIGS[name]
    fx = gu.load(name)
    sd = weights.generate_state_dict(cfg["variant"], cfg["seed"])
    grd, sat = weights.generate_inputs(cfg["variant"], cfg["batch"], cfg["seed"], cfg["fov"])
    # generator drift guard: fixtures are only meaningful for the same synthetic weights / inputs
    assert abs(np.abs(grd.astype(np.float64)).sum() - fx["meta/grd_abs_sum"][0]) < 1e-6 * fx["meta/grd_abs_sum"][0]
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values()) - fx["meta/weight_abs_sum"][0]) < 1e-9 * fx["meta/weight_abs_sum"][0]
    torch.set_num_threads(8)
    taps = {}
    outs = orc.forward(cfg["variant"], sd, torch.from_numpy(grd), torch.from_numpy(sat), cfg["circular"], cfg["ori_noise"], taps)
    for n, t in zip(gu.OUTPUT_NAMES, outs):
        if n == "ori":
            continue
        gu.compare(n, fx, t.numpy(), RTOL)
    for tap in ["sat_block0", "sat_block15", "grd_desc1", "grd_desc6", "loc_level6", "loc_level2", "ori_level6", "ori_level1"]:
        gu.compare("tap_" + tap, fx, taps[tap].numpy(), RTOL)
    # orientation: error weighted by the un-normalised magnitude (F.normalize is ill-conditioned near 0)
    ori = outs[2].numpy().reshape(-1)
    idx = gu.lattice(ori.size)
    mag = fx["ori/magnitude"].astype(np.float64)
    err = np.abs(ori[idx].astype(np.float64) - fx["ori/values"].astype(np.float64)) * mag
    assert err.max() <= RTOL * mag.max()
    # post-processing
    idx_, prob, cs, sn, _ = orc.postprocess(outs[1], outs[2])
    assert np.array_equal(idx_.numpy(), fx["post/index"])
    assert np.allclose(prob.numpy(), fx["post/prob"], rtol=1e-4)
