"""Manual parity script (GPU box; lives under tests/ because it uses the oracle): run the HIP forward with debug taps and compare every tap with the oracle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccvpe_amd import models, weights
from oracle import ccvpe_oracle as orc

CASES = {
    "vigor_ori_prior": (models.CVM_VIGOR_ori_prior, dict(ori_noise=180.0, circular_padding=True), 360.0),
    "vigor_ori_prior_fov108": (models.CVM_VIGOR_ori_prior, dict(ori_noise=72.0, circular_padding=False), 108.0),
    "vigor": (models.CVM_VIGOR, dict(circular_padding=True), 360.0),
    "kitti": (models.CVM_KITTI, {}, 360.0),
    "oxford": (models.CVM_OxfordRobotCar, {}, 360.0),
}


def run(case, B=1, debug=True):
    cls, kw, fov = CASES[case]
    variant = cls._variant
    sd = weights.generate_state_dict(variant, 0)
    grd, sat = weights.generate_inputs(variant, B, 0, fov)
    grd, sat = torch.from_numpy(grd), torch.from_numpy(sat)
    taps = {}
    t0 = time.time()
    ref = orc.forward(variant, sd, grd, sat, kw.get("circular_padding", False), kw.get("ori_noise"), taps)
    print(f"[{case}] oracle {time.time()-t0:.2f}s", flush=True)
    m = cls("cuda", **kw)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    m.set_debug(debug)
    t0 = time.time()
    out = m(grd.cuda(), sat.cuda())
    torch.cuda.synchronize()
    print(f"[{case}] hip first call {time.time()-t0:.2f}s", flush=True)
    names = ["logits", "heatmap", "ori", "ms1", "ms2", "ms3", "ms4", "ms5", "ms6"]
    worst = 0.0
    if debug:
        order = ["grd_block0", "grd_block1", "grd_block15", "grd_volume", "grd_desc1", "grd_desc6", "sat_block0", "sat_block2", "sat_block4", "sat_block10",
                 "sat_block15", "sat_volume", "sat_descriptor_map", "loc_level6", "loc_level5", "loc_level4", "loc_level3", "loc_level2",
                 "ori_level6", "ori_level5", "ori_level4", "ori_level3", "ori_level2", "ori_level1_nchw"]
        for k in order:
            try:
                t = m.read_tap(k)
            except Exception as e:
                print(f"  tap {k}: {e}")
                continue
            key = {"ori_level1_nchw": "ori_level1"}.get(k, k)
            if key.startswith("grd_block"):
                continue
            r = taps[key]
            if r.dim() == 2:
                r = r[:, :, None, None]
            d = (t - r).abs().max().item()
            s = r.abs().max().item()
            print(f"  tap {k:20s} {tuple(t.shape)} maxdiff {d:.3g} rel {d/s:.3g}")
    for n, a, b in zip(names, ref, out):
        b = b.cpu()
        d = (a - b).abs().max().item()
        s = a.abs().max().item()
        rel = d / s
        if n != "ori":
            worst = max(worst, rel)
        print(f"  out {n:8s} {tuple(b.shape)} maxdiff {d:.3g} rel {rel:.3g} nan={bool(torch.isnan(b).any())}")
    return worst


if __name__ == "__main__":
    cases = sys.argv[1:] or list(CASES)
    bad = 0
    for c in cases:
        w = run(c)
        print(f"[{c}] worst rel (non-ori) {w:.3g}")
        bad += w > 1e-3
    sys.exit(1 if bad else 0)
