"""Manual parity script (GPU box; uses the oracle): every MBConv block output of both encoders against the oracle's encoder."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccvpe_amd import models, weights
from oracle import ccvpe_oracle as orc

variant = sys.argv[1] if len(sys.argv) > 1 else "vigor_ori_prior"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cls = {"vigor_ori_prior": models.CVM_VIGOR_ori_prior, "kitti": models.CVM_KITTI, "oxford": models.CVM_OxfordRobotCar}[variant]
kw = dict(ori_noise=180.0, circular_padding=True) if variant == "vigor_ori_prior" else {}
sd = weights.generate_state_dict(variant, 0)
grd, sat = weights.generate_inputs(variant, B, 0, 360.0)
grd, sat = torch.from_numpy(grd), torch.from_numpy(sat)
m = cls("cuda", **kw)
m.load_state_dict(sd)
m.to("cuda").eval()
m.set_debug(True)
m(grd.cuda(), sat.cuda())
torch.cuda.synchronize()
for tag, x, circ in (("grd", grd, kw.get("circular_padding", False)), ("sat", sat, False)):
    with torch.no_grad():
        _, taps = orc.encoder(x, sd, tag + "_efficientnet", circ)
    for i, r in enumerate(taps):
        t = m.read_tap(f"{tag}_block{i}")
        d = (t - r).abs().max().item()
        print(f"{tag}_block{i:<2d} {tuple(t.shape)} rel {d / r.abs().max().item():.3g}")
