"""Dev tool (GPU box): per-launch device times of the headline workload's plan (ccvpe_profile_forward: one hipEvent pair around every
launch, single issue order), median over a few runs, filtered by a regular expression on the launch name.
usage: time_ops.py [regex] [runs] [batch]      e.g.  CCVPE_LIB_PATH=variants/libccvpe_x.so python tools/time_ops.py 'b1[2-5]\\.' 7"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ccvpe_amd import models, weights

rx = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 7
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
m = models.CVM_VIGOR_ori_prior("cuda", 180.0, True)
m.load_state_dict(weights.generate_state_dict("vigor_ori_prior", 0))
m.to("cuda").eval()
g, s = weights.generate_inputs("vigor_ori_prior", batch, 0, 360.0)
g, s = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()
m(g, s)
times = {}
order = []
for _ in range(runs):
    for name, ms, fl, by, iss in m.profile(g, s):
        if name not in times:
            times[name] = []
            order.append(name)
        times[name].append(ms)
total = 0.0
sel = 0.0
for name in order:
    med = float(np.median(times[name]))
    total += med
    if rx.search(name):
        sel += med
        print(f"{name:32s} {med * 1e3:9.1f} us")
print(f"selected {sel:.3f} ms of {total:.3f} ms (sum of medians, {runs} runs, batch {batch})")
