"""Dev tool (GPU box): run a few batch-32 forwards of the headline workload so rocprofv3 (--pmc or --kernel-trace) can attribute
counters / durations to every kernel of the path.  usage: pmc_forward.py [steps] [batch]
Single-stream issue order (CCVPE_STREAMS=1 is set here) so dispatches of the two chains do not overlap."""
import os
import sys

os.environ.setdefault("CCVPE_STREAMS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccvpe_amd import models, weights

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
m = models.CVM_VIGOR_ori_prior("cuda", 180.0, True)
m.load_state_dict(weights.generate_state_dict("vigor_ori_prior", 0))
m.to("cuda").eval()
g, s = weights.generate_inputs("vigor_ori_prior", batch, 0, 360.0)
g, s = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()
for _ in range(steps):
    o = m(g, s)
    m.postprocess(o[1], o[2])      # one postprocess_kernel per step: the step marker of the summaries
torch.cuda.synchronize()
print("done", steps, batch)
