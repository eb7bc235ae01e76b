"""Dev tool (GPU box): 40 batch-1 frames (hipGraph replay) for `rocprofv3 --kernel-trace -- python3 tools/b1_frames.py`; summarise with
tools/b1_trace_summary.py <trace dir>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccvpe_amd import models, weights
m = models.CVM_VIGOR_ori_prior("cuda", 180.0, True); m.load_state_dict(weights.generate_state_dict("vigor_ori_prior", 0)); m.to("cuda").eval()
g, s = weights.generate_inputs("vigor_ori_prior", 1, 0, 360.0)
g, s = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()
for _ in range(40):
    o = m(g, s); m.postprocess(o[1], o[2])
torch.cuda.synchronize()
