"""Dev tool (GPU box): launch ONE conv shape/tile a few times so rocprofv3 --pmc can attribute counters.
usage: pmc_conv.py B H W Cin Cout K tile [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccvpe_amd import _lib

B, H, W, Cin, Cout, K, tile = [int(v) for v in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 5
x = torch.randn(B, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, K, K, device="cuda") / (Cin * K * K) ** 0.5
b = torch.randn(Cout, device="cuda")
out, ms = _lib.op_conv2d(x, w, b, 1, K // 2, 0, tile, iters=iters)
torch.cuda.synchronize()
print(f"{_lib.load().ccvpe_op_tile_name(tile).decode()} {2.0*B*H*W*Cin*Cout*K*K/ms/1e9:.1f} TF {ms:.4f} ms")
