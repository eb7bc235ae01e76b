"""Dev tool (GPU box): the batch-1 weight-streaming layers through ccvpe_op_conv2d, tile by tile and split by split (mean of 200 back-to-back launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccvpe_amd import _lib

lib = _lib.load()
names = {lib.ccvpe_op_tile_name(t).decode(): t for t in range(1, lib.ccvpe_op_num_tiles() + 1)}
SHAPES = [("descmap", (1, 16, 16, 1280, 1280, 2, 2)), ("loc6.deconv as 1x1", (1, 8, 8, 1312, 4096, 1, 1)), ("loc5.deconv as 1x1", (1, 16, 16, 672, 1280, 1, 1))]
for label, (B, H, W, Cin, Cout, K, stride) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, K, K, device="cuda", generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    for name in ("conv_projl_1", "conv_projl_2", "conv_projl_4", "conv_projl_r2", "conv_projl_r4", "conv_igemm_64x32_m16", "conv_igemm_64x32_m16_s1"):
        row = []
        for S in (1, 2, 4, 8, 16):
            code = names[name] | (((64 + S) if S > 1 else 1) << 8)
            _, ms = _lib.op_conv2d(x, w, b, stride, 0, 0, code, iters=200)
            row.append(f"S{S}: {1e3 * ms:6.1f}")
        print(f"{label:22s} {name:26s} " + "  ".join(row) + "  us", flush=True)
