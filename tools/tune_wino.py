"""Dev tool (GPU box): Winograd F(2x2,3x3) kernel vs the implicit GEMM on the decoder 3x3 shapes of the VIGOR
workload: correctness against torch fp64 and effective TFLOP/s (direct-convolution FLOPs / time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from ccvpe_amd import _lib

_l = _lib.load()
NAMES = {t: _l.ccvpe_op_tile_name(t).decode() for t in range(1, _l.ccvpe_op_num_tiles() + 1)}
WINO = [t for t, n in NAMES.items() if "wino" in n]

SHAPES = [
    ("conv6.0", 32, 16, 16, 1344, 640),
    ("conv6.2", 32, 16, 16, 640, 640),
    ("conv5.0", 32, 32, 32, 432, 320),
    ("conv5.2", 32, 32, 32, 320, 320),
    ("conv4.0", 32, 64, 64, 200, 160),
    ("conv4.2", 32, 64, 64, 160, 160),
    ("conv3.0", 32, 128, 128, 104, 80),
    ("conv3.2", 32, 128, 128, 80, 80),
    ("conv2.0", 32, 256, 256, 56, 40),
    ("conv2.2", 32, 256, 256, 40, 40),
    ("conv5_ori.0", 32, 32, 32, 368, 256),
    ("conv4_ori.0", 32, 64, 64, 168, 128),
    ("conv3_ori.0", 32, 128, 128, 88, 64),
    ("conv2_ori.0", 32, 256, 256, 48, 32),
]


def main():
    only = sys.argv[1:]
    torch.manual_seed(0)
    for name, B, H, W, Cin, Cout in SHAPES:
        if only and name not in only:
            continue
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
        b = torch.randn(Cout, device="cuda")
        flops = 2.0 * B * H * W * Cin * Cout * 9
        ref = F.conv2d(x[:2].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
        line = f"{name:12s}"
        out, ms = _lib.op_conv2d(x, w, b, 1, 1, 0, 0, iters=10)
        err = (out[:2] - ref).abs().max().item() / ref.abs().max().item()
        line += f" igemm(auto): {ms:6.3f} ms {flops / ms / 1e9:6.1f} TF err {err:.1e} |"
        for t in WINO:
            bn = int(NAMES[t].split("x")[-1])
            if Cout / (-(-Cout // bn) * bn) < 0.6:
                continue
            out, ms = _lib.op_conv2d(x, w, b, 1, 1, 0, t, iters=10)
            err = (out[:2] - ref).abs().max().item() / ref.abs().max().item()
            line += f" {NAMES[t][10:]}: {ms:6.3f} ms {flops / ms / 1e9:6.1f} TF err {err:.1e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
