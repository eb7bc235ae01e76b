"""Dev tool (GPU box): time the implicit-GEMM conv kernel per tile on the decoder/encoder shapes of the
VIGOR workload and, for orientation, torch's own fp32 conv (MIOpen) on the same shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from ccvpe_amd import _lib

_l = _lib.load()
import re

TILES = {t: _l.ccvpe_op_tile_name(t).decode().replace("conv_igemm_", "") for t in range(1, _l.ccvpe_op_num_tiles() + 1)}


def tile_dims(name):
    """(rows, cols) of a tile from its name; Winograd tiles are named <2x2 tiles>x<channels> (4 output pixels per tile)."""
    m = re.search(r"(\d+)x(\d+)", name)
    bm, bn = int(m.group(1)), int(m.group(2))
    return (bm * 4 if "wino" in name else bm), bn

# (name, B, H, W, Cin, Cout, K)
SHAPES = [
    ("conv6.0", 32, 16, 16, 1344, 640, 3),
    ("conv6.2", 32, 16, 16, 640, 640, 3),
    ("conv5.0", 32, 32, 32, 432, 320, 3),
    ("conv4.0", 32, 64, 64, 200, 160, 3),
    ("conv4.2", 32, 64, 64, 160, 160, 3),
    ("conv3.0", 32, 128, 128, 104, 80, 3),
    ("conv2.0", 32, 256, 256, 56, 40, 3),
    ("conv2.2", 32, 256, 256, 40, 40, 3),
    ("conv1.0", 32, 512, 512, 16, 16, 3),
    ("conv5_ori.2", 32, 32, 32, 256, 256, 3),
    ("conv2_ori.0", 32, 256, 256, 48, 32, 3),
    ("b1.expand", 32, 256, 256, 16, 96, 1),
    ("b5.project", 32, 32, 32, 240, 80, 1),
    ("head", 32, 16, 16, 320, 1280, 1),
]


def main():
    only = sys.argv[1:]
    torch.manual_seed(0)
    for name, B, H, W, Cin, Cout, K in SHAPES:
        if only and name not in only:
            continue
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, K, K, device="cuda") / (Cin * K * K) ** 0.5
        b = torch.randn(Cout, device="cuda")
        flops = 2.0 * B * H * W * Cin * Cout * K * K
        ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=K // 2)
        line = f"{name:12s} M={B*H*W:8d} N={Cout:5d} K={Cin*K*K:6d} |"
        M_, N_ = B * H * W, Cout
        for tid, tn in TILES.items():
            bm, bn = tile_dims(tn)
            if "wino" in tn and (K != 3 or H % 16 or W % 16):
                continue
            util = (M_ * N_) / (-(-M_ // bm) * bm * -(-N_ // bn) * bn)
            if util < 0.45:
                continue
            try:
                out, ms = _lib.op_conv2d(x, w, b, 1, K // 2, 0, tid, iters=10)
            except Exception as e:  # noqa: BLE001
                line += f" {tn}: ERR"
                continue
            err = (out.permute(0, 3, 1, 2) - ref).abs().max().item() / ref.abs().max().item()
            line += f" {tn}:{flops / ms / 1e9:6.1f}TF" + ("" if err < 1e-4 else f"(err {err:.1e})")
        out, ms = _lib.op_conv2d(x, w, b, 1, K // 2, 0, 0, iters=10)
        line += f" | auto:{flops / ms / 1e9:6.1f}"
        # torch reference timing (channels_last)
        xc = x.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        wc = w.contiguous(memory_format=torch.channels_last)
        for _ in range(3):
            F.conv2d(xc, wc, b, padding=K // 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            F.conv2d(xc, wc, b, padding=K // 2)
        e1.record()
        torch.cuda.synchronize()
        line += f" | torch:{flops / (e0.elapsed_time(e1) / 10) / 1e9:6.1f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
