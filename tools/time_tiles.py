"""Dev tool (GPU box): time given tile ids on given conv shapes.  usage: time_tiles.py "B,H,W,Cin,Cout" ... -- tile ids"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccvpe_amd import _lib
lib = _lib.load()
args = sys.argv[1:]
k = args.index("--")
shapes = [tuple(int(v) for v in a.split(",")) for a in args[:k]]
tiles = [int(v) for v in args[k + 1:]]
for B, H, W, Cin, Cout in shapes:
    x = torch.randn(B, H, W, Cin, device="cuda")
    mode = os.environ.get("TT_DATA", "randn")   # relu: half zeros like a post-ReLU activation; zeros: the DVFS best case
    if mode == "relu": x = x.clamp_min(0)
    if mode == "zeros": x = torch.zeros_like(x)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda")
    for t in tiles:
        try:
            _, ms = _lib.op_conv2d(x, w, b, 1, 1, 0, t, iters=20)
            print(f"{B}x{H}x{W} {Cin}->{Cout}  {lib.ccvpe_op_tile_name(t & 255).decode():22s} split {t >> 8}: {ms:.4f} ms  {2.0*B*H*W*Cin*Cout*9/ms/1e9:7.1f} TF/s alg", flush=True)
        except Exception as e:
            print(f"{B}x{H}x{W} {Cin}->{Cout} tile {t}: {e}")
