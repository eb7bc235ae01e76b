import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ccvpe_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
for shape in [(3,16,16,64,88),(1,16,16,16,64),(1,16,16,16,128),(2,32,32,32,48)]:
    B,H,W,Cin,Cout = shape
    x = torch.randn(B,H,W,Cin,device="cuda"); w = torch.randn(Cout,Cin,3,3,device="cuda")/(Cin*9)**0.5; b = torch.randn(Cout,device="cuda")
    ref = torch.nn.functional.conv2d(x.permute(0,3,1,2).double(), w.double(), b.double(), padding=1).permute(0,2,3,1).float()
    for t in (42,43):
        out,_ = _lib.op_conv2d(x,w,b,1,1,0,t)
        e = (out-ref).abs()
        print(shape, lib.ccvpe_op_tile_name(t).decode(), "max err", e.max().item()/ref.abs().max().item())
        per_ch = e.amax(dim=(0,1,2)); bad = (per_ch > 1e-3).nonzero().flatten().tolist()
        print("   bad channels:", bad[:40])
        per_px = e.amax(dim=(0,3)); print("   bad pixels:", (per_px>1e-3).sum().item(), "of", H*W)
B,H,W,Cin,Cout = 1,16,16,16,64
x = torch.randn(B,H,W,Cin,device="cuda"); w = torch.randn(Cout,Cin,3,3,device="cuda")/(Cin*9)**0.5; b = torch.zeros(Cout,device="cuda")
ref = torch.nn.functional.conv2d(x.permute(0,3,1,2).double(), w.double(), b.double(), padding=1).permute(0,2,3,1).float()
out,_ = _lib.op_conv2d(x,w,b,1,1,0,43)
e = (out-ref).abs()
bad = (e > 1e-3).nonzero().tolist()
import collections
print("bad (y,x):", sorted(set((r[1], r[2]) for r in bad)))
print("bad ch at first bad px:", [r[3] for r in bad if (r[1], r[2]) == (bad[0][1], bad[0][2])])
y0, x0, c0 = bad[0][1], bad[0][2], bad[0][3]
print("out", out[0,y0,x0,c0-1:c0+3].tolist(), "ref", ref[0,y0,x0,c0-1:c0+3].tolist())
# is the wrong value some other pixel's value?
val = out[0,y0,x0,c0].item()
close = ((ref[0,:,:,c0]-val).abs() < 1e-4).nonzero().tolist()
print("value equals ref at pixels (same channel):", close)
close2 = ((ref[0,y0,x0,:]-val).abs() < 1e-4).nonzero().tolist()
print("value equals ref at channels (same pixel):", close2)
