"""Dev tool (GPU box): cost of writing channel slices into a wider NHWC buffer (the concat buffers of the decoder: 40 of 56 channels per
pixel = 160-byte pieces every 224 bytes) against dense writes of the same bytes."""
import torch

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

N = 32 * 256 * 256
for ld, c, off in ((40, 40, 0), (56, 40, 0), (56, 16, 40), (64, 40, 0), (48, 32, 0), (48, 16, 32)):
    buf = torch.zeros(N, ld, device="cuda")
    x = torch.randn(N, c, device="cuda")
    s = t(lambda: buf[:, off:off + c].copy_(x))
    mb = N * c * 4 / 1e6
    print(f"write {c:3d} of {ld:3d} channels at offset {off:2d}: {s * 1e6:7.1f} us for {mb:.0f} MB read + {mb:.0f} MB written = {2 * mb / s / 1e6:5.2f} TB/s")
