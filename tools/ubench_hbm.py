"""Dev tool (GPU box): achievable HBM rates of plain streaming kernels (torch fill / copy / sum) on buffers far beyond the 256 MB Infinity
Cache - the write-only, read + write and read-only ceilings the memory-bound launches are judged against (profiles/r03_hbm_kernels.md)."""
import torch

def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for gb in (1, 4):
    n = gb * (1 << 30) // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    s = t(lambda: x.fill_(1.0)); print(f"{gb} GiB fill  (write only)  {gb * 1.0737 / s / 1e3:6.2f} TB/s written")
    s = t(lambda: y.copy_(x));   print(f"{gb} GiB copy  (read + write) {2 * gb * 1.0737 / s / 1e3:6.2f} TB/s moved ({gb * 1.0737 / s / 1e3:.2f} written)")
    s = t(lambda: x.sum());      print(f"{gb} GiB sum   (read only)   {gb * 1.0737 / s / 1e3:6.2f} TB/s read")
    s = t(lambda: torch.add(x, y, out=y)); print(f"{gb} GiB add   (2 reads + write) {3 * gb * 1.0737 / s / 1e3:6.2f} TB/s moved")
    del x, y
