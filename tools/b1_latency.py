"""Dev tool (GPU box): batch-1 latency exactly as bench.py reports it (`batch1` / `config5`): p50 / p99 of a synchronised step
(forward + device post-processing) and the back-to-back rate, for the VIGOR headline model and the Oxford streaming model.
usage: python tools/b1_latency.py [n]      (environment switches such as CCVPE_SE_TICKET=0 apply)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccvpe_amd import models, weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for variant, cls, kw, fov in (("vigor_ori_prior", models.CVM_VIGOR_ori_prior, dict(ori_noise=180.0, circular_padding=True), 360.0),
                              ("oxford", models.CVM_OxfordRobotCar, {}, 360.0)):
    m = cls("cuda", **kw)
    m.load_state_dict(weights.generate_state_dict(variant, 0))
    m.to("cuda").eval()
    g, s = weights.generate_inputs(variant, 1, 0, fov)
    g, s = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()

    def step():
        o = m(g, s)
        return m.postprocess_rows(o[1], o[2])

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    lat = []
    for _ in range(n):
        t = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t))
    lat.sort()
    t = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    thr = n / (time.perf_counter() - t)
    # forward alone, GPU time by events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev = []
    for _ in range(100):
        e0.record()
        m(g, s)
        e1.record()
        torch.cuda.synchronize()
        ev.append(e0.elapsed_time(e1))
    ev.sort()
    print(f"{variant:16s} p50 {lat[n // 2]:.3f} ms  p99 {lat[int(n * 0.99)]:.3f} ms  back-to-back {thr:.0f} /s ({1e3 / thr:.3f} ms)  forward by events p50 {ev[50]:.3f} ms", flush=True)
    del m
