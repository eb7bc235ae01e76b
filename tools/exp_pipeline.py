#!/usr/bin/env python3
"""Experiment: does overlapping the HBM/latency-bound encoder phase of one half-batch with the MFMA-bound decoder phase of
the other pay?  Two handles at batch 16 on two torch streams, issued alternately and continuously (the second lags the
first by roughly half a step), against one handle at batch 32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccvpe_amd import models, weights

def build():
    m = models.CVM_VIGOR_ori_prior("cuda", 180.0, True)
    m.load_state_dict(weights.generate_state_dict("vigor_ori_prior", 0))
    return m.to("cuda").eval()

dev = torch.device("cuda", 0)
g, s = weights.generate_inputs("vigor_ori_prior", 32, 0)
g, s = torch.from_numpy(g).to(dev), torch.from_numpy(s).to(dev)
steps = 40

def run_single(B):
    m = build()
    gg, ss = g[:B].contiguous(), s[:B].contiguous()
    for _ in range(5): m(gg, ss)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps): m(gg, ss)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"single handle  batch {B:2d}: {steps*B/dt:8.1f} queries/s  {1e3*dt/steps:7.3f} ms/step", flush=True)
    del m; torch.cuda.empty_cache()

def run_pair(B, lag_ms):
    ma, mb = build(), build()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ga, sa_in = g[:B].contiguous(), s[:B].contiguous()
    gb, sb_in = (g[B:2*B].contiguous(), s[B:2*B].contiguous()) if 2 * B <= g.shape[0] else (ga.clone(), sa_in.clone())
    for _ in range(5):
        with torch.cuda.stream(sa): ma(ga, sa_in)
        with torch.cuda.stream(sb): mb(gb, sb_in)
    torch.cuda.synchronize()
    filler = torch.randn(8192, 8192, device=dev)
    t = time.perf_counter()
    if lag_ms > 0:
        with torch.cuda.stream(sb):
            for _ in range(int(lag_ms / 0.45) + 1): filler @ filler   # ~0.45 ms each on the fp32 path: delays stream b
    for _ in range(steps):
        with torch.cuda.stream(sa): ma(ga, sa_in)
        with torch.cuda.stream(sb): mb(gb, sb_in)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"two handles  2 x batch {B:2d}, initial lag ~{lag_ms} ms: {steps*2*B/dt:8.1f} queries/s  {1e3*dt/steps:7.3f} ms per pair of steps", flush=True)
    del ma, mb; torch.cuda.empty_cache()

run_single(32)
run_single(16)
run_pair(16, 0)
run_pair(16, 4)
run_pair(8, 2)
run_pair(32, 0)
run_pair(32, 7)
