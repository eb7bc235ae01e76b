#!/usr/bin/env python3
"""Experiment: partition the CUs between an encoder stream and a decoder stream (hipExtStreamCreateWithCUMask).
How do (a) the aerial encoder alone and (b) the full forward scale with the number of CUs, and do they overlap without
slowing each other when they run on disjoint CU sets?  Single-stream issue order inside the library (CCVPE_STREAMS=1), so
every launch of a call lands on the caller's masked stream."""
import ctypes as C, os, sys, time
os.environ["CCVPE_STREAMS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccvpe_amd import models, weights

hip = C.CDLL("libamdhip64.so")

def masked_stream(cus):
    """stream restricted to the given CU indices (bit i of the mask = CU i; 256 CUs -> 8 words)"""
    words = (C.c_uint32 * 8)()
    for cu in cus:
        words[cu // 32] |= 1 << (cu % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

def interleaved(n, total=256):
    """n CUs spread over the XCDs: every (total/n)-th CU"""
    step = total / n
    return sorted({int(i * step) for i in range(n)})

def build():
    m = models.CVM_VIGOR_ori_prior("cuda", 180.0, True)
    m.load_state_dict(weights.generate_state_dict("vigor_ori_prior", 0))
    return m.to("cuda").eval()

dev = torch.device("cuda", 0)
g, s = weights.generate_inputs("vigor_ori_prior", 32, 0)
g, s = torch.from_numpy(g).to(dev), torch.from_numpy(s).to(dev)
ma, mb = build(), build()
for _ in range(3):
    ma(g, s); mb.encode_aerial(s)
torch.cuda.synchronize()

def timed(fn, stream, n=10):
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n

allc = list(range(256))
for n in (256, 224, 192, 160, 128):
    st = masked_stream(allc[:n])
    print(f"forward (single stream) on the first {n:3d} CUs: {timed(lambda: ma(g, s), st):7.3f} ms", flush=True)
for n in (256, 128, 64, 32):
    st = masked_stream(allc[256 - n:])
    print(f"aerial encoder on the last {n:3d} CUs: {timed(lambda: mb.encode_aerial(s), st):7.3f} ms", flush=True)
for nf in (192, 176, 160):
    sf, se = masked_stream(allc[:nf]), masked_stream(allc[nf:])
    for _ in range(2):
        with torch.cuda.stream(sf): ma(g, s)
        with torch.cuda.stream(se): mb.encode_aerial(s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 10
    for _ in range(n):
        with torch.cuda.stream(sf): ma(g, s)
        with torch.cuda.stream(se): mb.encode_aerial(s); mb.encode_aerial(s)
    torch.cuda.synchronize()
    print(f"concurrently: forward on {nf} CUs + 2 x aerial encoder on {256 - nf} CUs: {1e3 * (time.perf_counter() - t) / n:7.3f} ms per round", flush=True)
