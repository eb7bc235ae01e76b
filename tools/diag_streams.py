#!/usr/bin/env python3
"""Diagnostic for the two-stream schedule: run ONE handle in program order and on two streams, dump a checksum of
every plan tensor after each run (ccvpe_debug_dump_plan) and name the first launch whose tensors differ.

    python tools/diag_streams.py [--batch 32] [--out gpurun_out/diag]

Experiments (each its own handle; environment switches are read by ccvpe_create):
    bf16x3        precision bf16x3 (two streams by default since round 2)
    bf16x3_nosplit  same + CCVPE_NO_SPLIT_PLANES=1 (register-staged kernel only, no LDS-DMA)
    fp32_igemm    exact fp32, CCVPE_WINOGRAD=0 (small-grid implicit GEMM decoder under two streams)
"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ccvpe_amd import _lib, models, weights  # noqa: E402

EXPERIMENTS = {
    "bf16x3": ("bf16x3", {}),
    "bf16x3_nosplit": ("bf16x3", {"CCVPE_NO_SPLIT_PLANES": "1"}),
    "fp32_igemm": ("fp32", {"CCVPE_WINOGRAD": "0"}),
    # second round of narrowing (register-staged bf16x3 kernel only)
    "snap_match2": ("bf16x3", {"CCVPE_NO_SPLIT_PLANES": "1", "CCVPE_DIAG_SNAP": "match2"}),
    "sync_match2": ("bf16x3", {"CCVPE_NO_SPLIT_PLANES": "1", "CCVPE_DIAG_SYNC_BEFORE": "match2"}),
    "sync_ori6": ("bf16x3", {"CCVPE_NO_SPLIT_PLANES": "1", "CCVPE_DIAG_SYNC_BEFORE": "ori6.deconv"}),
}
ENV_KEYS = ["CCVPE_STREAMS", "CCVPE_NO_SPLIT_PLANES", "CCVPE_WINOGRAD", "CCVPE_TUNE_SPLITK", "CCVPE_DIAG_SNAP", "CCVPE_DIAG_SYNC_BEFORE"]


def pattern(a, b, name):
    """Where two [B,R,H,W] score tensors differ: per sample, per roll, per 8-pixel block."""
    d = (a != b)
    if not d.any():
        print(f"     {name}: identical")
        return
    B, R, H, W = a.shape
    err = (a - b).abs()
    print(f"     {name}: {int(d.sum())} of {d.numel()} elements differ; max abs {err.max().item():.3g}; "
          f"samples hit {d.flatten(1).any(1).nonzero().flatten().tolist()}")
    print(f"        rolls hit {d.permute(1, 0, 2, 3).flatten(1).any(1).nonzero().flatten().tolist()}")
    blk = d.any(1).flatten(1).reshape(B, -1, 8).any(2)      # [B, HW/8] blocks of 8 pixels (one workgroup of match_kernel at C=640)
    print(f"        8-pixel blocks hit per sample: {blk.sum(1).tolist()} of {blk.shape[1]}")
    whole = (d.any(1).flatten(1).reshape(B, -1, 8).all(2) == blk).all().item()
    print(f"        a hit block is always hit in all 8 pixels: {whole}")


def parse(path):
    ops = []
    for line in open(path):
        if not line.startswith("op "):
            continue
        f = line.split()
        tile = next((x[5:] for x in f if x.startswith("tile=")), "")
        ops.append({"i": int(f[1]), "name": f[2], "stream": f[3], "wait": f[4], "tile": tile,
                    "tensors": {int(a): b for a, b in re.findall(r"t(\d+)\[[^\]]*\]=([0-9a-f]+)", line)}})
    return ops


def run(name, batch, outdir, sequence):
    prec, env = EXPERIMENTS[name]
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ["CCVPE_NO_REUSE"] = "1"   # every tensor keeps its memory, so the end-of-run checksums mean something
    m = models.CVM_VIGOR_ori_prior("cuda", 180.0, True, precision=prec)
    m.load_state_dict(weights.generate_state_dict("vigor_ori_prior", 0))
    m.to("cuda").eval()
    g, s = weights.generate_inputs("vigor_ori_prior", batch, 0)
    g, s = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()
    lib = _lib.load()
    dumps, outs = [], []
    for i, ns in enumerate(sequence):
        m.set_streams(ns)
        o = None
        for _ in range(2):
            o = m(g, s)
        torch.cuda.synchronize()
        path = os.path.join(outdir, f"{name}_{i}_s{ns}.txt")
        _lib.check(lib.ccvpe_debug_dump_plan(m._handle, path.encode()), "ccvpe_debug_dump_plan")
        dumps.append(parse(path))
        outs.append([t.clone() for t in o])
    ref = dumps[0]
    print(f"== {name}: precision {prec}, env {env}, batch {batch}, sequence {sequence}", flush=True)
    for i in range(1, len(sequence)):
        bad_ops = []
        for a, b in zip(ref, dumps[i]):
            diff = [t for t in a["tensors"] if a["tensors"][t] != b["tensors"].get(t)]
            if diff:
                bad_ops.append((a, diff))
        out_bad = [k for k, (x, y) in enumerate(zip(outs[0], outs[i])) if not torch.equal(x, y)]
        worst = max(((x - y).abs().max().item() / max(x.abs().max().item(), 1e-30)) for x, y in zip(outs[0], outs[i]))
        print(f"  run {i} (streams={sequence[i]}) vs run 0 (streams={sequence[0]}): {len(bad_ops)} launches touch a differing tensor; "
              f"outputs differing {out_bad}, worst {worst:.3g} of scale", flush=True)
        if out_bad:
            for k in (4, 5):
                if k in out_bad:
                    pattern(outs[0][k].cpu(), outs[i][k].cpu(), f"ms{k - 2}")
                    break
        for ln in open(os.path.join(outdir, f"{name}_{i}_s{sequence[i]}.txt")):
            if ln.startswith("snap_"):
                print("     " + ln.strip())
                final = {t: v for o in dumps[i] for t, v in o["tensors"].items()}
                serial = {t: v for o in ref for t, v in o["tensors"].items()}
                for t, v in re.findall(r"t(\d+)=([0-9a-f]+)", ln):
                    print(f"        t{t}: snapshot {'==' if v == final[int(t)] else '!='} final of this run, {'==' if v == serial[int(t)] else '!='} serial run")
        seen = set()
        for a, diff in bad_ops[:6]:
            new = [t for t in diff if t not in seen]
            seen.update(diff)
            print(f"     op {a['i']:3d} {a['name']:24s} {a['stream']} {a['wait']:9s} tile={a['tile']:28s} differing tensors {diff} (first seen here: {new})")
    del m
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "diag"))
    ap.add_argument("--exp", nargs="*", default=["bf16x3", "bf16x3_nosplit", "fp32_igemm"])
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for e in args.exp:
        run(e, args.batch, args.out, [1, 2, 2])


if __name__ == "__main__":
    main()
