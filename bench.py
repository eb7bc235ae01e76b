#!/usr/bin/env python3
"""Headline benchmark: query image pairs / second through forward(grd, sat) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): VIGOR same-area inference, CVM_VIGOR_ori_prior(ori_noise 180,
circular padding, HFoV 360), ground 3x320x640, aerial 3x512x512, batch 32 per GPU, synthetic
standard-normal inputs resident in HBM and deterministic synthetic weights (no network for datasets
or checkpoints).  One step = one forward over the batch + device-side post-processing (argmax,
cos/sin lookup) + the only collective of the path, an all_gather of the 20-byte-per-query results.
Weak scaling: every rank processes its own batch; value = all ranks' queries / max-over-ranks time.

Besides the contract line this prints a `roofline` object for the dominant kernel (the fp32-MFMA
implicit-GEMM convolution, timed per launch with hipEvents on the launch stream in a dedicated
profiled step right after the timed region) and a `cpu_baseline` object (the CPU oracle - a port of
the reference's PyTorch path - timed on this box's host cores, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ccvpe_amd import distributed as D  # noqa: E402
from ccvpe_amd import models, weights  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (variant, ctor kwargs, fov)
    "vigor_samearea_fov360_b32": ("vigor_ori_prior", dict(ori_noise=180.0, circular_padding=True), 360.0),
    "vigor_samearea_fov108_noise72": ("vigor_ori_prior", dict(ori_noise=72.0, circular_padding=False), 108.0),
    "kitti_test1": ("kitti", {}, 360.0),
    "oxford_stream": ("oxford", {}, 360.0),
}


def build_model(variant, kw, dev, micro_batch, precision="fp32"):
    cls = {"vigor": models.CVM_VIGOR, "vigor_ori_prior": models.CVM_VIGOR_ori_prior, "kitti": models.CVM_KITTI,
           "oxford": models.CVM_OxfordRobotCar}[variant]
    if variant == "vigor":
        m = cls(dev, kw.get("circular_padding", True), micro_batch=micro_batch, precision=precision)
    elif variant == "vigor_ori_prior":
        m = cls(dev, kw["ori_noise"], kw["circular_padding"], micro_batch=micro_batch, precision=precision)
    else:
        m = cls(dev, micro_batch=micro_batch, precision=precision)
    m.load_state_dict(weights.generate_state_dict(variant, 0))
    return m.to(dev).eval()


def traffic_from_profiles(kernel: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (bench.py cannot run
    the profiler on itself): profiles/r01_traffic.json = 2*FETCH_SIZE + WRITE_SIZE for one representative launch
    (decoder conv4.0 at batch 32; its algorithmic bytes are in the same file).  None if the dominant kernel of this
    run is a different instantiation."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
            t = json.load(fh)
        return t["hbm_bytes_per_launch"] if t.get("kernel") == kernel else None
    except (OSError, ValueError, KeyError):
        return None


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (os.cpu_count()
    reports the host's cores on a shared box and oversubscribes the intra-op pool badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(variant, kw, fov, budget_s=15.0):
    """The oracle (CPU port of the reference graph, proven equal to it in the build container) on the
    host cores.  Bounded sample: batch-2 forwards until ~budget_s of wall time has been spent."""
    from oracle import ccvpe_oracle as orc   # checker / baseline only - never on the product path
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = weights.generate_state_dict(variant, 0)
    b = 2
    g, s = weights.generate_inputs(variant, b, 0, fov)
    g, s = torch.from_numpy(g), torch.from_numpy(s)
    orc.forward(variant, sd, g[:1], s[:1], kw.get("circular_padding", False), kw.get("ori_noise"))   # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        orc.forward(variant, sd, g, s, kw.get("circular_padding", False), kw.get("ori_noise"))
        n += b
        if time.perf_counter() - t0 > budget_s or n >= 64:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{n} queries as batch-{b} forwards of the same workload, torch {torch.__version__} CPU fp32, no_grad"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="queries per GPU per step")
    ap.add_argument("--micro-batch", type=int, default=0)
    ap.add_argument("--workload", default="vigor_samearea_fov360_b32", choices=list(WORKLOADS))
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="fp32 = exact fp32 MFMA (default, the headline); bf16x3 = 3-term bf16 split (opt-in, ~1e-5 rel.)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-precision", action="store_true")
    ap.add_argument("--cached-aerial", action="store_true",
                    help="streaming mode: the aerial tile is encoded once (outside the timed region) and every step runs forward_cached")
    ap.add_argument("--breakdown", action="store_true", help="print the per-launch profile table to stderr")
    args = ap.parse_args()

    rank, local_rank, world = D.init_from_env()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # CCVPE_BENCH_SHARE_GPU=1 (test rigs only) lets several ranks share one GPU, with CCVPE_DIST_BACKEND=gloo
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % ndev if os.environ.get("CCVPE_BENCH_SHARE_GPU") else local_rank)
    torch.cuda.set_device(dev)

    variant, kw, fov = WORKLOADS[args.workload]
    model = build_model(variant, kw, dev, args.micro_batch, args.precision)
    g, s = weights.generate_inputs(variant, args.batch, rank, fov)
    grd, sat = torch.from_numpy(g).to(dev), torch.from_numpy(s).to(dev)

    cache = model.encode_aerial(sat) if args.cached_aerial else None

    def step():
        outs = model.forward_cached(grd, cache) if cache is not None else model(grd, sat)
        post = model.postprocess(outs[1], outs[2])
        rows = torch.stack([post["index"].to(torch.float32), post["prob"], post["cos"], post["sin"], post["angle_deg"]], dim=1)
        return D.gather_results(rows)

    for _ in range(args.warmup):
        step()
    D.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize(dev)
    D.barrier()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, dev)
    assert res.shape[0] == world * args.batch

    total_queries = world * args.batch * args.steps
    line = {
        "metric": "query images/sec (VIGOR, 512x512 sat / 320x640 grd)" if variant.startswith("vigor") else "query images/sec",
        "value": total_queries / dt,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "f32 via bf16x3 split (3 bf16 MFMA per product, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": args.workload + ("+cached_aerial" if args.cached_aerial else ""), "variant": variant, "batch_per_gpu": args.batch,
                   "global_batch": world * args.batch, "grd": list(grd.shape[1:]), "sat": list(sat.shape[1:]),
                   "parallelism": f"image-parallel x{world}, all_gather of 20 B/query results",
                   "schedule": "single stream" if os.environ.get("CCVPE_STREAMS") == "1" or args.precision != "fp32" else "two streams per GPU (aerial encoder + orientation decoder on the second)"},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel: one extra profiled step, hipEvents around every launch ----
        rows = model.profile(grd, sat)
        groups = {}
        for name, ms, fl, by in rows:
            tag = name.split("|")[1] if "|" in name else name.split(".")[-1]
            gr = groups.setdefault(tag, [0.0, 0.0, 0.0, 0])
            gr[0] += ms; gr[1] += fl; gr[2] += by; gr[3] += 1
        total_ms = sum(v[0] for v in groups.values())
        mfma = {k: v for k, v in groups.items() if k.startswith(("conv_igemm", "conv_bf16x3", "conv_wino"))}
        dom = max(mfma, key=lambda k: mfma[k][0])
        ms, fl, by, cnt = mfma[dom]
        all_ms = sum(v[0] for v in mfma.values())
        all_fl = sum(v[1] for v in mfma.values())
        line["roofline"] = {
            "kernel": dom, "bound": "mfma", "launches_per_step": cnt,
            "achieved": fl / (ms * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "avg_launch_ms": ms / cnt, "flops_per_launch": fl / cnt,
            "traffic": traffic_from_profiles(dom),
            "all_mfma_kernels": {"achieved": all_fl / (all_ms * 1e-3) / 1e12, "frac": all_fl / (all_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                 "share_of_step": all_ms / total_ms},
            "hbm_kernels_share_of_step": 1.0 - all_ms / total_ms,
        }
        if dom.startswith("conv_wino"):
            # `achieved` counts the layer's direct-convolution FLOPs (the algorithmic work); the Winograd
            # F(2x2,3x3) form issues 16/36 of them on the matrix pipe, so frac can exceed 1 - the pipe's own
            # utilisation is reported next to it
            ex = fl / 2.25 / (ms * 1e-3) / 1e12
            line["roofline"]["matrix_pipe"] = {"executed": ex, "frac": ex / PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                               "note": "Winograd F(2x2,3x3): executed = algorithmic / 2.25 (channel padding not counted)"}
        if args.breakdown:
            print(f"{'launch':40s} {'ms':>9s} {'TFLOP/s':>9s} {'GB/s':>9s}", file=sys.stderr)
            for name, ms_, fl_, by_ in rows:
                print(f"{name:40s} {ms_:9.4f} {fl_ / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0:9.2f} {by_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0:9.1f}", file=sys.stderr)
            for k, v in sorted(groups.items(), key=lambda kv: -kv[1][0]):
                print(f"  group {k:28s} {v[0]:9.3f} ms  {100 * v[0] / total_ms:5.1f}%  n={v[3]}", file=sys.stderr)
        if world == 1 and args.precision == "fp32" and not args.no_alt_precision:
            # opt-in mode, reported beside the headline (never as `value`): same workload, dense contractions as
            # a 3-term bf16 split on the bf16 matrix cores (error ~1e-5 of scale; tests hold it to 5e-4)
            del model
            torch.cuda.empty_cache()
            alt = build_model(variant, kw, dev, args.micro_batch, "bf16x3")
            for _ in range(max(args.warmup, 1)):
                alt(grd, sat)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                o = alt(grd, sat)
                alt.postprocess(o[1], o[2])
            torch.cuda.synchronize(dev)
            dta = time.perf_counter() - t0
            line["alt_precision"] = {"mode": "bf16x3", "value": args.batch * args.steps / dta, "unit": "queries/s",
                                     "ms_per_step": 1e3 * dta / args.steps,
                                     "note": "fp32 operands split into 2 bf16, 3 bf16 MFMAs per product, fp32 accumulate; not the headline"}
            del alt
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(variant, kw, fov)
        print(json.dumps(line), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
