#!/usr/bin/env python3
"""Headline benchmark: query image pairs / second through forward(grd, sat) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

With --gpus N > 1 and no WORLD_SIZE in the environment bench.py starts its N ranks itself (one child
process per GPU, before anything touches HIP) and relays rank 0's JSON line; it also runs under
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): VIGOR same-area inference, CVM_VIGOR_ori_prior(ori_noise 180,
circular padding, HFoV 360), ground 3x320x640, aerial 3x512x512, batch 32 per GPU, synthetic
standard-normal inputs resident in HBM and deterministic synthetic weights (no network for datasets
or checkpoints).  One step = one forward over the batch + device-side post-processing (argmax,
cos/sin lookup) + the only collective of the path, an all_gather of the 20-byte-per-query results.
Weak scaling: every rank processes its own batch; value = all ranks' queries / max-over-ranks time.

Besides the contract line rank 0 prints (N = 1 only for everything but `roofline`):
  roofline       dominant MFMA kernel, timed per launch with hipEvents on the launch stream in one serial profiled
                 step after the timed region.  `achieved` / `frac` = FLOPs the kernel ISSUES on the matrix pipe
                 (tile padding included; Winograd F(2x2,3x3) issues 16 products per 2x2 tile) over its time and
                 over the fp32-MFMA peak - always <= 1; `algorithmic_tflops` = the layer's direct-convolution FLOPs
                  over the same time; `end_to_end` = 56.37 GFLOP/query x queries/s over the peak.
  batch1         the same model at batch 1 (latency mode): queries/s, p50 / p99 ms of a synchronised step, kernel launches per frame
  configs        BASELINE.json configs 3-5: VIGOR FoV 108 / noise 72 (batch 32), KITTI (batch 32), Oxford streaming
                 (batch 1, p50 / p99, FPS against the reference README's 14 FPS)
  alt_precision  the opt-in bf16x3 mode (never the headline)
  cpu_baseline   the CPU oracle (a port of the reference's PyTorch graph) on this box's host cores, batch 1
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA (bf16x3 mode only)
PEAK_HBM_GBS = 8000.0
GFLOP_PER_QUERY = {"vigor_samearea_fov360_b32": 56.37, "vigor_samearea_fov108_noise72": 54.13, "kitti_test1": 54.45,
                   "oxford_stream": 53.66}   # SURVEY 8d / BASELINE.md section 3

WORKLOADS = {
    # name: (variant, ctor kwargs, fov)
    "vigor_samearea_fov360_b32": ("vigor_ori_prior", dict(ori_noise=180.0, circular_padding=True), 360.0),
    "vigor_samearea_fov108_noise72": ("vigor_ori_prior", dict(ori_noise=72.0, circular_padding=False), 108.0),
    "kitti_test1": ("kitti", {}, 360.0),
    "oxford_stream": ("oxford", {}, 360.0),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="queries per GPU per step")
    ap.add_argument("--micro-batch", type=int, default=0)
    ap.add_argument("--workload", default="vigor_samearea_fov360_b32", choices=list(WORKLOADS))
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="fp32 = exact fp32 MFMA (default, the headline); bf16x3 = 3-term bf16 split (opt-in, ~1e-5 rel.)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-precision", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the batch-1 and config 3-5 sub-results")
    ap.add_argument("--cached-aerial", action="store_true",
                    help="streaming mode: the aerial tile is encoded once (outside the timed region) and every step runs forward_cached")
    ap.add_argument("--breakdown", action="store_true", help="print the per-launch profile table to stderr")
    ap.add_argument("--dry-run", action="store_true",
                    help="test rigs only: exercise rank start-up, rendezvous (gloo), the per-step gather and the single-line "
                         "report on the CPU with fake result rows - measures nothing, prints no metric value")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves (nothing here may touch HIP / torch.cuda)
# ---------------------------------------------------------------------------------------------------------------
def self_launch(n: int) -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0's stdout carries the JSON line; the other ranks must not write to ours
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    rc = 0
    out0 = None
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            p = procs[r]
            if r == 0 and out0 is None and p.poll() is not None:
                out0 = p.stdout.read()
            if p.poll() is not None:
                alive.discard(r)
                if p.returncode != 0:
                    rc = rc or p.returncode
                    print(f"bench.py: rank {r} exited with {p.returncode}", file=sys.stderr)
        if rc and alive:   # a failed rank leaves the others waiting in the collective: stop exactly the children we started
            for r in alive:
                procs[r].kill()
            for r in alive:
                procs[r].wait()
            break
        time.sleep(0.2)
    if out0 is None and procs[0].stdout is not None:
        out0 = procs[0].stdout.read()
    for ln in (out0 or "").splitlines():   # our stdout carries the JSON record only (gloo, for one, chats on stdout)
        (sys.stdout if ln.startswith("{") and rc == 0 else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    return rc


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


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main() -> int:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)

    import torch
    from ccvpe_amd import distributed as D
    from ccvpe_amd import models, weights

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
        the profiler on itself; the file names its command): 2*FETCH_SIZE + WRITE_SIZE of one representative launch."""
        for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    t = json.load(fh)
                if t.get("kernel") == kernel or t.get("kernel", "").split("_splitk")[0].replace("_tailsplit", "") == kernel:
                    return t["hbm_bytes_per_launch"], t.get("algorithmic_bytes_per_launch"), "profiles/" + name
            except (OSError, ValueError, KeyError):
                continue
        return None, None, None

    def cpu_baseline(variant, kw, fov):
        """The oracle (CPU port of the reference graph, proven equal to it in the build container) on the host cores,
        as BASELINE.md section 4 plans it: batch 1 (BASELINE.json configs[0]), warm-up 1, median of 5 under no_grad,
        3 forwards with autograd recording (as train_VIGOR.py:282 runs it), batch 8 (warm-up 1, median of 3) and the headline
        batch 32 (warm-up 1, one timed forward: ~20 s each) - about a minute of CPU work in all."""
        from oracle import ccvpe_oracle as orc   # checker / baseline only - never on the product path
        cores = usable_cores()
        torch.set_num_threads(cores)
        sd = weights.generate_state_dict(variant, 0)
        g, s = weights.generate_inputs(variant, 32, 0, fov)
        g, s = torch.from_numpy(g), torch.from_numpy(s)
        circ, noise = kw.get("circular_padding", False), kw.get("ori_noise")

        def timed(fn, n):
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2]

        orc.forward(variant, sd, g[:1], s[:1], circ, noise)   # warm-up
        t1 = timed(lambda: orc.forward(variant, sd, g[:1], s[:1], circ, noise), 5)
        sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v) for k, v in sd.items()}   # parameters, not BN buffers
        tg = timed(lambda: orc.forward(variant, sdg, g[:1], s[:1], circ, noise, grad=True), 3)
        orc.forward(variant, sd, g[:8], s[:8], circ, noise)   # warm-up (allocator, thread pool at this size)
        t8 = timed(lambda: orc.forward(variant, sd, g[:8], s[:8], circ, noise), 3)
        orc.forward(variant, sd, g, s, circ, noise)           # warm-up
        t32 = timed(lambda: orc.forward(variant, sd, g, s, circ, noise), 1)
        return {"value": 1.0 / t1, "unit": "queries/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
                "sample": f"batch-1 forwards of the same model (BASELINE.json configs[0]): warm-up 1, median of 5, torch {torch.__version__} CPU fp32, no_grad",
                "batch1_autograd_on": {"value": 1.0 / tg, "note": "median of 3, autograd recording as in train_VIGOR.py:282"},
                "batch8": {"value": 8.0 / t8, "note": "batch-8 forwards: warm-up 1, median of 3"},
                "batch32": {"value": 32.0 / t32, "note": "the headline batch (BASELINE.json configs[1] on the CPU): warm-up 1, one timed forward, no_grad"}}

    if args.dry_run:
        rank, local_rank, world = D.init_from_env("gloo")
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        for _ in range(args.steps):
            res = D.gather_results(torch.full((args.batch, 5), float(rank)))
        D.barrier()
        assert res.shape[0] == world * args.batch and float(res[-1, 0]) == world - 1
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "rows_per_step": int(res.shape[0])}), flush=True)
        D.barrier()
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return 0

    rank, local_rank, world = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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

    def make_step(m, gg, ss, cc=None):
        def step():
            outs = m.forward_cached(gg, cc) if cc is not None else m(gg, ss)
            rows = m.postprocess_rows(outs[1], outs[2])   # [B, 5] float rows written by the post-processing launch itself
            return D.gather_results(rows)
        return step

    step = make_step(model, grd, sat, cache)
    # first call of the headline plan: weight ingestion + plan build (+ measuring whatever launches the tuning table does not know)
    torch.cuda.synchronize(dev)
    t_first = time.perf_counter()
    if args.warmup > 0:
        step()
        torch.cuda.synchronize(dev)
    t_first = time.perf_counter() - t_first
    from ccvpe_amd import _lib as _cl
    tuned_plans = _cl.load().ccvpe_tuning_generation(model._handle) if model._handle is not None else -1
    for _ in range(max(args.warmup - 1, 0)):
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
    two_streams = not (os.environ.get("CCVPE_STREAMS") == "1")
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
                   "schedule": "two streams per GPU (aerial encoder + orientation decoder on the second)" if two_streams else "single stream"},
    }

    line["startup"] = {"first_step_s": t_first, "plans_measured_at_startup": tuned_plans,
                       "note": "first call of the batch-32 plan: state_dict ingestion + weight packing + plan build; 0 plans measured = every launch came from the tuning table"}

    def latency_run(m, gg, ss, n=200, warm=12):
        """Per-call latency (synchronised after every step) and back-to-back throughput of one model / input."""
        st = make_step(m, gg, ss)
        for _ in range(warm):
            st()
        torch.cuda.synchronize(dev)
        lat = []
        for _ in range(n):
            t = time.perf_counter()
            st()
            torch.cuda.synchronize(dev)
            lat.append(1e3 * (time.perf_counter() - t))
        lat.sort()
        t = time.perf_counter()
        for _ in range(n):
            st()
        torch.cuda.synchronize(dev)
        thr = n * gg.shape[0] / (time.perf_counter() - t)
        from ccvpe_amd import _lib as _cl2
        c0 = _cl2.load().ccvpe_launch_count()
        st()
        torch.cuda.synchronize(dev)
        launches = int(_cl2.load().ccvpe_launch_count() - c0)
        return {"queries_per_s": thr, "p50_ms": lat[n // 2], "p99_ms": lat[min(n - 1, int(n * 0.99))], "steps": n, "batch": gg.shape[0],
                "launches_per_frame": launches,
                "launches_note": "kernel launches the library issues for one step (forward + postprocess; ccvpe_launch_count), eager issue interleaved over two streams"}

    def throughput_run(name, batch, n=30, warm=5):
        v, k, f = WORKLOADS[name]
        m = build_model(v, k, dev, args.micro_batch, "fp32")
        a, b = weights.generate_inputs(v, batch, 0, f)
        a, b = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        if batch == 1:
            r = latency_run(m, a, b)
        else:
            st = make_step(m, a, b)
            for _ in range(warm):
                st()
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for _ in range(n):
                st()
            torch.cuda.synchronize(dev)
            d = time.perf_counter() - t
            r = {"queries_per_s": n * batch / d, "ms_per_step": 1e3 * d / n, "steps": n, "batch": batch}
        r["grd"] = list(a.shape[1:])
        r["end_to_end_frac_of_fp32_mfma_peak"] = r["queries_per_s"] * GFLOP_PER_QUERY[name] / 1e3 / PEAK_FP32_MFMA_TFLOPS
        del m
        torch.cuda.empty_cache()
        return r

    def pipeline_run(m, n=20, warm=3):
        """The whole on-device test loop for one batch (the counterpart of train_VIGOR.py:265-326): decoded uint8 images (1024 x 2048
        panoramas, 640 x 640 aerial tiles, VIGOR's native sizes) -> PIL-exact resize + ToTensor + Normalize + roll (ccvpe_preprocess_resize)
        -> forward -> argmax / orientation lookup -> ground-truth metrics (ccvpe_eval_metrics) -> gather of the compact results."""
        from ccvpe_amd import _lib
        B = args.batch
        gen = torch.Generator().manual_seed(17)
        pano = torch.randint(0, 256, (B, 1024, 2048, 3), dtype=torch.uint8, generator=gen).to(dev)
        tile = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, generator=gen).to(dev)
        shift = torch.randint(0, 640, (B,), dtype=torch.int32, generator=gen).to(dev)
        gt_index = torch.randint(0, 512 * 512, (B,), dtype=torch.int32, generator=gen).to(dev)
        ang = torch.rand(B, generator=gen) * 6.2831853
        gt_cs = torch.stack([torch.cos(ang), torch.sin(ang)], dim=1).to(dev)
        mpp = 0.113248 / 512 * 640     # train_VIGOR.py:301-308 (NewYork)

        def st():
            gg = _lib.preprocess_resize(pano, (320, 640), shift=shift)
            ss = _lib.preprocess_resize(tile, (512, 512))
            outs = m(gg, ss)
            ev = m.evaluate(outs[1], outs[2], gt_index, mpp, gt_cos_sin=gt_cs)
            rows = torch.stack([ev["index"].to(torch.float32), ev["meter_distance"].to(torch.float32), ev["prob_at_gt"].to(torch.float32),
                                ev["orientation_error_deg"].to(torch.float32), ev["angle_pred_deg"].to(torch.float32)], dim=1)
            return D.gather_results(rows)
        for _ in range(warm):
            st()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(n):
            r = st()
        torch.cuda.synchronize(dev)
        d = time.perf_counter() - t
        assert r.shape == (B, 5) and bool(torch.isfinite(r).all())
        return {"queries_per_s": n * B / d, "ms_per_step": 1e3 * d / n, "steps": n, "batch": B,
                "stages": "uint8 1024x2048 + 640x640 (HBM resident) -> preprocess_resize -> forward -> postprocess -> eval_metrics -> gather",
                "input_bytes_per_query": 1024 * 2048 * 3 + 640 * 640 * 3}

    if rank == 0:
        peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
        # ---- roofline of the dominant kernel: one extra profiled step, hipEvents around every launch ----
        rows = model.profile(grd, sat)
        groups = {}
        for name, ms, fl, by, iss in rows:
            tag = name.split("|")[1] if "|" in name else name.split(".")[-1]
            gr = groups.setdefault(tag, [0.0, 0.0, 0.0, 0, 0.0])
            gr[0] += ms; gr[1] += fl; gr[2] += by; gr[3] += 1; gr[4] += iss
        total_ms = sum(v[0] for v in groups.values())
        # tiled GEMM / Winograd launches (tag = tile name) - the candidates for the dominant kernel ...
        mfma = {k: v for k, v in groups.items() if k.startswith(("conv_igemm", "conv_bf16x3", "conv_wino", "conv_pw", "conv_proj"))}
        # the dominant KERNEL is a template instantiation (= tile name: conv_wino4_16x128 is conv_wino4_kernel<8>, conv_wino4_16x64 is
        # conv_wino4_kernel<4> in a rocprofv3 trace); its launches may differ in split-K, which only changes the grid: the tile-tag
        # groups (tile + split) are kept as `sub_groups`
        def instantiation(tag):
            return tag.split("_splitk")[0].replace("_tailsplit", "")
        inst = {}
        for k, v in mfma.items():
            gr = inst.setdefault(instantiation(k), [0.0, 0.0, 0.0, 0, 0.0])
            for i in range(5):
                gr[i] += v[i]
        dom = max(inst, key=lambda k: inst[k][0])
        ms, fl, by, cnt, iss = inst[dom]
        sub_groups = {k: {"launches_per_step": v[3], "ms_per_step": v[0], "issued_tflops": v[4] / (v[0] * 1e-3) / 1e12,
                          "frac": v[4] / (v[0] * 1e-3) / 1e12 / peak} for k, v in sorted(mfma.items()) if instantiation(k) == dom}
        # ... and every launch whose arithmetic runs on the matrix cores: + the fused last decoder level (level1_kernel), the fused
        # MBConv fronts (expand GEMM + depthwise) and the MFMA rolling match of the wide levels
        def on_matrix_cores(tag):
            return tag in mfma or tag in ("fused", "expand_dw") or tag in ("match1", "match2", "match3", "match4")
        allm = {k: v for k, v in groups.items() if on_matrix_cores(k)}
        all_ms = sum(v[0] for v in allm.values())
        all_fl = sum(v[1] for v in allm.values())
        all_iss = sum(v[4] for v in allm.values())
        # the most time-consuming launch group that is NOT on the matrix cores: the HBM-side entry of the roofline
        # (groups of launches under 20 us each - squeeze-excite, descriptor heads - are launch-latency bound: listed separately)
        hbm = {k: v for k, v in groups.items() if not on_matrix_cores(k) and v[2] > 0 and v[0] / v[3] >= 0.02}
        tiny = {k: v for k, v in groups.items() if not on_matrix_cores(k) and v[0] / v[3] < 0.02}
        hdom = max(hbm, key=lambda k: hbm[k][0]) if hbm else None
        traffic, traffic_alg, traffic_src = traffic_from_profiles(dom)
        line["roofline"] = {
            "kernel": dom, "bound": "mfma", "launches_per_step": cnt,
            "achieved": iss / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
            "frac": iss / (ms * 1e-3) / 1e12 / peak,
            "achieved_is": "FLOPs issued on the matrix pipe (tile padding included; Winograd F(4x4,3x3) tiles = 36 products per 4x4 output tile and channel pair, F(2x2,3x3) = 16 per 2x2 tile: a quarter / 4 ninths of the direct-convolution count in algorithmic_tflops)",
            "algorithmic_tflops": fl / (ms * 1e-3) / 1e12,
            "avg_launch_ms": ms / cnt, "issued_flops_per_launch": iss / cnt, "algorithmic_flops_per_launch": fl / cnt,
            "traffic": traffic, "traffic_algorithmic": traffic_alg, "traffic_source": traffic_src,
            "sub_groups": sub_groups,
            "all_mfma_kernels": {"issued_tflops": all_iss / (all_ms * 1e-3) / 1e12, "frac": all_iss / (all_ms * 1e-3) / 1e12 / peak,
                                 "algorithmic_tflops": all_fl / (all_ms * 1e-3) / 1e12, "share_of_serial_step": all_ms / total_ms},
            "hbm_bound": None if hdom is None else {
                "kernel": hdom, "bound": "hbm", "launches_per_step": hbm[hdom][3], "ms_per_step": hbm[hdom][0],
                "algorithmic_bytes_per_step": hbm[hdom][2], "achieved": hbm[hdom][2] / (hbm[hdom][0] * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": hbm[hdom][2] / (hbm[hdom][0] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "note": "most time-consuming launch group off the matrix cores; algorithmic bytes (each tensor once) over the hipEvent time; "
                        "8 TB/s spec, ~6.3 TB/s achievable (MI355X_MICROARCH.md); PMC FETCH/WRITE sizes per kernel: profiles/r04_hbm_kernels.md"},
            "latency_bound_launches": {"groups": sorted(tiny), "launches_per_step": sum(v[3] for v in tiny.values()), "ms_per_step": sum(v[0] for v in tiny.values())},
            "serial_step_ms": total_ms,
            "end_to_end": {"gflop_per_query": GFLOP_PER_QUERY[args.workload],
                           "algorithmic_tflops": line["value"] / world * GFLOP_PER_QUERY[args.workload] / 1e3,
                           "frac_of_fp32_mfma_peak": line["value"] / world * GFLOP_PER_QUERY[args.workload] / 1e3 / PEAK_FP32_MFMA_TFLOPS,
                           # what the matrix pipe actually executes per query (Winograd tiles issue 1/4 or 4/9 of the direct count, tile
                           # and K padding included) x the TIMED two-stream rate: the whole step's matrix-pipe utilisation, always <= 1
                           "issued_gflop_per_query": all_iss / args.batch / 1e9,
                           "issued_tflops": line["value"] / world * all_iss / args.batch / 1e12,
                           "issued_frac": line["value"] / world * all_iss / args.batch / 1e12 / peak},
        }
        if args.breakdown:
            print(f"{'launch':48s} {'ms':>9s} {'alg TF/s':>9s} {'iss TF/s':>9s} {'GB/s':>9s}", file=sys.stderr)
            for name, ms_, fl_, by_, is_ in rows:
                print(f"{name:48s} {ms_:9.4f} {fl_ / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0:9.2f} {is_ / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0:9.2f} "
                      f"{by_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0:9.1f}", file=sys.stderr)
            for k, v in sorted(groups.items(), key=lambda kv: -kv[1][0]):
                print(f"  group {k:28s} {v[0]:9.3f} ms  {100 * v[0] / total_ms:5.1f}%  n={v[3]}", file=sys.stderr)
        extras = world == 1 and args.precision == "fp32" and not args.cached_aerial
        if extras and not args.no_extra:
            # BASELINE.json metric asks batch 1 AND 32; BASELINE.md section 5 lists configs 3-5
            line["batch1"] = latency_run(model, grd[:1].contiguous(), sat[:1].contiguous())
            line["batch1"]["note"] = "same model, batch 1 (latency mode: eager launches interleaved over two streams; CCVPE_GRAPH=1 replays a hipGraph instead); latency = one synchronised step (forward + post-processing)"
            if variant.startswith("vigor") and fov == 360.0:
                line["pipeline"] = pipeline_run(model)
        del model
        torch.cuda.empty_cache()
        if extras and not args.no_extra:
            cfgs = {}
            if args.workload != "vigor_samearea_fov108_noise72":
                cfgs["config3_vigor_fov108_noise72_b32"] = throughput_run("vigor_samearea_fov108_noise72", 32)
            cfgs["config4_kitti_b32_per_gpu"] = throughput_run("kitti_test1", 32)
            ox = throughput_run("oxford_stream", 1)
            ox["fps_vs_reference_readme_14fps"] = ox["queries_per_s"] / 14.0
            ox["note"] = "streaming, batch 1; the reference's 14 FPS (README.md:21) is on an unnamed GPU and includes data loading"
            cfgs["config5_oxford_stream_b1"] = ox
            line["configs"] = cfgs
        if extras and not args.no_alt_precision:
            # opt-in mode, reported beside the headline (never as `value`): same workload, dense contractions as
            # a 3-term bf16 split on the bf16 matrix cores (error ~1e-5 of scale; tests hold it to 5e-4)
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
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(variant, kw, fov)
        print(json.dumps(line), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
