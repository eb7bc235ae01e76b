"""bench.py must start its own ranks when the driver calls `python bench.py --gpus N` without a launcher
(no WORLD_SIZE in the environment), relay exactly one JSON line from rank 0 and fail loudly when a rank fails."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CCVPE_DIST_BACKEND"] = "gloo"
    return env


def test_self_launch_two_ranks_dry_run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--batch", "4", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rows_per_step"] == 8


def test_self_launch_eight_ranks_dry_run():
    """The round-end scaling run is N = 8: eight ranks start at once (rendezvous on 127.0.0.1, one gather per step, one JSON line)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--batch", "32", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["rows_per_step"] == 256


def test_torchrun_style_launch_eight_ranks_dry_run():
    """... and the driver's own form: python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--batch", "4", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["rows_per_step"] == 32


def test_self_launch_reports_a_failed_rank():
    # no GPU in the CPU test container: every rank refuses to run (no CPU fallback) and the parent must say so
    # instead of hanging or printing a metric
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: ranks would run the real benchmark")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "rank" in r.stderr
