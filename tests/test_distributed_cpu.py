"""N>1 path on CPU: world_size-2 gloo processes shard a query range and all_gather compact results."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ccvpe_amd import distributed as D


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 32, 52605):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_pose(idx: torch.Tensor) -> torch.Tensor:
    f = idx.to(torch.float32)
    return torch.stack([f, f * 0.5, torch.cos(f), torch.sin(f), f % 360], dim=1)


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        r, lr, w = D.init_from_env("gloo")
        lo, hi = D.shard_range(n_items, r, w)
        local = _fake_pose(torch.arange(lo, hi))
        full = D.gather_results(local, n_items)
        D.barrier()
        t = D.max_over_ranks(float(rank + 1))
        ok = torch.equal(full, _fake_pose(torch.arange(n_items))) and t == float(world)
        q.put((rank, bool(ok), tuple(full.shape)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 7])
def test_two_rank_gloo_gather(n_items):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res) and all(shape == (n_items, 5) for _, _, shape in res)
