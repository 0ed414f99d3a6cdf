"""The data-parallel path on CPU: world_size 2, gloo backend (the N>1 path of bench.py / trainer.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from fbk_fairseq_st_amd import distributed as D
    r = D.distributed_init("gloo", "cpu")
    assert r == rank and D.get_world_size() == world
    # bucketed gradient all-reduce: every element reduced exactly once, buckets launched from "backward"
    n = 5000
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = D.BucketedGradReducer(g, bucket_bytes=4 * 1200)
    for s, e in [(4000, 5000), (3500, 4000), (1500, 3500), (1000, 1500)]:      # descending, as backward produces them
        red.notify(s, e)
    launched_early = len(red.launched)
    red.finish()
    expect = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1))
    ok_sum = bool(torch.equal(g, expect))
    stats = D.all_reduce_stats({"sample_size": 3.0 + rank, "nframes": 100.0 * (rank + 1)})
    same = D.check_grad_norms(float(g.norm()))
    bad = False
    try:
        D.check_grad_norms(1.0 + rank)
    except FloatingPointError:
        bad = True
    q.put((rank, ok_sum, launched_early, stats, same, bad))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_bucketed_allreduce_and_stats():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_sum, launched_early, stats, same, bad in res:
        assert ok_sum, "all-reduced gradient differs from the sum over ranks"
        assert launched_early >= 2, "buckets must be launched before backward ends (overlap)"
        assert stats == {"nframes": 300.0, "sample_size": 7.0}
        assert same and bad


def test_projected_exposed_allreduce_is_plain_queueing_arithmetic():
    """bench.py's N = 1 `data_parallel.dry_run.projection` (a labelled projection, not a measurement): buckets run back to back on one
    RCCL stream from the moment each became launchable; what sticks out past the end of backward is exposed"""
    from fbk_fairseq_st_amd.distributed import project_exposed_allreduce
    # 8 ranks, 100 GB/s bus: a bucket of 25e6 f32 elements = 100 MB takes 2 * 7/8 * 1e8 / 1e11 s = 1.75 ms
    exposed, total = project_exposed_allreduce([(25_000_000, 0.0)], finish_ms=5.0, n_ranks=8, bus_gbps=100.0)
    assert abs(total - 1.75) < 1e-9 and exposed == 0.0                       # fully under the remaining backward
    exposed, total = project_exposed_allreduce([(25_000_000, 0.0), (25_000_000, 1.0), (25_000_000, 4.0)], 4.0, 8, 100.0)
    assert abs(total - 5.25) < 1e-9 and abs(exposed - (5.75 - 4.0)) < 1e-9   # 0 -> 1.75 -> 3.5, third starts at 4.0 -> 5.75
    exposed, _ = project_exposed_allreduce([(25_000_000, 3.0)], 3.0, 2, 100.0)  # launched by finish(): all of it exposed (2 ranks: 1.0 ms)
    assert abs(exposed - 1.0) < 1e-9
