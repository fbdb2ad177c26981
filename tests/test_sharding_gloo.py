"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding helpers
bench.py uses (disjoint shards, barrier, MAX of elapsed, SUM of units).  The step
itself needs a GPU; what is covered here is that ranks agree on the job-level
numbers and own disjoint environments."""
import os
import socket

import torch.multiprocessing as mp

from rvo3d_amd import sharding, synthetic_world


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = sharding.init_process_group("gloo")
    r, lr, w = sharding.rank_info()
    assert (r, w) == (rank, world)
    wld = synthetic_world(4, 6, (20, 20, 8), seed=sharding.shard_seed(1234, rank))
    dist.barrier()
    elapsed = 1.0 + 0.5 * rank  # rank 1 is the slow one
    job = sharding.max_over_ranks(dist, elapsed)
    units = sharding.sum_over_ranks(dist, 4 * 6 * 10)
    lo, hi = sharding.shard_env_range(10, rank, world)
    q.put((rank, job, units, lo, hi, float(wld.waypoints.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, job0, units0, lo0, hi0, s0), (r1, job1, units1, lo1, hi1, s1) = out
    assert job0 == job1 == 1.5                 # MAX over ranks, identical on every rank
    assert units0 == units1 == 2 * 4 * 6 * 10  # whole-job units
    assert (lo0, hi0, lo1, hi1) == (0, 5, 5, 10)
    assert s0 != s1                            # different shards = different worlds


def test_shard_ranges_cover_and_are_disjoint():
    for total in (1, 7, 4096, 32768):
        for world in (1, 2, 3, 8):
            r = [sharding.shard_env_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
