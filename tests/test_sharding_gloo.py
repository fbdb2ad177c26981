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


def _bucket_worker(rank, world, port, q):
    import torch
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = sharding.init_process_group("gloo")
    import bench  # the driver's script: its multi-GPU leg is this class

    class Env:
        E, N, W, device = 2, 3, 102, torch.device("cpu")
    b = bench.GradBucket(Env(), dist)
    for p in b.ac.parameters():
        p.grad.fill_(float(rank + 1))                  # (views of the bucket) rank 0: 1, rank 1: 2 -> mean 1.5
    b.tr._bucket()[-1] = float(rank)                   # rank 0: 0, rank 1: 1 -> mean 0.5, in the bucket's last slot
    b.step()
    seen = b.ranks_seen()
    g = torch.cat([p.grad.reshape(-1) for p in b.ac.parameters()])
    q.put((rank, seen, float(g.min()), float(g.max()), float(b.tr._bucket()[-1]), b.nbytes))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_gradient_bucket_collective_two_ranks():
    """bench.py's N > 1 leg on CPU (gloo, world_size 2): the trainer's flattened gradient bucket
    is averaged over the ranks in ONE all-reduce, the KL estimate riding in its last slot, and
    ranks_seen comes from a real all-reduce.  0.74 MB = the MLP(256,256) policy of BASELINE config 3."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen, gmin, gmax, kl, nbytes in out:
        assert seen == 2 and gmin == gmax == 1.5 and kl == 0.5
        assert 0.70e6 < nbytes < 0.80e6


def test_shard_ranges_cover_and_are_disjoint():
    for total in (1, 7, 4096, 32768):
        for world in (1, 2, 3, 8):
            r = [sharding.shard_env_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
