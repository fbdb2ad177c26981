#!/usr/bin/env python3
"""bench.py -- env.step throughput of the MI355X-native 3D-RVO drone environment.

Metric (BASELINE.json): drone-steps/sec of `env.step` at 64 drones x 4096 envs
per GPU (config 3 of BASELINE.json; synthetic worlds/actions of SURVEY.md 8(d)).
One "step" = one launch of the fused HIP step (RVO reward sweep -> integrate ->
observation / reward / termination sweep -> auto-reset + re-observe) over the
whole batch, with the actions already resident in HBM.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Envs are independent, so N GPUs = N shards of 4096 envs each (weak scaling) and
no data-path collective; the only collectives are the timing barrier and the
max-over-ranks of the elapsed time.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : algorithmic bytes per launch / average kernel time (HIP events on
                 the launch stream) against the 8 TB/s HBM peak
  cpu_baseline : the CPU oracle (oracle/, the validated C restatement of the
                 reference step) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd")]

import numpy as np
import torch

ALLOW_DIAG = False  # set by tools/bench_diag.py only (the -DRVO3D_DIAG build of the library)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def algorithmic_bytes_per_drone_step(nm: int, nb: int, N: int) -> float:
    """B = 359 + 36*nm (+ 32*nb/N)."""
    return 359.0 + 36.0 * nm + 32.0 * nb / N


def cpu_baseline(N, nm, map_size, seconds_target=12.0):
    """The oracle (kind 'port') on this host: E scaled down, same generator."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc  # test infrastructure, used here only as the timed CPU baseline
    from rvo3d_amd import synthetic_actions, synthetic_world
    E = 64
    w = synthetic_world(E, N, map_size)
    best = None
    ncpu = len(os.sched_getaffinity(0))
    for threads in sorted({1, min(ncpu, 16)}):
        env = orc.OracleEnv(w.waypoints, w.n_points, w.map_size, w.buildings, nm=nm, threads=threads)
        acts = [synthetic_actions(E, N, t) for t in range(8)]
        for t in range(3):
            env.step_autoreset(acts[t])
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < seconds_target / 2:
            env.step_autoreset(acts[steps % len(acts)])
            steps += 1
        dt = time.perf_counter() - t0
        rate = E * N * steps / dt
        rec = dict(value=rate, cores=threads, steps=steps)
        if threads == 1:
            one = rate
        if best is None or rate > best["value"]:
            best = rec
    return {"value": round(best["value"], 1), "unit": "drone-steps/s", "cores": best["cores"],
            "kind": "port",
            "sample": f"{N} drones x {E} envs x {best['steps']} fused steps (oracle/rvo3d_oracle.c, "
                      f"OpenMP over envs; single-thread rate {one:.0f}/s; host has {ncpu} cpus)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm", type=int, default=300,
                    help="untimed steps before the W warmup steps (lets the GPU clocks settle; "
                         "a launch is ~60 us, so W alone is a few ms)")
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--drones", type=int, default=64)
    ap.add_argument("--nm", type=int, default=10)
    ap.add_argument("--buildings", type=int, default=0)
    ap.add_argument("--map", type=float, nargs=3, default=[50.0, 50.0, 10.0])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autoreset", action="store_true")
    args = ap.parse_args()

    from rvo3d_amd import BatchedDroneEnv, _lib, sharding, synthetic_actions, synthetic_world

    # The product library has no diagnostics (no phase ablation, no stamps, reads no environment
    # variable); a diagnostics build can only get here through tools/bench_diag.py.
    diag = _lib.is_diag_build()
    if diag and not ALLOW_DIAG:
        raise SystemExit("bench.py refuses the diagnostics build of the library (use tools/bench_diag.py)")

    rank, local_rank, world = sharding.rank_info()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist = sharding.init_process_group("nccl", dev)  # nccl == RCCL on ROCm

    E, N, nm, nb = args.envs, args.drones, args.nm, args.buildings
    K, W = args.steps, args.warmup
    # shard = this rank's own envs: seed offset per rank
    wld = synthetic_world(E, N, tuple(args.map), nb=nb, seed=sharding.shard_seed(1234, rank))
    env = BatchedDroneEnv(wld, neighbors_num=nm, device=dev, action_decimals=2)
    # actions for every step, resident in HBM before the timed region (float32,
    # re-quantised on device to the 2-decimal fp64 values the reference steps with)
    n_act = min(K + W, 64)
    acts = torch.stack([torch.from_numpy(synthetic_actions(E, N, t, seed=1234 + rank).astype(np.float32))
                        for t in range(n_act)]).to(dev)
    autoreset = not args.no_autoreset
    env.observe()
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for t in range(args.prewarm):
        env.step(acts[t % n_act], autoreset=autoreset)
    for t in range(W):
        env.step(acts[t % n_act], autoreset=autoreset)
    barrier()
    t0 = time.perf_counter()
    for t in range(K):
        env.step(acts[(W + t) % n_act], autoreset=autoreset)
    barrier()
    elapsed = sharding.max_over_ranks(dist, time.perf_counter() - t0, dev)

    # kernel time: HIP events on the launch stream (torch's current stream), one pair per launch
    stream = torch.cuda.current_stream(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(K)]
    for t, (a, b) in enumerate(evs):
        a.record(stream)
        env.step(acts[(W + t) % n_act], autoreset=autoreset)
        b.record(stream)
    torch.cuda.synchronize()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    flags = env.error_flags()

    if rank == 0:
        total_units = world * E * N * K
        value = total_units / elapsed
        B = algorithmic_bytes_per_drone_step(nm, nb, N)
        bytes_per_launch = E * N * B
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{N}x{E}")
            except Exception:
                traffic = None
        out = {
            "metric": "drone-steps/sec (env.step throughput)",
            "value": round(value, 1), "unit": "drone-steps/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE config 3 env.step: {N} drones x {E} envs per GPU, "
                                   f"nm={nm}, {nb} buildings, map {args.map}, fused step"
                                   f"{'+auto-reset' if autoreset else ''}, f32 actions in HBM",
                       "envs_per_gpu": E, "drones": N, "launch": env.launch_info(), "diag_build": diag,
                       "ablate": int(os.environ.get("RVO3D_ABLATE", "0")) if diag else 0,
                       "device_error_word": flags},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "kernel": ("rvo3d::env_kernel<2, 1, 64> (fused step + auto-reset)" if N == 64 else
                                    "rvo3d::env_kernel<2, 1> (fused step + auto-reset)") if N <= 64 else
                                   "rvo3d::env_kernel<2, NW> (fused step + auto-reset)",
                         "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "bytes_per_drone_step": B},
        }
        if not args.no_cpu_baseline and world == 1:  # reported once, at N = 1
            out["cpu_baseline"] = cpu_baseline(N, nm, tuple(args.map))
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
