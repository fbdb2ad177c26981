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

Envs are independent, so N GPUs = N shards of 4096 envs each (weak scaling) and the
step itself needs no data-path collective.  The one collective of the north-star
design - the RCCL all-reduce of the flattened policy-gradient bucket, the KL estimate
in its last slot, once per optimizer step (rvo3d_amd.policy.multi_ppo._allreduce_grads /
update) - is put INSIDE every timed step whenever N > 1 (or with
--grad-allreduce): the trainer's own code path on the gradients of the MLP(256,256)
policy of BASELINE config 3 (0.74 MB fp32).  That is the worst case - one optimizer
step per env step; training does one per several hundred - so the multi-GPU record
contains real xGMI traffic and the scaling efficiency read from it is a lower bound.
The collective is issued on a second HIP stream (it depends on nothing the env step
produces), so it overlaps the next env step the way it would overlap a backward pass;
--serial-collective puts it on the step's stream instead.
At N = 1 the step is the env step alone (the headline).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : algorithmic bytes per launch / average kernel time (HIP events on
                 the launch stream) against the 8 TB/s HBM peak; `traffic` (HBM bytes per
                 launch) and `fp64_valu_tflops` come from the PMC record of this very
                 command (profiles/traffic.json, written by tools/pmc_record.sh: separate
                 rocprofv3 --pmc passes) - `traffic_source` says so
  cpu_baseline : the CPU oracle (oracle/, the validated C restatement of the
                 reference step) timed on this host on a bounded sample, on 1 thread
                 and on every core the process may use
  collective   : (N > 1) ranks_seen, allreduce_us, bucket bytes.
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
    """The oracle (kind 'port') on this host, same generator, E scaled down: one thread (the
    scalar port) and every core this process may run on (OpenMP over envs)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc  # test infrastructure, used here only as the timed CPU baseline
    from rvo3d_amd import synthetic_actions, synthetic_world
    ncpu = len(os.sched_getaffinity(0))
    recs = {}
    # 1 thread, the GPU box's per-GPU CPU share (16) and every cpu the process may run on: on a
    # shared host the last can be slower than the share (oversubscription); the best is reported
    for threads in sorted({1, min(ncpu, 16), ncpu}):
        E = 64 if threads == 1 else max(64, 4 * threads)  # several envs per thread
        w = synthetic_world(E, N, map_size)
        env = orc.OracleEnv(w.waypoints, w.n_points, w.map_size, w.buildings, nm=nm, threads=threads)
        acts = [synthetic_actions(E, N, t) for t in range(8)]
        for t in range(3):
            env.step_autoreset(acts[t])
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < seconds_target / 3:
            env.step_autoreset(acts[steps % len(acts)])
            steps += 1
        dt = time.perf_counter() - t0
        recs[threads] = dict(value=E * N * steps / dt, cores=threads, steps=steps, envs=E)
    best = max(recs.values(), key=lambda r: r["value"])
    one = recs[1]
    return {"value": round(best["value"], 1), "unit": "drone-steps/s", "cores": best["cores"],
            "kind": "port", "value_1core": round(one["value"], 1), "nproc": ncpu,
            "by_threads": {str(k): round(v["value"], 1) for k, v in sorted(recs.items())},
            "sample": f"{N} drones x {best['envs']} envs x {best['steps']} fused steps on {best['cores']} threads "
                      f"(oracle/rvo3d_oracle.c, OpenMP over envs); 1 thread: {N} drones x {one['envs']} envs x "
                      f"{one['steps']} steps = {one['value']:.0f}/s; this process may use {ncpu} cpus"}


class GradBucket:
    """The policy-gradient collective of one optimizer step, through the trainer's own code
    (multi_ppo._allreduce_grads + the KL mean of multi_ppo.update): MLP(256,256) actor-critic of
    BASELINE config 3 on the 12 + 9*nm observation, gradients filled once (their values do not
    matter to the collective's cost)."""

    def __init__(self, env, dist):
        from rvo3d_amd.policy import mlp_ac, multi_ppo
        self.dist = dist
        self.ac = mlp_ac(env.W).to(env.device)
        self.tr = multi_ppo(env, self.ac, steps_per_epoch=1, dist=dist)
        for p in self.ac.parameters():
            p.grad = torch.ones_like(p)
        self.kl32 = torch.zeros(1, dtype=torch.float32, device=env.device)
        self.ones = torch.ones(1, dtype=torch.float64, device=env.device)
        self.nbytes = (sum(p.numel() for p in self.ac.parameters()) + 1) * 4

    def step(self):
        # ONE collective: the flattened bucket (sum, then / world) with the KL estimate in its last
        # slot, exactly as multi_ppo's policy pass issues it (the mean KL is left on the device here:
        # the trainer's host read of it is not part of the collective's cost)
        d = self.dist
        if d is None:
            return
        flat = self.tr._bucket()
        flat[-1:].copy_(self.kl32)
        d.all_reduce(flat)
        flat /= d.get_world_size()

    def ranks_seen(self):
        t = self.ones.clone()
        if self.dist is not None:
            self.dist.all_reduce(t)
        return int(round(float(t.item())))


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
    ap.add_argument("--grad-allreduce", action="store_true",
                    help="every timed step also runs the trainer's gradient-bucket all-reduce + KL mean "
                         "(default whenever --gpus > 1)")
    ap.add_argument("--no-grad-allreduce", action="store_true", help="N > 1 without the collective")
    ap.add_argument("--serial-collective", action="store_true",
                    help="issue the collective on the env step's own stream (default: a second HIP stream, so "
                         "that it overlaps the next env step as it would overlap any other compute)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse the N > 1 code path where the ranks cannot have a GPU each)")
    ap.add_argument("--all-ranks-on-device", type=int, default=None,
                    help="rehearsal only: every rank uses this one GPU (needs --backend gloo)")
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
    dev_index = local_rank if args.all_ranks_on_device is None else args.all_ranks_on_device
    if args.all_ranks_on_device is not None and args.backend == "nccl" and world > 1:
        raise SystemExit("--all-ranks-on-device needs --backend gloo (RCCL wants one GPU per rank)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        dist = sharding.init_process_group(args.backend, dev)  # nccl == RCCL on ROCm

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
    with_coll = (world > 1 or args.grad_allreduce) and not args.no_grad_allreduce
    bucket = GradBucket(env, dist) if with_coll else None
    torch.cuda.synchronize()

    # The collective depends on nothing the env step produces (and vice versa): it goes to its own
    # HIP stream and overlaps the following env step(s); both are inside the timed region.
    side = None
    if bucket is not None and not args.serial_collective:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))

    def one_step(t):
        env.step(acts[t % n_act], autoreset=autoreset)
        if bucket is not None:
            if side is not None:
                with torch.cuda.stream(side):
                    bucket.step()
            else:
                bucket.step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for t in range(args.prewarm):
        one_step(t)
    for t in range(W):
        one_step(t)
    barrier()
    t0 = time.perf_counter()
    for t in range(K):
        one_step(W + t)
    barrier()
    elapsed = sharding.max_over_ranks(dist, time.perf_counter() - t0, dev)
    elapsed_step_only = None
    if bucket is not None:  # the same K steps without the collective, for comparison (not `value`)
        barrier()
        t1 = time.perf_counter()
        for t in range(K):
            env.step(acts[(W + t) % n_act], autoreset=autoreset)
        barrier()
        elapsed_step_only = sharding.max_over_ranks(dist, time.perf_counter() - t1, dev)

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
    coll = None
    if bucket is not None:  # the collective alone, same stream, K repetitions
        ce = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for a, b in ce:
            a.record(stream)
            bucket.step()
            b.record(stream)
        torch.cuda.synchronize()
        coll = dict(ranks_seen=bucket.ranks_seen(),
                    backend=("nccl (RCCL)" if args.backend == "nccl" else args.backend) if world > 1 else "none (1 rank)",
                    allreduce_us=round(float(np.mean([a.elapsed_time(b) for a, b in ce])) * 1e3, 2),
                    overlapped=side is not None,
                    bucket_bytes=bucket.nbytes, per_step="ONE all-reduce of the gradient bucket with the KL estimate in its last slot "
                    "(multi_ppo._allreduce_grads), once per timed env step")

    if rank == 0:
        total_units = world * E * N * K
        value = total_units / elapsed
        B = algorithmic_bytes_per_drone_step(nm, nb, N)
        bytes_per_launch = E * N * B
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        traffic = fp64_flops = None
        traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        key = f"{N}x{E}" + (f"+{nb}b" if nb else "")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath)).get(key)
                if isinstance(rec, dict):
                    traffic, fp64_flops = rec.get("hbm_bytes"), rec.get("fp64_flops")
                    traffic_source = ("profiles/traffic.json[%s]: builder's rocprofv3 --pmc passes of this "
                                      "command (%s), not measured in this run" % (key, rec.get("source", "tools/pmc_record.sh")))
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
                       "envs_per_gpu": E, "drones": N, "launch": env.launch_info(),
                       "untimed_steps_before_warmup": args.prewarm, "diag_build": diag,
                       "ablate": int(os.environ.get("RVO3D_ABLATE", "0")) if diag else 0,
                       "device_error_word": flags},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "fp64_valu_tflops": (round(fp64_flops / (kern_ms * 1e-3) / 1e12, 3)
                                              if fp64_flops else None),
                         "fp64_valu_peak_tflops": 78.6,
                         "kernel": "rvo3d::env_kernel<%d, %d, %d, true> (fused step%s)" % (
                             2 if autoreset else 1, 1 if N <= 64 else 2 if N <= 128 else 4 if N <= 256 else 8,
                             N if N in (16, 32, 64, 128, 256) and env.launch_info()["envs_per_block"] == max(64 // N, 1) else 0,
                             " + auto-reset" if autoreset else ""),
                         "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "bytes_per_drone_step": B},
        }
        if coll is not None:
            coll["value_env_step_only"] = round(total_units / elapsed_step_only, 1)
            coll["ms_per_step_env_step_only"] = round(elapsed_step_only / K * 1e3, 4)
            out["collective"] = coll
            out["config"]["workload"] += ("; + gradient-bucket all-reduce and KL mean per step" +
                                          (" on a second stream" if side is not None else ""))
        if not args.no_cpu_baseline and world == 1:  # reported once, at N = 1
            out["cpu_baseline"] = cpu_baseline(N, nm, tuple(args.map))
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
