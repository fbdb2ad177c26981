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
update) - is put INSIDE the timed region whenever N > 1 (or with
--grad-allreduce): the trainer's own code path on the gradients of the MLP(256,256)
policy of BASELINE config 3 (0.74 MB fp32).  One collective per --allreduce-every env
steps (default 3: the training ratio with the reference's defaults - 300 env steps per epoch, then 50 + 50
optimizer steps; 1 = one per env step, the worst case), so the multi-GPU record contains real xGMI traffic
at the rate training produces it.
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


def usable_cpus():
    """CPUs this process can really use: min(affinity mask, cgroup CPU quota).  On a shared host the
    affinity mask shows every hardware thread while the container's quota (cpu.max, cgroup v2; cfs_quota
    for v1) allows far fewer: running OpenMP over the mask then oversubscribes the quota."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    n = aff if quota is None else max(1, min(aff, int(quota + 0.999)))
    return n, aff, quota


def cpu_baseline(N, nm, map_size, seconds_target=12.0):
    """The oracle (kind 'port') on this host, same generator, E scaled down: one thread (the
    scalar port) and every core this process may run on (OpenMP over envs)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc  # test infrastructure, used here only as the timed CPU baseline
    from rvo3d_amd import synthetic_actions, synthetic_world
    ncpu, affinity, quota = usable_cpus()
    recs = {}
    # 1 thread, the GPU box's per-GPU CPU share (16) and every cpu the process can really use
    # (min(affinity, cgroup quota)); the best is reported
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
            "affinity_cpus": affinity, "cgroup_cpu_quota": quota,
            "by_threads": {str(k): round(v["value"], 1) for k, v in sorted(recs.items())},
            "sample": f"{N} drones x {best['envs']} envs x {best['steps']} fused steps on {best['cores']} threads "
                      f"(oracle/rvo3d_oracle.c, OpenMP over envs); 1 thread: {N} drones x {one['envs']} envs x "
                      f"{one['steps']} steps = {one['value']:.0f}/s; this process can use {ncpu} cpus "
                      f"(affinity {affinity}, cgroup quota {quota})"}


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
        self.tr._bucket()[-1:].copy_(self.kl32)  # the KL slot (its value does not matter to the cost)
        self.ones = torch.ones(1, dtype=torch.float64, device=env.device)
        self.nbytes = (sum(p.numel() for p in self.ac.parameters()) + 1) * 4

    def step(self):
        # ONE collective: the flattened bucket averaged over the ranks, the KL estimate in its last slot, as
        # multi_ppo's policy pass issues it (multi_ppo._allreduce_grads; the mean KL is left on the device here:
        # the trainer's host read of it is not part of the collective's cost).  ONE call per step on RCCL
        # (ReduceOp.AVG); three launches from Python per step made the loop host-bound at one rank.
        if self.dist is None:
            return
        self.tr._allreduce_bucket()

    def ranks_seen(self):
        t = self.ones.clone()
        if self.dist is not None:
            self.dist.all_reduce(t)
        return int(round(float(t.item())))


def parse_args(argv=None):
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
    ap.add_argument("--allreduce-every", type=int, default=3,
                    help="env steps per gradient all-reduce when the collective is on.  Default 3 = the training ratio "
                         "with the reference's defaults (train_process.py:21-79: 300 env steps per epoch, then 50 policy "
                         "+ 50 value optimizer steps, one collective each); 1 = one per env step (worst case)")
    ap.add_argument("--serial-collective", action="store_true",
                    help="issue the collective on the env step's own stream (default: a second HIP stream, so "
                         "that it overlaps the next env step as it would overlap any other compute)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse the N > 1 code path where the ranks cannot have a GPU each)")
    ap.add_argument("--all-ranks-on-device", type=int, default=None,
                    help="rehearsal only: every rank uses this one GPU (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: initialise torch.distributed even with one rank (with --backend nccl this runs the "
                         "gradient-bucket collective through RCCL on a single GPU: library, device binding and call "
                         "sequence are the ones of an N-GPU run)")
    ap.add_argument("--no-cold", action="store_true",
                    help="skip the cache-cold kernel timing (roofline.frac_cold)")
    ap.add_argument("--cold-mb", type=int, default=1024,
                    help="bytes (MiB) of scratch streamed between two timed launches of the cold measurement "
                         "(>= 512: twice the 256 MiB Infinity Cache, so state and observation lines are evicted)")
    ap.add_argument("--cold-mode", default="rw", choices=["rw", "read"],
                    help="rw: the scratch is read and rewritten (the cache is left full of dirty lines, as after "
                         "the policy GEMMs of a rollout); read: only read (clean lines)")
    ap.add_argument("--cold-all", action="store_true",
                    help="profiling only: the scratch is streamed before EVERY launch (warm-up and timed loop included), "
                         "so that a rocprofv3 trace of the run shows cache-cold launches only; `value` then includes the "
                         "flush and is not the metric")
    ap.add_argument("--no-rollout", action="store_true",
                    help="skip the `rollout` block (BASELINE config 3 as written: policy forward -> env step -> "
                         "buffer stores, GAE, one PPO update), which runs after the headline timed region")
    ap.add_argument("--rollout-steps", type=int, default=16)
    ap.add_argument("--rollout-inline", action="store_true",
                    help="the rollout block in this process (default: a child process, so that nothing that happens "
                         "in the secondary measurement - a fault, a hang - can cost the headline line)")
    ap.add_argument("--rollout-only", action="store_true", help=argparse.SUPPRESS)  # the child's mode
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="seconds the launcher (--gpus N > 1 without WORLD_SIZE) waits for its ranks")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="N > 1 plumbing only (rendezvous, ranks_seen all-reduce, JSON relay, exit status); no env "
                         "step, no GPU: what the CPU test drives with --backend gloo")
    return ap.parse_args(argv)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): this process
    starts the N ranks itself - child processes of this very script with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, one per GPU - relays rank 0's JSON line and exits with the ranks' status.  It never
    touches the GPU (no HIP call; torch.cuda.device_count() does not initialise it) and replaces no
    process.  The line is refused (exit 3) unless it says n_gpus == N == the ranks an all-reduce saw
    (the reference's intent: train/policy/multi_ppo.py:179-181, 320-325)."""
    import subprocess
    N = args.gpus
    if not args.launcher_selftest and args.all_ranks_on_device is None:
        have = torch.cuda.device_count()
        if have < N:
            print(f"bench.py: --gpus {N} but this node shows {have} GPU(s): refusing to report fewer ranks "
                  "than asked for", file=sys.stderr)
            return 2
    env = dict(os.environ, WORLD_SIZE=str(N), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(N):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + args.launch_timeout
    rc, out0 = 0, ""
    try:
        # rank 0's pipe is drained by communicate(); the others are polled so that one failed rank
        # ends the job instead of leaving the rest in a collective
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.update(out=procs[0].communicate()[0]), daemon=True)
        th.start()
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                rc = next(c for c in codes if c not in (None, 0))
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                print("bench.py: the ranks did not finish in --launch-timeout seconds", file=sys.stderr)
                rc = 124
                break
            time.sleep(0.2)
    finally:
        for p in procs:  # exactly the processes started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
    th.join(timeout=20)
    out0 = box.get("out") or ""
    line = next((l for l in reversed(out0.splitlines()) if l.startswith("{")), None)
    if rc == 0 and line is None:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 3
    if line is not None:
        try:
            rec = json.loads(line)
            seen = (rec.get("collective") or {}).get("ranks_seen")
            if rc == 0 and not (rec.get("n_gpus") == N and seen == N):
                print(f"bench.py: asked for {N} ranks, the line says n_gpus {rec.get('n_gpus')}, "
                      f"ranks_seen {seen}", file=sys.stderr)
                rc = 3
            rec["launcher"] = "bench.py started the ranks itself (one child process per GPU)"
            line = json.dumps(rec)
        except ValueError:
            rc = rc or 3
        print(line, flush=True)
    return rc


def selftest_rank(args):
    """--launcher-selftest: the N > 1 plumbing without the env step (no GPU needed): rendezvous,
    ranks_seen from a real all-reduce, rank 0's JSON line."""
    from rvo3d_amd import sharding
    rank, local_rank, world = sharding.rank_info()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.backend == "nccl":
        raise SystemExit("--launcher-selftest runs on the CPU: use --backend gloo")
    dist = sharding.init_process_group(args.backend) if world > 1 else None
    seen = int(round(sharding.sum_over_ranks(dist, 1.0)))
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "drone-steps/sec (env.step throughput)", "value": None, "selftest": True,
                          "n_gpus": seen, "local_ranks": world,
                          "collective": {"ranks_seen": seen, "backend": args.backend}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if seen == args.gpus else 3


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    if args.launcher_selftest:
        return selftest_rank(args)
    if args.rollout_only:
        return rollout_child_main(args)
    return run_rank(args)


def rollout_child_main(args):
    """The child of rollout_in_child: builds its own env of the same shape and prints the rollout record."""
    from rvo3d_amd import BatchedDroneEnv, synthetic_world
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    wld = synthetic_world(args.envs, args.drones, tuple(args.map), nb=args.buildings, seed=1234)
    env = BatchedDroneEnv(wld, neighbors_num=args.nm, device=dev, action_decimals=2)
    print("ROLLOUT_RECORD " + json.dumps(rollout_block(env, args)), flush=True)
    return 0


def rollout_in_child(args):
    """rollout_block in a child process (started, not exec'ed; at most two processes on the GPU): its record, or an
    error record - the headline line is printed either way."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--rollout-only", "--envs", str(args.envs), "--drones", str(args.drones),
           "--nm", str(args.nm), "--buildings", str(args.buildings), "--map", *[str(x) for x in args.map],
           "--rollout-steps", str(args.rollout_steps)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("ROLLOUT_RECORD "):
                return json.loads(line[len("ROLLOUT_RECORD "):])
        return {"error": f"the rollout child ended with status {r.returncode} and no record", "stderr_tail": r.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"error": "the rollout child did not finish within 900 s"}
    except Exception as ex:
        return {"error": f"{type(ex).__name__}: {ex}"}


def run_rank(args):
    from rvo3d_amd import BatchedDroneEnv, _lib, sharding, synthetic_actions, synthetic_world

    # The product library has no diagnostics (no phase ablation, no stamps, reads no environment
    # variable); a diagnostics build can only get here through tools/bench_diag.py.
    diag = _lib.is_diag_build()
    if diag and not ALLOW_DIAG:
        raise SystemExit("bench.py refuses the diagnostics build of the library (use tools/bench_diag.py)")

    rank, local_rank, world = sharding.rank_info()
    if world != args.gpus:  # (a launcher-less --gpus N > 1 never gets here: main() starts the ranks)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    dev_index = local_rank if args.all_ranks_on_device is None else args.all_ranks_on_device
    if args.all_ranks_on_device is not None and args.backend == "nccl" and world > 1:
        raise SystemExit("--all-ranks-on-device needs --backend gloo (RCCL wants one GPU per rank)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1 or args.force_dist:
        # The bucket is 0.74 MB: latency-bound, a few channels carry it; every channel is a workgroup that holds wave
        # slots on a CU while the env launch wants ALL of them for its one round of 4096 waves (measured at one rank:
        # a collective kernel alongside costs the env step up to 24 %).  RCCL's own choice can be restored by setting
        # the variable in the environment.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "4")
        if world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
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
    with_coll = (world > 1 or args.grad_allreduce or args.force_dist) and not args.no_grad_allreduce
    bucket = GradBucket(env, dist) if with_coll else None
    torch.cuda.synchronize()

    # The collective depends on nothing the env step produces (and vice versa): it goes to its own
    # HIP stream and overlaps the following env step(s); both are inside the timed region.
    side = None
    if bucket is not None and not args.serial_collective:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))

    flush = None
    if args.cold_all:
        flush_buf = torch.zeros(max(args.cold_mb, 1) << 20, dtype=torch.uint8, device=dev)
        flush = (lambda: flush_buf.add_(1)) if args.cold_mode == "rw" else (lambda: flush_buf.sum())

    every = max(1, args.allreduce_every)

    def one_step(t):
        if flush is not None:
            flush()
        env.step(acts[t % n_act], autoreset=autoreset)
        if bucket is not None and t % every == 0:
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

    def timed_launches(between=None):
        for t in range(5):  # (untimed: the first launches behind a synchronisation carry its wake-up)
            if between is not None:
                between()
            env.step(acts[t % n_act], autoreset=autoreset)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for t, (a, b) in enumerate(evs):
            if between is not None:
                between()  # same stream, outside the event pair
            a.record(stream)
            env.step(acts[(W + t) % n_act], autoreset=autoreset)
            b.record(stream)
        torch.cuda.synchronize()
        ts = np.asarray([a.elapsed_time(b) for a, b in evs])
        # An event pair brackets the HOST's enqueue as well: when the process is descheduled between
        # a.record() and the launch (shared host), the GPU waits inside the pair and one sample reads
        # milliseconds.  Such samples (> 3 x the median) are not launch durations: they are counted and left
        # out of the mean (the rocprofv3 average of the same command, profiles/, has no such samples).
        med = float(np.median(ts))
        ok = ts <= 3.0 * med
        return float(ts[ok].mean()), med, float(ts.mean()), int((~ok).sum())

    kern_ms, kern_med_ms, kern_mean_all_ms, kern_dropped = timed_launches(flush)
    # cache-cold: back-to-back launches find the ~54 MB of state the previous launch wrote still in the
    # 256 MiB Infinity Cache; a trainer runs policy GEMMs over several hundred MB between two env steps.
    # Here a scratch buffer of --cold-mb MiB is read and rewritten between two timed launches.
    kern_cold_ms = None
    if not args.no_cold:
        scratch = torch.zeros(max(args.cold_mb, 1) << 20, dtype=torch.uint8, device=dev)
        kern_cold_ms = timed_launches((lambda: scratch.add_(1)) if args.cold_mode == "rw" else (lambda: scratch.sum()))[0]
        del scratch
    flags = env.error_flags()
    coll = None
    ranks_seen = 1
    if dist is not None:  # from a real all-reduce, whatever else is switched off
        ones = torch.ones(1, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(round(float(ones.item())))
    if bucket is not None:  # the collective alone, same stream, K repetitions
        ce = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for a, b in ce:
            a.record(stream)
            bucket.step()
            b.record(stream)
        torch.cuda.synchronize()
        coll = dict(ranks_seen=ranks_seen,
                    backend=("nccl (RCCL)" if args.backend == "nccl" else args.backend) if dist is not None else "none (1 rank)",
                    allreduce_us=round(float(np.mean([a.elapsed_time(b) for a, b in ce])) * 1e3, 2),
                    overlapped=side is not None,
                    bucket_bytes=bucket.nbytes, env_steps_per_allreduce=every,
                    per_step="ONE all-reduce of the gradient bucket with the KL estimate in its last slot "
                    f"(multi_ppo._allreduce_grads) per {every} timed env step(s)")
    elif world > 1:
        coll = dict(ranks_seen=ranks_seen, backend="nccl (RCCL)" if args.backend == "nccl" else args.backend,
                    per_step="none (--no-grad-allreduce)")

    rollout = None
    if not args.no_rollout and rank == 0 and world == 1:
        rollout = rollout_in_child(args) if not args.rollout_inline else rollout_block(env, args)

    status = 0
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
            "value": round(value, 1), "unit": "drone-steps/s", "n_gpus": ranks_seen, "steps": K,
            "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE config 3 env.step: {N} drones x {E} envs per GPU, "
                                   f"nm={nm}, {nb} buildings, map {args.map}, fused step"
                                   f"{'+auto-reset' if autoreset else ''}, f32 actions in HBM",
                       "envs_per_gpu": E, "drones": N, "launch": env.launch_info(),
                       "untimed_steps_before_warmup": args.prewarm, "diag_build": diag,
                       "cold_all": bool(args.cold_all),
                       "ablate": int(os.environ.get("RVO3D_ABLATE", "0")) if diag else 0,
                       "device_error_word": flags},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "fp64_valu_tflops": (round(fp64_flops / (kern_ms * 1e-3) / 1e12, 3)
                                              if fp64_flops else None),
                         "fp64_valu_peak_tflops": 78.6,
                         # the instantiation the library launches for this handle (rvo3d_kernel_name)
                         "kernel": env.kernel_name("step_autoreset" if autoreset else "step"),
                         "kernel_ms": round(kern_ms, 4),  # mean over the K event pairs without host-stall samples
                         "kernel_ms_median": round(kern_med_ms, 4), "kernel_ms_mean_all_samples": round(kern_mean_all_ms, 4),
                         "host_stall_samples_dropped": kern_dropped,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "bytes_per_drone_step": B},
        }
        if kern_cold_ms is not None:
            ach_cold = bytes_per_launch / (kern_cold_ms * 1e-3) / 1e9
            out["roofline"].update({
                "kernel_ms_cold": round(kern_cold_ms, 4), "achieved_cold": round(ach_cold, 2),
                "frac_cold": round(ach_cold / HBM_PEAK_GBS, 5),
                "cold": f"{args.cold_mb} MiB of scratch {'read and rewritten' if args.cold_mode == 'rw' else 'read'} on the launch stream between two "
                        "timed launches (outside the event pairs): nothing of the previous step is left in "
                        "the 256 MiB Infinity Cache - the regime of a rollout with policy GEMMs between env steps"})
        if coll is not None:
            if elapsed_step_only is not None:
                coll["value_env_step_only"] = round(total_units / elapsed_step_only, 1)
                coll["ms_per_step_env_step_only"] = round(elapsed_step_only / K * 1e3, 4)
                out["config"]["workload"] += (f"; + gradient-bucket all-reduce and KL mean every {every} step(s)" +
                                              (" on a second stream" if side is not None else ""))
            out["collective"] = coll
        if rollout is not None:
            out["rollout"] = rollout
        if not args.no_cpu_baseline and world == 1:  # reported once, at N = 1
            out["cpu_baseline"] = cpu_baseline(N, nm, tuple(args.map))
        if ranks_seen != args.gpus:
            out["error"] = f"--gpus {args.gpus} but the all-reduce saw {ranks_seen} rank(s)"
            status = 3
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return status


def policy_kernel_roofline(tr, env):
    """rvo3d_policy_mlp_sample (csrc/rvo3d_policy_mlp.hpp) on the rollout buffer's first slot, as the loop calls it (the
    env's rows with their vo_count: zero column groups skipped) and on dense random rows of the same shape.  Bound:
    MFMA.  Algorithmic flops = 2 x (W x 256 + 256 x 256 + 256 x out) per row and network (out = 3 / 1) - what the two
    MLPs need, not what the kernel issues (padded k-steps, 32-row head tiles)."""
    import ctypes as C
    from rvo3d_amd import _lib
    if tr._fused_mode() != "mlp":
        return None
    L = _lib.lib()
    E, N, W = env.E, env.N, env.W
    B = E * N
    mb = tr.ac.mlp_blob()
    dev = env.device
    act = torch.empty((B, 3), device=dev); logp = torch.empty(B, device=dev); val = torch.empty(B, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def timed(x, cnt):
        call = lambda i: _lib.check(L.rvo3d_policy_mlp_sample(p(mb["blob"]), W, p(x), x.stride(0), B, cnt, 12, 9,
                                                              1 if mb["tanh"] else 0, p(tr.ac.log_std), 1.0, 1, i, p(act),
                                                              p(logp), p(val), None, None, st), "rvo3d_policy_mlp_sample")
        for i in range(3):
            call(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            call(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e-3
    x_env, c_env = tr.buf.obs[0].view(B, W), tr.buf.cnt[0]
    t_env = timed(x_env, p(c_env))
    t_dense = timed(torch.randn((B, W), device=dev), None)
    flop = 2.0 * B * ((W * 256 + 256 * 256 + 256 * 3) + (W * 256 + 256 * 256 + 256 * 1))
    peak = 2500.0
    return {"kernel": f"rvo3d::policy_mlp_kernel<{(W + 16) // 16}, 8>", "bound": "mfma", "unit": "TFLOP/s", "peak": peak,
            "algorithmic_gflop_per_launch": round(flop / 1e9, 2),
            "rollout_rows": {"kernel_ms": round(t_env * 1e3, 4), "achieved": round(flop / t_env / 1e12, 1),
                             "frac": round(flop / t_env / 1e12 / peak, 4),
                             "rows_without_vo_rows": round(float((c_env == 0).float().mean()), 5)},
            "dense_random_rows": {"kernel_ms": round(t_dense * 1e3, 4), "achieved": round(flop / t_dense / 1e12, 1),
                                  "frac": round(flop / t_dense / 1e12 / peak, 4)}}


def rollout_block(env, args):
    """BASELINE config 3 AS WRITTEN - "full MA-PPO rollout + update, MLP(256,256) policy" - measured
    after the headline region, on the same env: T rollout steps of the trainer's own loop
    (rvo3d_amd.policy.multi_ppo.collect: policy forward in bf16 -> env step from the policy samples ->
    buffer stores; reference loop: train/policy/multi_ppo.py:183-281), one GAE scan and one clipped-PPO
    update of 2 + 2 optimizer iterations over the T * E * N samples (reference: :341-376).  Reported
    beside the headline, never as `value`."""
    from rvo3d_amd.policy import mlp_ac, multi_ppo
    T, E, N = args.rollout_steps, env.E, env.N
    dev = env.device
    try:
        ac = mlp_ac(env.W).to(dev)
        tr = multi_ppo(env, ac, train_epoch=0, steps_per_epoch=T, max_ep_len=500, train_pi_iters=2,
                       train_v_iters=2, target_kl=1e9, minibatch_size=E * N, save_freq=10 ** 9, amp=True)
        env.reset(); env.observe()
        tr.collect(); tr.buf.get()  # warm-up: allocator, first calls
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.collect()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        data = tr.buf.get()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tr.update(data)  # the first update of a process pays the allocator and the GEMM heuristics: not timed
        torch.cuda.synchronize(); t2b = time.perf_counter()
        tr.update(data)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        rec = {"workload": f"{N} drones x {E} envs, MLP(256,256) actor-critic, bf16 policy products, T = {T} steps; "
                           "update = 2 policy + 2 value iterations, minibatch E*N",
               "path": {"mlp": "fused: rvo3d_policy_mlp_sample (the whole policy step, one MFMA kernel) + "
                               "rvo3d_step_policy + rvo3d_rollout_account per step",
                        "heads": "fused: library GEMMs + rvo3d_policy_sample + rvo3d_step_policy + rvo3d_rollout_account per step",
                        "direct": "fused: the module's forward + rvo3d_policy_sample (direct) + rvo3d_step_policy + "
                                  "rvo3d_rollout_account per step"}.get(tr._fused_mode(), "module (PyTorch glue)"),
               "drone_steps_per_s": round(E * N * T / (t1 - t0), 1),
               "ms_per_step": round((t1 - t0) / T * 1e3, 4),
               "gae_ms": round((t2 - t1) * 1e3, 3), "update_s": round(t3 - t2b, 4),
               "update_first_call_s": round(t2b - t2, 4),
               "update_samples_per_s": round(E * N * T * 4 / (t3 - t2b), 1)}
        rec.update(tr.rollout_profile())  # env_kernel_us, launches_per_step (None when not measurable)
        try:  # the policy step's own kernel against the matrix-core roofline, HIP events around 20 launches
            rec["policy_kernel"] = policy_kernel_roofline(tr, env)
        except Exception as ex:
            rec["policy_kernel"] = {"error": f"{type(ex).__name__}: {ex}"}
        # the reference's own architecture (biGRU reader 9 -> 256, LayerNorm(268), actor / critic 268-256-256;
        # train/policy/policy_rnn_ac.py:31-257) through the same loop: rollout only, 8 steps
        try:
            del tr, data
            torch.cuda.empty_cache()
            from rvo3d_amd.policy import rnn_ac

            class _Space:
                shape = (3,)
            ac2 = rnn_ac(None, _Space(), 12, 9, 256, (256, 256), (256, 256), torch.nn.ReLU, torch.nn.Tanh,
                         torch.nn.Identity, use_gpu=False, rnn_mode="biGRU").to(dev)
            tr2 = multi_ppo(env, ac2, train_epoch=0, steps_per_epoch=8, max_ep_len=500, save_freq=10 ** 9, amp=True)
            env.reset(); env.observe()
            tr2.collect(); tr2.buf.ptr = 0; tr2.buf.cut.zero_()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tr2.collect()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            rec["reference_architecture"] = {"policy": "rnn_ac biGRU(9 -> 256) + LayerNorm(268) + MLP(256,256) heads, bf16",
                                             "path": tr2._fused_mode(), "ms_per_step": round((t1 - t0) / 8 * 1e3, 4),
                                             "drone_steps_per_s": round(E * N * 8 / (t1 - t0), 1)}
        except Exception as ex:
            rec["reference_architecture"] = {"error": f"{type(ex).__name__}: {ex}"}
        return rec
    except Exception as ex:  # the headline must not die with the secondary measurement
        return {"error": f"{type(ex).__name__}: {ex}"}


if __name__ == "__main__":
    sys.exit(main() or 0)
