"""Secondary measurement: kernel time (HIP events on the launch stream) of the entry points next to
the fused step at the benchmark shape - observe, plain step, step_policy (trainer glue fused in),
reset + observe, des_vel, the classical RVO velocity selection.
usage: python tools/bench_aux.py [--envs 4096 --drones 64]"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd"))
from rvo3d_amd import BatchedDroneEnv, synthetic_actions, synthetic_world

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--drones", type=int, default=64)
ap.add_argument("--reps", type=int, default=100)
args = ap.parse_args()
E, N = args.envs, args.drones
env = BatchedDroneEnv(synthetic_world(E, N, (50.0, 50.0, 10.0)), neighbors_num=10, action_decimals=2)
acts = [torch.from_numpy(synthetic_actions(E, N, t).astype(np.float32)).cuda() for t in range(16)]
env.observe()
for t in range(600):
    env.step(acts[t % 16], autoreset=True)


def timed(name, fn, bytes_per_drone=None):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    ts = np.asarray([a.elapsed_time(b) for a, b in ev])
    us = float(ts[ts <= 3 * np.median(ts)].mean()) * 1e3  # (without host-stall samples, see bench.py)
    extra = "" if bytes_per_drone is None else "  %.0f GB/s of its %d algorithmic bytes per drone" % (E * N * bytes_per_drone / us / 1e3, bytes_per_drone)
    print("%-44s %8.1f us  %.3e drones/s%s" % (name, us, E * N / (us * 1e-6), extra), flush=True)


W = env.W
timed("step + auto-reset (fused, f32 actions)", lambda i: env.step(acts[i % 16], autoreset=True), 719)
timed("step_policy + auto-reset (trainer glue)", lambda i: env.step_policy(acts[i % 16], autoreset=True), 719)
timed("observe (env_observation, action 0)", lambda i: env.observe(), 64 + 4 * W + 4 + 24)


def manual(i):  # the reference's protocol without the fused auto-reset: step, reset the ended drones, observe again
    _, _, _, done, _, fin = env.step(acts[i % 16])
    env.reset_drones(done | fin)
    env.observe()


timed("plain step + reset_drones(done | finish) + observe", manual)
mask = torch.zeros((E, N), dtype=torch.uint8, device="cuda"); mask[:, ::7] = 1
timed("reset_drones (1/7 of the drones) + observe", lambda i: (env.reset_drones(mask), env.observe()))
timed("des_vel (cal_des_list)", lambda i: env.des_vel(), 48 + 24)
timed("rvo_vel (classical RVO velocity selection)", lambda i: env.rvo_vel(vmax=(2.0, 2.0, 2.0), acceler=0.5))
flags = env.error_flags()
print("device error word", flags)
