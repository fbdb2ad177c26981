"""Secondary measurement (BASELINE config 3 as written: rollout + PPO update with an
MLP(256,256) policy): drone-steps/s of the on-device rollout loop (policy forward ->
rvo3d_step_policy -> buffer store) and the time of one clipped-PPO update.
Not the bench.py headline (that is env.step alone)."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd"))
from rvo3d_amd import BatchedDroneEnv, synthetic_world
from rvo3d_amd.policy import mlp_ac, multi_ppo, rnn_ac

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--drones", type=int, default=64)
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--policy", default="mlp")
ap.add_argument("--minibatch", type=int, default=262144)
ap.add_argument("--amp", action="store_true")
ap.add_argument("--no-update", action="store_true", help="rollout only (for profiling the loop)")
ap.add_argument("--no-tune", action="store_true", help="hipBLASLt's heuristic kernels instead of TunableOp's pick "
                "(a rocprofv3 trace then holds no tuning runs; the GEMMs are ~15 %% slower)")
args = ap.parse_args()
E, N, T = args.envs, args.drones, args.steps
env = BatchedDroneEnv(synthetic_world(E, N, (50, 50, 10)))


class Space:
    shape = (3,)


ac = (mlp_ac(env.W) if args.policy == "mlp" else
      rnn_ac(None, Space(), 12, 9, 256, (256, 256), (256, 256), torch.nn.ReLU, torch.nn.Tanh,
             torch.nn.Identity, use_gpu=False, rnn_mode="biGRU")).cuda()
tr = multi_ppo(env, ac, train_epoch=0, steps_per_epoch=T, max_ep_len=500, train_pi_iters=2,
               train_v_iters=2, target_kl=1e9, minibatch_size=args.minibatch, save_freq=10**9, amp=args.amp, tune_gemms=not args.no_tune)
env.reset(); env.observe()
tr.collect(); tr.buf.get()            # warm-up (allocator, hipBLASLt heuristics)
torch.cuda.synchronize(); t0 = time.perf_counter()
tr.collect()
torch.cuda.synchronize(); t1 = time.perf_counter()
data = tr.buf.get()
torch.cuda.synchronize(); t2 = time.perf_counter()
if not args.no_update:
    tr.update(data)
torch.cuda.synchronize(); t3 = time.perf_counter()
print(json.dumps({"policy": args.policy, "amp": args.amp, "envs": E, "drones": N, "steps": T,
                  "rollout_drone_steps_per_s": E * N * T / (t1 - t0),
                  "rollout_ms_per_step": (t1 - t0) / T * 1e3, "gae_ms": (t2 - t1) * 1e3,
                  "update_s": t3 - t2,
                  "update_samples_per_s": E * N * T * 4 / (t3 - t2)}))
