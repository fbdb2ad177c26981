"""Per-kernel GPU time of the rollout loop (BASELINE config 3 as written: MLP(256,256), bf16), from
torch.profiler: which launches make up one rollout step and what each costs.
usage (GPU box): python tools/rollout_profile.py [--steps 8] [--fp32] [--module-path]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd"))
from rvo3d_amd import BatchedDroneEnv, synthetic_world
from rvo3d_amd.policy import mlp_ac, multi_ppo
from torch.profiler import ProfilerActivity, profile

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--drones", type=int, default=64)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--fp32", action="store_true")
ap.add_argument("--policy", default="mlp", help="mlp | rnn (the reference's biGRU actor-critic, 256 hidden)")
ap.add_argument("--chunk", type=int, default=0, help="rows per policy pass (0: all at once)")
ap.add_argument("--tunable", action="store_true", help="torch.cuda.tunable: let TunableOp pick the hipBLASLt solution per GEMM shape")
ap.add_argument("--module-path", action="store_true", help="the unfused loop (PyTorch glue)")
args = ap.parse_args()
if args.tunable:
    import torch.cuda.tunable as tun
    tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(2000); tun.set_filename(os.path.join(ROOT, 'gpurun_out', 'tunableop_results.csv'))
env = BatchedDroneEnv(synthetic_world(args.envs, args.drones, (50, 50, 10)))
if args.policy == "mlp":
    ac = mlp_ac(env.W).cuda()
else:
    from rvo3d_amd.policy import rnn_ac

    class Space:
        shape = (3,)
    ac = rnn_ac(None, Space(), 12, 9, 256, (256, 256), (256, 256), torch.nn.ReLU, torch.nn.Tanh, torch.nn.Identity,
                use_gpu=False, rnn_mode="biGRU").cuda()
tr = multi_ppo(env, ac, steps_per_epoch=args.steps, max_ep_len=500, amp=not args.fp32,
               fused_rollout=not args.module_path, rollout_chunk=args.chunk,
               graph_rollout=False)   # (eager launches: the profiler lists the kernels of a step)
env.reset(); env.observe()
tr.collect(final_reset=False); tr.buf.ptr = 0
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    tr.collect(final_reset=False)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
tot = {}
for e in ev:
    k = e.name[:110]
    t = tot.setdefault(k, [0, 0.0])
    t[0] += 1; t[1] += e.device_time
allt = sum(v[1] for v in tot.values())
print(f"{len(ev) / args.steps:.1f} launches and {allt / args.steps:.1f} us of GPU time per rollout step "
      f"({args.drones} x {args.envs}, {'fp32' if args.fp32 else 'bf16'}, {'module' if args.module_path else 'fused'} path)")
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / args.steps:9.1f} us/step  {n / args.steps:5.2f} x {t / n:8.1f} us  {k}")
