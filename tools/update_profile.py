#!/usr/bin/env python3
"""usage: tools/update_profile.py [--no-split]
torch.profiler table of ONE PPO update of the bench line's shape (64 drones x 4096 envs, T = 16, MLP(256,256),
2 + 2 passes in minibatches of E * N: 64 optimizer steps, float32) on the GPU box.  --no-split: nn.Linear's own
weight-gradient GEMMs instead of the row-sliced ones (policy_rnn_ac._Linear) - the trace DESIGN.md section 6 starts from."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd"))
from rvo3d_amd import BatchedDroneEnv, synthetic_world  # noqa: E402
from rvo3d_amd.policy import mlp_ac, multi_ppo, policy_rnn_ac  # noqa: E402

if "--no-split" in sys.argv:
    policy_rnn_ac._Linear.split_rows = 1 << 62
E_, N_ = 4096, 64
env = BatchedDroneEnv(synthetic_world(E_, N_, (50, 50, 10)))
ac = mlp_ac(env.W).cuda()
tr = multi_ppo(env, ac, steps_per_epoch=16, max_ep_len=500, amp=True, train_pi_iters=2, train_v_iters=2, target_kl=1e9,
               minibatch_size=E_ * N_)
env.reset(); env.observe()
tr.collect(); data = tr.buf.get()
tr.update(data); torch.cuda.synchronize()
t0 = time.perf_counter(); tr.update(data); torch.cuda.synchronize()
print(f"update: {time.perf_counter() - t0:.4f} s for 64 optimizer steps of {E_ * N_} samples"
      f" ({'nn.Linear weight gradients' if '--no-split' in sys.argv else 'row-sliced weight gradients'})")
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    tr.update(data); torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
ev.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in ev)
print(f"GPU time {tot * 1e-3:.1f} ms in {sum(e.count for e in ev)} launches; top kernels:")
for e in ev[:16]:
    print(f"  {e.device_time_total * 1e-3:8.2f} ms {100 * e.device_time_total / tot:5.1f} %  {e.count:4d} x {e.device_time_total / e.count:8.1f} us  {e.key[:110]}")
