"""Diagnostic: where does a wave spend its life?  Attaches the stamp buffer
(rvo3d_debug_stamps) and prints per-phase mean cycles over all workgroups."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd"))
from rvo3d_amd import BatchedDroneEnv, synthetic_actions, synthetic_world, _lib
E, N = 4096, 64
env = BatchedDroneEnv(synthetic_world(E, N, (50, 50, 10)), action_decimals=2)
acts = [torch.from_numpy(synthetic_actions(E, N, t).astype(np.float32)).cuda() for t in range(12)]
env.observe()
for t in range(10):
    env.step(acts[t], autoreset=True)
nb = env.launch_info()["blocks"]
buf = torch.zeros((nb, 16), dtype=torch.int64, device="cuda")
_lib.check(_lib.lib().rvo3d_debug_stamps(env._h, C.c_void_p(buf.data_ptr())), "stamps")
env.step(acts[10], autoreset=True)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.int64)
names = ["load+dronestate", "stage", "sweepA", "integrate", "lite/sweepB", "reset", "final sweep", "store", "zero-fill"]
d = np.diff(s[:, :10], axis=1)
print("phase mean / p50 / p95 cycles (s_memtime ticks):")
for i, n in enumerate(names):
    print(f"  {n:16s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {np.percentile(d[:, i], 95):9.0f}")

life = s[:, 9] - s[:, 0]
print("wave life mean", life.mean(), "start spread", s[:, 0].max() - s[:, 0].min(), "end-start", s[:, 9].max() - s[:, 0].min())
