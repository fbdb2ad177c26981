import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd"), os.path.join(ROOT, "oracle")]
import oracle as orc
from rvo3d_amd import BatchedDroneEnv, synthetic_world
world = synthetic_world(128, 16, (20, 20, 8), seed=11)
E, N, _ = world.shape
env = BatchedDroneEnv(world)
ref = orc.OracleEnv(world.waypoints, world.n_points, world.map_size, world.buildings, threads=8)
env.observe(); ref.observe()
rng = np.random.default_rng(3)
np.set_printoptions(precision=17)
for t in range(40):
    a_inc = rng.normal(0, 0.6, (E, N, 3)).clip(-1, 1).astype(np.float32)
    vel = ref.get_state()["vel"]; gvel = env.get_state()["vel"].cpu().numpy()
    abs_action = np.round(env.acceler * np.round(a_inc, 2) + vel, 2)
    gabs = np.round(env.acceler * np.round(a_inc, 2) + gvel, 2)
    obs, cnt, rew, done, info, fin = env.step_policy(torch.from_numpy(a_inc).cuda(), autoreset=True)
    ro, rcnt, rr, rd, ri, rf, rm = ref.step_autoreset(abs_action)
    bad = np.argwhere(rew.cpu().numpy() != rr.astype(np.float32))
    bad = [b for b in bad if not (np.isnan(rr[tuple(b)]) )]
    if len(bad):
        for b in bad[:3]:
            b = tuple(b)
            print("t", t, b, "gpu", rew.cpu().numpy()[b], "ref", rr[b], "margin", ref.margin_sites()[0][b], ref.margin_sites()[1][b])
            print("  a_inc", a_inc[b], "vel ref", vel[b], "vel gpu", gvel[b], "abs ref", abs_action[b], "abs from gpu vel", gabs[b])
        break
