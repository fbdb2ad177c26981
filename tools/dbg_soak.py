import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"),
                os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd")]
import numpy as np, torch
import oracle as orc
from rvo3d_amd import BatchedDroneEnv, synthetic_world, synthetic_actions
E, N, Tn = 64, 64, 60
mode = sys.argv[1] if len(sys.argv) > 1 else "cached"
w = synthetic_world(E, N, (50, 50, 10), seed=1234)
env = BatchedDroneEnv(w, neighbors_num=10, action_decimals=2)
ref = orc.OracleEnv(w.waypoints, w.n_points, w.map_size, w.buildings, nm=10, threads=8)
env.observe(); ref.observe()
for t in range(Tn):
    a = synthetic_actions(E, N, t, 1234)
    if mode == "nocache":
        env.set_state()
    obs, cnt, rew, done, info, fin = env.step(torch.from_numpy(a.astype(np.float32)).cuda(), autoreset=True)
    ro, rcnt, rr, rd, ri, rf, rm = ref.step_autoreset(a)
    o = obs.cpu().numpy()
    bad = ~np.isclose(o, ro, rtol=1e-5, atol=1e-6, equal_nan=True)
    bad &= (ref.margin() >= 1e-9)[:, :, None]
    if bad.any():
        idx = np.argwhere(bad.any(axis=-1))
        print(mode, "t", t, "mismatching drones", idx[:6].tolist(), "margin", ref.margin()[tuple(idx[0])])
        e, d = idx[0]
        print(" cnt hip", cnt.cpu().numpy()[e, d], "ref", rcnt[e, d], "reset_mask hip", env.reset_mask.cpu().numpy()[e].nonzero()[0].tolist(), "ref", rm[e].nonzero()[0].tolist())
        cols = np.nonzero(bad[e, d])[0]
        print(" cols", cols.tolist()); print(" hip", o[e, d, cols[:12]]); print(" ref", ro[e, d, cols[:12]])
        print(" hip row0", o[e, d, 12:21], "\n ref row0", ro[e, d, 12:21])
        print(" hip proprio", o[e, d, :12], "\n ref proprio", ro[e, d, :12])
        sh, sr = env.get_state(), ref.get_state()
        for k in ("pos", "vel", "max_dev", "wp_idx"):
            print(" state", k, sh[k].cpu().numpy()[e, d], sr[k][e, d])
        for k in ("vel", "yaw", "pitch", "real_len"):
            hv, rv = np.atleast_1d(sh[k].cpu().numpy()[e, d]), np.atleast_1d(sr[k][e, d])
            print(" hex", k, [float(x).hex() for x in hv], [float(x).hex() for x in rv])
        print(" action", a[e, d], "prev vel?", )
        break
else:
    print(mode, "no mismatch in", Tn, "steps")
