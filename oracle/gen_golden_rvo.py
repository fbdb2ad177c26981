#!/usr/bin/env python3
"""Golden vectors for the classical RVO velocity selection (tests/golden/rvo_vel.npz).

The reference class uaisa_env/vel_obs/reciprocal_vel_obs.py cannot run (see the header of
orc_rvo_vel in rvo3d_oracle.c), so the DRIVER LOOP below is the algorithm that file spells
out, written here once more in Python; every arithmetic building block it calls is the
reference's own function, imported from /root/reference (vel_obs3D.get_alpha, get_PAA,
get_rvo_array, get_beta, cal_exp_tim).  The vectors therefore pin the helpers and the
numpy arange / clip / min semantics, not the (non-running) reference loop: parity of the
loop itself stays "unpinned" and is documented so.

Round 3 - what DOES run on an instance of the reference class is pinned call by call
(tests/golden/rvo_calls.npz): `reciprocal_vel_obs.preprocess` (:32-54, incl. this class's own
building gate h > z - 1, <= 10), `.penalty` (:126-147), `.distance` (:149-151), `.vel_candidate`
with an empty VO list (:85-101: the candidate grid), `.vel_select` with an empty vo_outside
(:119-124: the only branch of it that returns a velocity).  `config_vo` (:63-83) raises ValueError
(state[0:4] has four values, get_PAA unpacks three), `vel_candidate` with a non-empty VO list raises
AttributeError (vo_out2, :109) and `vel_select` with a non-empty vo_outside returns None: recorded
as such in the file (`raises_*`), and the loop around them stays "parity unpinned".

    python oracle/gen_golden_rvo.py        # needs /root/reference (this container only)
"""
import os
import sys
from math import sqrt

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
from uaisa_env.vel_obs.vel_obs3D import (get_alpha, get_PAA, get_rvo_array, get_beta,  # noqa: E402
                                         cal_exp_tim)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def distance(p1, p2):  # reciprocal_vel_obs.distance (:150-152)
    return sqrt((p2[0] - p1[0]) ** 2 + (p2[1] - p1[1]) ** 2 + (p2[2] - p1[2]) ** 2)


def cal_vel(i, P, V, R, PR, DES, vmax, acceler, delta_t=1):
    """cal_vel (:19-31) for agent i of one env, as intended."""
    Pa, Va, ra, pra = list(P[i]), list(V[i]), R[i], PR[i]
    odro = [j for j in range(len(P)) if j != i and np.linalg.norm(np.array(Pa) - np.array(P[j])) <= 10]
    vo_list = []
    for j in odro:  # config_vo (:57-76)
        Pb, Vb = list(P[j]), list(V[j])
        try:
            alpha = get_alpha(Pa, Pb, ra, R[j])
        except ValueError:  # asin domain: overlapping spheres (the reference raises)
            alpha = 1.57
        vo_list.append(get_PAA(Pa, pra, PR[j], Va, Vb) + get_rvo_array(Pa, Pb)
                       + [alpha, cal_exp_tim(Pa, Pb, Va, Vb, ra, R[j])])
    rng = [np.clip([Va[k] - acceler, Va[k] + acceler], -vmax[k], vmax[k]) for k in range(3)]
    outside, inside = [], []
    for vx in np.arange(rng[0][0], rng[0][1], 0.5):  # vel_candidate (:79-101)
        for vy in np.arange(rng[1][0], rng[1][1], 0.5):
            for vz in np.arange(rng[2][0], rng[2][1], 0.5):
                if sqrt(vx ** 2 + vy ** 2 + vz ** 2) < 0.3:
                    continue
                pn = [Pa[0] + vx * delta_t, Pa[1] + vy * delta_t, Pa[2] + vz * delta_t]
                col = False
                for vo in vo_list:  # vo_out2 (:103-117)
                    w = [pn[k] - vo[k] for k in range(3)]
                    if vo[6] > get_beta(vo[3:6], w):
                        col = True
                (inside if col else outside).append([vx, vy, vz])
    des = list(DES[i])
    if outside:  # vel_select (:119-124)
        return min(outside, key=lambda v: distance(v, des))
    if inside:
        tc_min = min(vo[7] for vo in vo_list)  # penalty (:126-147)
        tc_inv = float("inf") if tc_min == 0 else 1 / tc_min
        return min(inside, key=lambda v: 1 * tc_inv + distance(des, v))
    return [0.0, 0.0, 0.0]


def call_vectors():
    """Return values of the reference class's own methods, for seeded arguments."""
    from uaisa_env.vel_obs.reciprocal_vel_obs import reciprocal_vel_obs
    rng = np.random.default_rng(77)
    out = {}

    def state(p, v, r=0.2, prio=5.0, des=(0.0, 0.0, 0.0)):
        return [float(x) for x in (*p, *v, r, prio, *des, 0.0)]

    # -- preprocess: neighbours around the 10 m gate (exact 10: 6-8-0 / 0-6-8 triangles), buildings
    #    around h = z - 1 and around 10 m in the plane
    inst = reciprocal_vel_obs()
    P = {"agent": [], "drones": [], "blds": [], "keep_d": [], "keep_b": []}
    for c in range(60):
        a = np.round(rng.uniform(2, 30, 3), 2)
        n, nb = 12, 8
        d = np.round(a + rng.normal(0, 6.5, (n, 3)), 2)
        d[0] = a + np.array([6.0, 8.0, 0.0]); d[1] = a + np.array([0.0, -6.0, 8.0])   # norm == 10 exactly
        d[2] = a + np.array([6.0, 8.0, 0.01]); d[3] = a                               # just outside; the same spot
        b = np.column_stack([np.round(a[0] + rng.normal(0, 7.5, nb), 2), np.round(a[1] + rng.normal(0, 7.5, nb), 2),
                             np.round(a[2] - 1 + rng.normal(0, 1.0, nb), 2), np.round(rng.uniform(0.5, 1.5, nb), 2)])
        b[0, :3] = [a[0] + 6.0, a[1] + 8.0, a[2] - 1.0]      # on the 10 m circle, h == z - 1 (not >)
        b[1, :3] = [a[0] + 6.0, a[1] - 8.0, a[2] - 0.99]     # on the circle, just high enough
        dl = [state(x, (0, 0, 0)) for x in d]
        bl = [list(map(float, x)) for x in b]
        od, ob = inst.preprocess(state(a, (0, 0, 0)), dl, bl)
        P["agent"].append(a); P["drones"].append(d); P["blds"].append(b)
        P["keep_d"].append([any(x is y for y in od) for x in dl])
        P["keep_b"].append([any(x is y for y in ob) for x in bl])
    out.update({"pre_" + k: np.asarray(v) for k, v in P.items()})

    # -- penalty / distance / vel_select(inside only): random close encounters, overlaps (tc = 0 ->
    #    inf), receding pairs (tc = inf -> 1/inf = 0)
    Q = {"agent": [], "odro": [], "n": [], "vel": [], "des": [], "pen": [], "dist": [], "inside": [],
         "n_in": [], "sel": []}
    for c in range(200):
        a = np.round(rng.uniform(2, 20, 3), 2)
        va = np.round(rng.uniform(-1.5, 1.5, 3), 2)
        des = np.round(rng.uniform(-1, 1, 3), 3)
        n = int(rng.integers(1, 6))
        od = np.zeros((5, 8))
        for j in range(n):
            sep = [0.3, 1.0, 3.0, 8.0][int(rng.integers(4))]
            od[j, :3] = np.round(a + rng.normal(0, sep, 3), 2)
            od[j, 3:6] = np.round(rng.uniform(-1.5, 1.5, 3), 2)
            od[j, 6], od[j, 7] = [0.2, 0.3, 0.5][int(rng.integers(3))], float(rng.integers(1, 9))
        ag = state(a, va, 0.2, 5.0, des)
        ol = [list(map(float, x)) for x in od[:n]]
        v = np.round(rng.uniform(-2, 2, 3) * 2) / 2
        Q["pen"].append(inst.penalty(list(v), list(des), ag, ol, 1))
        Q["dist"].append(reciprocal_vel_obs.distance(list(v), list(des)))
        n_in = int(rng.integers(1, 9))
        ins = np.zeros((8, 3))
        ins[:n_in] = np.round(rng.uniform(-2, 2, (n_in, 3)) * 2) / 2
        if n_in > 2 and c % 3 == 0:
            ins[n_in - 1] = ins[0]                                   # equal keys: the first one wins
        got = inst.vel_select(ag, [], [list(x) for x in ins[:n_in]], ol)
        sel = next(k for k in range(n_in) if list(ins[k]) == got)
        for k, x in (("agent", ag[:11]), ("odro", od), ("n", n), ("vel", v), ("des", des), ("inside", ins),
                     ("n_in", n_in), ("sel", sel)):
            Q[k].append(x)
    out.update({"pen_" + k: np.asarray(v) for k, v in Q.items()})
    assert np.isinf(out["pen_pen"]).any() and (np.asarray(Q["n_in"]) > 1).any()

    # -- vel_candidate with an empty VO list = the candidate grid, for several (vmax, acceler)
    G = {"vel": [], "vmax": [], "acc": [], "n": [], "cand": []}
    for c in range(120):
        vmax = [(2, 2, 2), (1, 1, 1), (1.5, 2, 1), (0.7, 2.2, 0.3)][c % 4]
        acc = [0.5, 1.0, 0.75, 0.3][(c // 4) % 4]
        inst2 = reciprocal_vel_obs(vxmax=vmax[0], vymax=vmax[1], vzmax=vmax[2], acceler=acc)
        v = np.round(rng.uniform(-2.3, 2.3, 3), 2)
        if c % 5 == 0:
            v = np.round(v * 2) / 2            # grid-aligned: arange end points, |v| == 0.5 exactly ...
        if c % 7 == 0:
            v[:] = 0.0                         # around the |v| < 0.3 hole
        outside, inside = inst2.vel_candidate(state((5, 5, 5), v), [])
        assert inside == []
        cand = np.zeros((125, 3)); cand[:len(outside)] = np.asarray(outside, dtype=np.float64).reshape(-1, 3)
        for k, x in (("vel", v), ("vmax", np.asarray(vmax, float)), ("acc", acc), ("n", len(outside)), ("cand", cand)):
            G[k].append(x)
    out.update({"cand_" + k: np.asarray(v) for k, v in G.items()})

    # -- what does not run: recorded, so that the claim is checked by whoever regenerates
    a, b = state((0, 0, 0), (1, 0, 0)), state((5, 0, 0), (-1, 0, 0))
    for name, fn in (("config_vo", lambda: inst.config_vo(a, b)),
                     ("vel_candidate_with_vo", lambda: inst.vel_candidate(a, [[0, 0, 0, 5, 0, 0, 0.08, 2.0]])),
                     ("cal_vel", lambda: inst.cal_vel(a, [b], []))):
        try:
            fn()
            out["raises_" + name] = np.array("")
        except Exception as ex:  # noqa: BLE001
            out["raises_" + name] = np.array(type(ex).__name__)
    out["vel_select_outside_returns_none"] = np.array(inst.vel_select(a, [[1, 0, 0]], [], [b]) is None)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rvo_calls.npz"), **out)
    print("rvo_calls.npz:", {k: v.shape for k, v in out.items() if v.ndim}, {k: str(v) for k, v in out.items() if not v.ndim})


def main():
    call_vectors()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd"))
    import oracle as orc
    from rvo3d_amd import synthetic_world
    rng = np.random.default_rng(2024)
    cases = []
    for (E, N, L, acc, vmax) in [(4, 12, 8.0, 0.5, (2, 2, 2)), (3, 16, 6.0, 1.0, (2, 2, 2)),
                                 (3, 8, 5.0, 0.5, (1, 1, 1)), (2, 24, 7.0, 0.75, (1.5, 2, 1))]:
        w = synthetic_world(E, N, (L, L, L), seed=int(rng.integers(1 << 30)), min_sep=0.5)
        env = orc.OracleEnv(w.waypoints, w.n_points, w.map_size, w.buildings, nm=10)
        pos = np.round(w.waypoints[:, :, 0] + rng.uniform(-0.3, 0.3, (E, N, 3)), 2)
        vel = np.round(rng.uniform(-1.2, 1.2, (E, N, 3)), 2)
        vel[rng.random((E, N)) < 0.15] = 0.0
        env.set_state(pos=pos, vel=vel)
        des = env.des_vel()
        out = np.zeros((E, N, 3))
        for e in range(E):
            R = np.full(N, 0.2); PR = np.full(N, 5.0)
            for i in range(N):
                out[e, i] = cal_vel(i, pos[e], vel[e], R, PR, des[e], vmax, acc)
        cases.append(dict(waypoints=w.waypoints, n_points=w.n_points, map_size=w.map_size,
                          pos=pos, vel=vel, des=des, out=out, acceler=acc, vmax=np.array(vmax, float)))
        inside_any = int((np.abs(out - des).sum(-1) > 1e-12).sum())
        print(f"E={E} N={N}: {E*N} agents, {inside_any} selections differ from des_vel")
    flat = {}
    for k, c in enumerate(cases):
        for name, v in c.items():
            flat[f"c{k}_{name}"] = np.asarray(v)
    flat["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rvo_vel.npz"), **flat)


if __name__ == "__main__":
    main()
