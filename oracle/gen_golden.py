#!/usr/bin/env python3
"""Generate golden vectors by running the PYTHON REFERENCE in this container.

Test infrastructure.  Runs only where /root/reference exists (the build
container); the resulting tests/golden/*.npz are committed and are the only
thing that travels.  Nothing here is imported by the product.

The reference imports `gym` (mdin.py:2, ir_gym.py:11) and `imageio`
(env_plot.py:5), neither installed here; both are replaced by inert in-memory
stand-ins (a base class and a Box record - no numerics), and the hard-coded
matplotlib plotter (env_base.py:21 `plot = True`) by a no-op object.

Each scenario = a world (data_1.json content), a per-step action list, a reset
schedule, and per-step reference outputs:
  obs (padded to 12+9*nm), vo_count, reward, done, info, finish, post-step
  state, the reset mask the caller applied, and the observations recomputed
  after those resets (ir_gym.env_observation, action = 0).

Hygiene: every scenario is replayed through the C oracle twice - the exact
arithmetic model and the ulp-perturbed variant (ORC_VARIANT: x*x for pow(x,2),
unfused dot products).  A scenario is kept only if the reference, the oracle
and the variant agree on every flag and every rounded float, i.e. no decision
in it sits within an ulp of its threshold.

Usage:  python oracle/gen_golden.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import copy
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
NM = 10


# --------------------------------------------------------------------------
# reference harness
# --------------------------------------------------------------------------
def _install_standins():
    os.environ.setdefault("MPLBACKEND", "Agg")
    gym = types.ModuleType("gym")

    class Env:  # gym.Env stand-in: mdin only subclasses it
        pass

    class Box:  # gym.spaces.Box stand-in: ir_gym only stores it
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.dtype = low, high, dtype
            self.shape = shape if shape is not None else np.shape(low)

    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    gym.Env, gym.spaces = Env, spaces
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    sys.modules["imageio"] = types.ModuleType("imageio")
    if REF not in sys.path:
        sys.path.insert(0, REF)


class _NoPlot:
    def __init__(self, *a, **k):
        pass

    def __deepcopy__(self, memo):
        return self

    def __getattr__(self, name):
        return lambda *a, **k: None


def make_reference_env(world: dict, nm: int = NM, **kw):
    """Write `world` as data_1.json into a scratch dir and build mdin on it."""
    _install_standins()
    import uaisa_env.drone_envs.env_base as eb
    eb.env_plot = _NoPlot
    from uaisa_env.drone_envs.mdin import mdin

    d = tempfile.mkdtemp(prefix="rvo3d_world_")
    try:
        with open(os.path.join(d, "data_1.json"), "w") as f:
            json.dump(world, f)
        np.save(os.path.join(d, "E3d.npy"), np.zeros((1, 1, 1)))       # loaded, never used
        np.save(os.path.join(d, "E3d_safe.npy"), np.zeros((1, 1, 1)))  # (env_base.py:42-47)
        env = mdin(neighbors_num=nm, base_dir=d, **kw)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return env


def ref_state(env):
    dl = env.ir_gym.drone_list
    f = lambda g, dt=np.float64: np.array([g(d) for d in dl], dtype=dt)
    return dict(
        pos=f(lambda d: np.asarray(d.state, dtype=np.float64)),
        vel=f(lambda d: np.asarray(d.vel, dtype=np.float64)),
        yaw=f(lambda d: float(d.yaw)), pitch=f(lambda d: float(d.pitch)),
        real_len=f(lambda d: float(d.real_route_len)),
        max_dev=f(lambda d: float(np.squeeze(d.max_deviation))),
        extra_len=f(lambda d: float(d.extra_len)),
        wp_idx=f(lambda d: d.i, np.int32),
        arrive=f(lambda d: bool(d.arrive_flag), np.uint8),
        dest=f(lambda d: bool(d.destination_arrive_flag), np.uint8))


def pad_obs(obs_list, nm=NM):
    W = 12 + 9 * nm
    out = np.zeros((len(obs_list), W))
    cnt = np.zeros(len(obs_list), np.int32)
    for i, o in enumerate(obs_list):
        o = np.asarray(o, dtype=np.float64)
        k = (len(o) - 12) // 9
        o = o[:W]  # nm == 0: the reference still appends one zero row (ir_gym.py:211-214)
        out[i, :len(o)] = o
        # k == 1 with an all-zero row means "no VO" (ir_gym.py:211-214)
        cnt[i] = 0 if (k == 1 and not np.any(o[12:21])) else k
    return out, cnt


# --------------------------------------------------------------------------
# worlds
# --------------------------------------------------------------------------
def load_ref_world(name):
    with open(os.path.join(REF, "uaisa_env", "world", name, "data_1.json")) as f:
        return json.load(f)


def synth_world(rng, N, map_size, n_points=2, nb=0, min_sep=1.0):
    L, Wd, H = map_size
    starts = []
    while len(starts) < N:
        s = np.round([rng.uniform(1, L - 1), rng.uniform(1, Wd - 1), rng.uniform(1, H - 1)], 2)
        if all(np.linalg.norm(s - t) >= min_sep for t in starts):
            starts.append(s)
    wps = []
    for s in starts:
        pts = [s.tolist()]
        for _ in range(n_points - 1):
            pts.append(np.round([rng.uniform(1, L - 1), rng.uniform(1, Wd - 1),
                                 rng.uniform(1, H - 1)], 2).tolist())
        wps.append(pts)
    blds = [np.round([rng.uniform(2, L - 2), rng.uniform(2, Wd - 2), rng.uniform(3, H),
                      rng.uniform(0.5, 1.5)], 2).tolist() for _ in range(nb)]
    return dict(drone_num=N, map_size=list(map_size), waypoints_list=wps,
                n_points_list=[n_points] * N, building_list=blds)


# --------------------------------------------------------------------------
# action generators (all return list[N] of float64[3], 2-decimal values)
# --------------------------------------------------------------------------
def act_random(env, rng):
    n = env.ir_gym.drone_num
    return [np.round(rng.uniform(-1, 1, 3) * np.array([1, 0.3, 0.15]), 2) for _ in range(n)]


def act_desvel(env, rng):
    # uaisa_env/gym_env_test.py:12 feeds cal_des_list() (3-decimal values)
    return [np.asarray(v, dtype=np.float64) for v in env.ir_gym.cal_des_list()]


def act_trainer(env, rng):
    # multi_ppo.py:196-208: abs = round(acceler * round(a, 2) + cur_vel, 2)
    out = []
    for d in env.ir_gym.drone_list:
        a = np.round(rng.normal(0, 0.6, 3).clip(-1, 1).astype(np.float32), 2)
        out.append(np.round(env.ir_gym.acceler * a + np.squeeze(d.vel), 2))
    return out


def act_follow(env, rng):
    # steer (acc, yaw-rate, pitch-rate) toward the current waypoint, with noise
    out = []
    for d in env.ir_gym.drone_list:
        dif = np.asarray(d.current_des, float) - np.asarray(d.state, float)
        want_yaw = math.degrees(math.atan2(dif[1], dif[0]))
        want_pit = math.degrees(math.atan2(dif[2], math.hypot(dif[0], dif[1])))
        dy = (want_yaw - d.yaw + 180.0) % 360.0 - 180.0
        dp = want_pit - d.pitch
        speed = float(np.linalg.norm(d.vel))
        dist = float(np.linalg.norm(dif))
        acc = np.clip(min(1.0, dist * 0.6) - speed, -1, 1)
        a = np.array([acc, dy / 90.0, dp / 90.0]) + rng.normal(0, 0.03, 3)
        out.append(np.round(np.clip(a, -1, 1), 2))
    return out


ACTORS = dict(random=act_random, desvel=act_desvel, trainer=act_trainer, follow=act_follow)


# --------------------------------------------------------------------------
# scenario runner
# --------------------------------------------------------------------------
def scatter_state(env, rng, spread):
    """Overwrite every drone with a random dense state (tests only: reaches VO
    geometry that route-following rarely visits: k > nm, t = -1, collisions)."""
    ms = np.asarray(env.ir_gym.map_size, dtype=np.float64)
    c = ms / 2
    N = env.ir_gym.drone_num
    pos = c + rng.uniform(-1, 1, (N, 3)) * np.minimum(spread, ms / 2 - 0.3)
    yaw = rng.uniform(0, 360, N)
    pitch = rng.uniform(-60, 60, N)
    speed = rng.uniform(0.0, 2.0, N)
    for i, d in enumerate(env.ir_gym.drone_list):
        d.state = pos[i].copy()
        d.yaw, d.pitch = float(yaw[i]), float(pitch[i])
        yr, pr = np.deg2rad(yaw[i]), np.deg2rad(pitch[i])
        d.vel = speed[i] * np.array([np.cos(pr) * np.cos(yr), np.cos(pr) * np.sin(yr), np.sin(pr)])
    return dict(pos=pos, vel=np.array([d.vel for d in env.ir_gym.drone_list]), yaw=yaw, pitch=pitch)


def act_velocity_like(env, rng):
    # candidate velocities near the current velocity or aimed at a neighbour
    dl = env.ir_gym.drone_list
    out = []
    for i, d in enumerate(dl):
        if rng.random() < 0.5:
            j = int(rng.integers(len(dl) - 1)); j += j >= i
            dirv = np.asarray(dl[j].state, float) - np.asarray(d.state, float)
            dirv = dirv / (np.linalg.norm(dirv) + 1e-9) * rng.uniform(0.2, 1.5)
            a = dirv + rng.normal(0, 0.1, 3)
        else:
            a = np.squeeze(d.vel) + rng.normal(0, 0.3, 3)
        out.append(np.round(a, 2))
    return out


def act_vslow(env, rng):
    # small yaw/pitch commands: heading barely changes, so the candidate keeps
    # pointing along +-x while the post-move velocity stays on its old heading
    n = env.ir_gym.drone_num
    return [np.round(np.array([rng.uniform(-1, 1), rng.normal(0, 0.05), rng.normal(0, 0.05)]), 2)
            for _ in range(n)]


ACTORS["vlike"] = act_velocity_like
ACTORS["vslow"] = act_vslow


def run_scenario(world, actor, T, seed, reset_on_finish=True, nm=NM, env_kw=None, scatter=None,
                 radius=0.2, state_fn=None, action_fn=None, observe_set=False, retry_on_raise=False):
    """state_fn(env, rng, t) -> dict(pos, vel, yaw, pitch): custom per-step state overwrite
    (instead of scatter_state); action_fn(env, rng, t): custom actions; observe_set: also
    record ir_gym.env_observation() right after the overwrite (action 0, no other side
    effect than dronestate's max_deviation, which the step repeats anyway).
    A ValueError("math domain error") out of the step (get_alpha with env_train=False,
    vel_obs3D.py:13) ends the scenario: the steps before it are kept and the inputs of the
    raising step are recorded (raised = 1, raise_*).  retry_on_raise (scatter scenarios):
    instead, the env is restored from a deep copy taken before the draw and a new state is
    drawn (the number of rejected draws is recorded)."""
    rng = np.random.default_rng(seed)
    env = make_reference_env(world, nm=nm, **(env_kw or {}))
    N = env.ir_gym.drone_num
    for d in env.ir_gym.drone_list:  # dronestate reports radius_collision (drone.py:256)
        d.radius_collision = radius
    P = max(len(w) for w in world["waypoints_list"])
    wp = np.zeros((N, P, 3))
    for i, w in enumerate(world["waypoints_list"]):
        wp[i, :len(w)] = np.asarray(w, dtype=np.float64)
        wp[i, len(w):] = np.asarray(w[-1], dtype=np.float64)
    rec = dict(actions=[], obs=[], vo_count=[], reward=[], done=[], info=[], finish=[],
               reset_mask=[], obs_after=[], vo_count_after=[])
    if scatter is not None or state_fn is not None:
        rec.update(set_pos=[], set_vel=[], set_yaw=[], set_pitch=[])
    if observe_set:
        rec.update(obs_set=[], vo_count_set=[])
    raised = None
    rejected = 0
    st = {k: [] for k in ("pos", "vel", "yaw", "pitch", "real_len", "max_dev", "extra_len",
                          "wp_idx", "arrive", "dest")}
    obs0, cnt0 = pad_obs(env.drone_reset(False), nm)
    with np.errstate(all="ignore"):
        for t in range(T):
            ss = oset = None
            attempts = 0
            while True:
                snap = copy.deepcopy(env) if retry_on_raise else None
                step_raise = None
                ss = None
                if state_fn is not None:
                    ss = state_fn(env, rng, t)
                    for i, d in enumerate(env.ir_gym.drone_list):
                        d.state = np.array(ss["pos"][i], dtype=np.float64)
                        d.vel = np.array(ss["vel"][i], dtype=np.float64)
                        d.yaw, d.pitch = float(ss["yaw"][i]), float(ss["pitch"][i])
                elif scatter is not None:
                    ss = scatter_state(env, rng, scatter)
                oset = None
                acts = None
                try:
                    if observe_set:
                        step_raise = "observe"
                        oset = pad_obs(env.ir_gym.env_observation(), nm)
                    acts = action_fn(env, rng, t) if action_fn is not None else ACTORS[actor](env, rng)
                    step_raise = "step"
                    o, r, dn, inf, fin = env.drone_step(acts)
                    step_raise = None
                except ValueError as ex:
                    if "math domain error" not in str(ex):
                        raise
                if step_raise is None:
                    break
                attempts += 1
                if retry_on_raise and attempts < 200:
                    env = snap  # the reference aborted mid-step: back to the state before it
                    rejected += 1
                    continue
                raised = dict(where=np.array(step_raise))
                if acts is not None:
                    raised["actions"] = np.asarray(acts, dtype=np.float64)
                if ss is not None:
                    raised.update({"set_" + k: v for k, v in ss.items()})
                break
            if raised is not None:
                break
            if ss is not None:
                for k, v in ss.items():
                    rec["set_" + k].append(v)
            if oset is not None:
                rec["obs_set"].append(oset[0]); rec["vo_count_set"].append(oset[1])
            po, pc = pad_obs(o, nm)
            rec["actions"].append(np.asarray(acts, dtype=np.float64))
            rec["obs"].append(po); rec["vo_count"].append(pc)
            rec["reward"].append(np.asarray(r, dtype=np.float64))
            rec["done"].append(np.asarray(dn, dtype=np.uint8))
            rec["info"].append(np.asarray(inf, dtype=np.uint8))
            rec["finish"].append(np.asarray(fin, dtype=np.uint8))
            for k, v in ref_state(env).items():
                st[k].append(v)
            mask = np.asarray(dn, dtype=bool).copy()
            if reset_on_finish:
                mask |= np.asarray(fin, dtype=bool)
            for i in np.nonzero(mask)[0]:
                env.drone_reset_one(False, int(i))
            if mask.any():
                try:
                    oa, ca = pad_obs(env.ir_gym.env_observation(), nm)
                except ValueError as ex:
                    if "math domain error" not in str(ex):
                        raise
                    # the re-observation after the resets raised: the step itself stays (it is
                    # the last one; its resets are dropped from the record, nothing follows)
                    raised = dict(where=np.array("observe_after"))
                    mask[:] = False
                    oa, ca = po, pc
            else:
                oa, ca = po, pc
            rec["reset_mask"].append(mask.astype(np.uint8))
            rec["obs_after"].append(oa); rec["vo_count_after"].append(ca)
            if raised is not None:
                break
    if not rec["actions"]:
        return None  # raised on the very first step: nothing to keep
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update({"state_" + k: np.stack(v) for k, v in st.items()})
    if retry_on_raise:
        out["rejected_raises"] = np.int32(rejected)  # state draws on which the reference raised
    if raised is not None:
        out["raised"] = np.uint8(1)
        out.update({"raise_" + k: v for k, v in raised.items()})
    out.update(waypoints=wp, n_points=np.asarray(world["n_points_list"], np.int32),
               buildings=np.asarray(world["building_list"], dtype=np.float64).reshape(-1, 4),
               map_size=np.asarray(world["map_size"], dtype=np.float64),
               obs0=obs0, vo_count0=cnt0, nm=np.int32(nm), actor=np.array(actor),
               seed=np.int64(seed), env_train=np.uint8(bool((env_kw or {}).get("env_train", True))),
               radius=np.float64(radius))
    return out


# --------------------------------------------------------------------------
# oracle replay (exact model + ulp-perturbed variant)
# --------------------------------------------------------------------------
def _variant_lib():
    so = os.path.join(ROOT, "oracle", "_build", "librvo3d_oracle_variant.so")
    src = os.path.join(ROOT, "oracle", "rvo3d_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-DORC_VARIANT=1",
                               "-ffp-contract=off", "-fno-builtin-pow", "-fopenmp", "-o", so, src, "-lm"])
    return so


def replay(fx, variant=False):
    """Replay a fixture through the oracle; returns list of mismatch strings."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    if variant:
        import ctypes as C
        saved = orc._lib
        orc._lib = None
        real_so = orc._SO
        orc._SO = _variant_lib()
        try:
            orc.lib()
            return _replay(fx, orc)
        finally:
            orc._lib, orc._SO = saved, real_so
    return _replay(fx, orc)


def _eq(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def _replay(fx, orc):
    bad = []
    env = orc.OracleEnv(fx["waypoints"][None], fx["n_points"][None], fx["map_size"],
                        fx["buildings"], nm=int(fx["nm"]), env_train=bool(fx["env_train"]),
                        radius=np.full(fx["n_points"][None].shape, float(fx["radius"])))
    o0, c0 = env.observe()
    if not (_eq(o0[0], fx["obs0"]) and np.array_equal(c0[0], fx["vo_count0"])):
        bad.append("obs0")
    T = fx["actions"].shape[0]
    for t in range(T):
        if "set_pos" in fx:
            env.set_state(pos=fx["set_pos"][t][None], vel=fx["set_vel"][t][None],
                          yaw=fx["set_yaw"][t][None], pitch=fx["set_pitch"][t][None])
        if "obs_set" in fx:
            os_, cs_ = env.observe()
            if not (_eq(os_[0], fx["obs_set"][t]) and np.array_equal(cs_[0], fx["vo_count_set"][t])):
                bad.append(f"t={t} obs_set")
        obs, cnt, rew, done, info, fin = env.step(fx["actions"][t][None])
        for name, got in (("obs", obs), ("reward", rew)):
            if not _eq(got[0], fx[name][t]):
                bad.append(f"t={t} {name}")
        for name, got in (("vo_count", cnt), ("done", done), ("info", info), ("finish", fin)):
            if not np.array_equal(got[0], fx[name][t]):
                bad.append(f"t={t} {name}")
        s = env.get_state()
        for k in ("wp_idx", "arrive", "dest"):
            if not np.array_equal(s[k][0], fx["state_" + k][t]):
                bad.append(f"t={t} state {k}")
        for k in ("pos", "vel", "yaw", "pitch", "real_len", "max_dev", "extra_len"):
            if not np.allclose(s[k][0], fx["state_" + k][t], rtol=1e-12, atol=1e-12, equal_nan=True):
                bad.append(f"t={t} state {k}")
        m = fx["reset_mask"][t]
        if m.any():
            env.reset_drones(m[None])
            oa, ca = env.observe()
            if not (_eq(oa[0], fx["obs_after"][t]) and np.array_equal(ca[0], fx["vo_count_after"][t])):
                bad.append(f"t={t} obs_after")
    return bad


def oracle_margins(fx):
    """Per (step, drone) minimum decision margin (oracle/rvo3d_oracle.h
    orc_get_margin), covering the step and the re-observation after resets."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    env = orc.OracleEnv(fx["waypoints"][None], fx["n_points"][None], fx["map_size"],
                        fx["buildings"], nm=int(fx["nm"]), env_train=bool(fx["env_train"]),
                        radius=np.full(fx["n_points"][None].shape, float(fx["radius"])))
    env.observe()
    m0 = env.margin()[0]
    out = []
    for t in range(fx["actions"].shape[0]):
        if "set_pos" in fx:
            env.set_state(pos=fx["set_pos"][t][None], vel=fx["set_vel"][t][None],
                          yaw=fx["set_yaw"][t][None], pitch=fx["set_pitch"][t][None])
        ms = None
        if "obs_set" in fx:
            env.observe()
            ms = env.margin()[0]
        env.step(fx["actions"][t][None])
        m = env.margin()[0]
        if ms is not None:
            m = np.minimum(m, ms)
        if fx["reset_mask"][t].any():
            env.reset_drones(fx["reset_mask"][t][None])
            env.observe()
            m = np.minimum(m, env.margin()[0])
        out.append(m)
    return m0, np.stack(out)


# --------------------------------------------------------------------------
# scenario list
# --------------------------------------------------------------------------
def scenarios():
    S = []
    for name in ("world_4", "world_8", "world_2_cross", "world_3", "world_2"):
        w = load_ref_world(name)
        for actor, T in (("desvel", 60), ("follow", 80), ("trainer", 80), ("random", 40)):
            S.append((f"{name}_{actor}", w, actor, T, 11, {}))
    # Q9: drones parked on their destination (no reset) -> des_vel = 0 -> inf/nan reward
    for name in ("world_4", "world_8"):
        S.append((f"{name}_follow_noreset", load_ref_world(name), "follow", 60, 12,
                  dict(reset_on_finish=False)))
    rng = np.random.default_rng(2024)
    specs = [  # N, map, n_points, nb
        (8, (12, 12, 6), 3, 2), (16, (20, 20, 8), 2, 0), (16, (14, 14, 6), 3, 4),
        (24, (25, 25, 8), 4, 6), (32, (30, 30, 10), 2, 10), (12, (8, 8, 5), 2, 0),
    ]
    for k, (N, ms, npnt, nb) in enumerate(specs):
        w = synth_world(rng, N, ms, npnt, nb)
        for actor, T in (("follow", 100), ("trainer", 60), ("random", 40)):
            S.append((f"synth{k}_n{N}_{actor}", w, actor, T, 100 + k, {}))
    # dense scatter: state overwritten every step, velocity-like candidates
    for k, (N, ms, nb, nm, spread) in enumerate([(8, (10, 10, 6), 0, 10, 1.5), (16, (12, 12, 6), 3, 3, 2.0),
                                                  (16, (12, 12, 6), 0, 10, 1.2), (32, (16, 16, 8), 5, 10, 3.0),
                                                  (12, (10, 10, 5), 2, 0, 1.5), (24, (14, 14, 7), 0, 5, 1.0)]):
        w = synth_world(rng, N, ms, 2, nb)
        S.append((f"scatter{k}_n{N}_nm{nm}", w, "vlike", 30, 300 + k, dict(nm=nm, scatter=spread)))
    for k, (N, ms, nb, nm, spread, rad) in enumerate([(32, (12, 12, 6), 0, 2, 1.2, 0.2), (32, (12, 12, 6), 0, 3, 2.5, 0.6),
                                                       (24, (12, 12, 6), 2, 1, 1.0, 0.2), (32, (14, 14, 8), 0, 10, 3.0, 0.8)]):
        w = synth_world(rng, N, ms, 2, nb)
        S.append((f"dense{k}_n{N}_nm{nm}", w, "vslow", 30, 400 + k, dict(nm=nm, scatter=spread, radius=rad)))
    return S


def gen_voinf_calls(nm=3, N=12, iters=120, seed=77):
    """Call-level vectors for rvo_inter.config_vo_inf (rvo_inter.py:20-61):
    dense random states, head-on line-ups (k > nm: truncation + ordering) and
    the survey's Q8 case (t = -1 still flags)."""
    rng = np.random.default_rng(seed)
    world = synth_world(rng, N, (12, 12, 6), 2, 2)
    env = make_reference_env(world, nm=nm)
    g = env.ir_gym
    rec = dict(pos=[], vel=[], action=[], rows=[], count=[], flag=[], tmin=[], collision=[])
    with np.errstate(all="ignore"):
        for it in range(iters):
            kind = it % 3
            if kind == 0:      # dense scatter
                pos = 6 + rng.uniform(-1.5, 1.5, (N, 3)); pos[:, 2] = 3 + rng.uniform(-1, 1, N)
                vel = rng.normal(0, 0.8, (N, 3))
                act = np.round(vel + rng.normal(0, 0.3, (N, 3)), 2)
            elif kind == 1:    # head-on line along a random axis, jittered
                ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
                pos = np.array([6, 6, 3]) + np.outer(np.arange(N) * rng.uniform(0.6, 1.1), ax)
                pos += rng.normal(0, 0.02, (N, 3))
                sp = rng.uniform(0.3, 1.5, N)
                vel = -np.outer(sp, ax); vel[0] = ax * sp[0]
                act = np.round(np.outer(rng.uniform(0.3, 1.5, N), ax) + rng.normal(0, 0.02, (N, 3)), 2)
            else:              # Q8 geometry: neighbour ahead and running away faster
                pos = rng.uniform(2, 10, (N, 3)); pos[:, 2] = rng.uniform(1, 5, N)
                pos[0] = [2, 6, 3]; pos[1] = [7, 6, 3]
                vel = rng.normal(0, 0.3, (N, 3)); vel[0] = [1, 0, 0]; vel[1] = [3, 0, 0]
                act = np.round(vel + rng.normal(0, 0.2, (N, 3)), 2); act[0] = [1.5, 0, 0]
            for i, d in enumerate(g.drone_list):
                d.state, d.vel = pos[i].copy(), vel[i].copy()
            states = g.components["drones"].total_states()
            rows = np.zeros((N, max(nm, 1), 9)); cnt = np.zeros(N, np.int32)
            flag = np.zeros(N, np.uint8); col = np.zeros(N, np.uint8); tmin = np.zeros(N)
            for i in range(N):
                others = [s_ for j, s_ in enumerate(states) if j != i]
                r, f, tm, c, _ = g.rvo.config_vo_inf(states[i], others, g.building_list, act[i])
                cnt[i] = len(r)
                for k, row in enumerate(r):
                    rows[i, k] = np.asarray(row, dtype=np.float64)
                flag[i], col[i], tmin[i] = f, c, tm
            for k, v in (("pos", pos), ("vel", vel), ("action", act), ("rows", rows), ("count", cnt),
                         ("flag", flag), ("tmin", tmin), ("collision", col)):
                rec[k].append(v)
    out = {k: np.stack(v) for k, v in rec.items()}
    wp = np.asarray(world["waypoints_list"], dtype=np.float64)
    out.update(waypoints=wp, n_points=np.asarray(world["n_points_list"], np.int32),
               buildings=np.asarray(world["building_list"], dtype=np.float64).reshape(-1, 4),
               map_size=np.asarray(world["map_size"], dtype=np.float64), nm=np.int32(nm))
    return out


def replay_voinf(fx):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    env = orc.OracleEnv(fx["waypoints"][None], fx["n_points"][None], fx["map_size"],
                        fx["buildings"], nm=int(fx["nm"]))
    bad = 0
    for it in range(fx["pos"].shape[0]):
        env.set_state(pos=fx["pos"][it][None], vel=fx["vel"][it][None])
        for i in range(fx["pos"].shape[1]):
            rows, f, tm, c = env.vo_inf(0, i, fx["action"][it, i])
            k = int(fx["count"][it, i])
            ok = (len(rows) == k and np.array_equal(rows, fx["rows"][it, i, :k]) and
                  f == bool(fx["flag"][it, i]) and c == bool(fx["collision"][it, i]) and
                  (tm == fx["tmin"][it, i]))
            bad += not ok
    return bad


def main():
    os.makedirs(OUT, exist_ok=True)
    calls = gen_voinf_calls()
    nbad = replay_voinf(calls)
    print("voinf calls:", calls["pos"].shape[0] * calls["pos"].shape[1], "mismatches", nbad,
          "max k", int(calls["count"].max()), "t=-1 cases", int((calls["tmin"] == -1).sum()),
          "flagged", int(calls["flag"].sum()))
    assert nbad == 0
    np.savez_compressed(os.path.join(OUT, "calls_voinf.npz"), **calls)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.build()
    kept, dropped = 0, 0
    summary = {}
    for name, world, actor, T, seed, kw in scenarios():
        fx = None
        for attempt in range(6):
            fx = run_scenario(world, actor, T, seed + 1000 * attempt, **kw)
            bad = replay(fx)
            badv = replay(fx, variant=True)
            if not bad and not badv:
                break
            print(f"  {name} seed {seed + 1000 * attempt}: oracle {bad[:3]} variant {badv[:3]} -> retry")
            dropped += 1
            fx = None
        if fx is None:
            print(f"DROPPED {name}")
            continue
        fx["margin0"], fx["margin"] = oracle_margins(fx)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
        kept += 1
        ev = dict(steps=int(fx["actions"].shape[0] * fx["actions"].shape[1]),
                  vo_rows=int(fx["vo_count"].sum()), max_k=int(fx["vo_count"].max()),
                  done=int(fx["done"].sum()), finish_new=int(np.diff(
                      np.concatenate([np.zeros((1,) + fx["finish"].shape[1:], np.int8),
                                      fx["finish"].astype(np.int8)]), axis=0).clip(0).sum()),
                  wp_switch=int((np.diff(fx["state_wp_idx"], axis=0) > 0).sum()),
                  resets=int(fx["reset_mask"].sum()),
                  nonfinite_reward=int((~np.isfinite(fx["reward"])).sum()),
                  knife_edge=int((fx["margin"] < 1e-9).sum()))
        summary[name] = ev
        print(f"kept {name}: {ev}")
    with open(os.path.join(OUT, "SUMMARY.json"), "w") as f:
        json.dump(dict(kept=kept, retried=dropped, scenarios=summary,
                       numpy=np.__version__), f, indent=1)
    print(f"kept {kept} scenarios, {dropped} retries")


if __name__ == "__main__":
    main()
