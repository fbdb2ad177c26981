"""GPU parity: the HIP path (through the C-ABI) against (1) the golden vectors
produced by the Python reference and (2) the CPU oracle on seeded synthetic
batches.  Bar: done / info / finish / vo_count bit-exact; observations and
rewards within 1e-5 relative (BASELINE.json north_star) - in practice they are
bit-identical after the float32 cast and the tests assert that too.

Knife edges: a (drone, step) sample whose smallest decision margin in the
oracle is below 1e-9 (orc_get_margin; e.g. speed = |v| + acc cancelling to
+-1e-17, so the sign of v.rel is libm rounding noise in the reference itself)
is exempt from the comparison.  The accounting (class Tally):
  * exempt samples are counted and bounded: <= 1 % of the compared samples, plus three
    standard deviations of a 1 % binomial for the small runs (the share is inherent to the
    inputs: with 2-decimal accelerations `speed + acc` cancels exactly with probability
    ~1/200 per drone-step, observed 0.4-0.6 %); the LAST test of this file bounds the share
    over all tests together at 1 % flat;
  * exempt samples that REALLY differed are counted per sample (knife_mismatch <= knife);
  * an env in which an exempt sample differed has a different future from then on.  It is
    neither carried along under the exemption nor abandoned: its state is RE-SYNCHRONISED -
    overwritten with the reference's recorded post-step state (golden replays) or the oracle's
    (`resync_from_oracle`) - counted in `resyncs`, and compared again from the next step on, so
    every step of every scenario is compared (`steps_compared == steps`);
  * every test's tally is written to gpurun_out/parity_tally.json (copied to
    profiles/rNN/parity_tally.json for the record).
Everything else must match exactly."""
import json
import os

import numpy as np
import pytest
import torch

import oracle as orc
from golden_util import eq_nan, load, scenario_files
from rvo3d_amd import BatchedDroneEnv, World, synthetic_actions, synthetic_world

pytestmark = pytest.mark.gpu
RTOL = 1e-5
KNIFE = 1e-9
KNIFE_SHARE = 0.01
FILES = scenario_files()
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_TALLY_FILE = os.path.join(_ROOT, "gpurun_out", "parity_tally.json")


def _record_tally(name, rec):
    try:
        os.makedirs(os.path.dirname(_TALLY_FILE), exist_ok=True)
        allr = {}
        if os.path.exists(_TALLY_FILE):
            with open(_TALLY_FILE) as f:
                allr = json.load(f)
        allr[name] = rec
        with open(_TALLY_FILE, "w") as f:
            json.dump(allr, f, indent=1, sort_keys=True)
    except OSError:
        pass


class Tally:
    """Per-sample comparison with knife-edge exemption, sample-level accounting and env
    dropping.  Shape of a margin array: [E, N] (or [N] for one env)."""

    def __init__(self, name, E=1, strict=False):
        self.name, self.strict = name, strict
        self.samples = self.knife = self.knife_mismatch = self.resyncs = 0
        self.dropped = np.zeros(E, bool)   # envs whose state has diverged and is not yet re-synchronised
        self.mg = None
        self.bad = None

    def begin(self, margin, count=True):
        """Start a group of checks that share `margin` (one step / one observe)."""
        mg = np.array(margin, dtype=np.float64, copy=True).reshape(len(self.dropped), -1)
        if self.strict:
            mg[:] = np.inf
        live = ~self.dropped
        self.mg = mg
        self.bad = np.zeros(mg.shape, bool)
        if count:
            self.samples += int(live.sum()) * mg.shape[1]
            self.knife += int((mg[live] < KNIFE).sum())

    def check(self, what, ok):
        ok = np.asarray(ok)
        ok = ok.reshape(self.mg.shape[0], self.mg.shape[1], -1).all(axis=-1)
        live = ~self.dropped[:, None]
        firm = (self.mg >= KNIFE) & live
        assert ok[firm].all(), (f"{self.name} {what}: {int((~ok & firm).sum())} firm mismatches at "
                                f"{np.argwhere(~ok & firm)[:4].tolist()} (margins {self.mg[~ok & firm][:4].tolist()}, "
                                f"resyncs so far {self.resyncs})")
        self.bad |= ~ok & ~firm & live

    def end(self):
        """Close the group: count exempt samples that differed, drop their envs."""
        self.knife_mismatch += int(self.bad.sum())
        self.dropped |= self.bad.any(axis=1)
        return self.dropped

    def resync(self):
        """The caller has brought the diverged envs back in line with the reference."""
        self.resyncs += int(self.dropped.sum())
        self.dropped[:] = False

    def finish(self, **extra):
        rec = dict(samples=self.samples, knife=self.knife, knife_mismatch=self.knife_mismatch,
                   dropped_envs=int(self.dropped.sum()), resyncs=self.resyncs, envs=len(self.dropped), **extra)
        _record_tally(self.name, rec)
        lim = KNIFE_SHARE * self.samples
        assert self.knife <= lim + 3.0 * np.sqrt(lim) + 2, rec
        assert self.knife_mismatch <= self.knife, rec
        return rec


STATE_KEYS = ("pos", "vel", "yaw", "pitch", "real_len", "max_dev", "extra_len", "wp_idx", "arrive", "dest")


def resync_from_oracle(env, ref, tl, extra=None):
    """Envs in which an exempt sample really differed (tl.dropped), or `extra`: overwrite their
    device state with the oracle's (all ten arrays) and take them back into the comparison."""
    d = tl.dropped.copy()
    if extra is not None:
        d |= np.asarray(extra, bool)
    if not d.any():
        return 0
    rs = ref.get_state()
    s = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    for k in STATE_KEYS:
        s[k][d] = rs[k][d]
    env.set_state(**s)
    tl.dropped |= d
    tl.resync()
    return int(d.sum())


def close(a, b):
    """|a-b| <= RTOL*|b| elementwise, NaN==NaN, inf==inf (survey Q9)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        ok = np.abs(a - b) <= RTOL * np.abs(b) + 1e-9
    return ok | eq_nan(a, b)


def world_of(fx, E=1):
    rep = lambda a: np.repeat(a[None], E, axis=0).copy()
    return World(rep(fx["waypoints"]), rep(fx["n_points"]), fx["map_size"], fx["buildings"])


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_replays_reference_golden(path):
    """Every scenario the Python reference produced (oracle/gen_golden.py,
    oracle/gen_golden_branches.py): both env_train modes, observe-after-set records, and - where
    the reference ended in ValueError("math domain error") - the raising call, which must set
    RVO3D_FLAG_DOMAIN_ERROR.  nanbeta_n8 pins cos = 1 + ulp -> arccos NaN -> outside the cone:
    its samples sit on that edge by construction and are compared without exemption."""
    fx = load(path)
    name = os.path.basename(path)[:-4]
    env = BatchedDroneEnv(world_of(fx), neighbors_num=int(fx["nm"]), radius=float(fx["radius"]),
                          env_train=bool(fx["env_train"]))
    obs, cnt = env.observe()
    tl = Tally("golden/" + name, strict=name.startswith("nanbeta"))
    tl.begin(fx["margin0"], count=False)
    tl.check("obs0", close(obs[0].cpu().numpy(), fx["obs0"]))
    tl.check("vo_count0", cnt[0].cpu().numpy() == fx["vo_count0"])
    tl.end()
    T = fx["actions"].shape[0]
    steps_compared = 0
    for t in range(T):
        tl.begin(fx["margin"][t])
        if "set_pos" in fx:
            env.set_state(pos=fx["set_pos"][t][None], vel=fx["set_vel"][t][None],
                          yaw=fx["set_yaw"][t][None], pitch=fx["set_pitch"][t][None])
        if "obs_set" in fx:
            os_, cs_ = env.observe()
            tl.check(f"obs_set t={t}", eq_nan(os_[0].cpu().numpy(), fx["obs_set"][t].astype(np.float32)))
            tl.check(f"vo_count_set t={t}", cs_[0].cpu().numpy() == fx["vo_count_set"][t])
        obs, cnt, rew, done, info, fin = env.step(torch.from_numpy(fx["actions"][t][None]))
        o, r = obs[0].cpu().numpy(), rew[0].cpu().numpy()
        tl.check(f"done t={t}", done[0].cpu().numpy() == fx["done"][t])
        tl.check(f"info t={t}", info[0].cpu().numpy() == fx["info"][t])
        tl.check(f"finish t={t}", fin[0].cpu().numpy() == fx["finish"][t])
        tl.check(f"vo_count t={t}", cnt[0].cpu().numpy() == fx["vo_count"][t])
        tl.check(f"obs t={t}", close(o, fx["obs"][t]))
        tl.check(f"reward t={t}", close(r, fx["reward"][t]))
        tl.check(f"obs f32-exact t={t}", eq_nan(o, fx["obs"][t].astype(np.float32)))
        tl.check(f"reward f32-exact t={t}", eq_nan(r, fx["reward"][t].astype(np.float32)))
        if tl.end()[0]:
            # an exempt (margin < 1e-9) sample really differed: from here on the device would follow
            # another trajectory than the reference did.  The fixture holds the reference's full
            # post-step state: take it over and go on comparing.
            env.set_state(**{k: fx["state_" + k][t][None] for k in STATE_KEYS})
            tl.resync()
        else:
            s = env.get_state()
            assert np.array_equal(s["wp_idx"][0].cpu().numpy(), fx["state_wp_idx"][t])
            for k in ("pos", "vel", "yaw", "pitch", "real_len", "max_dev", "extra_len"):
                np.testing.assert_allclose(s[k][0].cpu().numpy(), fx["state_" + k][t], rtol=1e-9, atol=1e-9,
                                           err_msg=f"{k} t={t}")
        m = fx["reset_mask"][t]
        if m.any():
            env.reset_drones(m[None])
            oa, ca = env.observe()
            tl.begin(fx["margin"][t], count=False)  # (margin[t] covers the re-observation as well)
            tl.check(f"obs_after t={t}", close(oa[0].cpu().numpy(), fx["obs_after"][t]))
            tl.check(f"vo_count_after t={t}", ca[0].cpu().numpy() == fx["vo_count_after"][t])
            if tl.end()[0]:
                tl.resync()  # an observation only: the state behind it is the recorded one
        steps_compared += 1
    assert steps_compared == T
    raised_checked = False
    flags = env.error_flags()
    assert not (flags & 2), "RVO3D_FLAG_DOMAIN_ERROR on a call the reference completed"
    if int(fx.get("raised", 0)) and str(fx["raise_where"]) in ("step", "observe") and not tl.dropped[0]:
        if "raise_set_pos" in fx:
            env.set_state(pos=fx["raise_set_pos"][None], vel=fx["raise_set_vel"][None],
                          yaw=fx["raise_set_yaw"][None], pitch=fx["raise_set_pitch"][None])
        if str(fx["raise_where"]) == "observe":
            env.observe()
        else:
            env.step(torch.from_numpy(fx["raise_actions"][None]))
        with pytest.raises(ValueError, match="math domain error"):
            env.check_finite()
        raised_checked = True
    tl.finish(steps=T, steps_compared=steps_compared, env_train=int(fx["env_train"]),
              raise_checked=raised_checked)
    env.close()


def run_vs_oracle(world, T, nm=10, autoreset=True, f32_actions=False, radius=None, seed=1234,
                  vlike=False, env_train=True, name=None, priority=None):
    E, N, _ = world.shape
    dec = 2 if f32_actions else -1
    env = BatchedDroneEnv(world, neighbors_num=nm, action_decimals=dec, radius=radius,
                          env_train=env_train, priority=priority)
    bc = lambda x: None if x is None else np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), (E, N)))
    ref = orc.OracleEnv(world.waypoints, world.n_points, world.map_size, world.buildings, nm=nm,
                        radius=bc(radius), priority=bc(priority), threads=8, env_train=env_train)
    o0, c0 = env.observe()
    r0, rc0 = ref.observe()
    tl = Tally(name or f"oracle/{N}x{E}_nm{nm}_T{T}_seed{seed}", E)
    tl.begin(ref.margin(), count=False)
    tl.check("obs0", close(o0.cpu().numpy(), r0))
    tl.check("cnt0", c0.cpu().numpy() == rc0)
    tl.end()
    stats = dict(steps=0, done=0, finish=0, vo_rows=0, resets=0)
    for t in range(T):
        a = synthetic_actions(E, N, t, seed)
        if vlike:  # trainer-like candidates near the current velocity (multi_ppo.py:203-205)
            a = np.round(ref.get_state()["vel"] + a, 2)
        ad = torch.from_numpy(a.astype(np.float32) if f32_actions else a).cuda()
        if autoreset:
            obs, cnt, rew, done, info, fin = env.step(ad, autoreset=True)
            ro, rcnt, rr, rd, ri, rf, rm = ref.step_autoreset(a)
            tl.begin(ref.margin())  # covers the step and the re-observation after resets
            tl.check(f"reset_mask t={t}", env.reset_mask.cpu().numpy() == rm)
        else:
            obs, cnt, rew, done, info, fin = env.step(ad)
            ro, rcnt, rr, rd, ri, rf = ref.step(a)
            rm = rd | rf
            tl.begin(ref.margin())
        o, r = obs.cpu().numpy(), rew.cpu().numpy()
        tl.check(f"done t={t}", done.cpu().numpy() == rd)
        tl.check(f"info t={t}", info.cpu().numpy() == ri)
        tl.check(f"finish t={t}", fin.cpu().numpy() == rf)
        tl.check(f"vo_count t={t}", cnt.cpu().numpy() == rcnt)
        tl.check(f"obs t={t}", close(o, ro))
        tl.check(f"reward t={t}", close(r, rr))
        tl.check(f"obs f32-exact t={t}", eq_nan(o, ro.astype(np.float32)))
        tl.check(f"reward f32-exact t={t}", eq_nan(r, rr.astype(np.float32)))
        tl.end()
        resync_from_oracle(env, ref, tl)  # (before the resets below: both sides reset the same drones)
        if not autoreset and rm.any():
            env.reset_drones(rm)
            ref.reset_drones(rm)
            oa, ca = env.observe()
            roa, rca = ref.observe()
            tl.begin(ref.margin(), count=False)
            tl.check(f"obs_after t={t}", close(oa.cpu().numpy(), roa))
            tl.check(f"cnt_after t={t}", ca.cpu().numpy() == rca)
            tl.end()
            resync_from_oracle(env, ref, tl)
        stats["steps"] += E * N
        stats["done"] += int(rd.sum()); stats["finish"] += int(rf.sum())
        stats["vo_rows"] += int(rcnt.sum()); stats["resets"] += int(rm.sum())
    s, rs = env.get_state(), ref.get_state()
    assert not tl.dropped.any()  # every divergence was re-synchronised
    keep = ~tl.dropped
    for k in ("wp_idx", "arrive", "dest"):
        assert np.array_equal(s[k].cpu().numpy()[keep], rs[k][keep]), k
    for k in ("pos", "vel", "yaw", "pitch", "real_len", "max_dev", "extra_len"):
        np.testing.assert_allclose(s[k].cpu().numpy()[keep], rs[k][keep], rtol=1e-9, atol=1e-9, err_msg=k)
    flags = env.error_flags()
    if tl.resyncs == 0:  # (a diverged env may have met - or missed - such an event on its own path)
        assert (flags & 1) == (1 if ref.nan_count else 0)
        assert bool(flags & 2) == (ref.domain_count > 0)
    env.close()
    stats.update(tl.finish(**stats), domain_pairs=ref.domain_count)
    print(stats)
    return stats


def test_values_on_file_follow_outside_state_changes():
    """The step keeps des_vel, the stage-G words and the current / previous waypoint on file
    between steps.  State set from outside (reset_drones, set_state with a new waypoint
    index) must be picked up by the very next step, with no observe() in between."""
    E, N, nm = 8, 16, 10
    world = synthetic_world(E, N, (20.0, 20.0, 8.0), n_points=4, seed=77)
    # short legs, so that waypoint switches happen within the run
    wp = world.waypoints.copy()
    for k in range(1, 4):
        wp[:, :, k] = np.clip(np.round(wp[:, :, k - 1] + (wp[:, :, k] - wp[:, :, k - 1]) * 0.08, 2),
                              1.0, [19.0, 19.0, 7.0])
    world = type(world)(wp, world.n_points, world.map_size, world.buildings)
    env = BatchedDroneEnv(world, neighbors_num=nm, action_decimals=-1)
    ref = orc.OracleEnv(world.waypoints, world.n_points, world.map_size, world.buildings, nm=nm,
                        threads=8)
    env.observe(); ref.observe()
    tl = Tally("values_on_file", E)
    rng = np.random.default_rng(5)
    switches = 0
    for t in range(40):
        if t % 5 == 2:  # host-side reset of some drones, then step directly
            m = rng.random((E, N)) < 0.2
            env.reset_drones(m); ref.reset_drones(m)
        if t % 5 == 4:  # teleport + new waypoint index, then step directly
            st = ref.get_state()
            pos = np.round(st["pos"] + rng.uniform(-0.5, 0.5, st["pos"].shape), 2)
            wi = np.clip(st["wp_idx"] + rng.integers(-1, 2, st["wp_idx"].shape), 1, 3).astype(np.int32)
            env.set_state(pos=pos, wp_idx=wi); ref.set_state(pos=pos, wp_idx=wi)
        a = np.round(ref.get_state()["vel"] * 0.5 + synthetic_actions(E, N, t, 99), 2)
        before = ref.get_state()["wp_idx"].copy()
        obs, cnt, rew, done, info, fin = env.step(torch.from_numpy(a).cuda(), autoreset=(t % 2 == 0))
        if t % 2 == 0:
            ro, rcnt, rr, rd, ri, rf, rm = ref.step_autoreset(a)
        else:
            ro, rcnt, rr, rd, ri, rf = ref.step(a)
        switches += int((ref.get_state()["wp_idx"] > before).sum())
        tl.begin(ref.margin())
        o, r = obs.cpu().numpy(), rew.cpu().numpy()
        tl.check(f"done t={t}", done.cpu().numpy() == rd)
        tl.check(f"info t={t}", info.cpu().numpy() == ri)
        tl.check(f"finish t={t}", fin.cpu().numpy() == rf)
        tl.check(f"vo_count t={t}", cnt.cpu().numpy() == rcnt)
        tl.check(f"obs f32-exact t={t}", eq_nan(o, ro.astype(np.float32)))
        tl.check(f"reward f32-exact t={t}", eq_nan(r, rr.astype(np.float32)))
        keep = ~tl.end()
        s, rs = env.get_state(), ref.get_state()
        assert np.array_equal(s["wp_idx"].cpu().numpy()[keep], rs["wp_idx"][keep]), t
        np.testing.assert_allclose(s["max_dev"].cpu().numpy()[keep], rs["max_dev"][keep], rtol=1e-9, atol=1e-9)
        resync_from_oracle(env, ref, tl)
    assert switches > 20, switches
    env.close()
    print(tl.finish(waypoint_switches=switches))


def test_cfg2_16x256_flags_bit_exact():
    """BASELINE config 2: 16 drones x 256 envs, bit-exact collision flags vs CPU."""
    st = run_vs_oracle(synthetic_world(256, 16, (20, 20, 8)), T=60, autoreset=False)
    assert st["done"] > 0 and st["vo_rows"] > 0, st


def test_cfg2_autoreset_f32_actions():
    st = run_vs_oracle(synthetic_world(256, 16, (20, 20, 8), seed=99), T=60, autoreset=True,
                       f32_actions=True)
    assert st["resets"] > 0, st


def test_cfg3_shape_64_drones():
    st = run_vs_oracle(synthetic_world(96, 64, (50, 50, 10)), T=40, autoreset=True)
    assert st["resets"] > 0, st


def test_cfg5_shape_256_drones_50_buildings():
    st = run_vs_oracle(synthetic_world(6, 256, (100, 100, 10), nb=50), T=25, autoreset=True)
    assert st["done"] > 0, st


@pytest.mark.parametrize("N,E,nm", [(3, 5, 10), (12, 33, 2), (24, 7, 0), (100, 3, 10), (128, 5, 3), (300, 2, 4),
                                    (16, 6, 10), (32, 3, 10), (8, 13, 12)])
def test_ragged_sizes(N, E, nm):
    """N not a power of two, E not a multiple of envs-per-block, nm = 0, N > 256."""
    L = 6 + 2 * int(np.sqrt(N))
    st = run_vs_oracle(synthetic_world(E, N, (L, L, 6), n_points=3, nb=2), T=30, nm=nm)
    assert st["samples"] > 0, st


@pytest.mark.parametrize("N,E,kernel", [(48, 9, "<2, 1, 64, true, true>"), (33, 5, "<2, 1, 64, true, true>"),
                                        (63, 4, "<2, 1, 64, true, true>"), (24, 11, "<2, 1, 32, true, true>"),
                                        (31, 8, "<2, 1, 32, true, true>"), (65, 4, "<2, 2, 128, true, true>"),
                                        (96, 6, "<2, 2, 128, true, true>"), (127, 3, "<2, 2, 128, true, true>"),
                                        (129, 3, "<2, 3, 192, true, true>"), (160, 4, "<2, 3, 192, true, true>"),
                                        (192, 3, "<2, 3, 192, true, true>"), (193, 2, "<2, 4, 256, true, true>"),
                                        (200, 4, "<2, 4, 256, true, true>"), (255, 2, "<2, 4, 256, true, true>")])
def test_padded_compile_time_kernels(N, E, kernel):
    """Envs smaller than the compile-time ring of their kernel (round 3): the lanes beyond N are ghost drones
    parked at infinity.  Dense worlds (many pairs across the ring's seam, resets every few steps), both reset
    protocols, both env_train modes, per-drone radii; the instantiation is checked by name."""
    L = 5 + 1.6 * np.sqrt(N)
    world = synthetic_world(E, N, (L, L, 6.0), n_points=3, nb=3, min_sep=0.6, seed=N)
    env = BatchedDroneEnv(world)
    assert env.kernel_name("step_autoreset").endswith(kernel), env.kernel_name("step_autoreset")
    env.close()
    for autoreset in (True, False):
        st = run_vs_oracle(world, T=24, autoreset=autoreset, vlike=not autoreset, seed=3,
                           name=f"padded/{N}x{E}_ar{int(autoreset)}")
        assert st["done"] > 0 and (autoreset or st["vo_rows"] > 0), st
    rng = np.random.default_rng(N)
    st = run_vs_oracle(world, T=16, autoreset=True, radius=np.round(rng.uniform(0.15, 0.4, (E, N)), 2),
                       priority=rng.integers(1, 9, (E, N)).astype(np.float64), env_train=False, seed=5,
                       name=f"padded/{N}x{E}_eval_rp")
    assert st["samples"] > 0, st


def test_dense_small_nm_truncation():
    """Crowded envs with big radii: many VO rows per drone, nm = 2 forces the
    keep-the-most-urgent truncation (rvo_inter.py:50-56)."""
    st = run_vs_oracle(synthetic_world(64, 32, (12, 12, 5), min_sep=0.8), T=40, nm=2, radius=0.3,
                       vlike=True, autoreset=False)
    assert st["vo_rows"] > 500, st


@pytest.mark.parametrize("N,E", [(16, 64), (64, 12), (100, 3), (128, 3), (256, 2)])
def test_per_drone_radius_and_priority(N, E):
    """radius / priority arrays that differ from drone to drone (the step then reads them per
    drone instead of taking the one value from its argument block; get_PAA's pr = pra / (pra + prb)
    is no longer 0.5, vel_obs3D.py:19-32; the fp32 cone filter stands down for unequal priorities)."""
    rng = np.random.default_rng(8)
    L = 6 + 2.5 * np.sqrt(N)
    world = synthetic_world(E, N, (L, L, 6.0), min_sep=1.3, seed=23)
    radius = np.round(rng.uniform(0.15, 0.5, (E, N)), 2)
    priority = rng.integers(1, 9, (E, N)).astype(np.float64)
    st = run_vs_oracle(world, T=30, autoreset=False, radius=radius, priority=priority, vlike=True,
                       name=f"per_drone_rp/{N}x{E}")
    assert st["vo_rows"] > 20 and st["done"] > 0, st
    st = run_vs_oracle(world, T=20, autoreset=True, radius=radius, priority=priority, seed=3,
                       name=f"per_drone_rp/{N}x{E}_autoreset")
    assert st["resets"] > 0, st


@pytest.mark.parametrize("N,E,nb,rmaxb", [(32, 48, 10, 1.5), (48, 24, 40, 1.5), (64, 16, 12, 6.5), (100, 4, 300, 0.6)])
def test_building_lists_short_long_and_overflowing(N, E, nb, rmaxb):
    """The per-cell building lists (a building is listed where a drone of the largest radius
    could touch it, at most the 5 m gate of rvo_inter.py:104): sparse lists, lists longer than
    the four entries tested from registers, cells that overflow (all buildings tested), building
    radii beyond the gate (r + br > 5: the gate, not the radius, decides) and per-drone radii."""
    rng = np.random.default_rng(100 + N)
    L = 24.0
    world = synthetic_world(E, N, (L, L, 8.0), min_sep=0.8, seed=41, nb=nb)
    b = world.buildings.copy()
    # unrounded axes / radii / heights: with 2-decimal buildings AND 2-decimal start positions,
    # dis == r + br holds exactly for several percent of the (drone, building) pairs of a
    # crowded map - knife edges by construction, which would only dilute the comparison
    b[:, :2] += rng.uniform(-0.004, 0.004, (nb, 2))
    b[:, 3] = rng.uniform(0.05, rmaxb, nb)
    b[:, 2] = rng.uniform(0.5, 9.0, nb)      # some lower than the drones fly, some above the map
    world.buildings = b
    radius = np.round(rng.uniform(0.1, 0.9, (E, N)), 2)
    st = run_vs_oracle(world, T=25, autoreset=True, radius=radius, name=f"buildings/{N}x{E}x{nb}")
    assert st["done"] > 0 and st["resets"] > 0, st
    st = run_vs_oracle(world, T=15, autoreset=False, radius=0.2, seed=9, name=f"buildings/{N}x{E}x{nb}_uniform")
    assert st["done"] > 0, st


def test_cfg2_velocity_like_actions():
    """Same world as config 2 with trainer-like candidate velocities: VO rows are common."""
    st = run_vs_oracle(synthetic_world(256, 16, (20, 20, 8), seed=5), T=40, vlike=True,
                       autoreset=False)
    assert st["vo_rows"] > 50, st


@pytest.mark.parametrize("N,E", [(16, 12), (100, 3), (128, 3), (256, 2)])
def test_drones_far_outside_the_map_bypass_the_fp32_filters(N, E):
    """The fp32 stages (G, X1) assume centred coordinates within `cmax` of the map; an env with a
    drone beyond that (here: clusters teleported hundreds of metres away, close enough to each
    other to interact) switches to the exact path for every pair.  Results must not change."""
    world = synthetic_world(E, N, (20.0, 20.0, 8.0), seed=31)
    env = BatchedDroneEnv(world, neighbors_num=10, action_decimals=-1)
    ref = orc.OracleEnv(world.waypoints, world.n_points, world.map_size, world.buildings, nm=10, threads=8)
    env.observe(); ref.observe()
    rng = np.random.default_rng(2)
    tl = Tally(f"far_bypass/{N}x{E}", E)
    rows = 0
    for t in range(12):
        if t % 3 == 0:  # every third step: half of the envs get a far-away cluster
            st = ref.get_state()
            pos, vel = st["pos"].copy(), st["vel"].copy()
            for e in range(0, E, 2):
                k = max(N // 3, 2)
                centre = np.array([400.0 + 50 * e, -300.0, 4.0])
                pos[e, :k] = np.round(centre + rng.uniform(-2.0, 2.0, (k, 3)), 2)
                vel[e, :k] = np.round(rng.normal(0, 0.6, (k, 3)), 2)
            env.set_state(pos=pos, vel=vel); ref.set_state(pos=pos, vel=vel)
        a = np.round(ref.get_state()["vel"] + synthetic_actions(E, N, t, 5), 2)
        auto = t % 2 == 1
        obs, cnt, rew, done, info, fin = env.step(torch.from_numpy(a).cuda(), autoreset=auto)
        if auto:
            ro, rcnt, rr, rd, ri, rf, rm = ref.step_autoreset(a)
        else:
            ro, rcnt, rr, rd, ri, rf = ref.step(a)
        tl.begin(ref.margin())
        tl.check(f"done t={t}", done.cpu().numpy() == rd)
        tl.check(f"vo_count t={t}", cnt.cpu().numpy() == rcnt)
        tl.check(f"obs t={t}", eq_nan(obs.cpu().numpy(), ro.astype(np.float32)))
        tl.check(f"reward t={t}", eq_nan(rew.cpu().numpy(), rr.astype(np.float32)))
        tl.end()
        resync_from_oracle(env, ref, tl)
        rows += int(rcnt.sum())
    assert rows > 0
    env.close()
    tl.finish(vo_rows=rows)


def test_empty_neighbourhood_single_drone():
    st = run_vs_oracle(synthetic_world(9, 1, (10, 10, 5)), T=20)
    assert st["vo_rows"] == 0


def test_action_shape_is_checked():
    env = BatchedDroneEnv(synthetic_world(2, 4, (10, 10, 5)))
    with pytest.raises(AssertionError):  # drone.py:101 `assert act.shape == (3,)`
        env.step(torch.zeros(2, 4, 2))
    env.close()


def test_nonfinite_observation_sets_error_flag():
    """The reference raises ValueError on NaN/Inf observations (ir_gym.py:232-239);
    here the device sets a flag word that check_finite() surfaces."""
    env = BatchedDroneEnv(synthetic_world(1, 4, (10, 10, 5)))
    env.set_state(pos=np.array([[[np.nan, 1, 1], [2, 2, 2], [3, 3, 3], [4, 4, 4]]]))
    env.observe()
    with pytest.raises(ValueError):
        env.check_finite()
    env.close()


def test_mdin_list_api_config1_world4():
    """BASELINE config 1 plumbing: the `mdin` list API (mdin.py:7-48) on world_4,
    driven exactly like uaisa_env/gym_env_test.py:7-21, against the golden run."""
    from rvo3d_amd.drone_envs.mdin import mdin
    fx = load([f for f in FILES if f.endswith("world_4_desvel.npz")][0])
    world = dict(drone_num=4, map_size=fx["map_size"].tolist(),
                 waypoints_list=fx["waypoints"].tolist(), n_points_list=fx["n_points"].tolist(),
                 building_list=[])
    env = mdin(world=world)
    obs_list = env.drone_reset(False)
    assert len(obs_list) == 4 and all(len(o) == 21 for o in obs_list)
    assert env.ir_gym.drone_num == 4 and env.observation_space.shape == (21,)
    for t in range(fx["actions"].shape[0]):
        vel_list = env.ir_gym.cal_des_list()
        np.testing.assert_array_equal(np.asarray(vel_list), fx["actions"][t])
        obs_list, reward_list, done_list, info_list, finish_list = env.drone_step(vel_list)
        assert done_list == [bool(x) for x in fx["done"][t]]
        assert finish_list == [bool(x) for x in fx["finish"][t]]
        for i, o in enumerate(obs_list):
            k = max(int(fx["vo_count"][t][i]), 1)
            assert len(o) == 12 + 9 * k
            assert close(o, fx["obs"][t][i][:len(o)]).all()
        assert eq_nan(np.asarray(reward_list), fx["reward"][t]).all()   # float64, bit for bit (mdin.py:28)
        # the trainer's reach-through (multi_ppo.py:202, 212, 246): drone_list[i].vel / .state, indicators_*
        for i in (0, 3):
            np.testing.assert_allclose(env.ir_gym.drone_list[i].vel, fx["state_vel"][t][i], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(env.ir_gym.drone_list[i].state, fx["state_pos"][t][i], rtol=1e-12, atol=1e-12)
            assert env.ir_gym.drone_list[i].i == int(fx["state_wp_idx"][t][i])
        np.testing.assert_allclose(env.ir_gym.indicators_deviation(), fx["state_max_dev"][t], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(env.ir_gym.indicators_extra_len(), fx["state_extra_len"][t], rtol=1e-12, atol=1e-12)
        for i in [i for i, d in enumerate(done_list) if d]:
            env.drone_reset_one(False, i)
    with pytest.raises(AssertionError):
        env.drone_step([[0, 0, 0]] * 4)   # python lists are rejected (drone.py:98-101)
    env.close()


def test_step_policy_trainer_glue():
    """abs_action = np.round(acceler * np.round(a_inc, 2) + vel, 2)
    (multi_ppo.py:196-205) computed on the device from raw float32 samples must equal
    numpy's own arithmetic fed to the oracle."""
    world = synthetic_world(128, 16, (20, 20, 8), seed=11)
    E, N, _ = world.shape
    env = BatchedDroneEnv(world)
    ref = orc.OracleEnv(world.waypoints, world.n_points, world.map_size, world.buildings, threads=8)
    env.observe(); ref.observe()
    rng = np.random.default_rng(3)
    tl = Tally("step_policy_glue", E)
    rows = 0
    ties = 0  # (env, step) pairs whose action itself was decided by a last bit of the velocity
    for t in range(40):
        a_inc = rng.normal(0, 0.6, (E, N, 3)).clip(-1, 1).astype(np.float32)
        vel = ref.get_state()["vel"]
        gvel = env.get_state()["vel"].cpu().numpy()
        a2 = np.round(a_inc, 2)                      # float32, as in the trainer
        abs_action = np.round(env.acceler * a2 + vel, 2)
        assert a2.dtype == np.float32 and abs_action.dtype == np.float64
        # The glue's own rounding is a decision too: 0.5 * a + vel can sit on a .5 tie
        # (vel is a decimal when pitch = +-90 deg) where the last bit of vel - libm noise
        # of sin/cos - decides.  An env whose two sides would round differently is brought back
        # onto the oracle's state first (the last bit of its velocity included), so that both step
        # with the same action; it is counted, not dropped.
        tainted = (np.round(env.acceler * a2 + gvel, 2) != abs_action).any(axis=(1, 2))
        ties += resync_from_oracle(env, ref, tl, extra=tainted)
        obs, cnt, rew, done, info, fin = env.step_policy(torch.from_numpy(a_inc).cuda(), autoreset=True)
        ro, rcnt, rr, rd, ri, rf, rm = ref.step_autoreset(abs_action)
        tl.begin(ref.margin())
        tl.check(f"done t={t}", done.cpu().numpy() == rd)
        tl.check(f"finish t={t}", fin.cpu().numpy() == rf)
        tl.check(f"vo_count t={t}", cnt.cpu().numpy() == rcnt)
        tl.check(f"obs t={t}", eq_nan(obs.cpu().numpy(), ro.astype(np.float32)))
        tl.check(f"reward t={t}", eq_nan(rew.cpu().numpy(), rr.astype(np.float32)))
        tl.end()
        resync_from_oracle(env, ref, tl)
        rows += int(rcnt.sum())
    assert ties < 0.01 * 40 * E
    s, rs = env.get_state(), ref.get_state()
    np.testing.assert_allclose(s["pos"].cpu().numpy(), rs["pos"], rtol=1e-9, atol=1e-9)
    assert rows > 0
    tl.finish(action_rounding_ties=ties, vo_rows=rows)
    env.close()


@pytest.mark.parametrize("kind", ["mlp", "rnn"])
def test_training_loop_end_to_end(kind, tmp_path):
    """8(f) rows 1-2 on the device: rollout through rvo3d_step_policy, batched GAE with the
    reference's path-cut rules, clipped-PPO update; checkpoint layout of the reference."""
    from rvo3d_amd.policy import mlp_ac, multi_ppo, rnn_ac
    env = BatchedDroneEnv(synthetic_world(64, 8, (20, 20, 8), n_points=3, seed=4))

    class Space:
        shape = (3,)
    ac = (mlp_ac(env.W) if kind == "mlp" else
          rnn_ac(None, Space(), 12, 9, 64, (64, 64), (64, 64), torch.nn.ReLU, torch.nn.Tanh,
                 torch.nn.Identity, use_gpu=False, rnn_mode="biGRU")).cuda()
    tr = multi_ppo(env, ac, pi_lr=3e-4, vf_lr=1e-3, train_epoch=1, steps_per_epoch=24,
                   max_ep_len=15, train_pi_iters=4, train_v_iters=4, target_kl=0.05,
                   save_path=str(tmp_path) + "/", save_name="t", save_freq=1)
    log = tr.training_loop()
    assert len(log) == 2 and all(np.isfinite(l["loss_v"]) for l in log)
    ck = torch.load(str(tmp_path / "t_check_point_1.pt"), weights_only=True)
    assert set(ck) == {"model_state", "pi_optimizer", "vf_optimizer"}   # multi_ppo.py:411-412
    assert env.error_flags() == 0
    env.close()


class _TablePolicy:
    """Scripted stand-in for the actor-critic: replays rows of an action table (float32 [steps,
    N, 3]); env e starts at row start[e].  Same role as the generator's get_action."""

    def __init__(self, table, starts):
        pad = np.zeros((4096,) + table.shape[1:], np.float32)
        self.table = torch.from_numpy(np.concatenate([table, pad])).cuda()
        self.starts = list(starts)
        self.t = 0

    def eval(self):
        return self

    def step_tensors(self, obs, std_factor=1):
        rows = torch.stack([self.table[s + self.t] for s in self.starts])   # [E, N, 3]
        self.t += 1
        return rows.view(-1, 3), None, None


POST_TRAIN = sorted(f for f in os.listdir(os.path.join(_ROOT, "tests", "golden")) if f.startswith("post_train_"))


@pytest.mark.parametrize("name", POST_TRAIN)
def test_post_train_matches_the_references_policy_test(name):
    """8(f) row 3: post_train.policy_test of the REFERENCE (post_train.py:38-128), run in the
    build container on the reference env (oracle/gen_golden_post_train.py; env_train=False as
    train/policy_test.py:46 builds it, and the trainer's env_train=True test_env), against
    rvo3d_amd.policy.post_train with E = 1 on the same action table: every episode's length and
    mean speed, the success rate and the rounded statistics of the result line."""
    from rvo3d_amd.policy import post_train
    fx = load(os.path.join(_ROOT, "tests", "golden", name))
    env = BatchedDroneEnv(world_of(fx), neighbors_num=10, env_train=bool(fx["env_train"]))
    pt = post_train(env, num_episodes=int(fx["num_episodes"]), max_ep_len=int(fx["max_ep_len"]),
                    acceler_vel=1.0, inf_print=False, std_factor=1e-5)
    got = pt.policy_test(policy=_TablePolicy(fx["table"], [0]), policy_name="scripted")
    env.close()
    arrived = fx["ep_arrived"].astype(bool)
    assert got["episodes"] == int(fx["num_episodes"]) == len(fx["ep_len"])
    assert got["ep_len"] == fx["ep_len"][arrived].tolist()
    np.testing.assert_allclose(got["speed"], fx["ep_speed"], rtol=1e-12)
    assert got["success_rate"] == pytest.approx(float(fx["success_rate"]), abs=1e-4)  # printed as xx.xx%
    assert got["success_rate"] == fx["ep_finished"].mean()
    for k in ("mean_len", "std_len", "average_speed", "std_speed"):
        assert got[k] == float(fx[k]), k


def test_post_train_batched_envs_contribute_equal_shares():
    """E > 1: every env contributes its first ceil(num_episodes / E) episodes, whenever they end
    (NOT "the first num_episodes episodes to end", which would favour collisions).  Env e replays
    the reference's action table from the start of the reference's episode e, so its episodes are
    the reference's episodes e, e + 1, ...; the pooled statistics must be those of exactly these."""
    from rvo3d_amd.policy import post_train
    fx = load(os.path.join(_ROOT, "tests", "golden", "post_train_world_4_eval.npz"))
    E, quota = 3, 2
    starts = np.concatenate([[0], np.cumsum(fx["ep_len"])])[:E].tolist()
    env = BatchedDroneEnv(world_of(fx, E), neighbors_num=10, env_train=False)
    pt = post_train(env, num_episodes=E * quota - 1, max_ep_len=int(fx["max_ep_len"]), acceler_vel=1.0,
                    inf_print=False)
    got = pt.policy_test(policy=_TablePolicy(fx["table"], starts))
    env.close()
    want = [e + k for e in range(E) for k in range(quota)]      # reference episode indices counted
    assert got["episodes"] == E * quota
    assert sorted(got["speed"]) == pytest.approx(sorted(fx["ep_speed"][want].tolist()), rel=1e-12)
    assert sorted(got["ep_len"]) == sorted(fx["ep_len"][want][fx["ep_arrived"][want].astype(bool)].tolist())
    assert got["success_rate"] == fx["ep_finished"][want].mean()
    # the short (collision) episodes are not over-represented: lengths 5 appear as often as in `want`
    assert (fx["ep_len"][want] == 5).sum() == 2


@pytest.mark.parametrize("E,N,nb,size", [(4096, 64, 0, (50.0, 50.0, 10.0)),
                                         (32768, 64, 0, (50.0, 50.0, 10.0)),
                                         (1024, 256, 50, (100.0, 100.0, 10.0))])
def test_full_size_launch_matches_oracle_on_sampled_envs(E, N, nb, size):
    """BASELINE configs 3, 4 (its whole 64 x 32768 workload on ONE GPU: the per-node shape of
    the 8-GPU config, exercised here because no multi-GPU node is available to the tests) and
    5 at their FULL sizes.  Environments never interact, so a random sample of envs of the
    full-size run must equal the oracle run on exactly those envs (same worlds, same actions):
    every output of every sampled drone, 6 fused steps."""
    T = 6
    world = synthetic_world(E, N, size, nb=nb)
    rng = np.random.default_rng(11)
    pick = np.sort(rng.choice(E, size=24 if N == 64 else 6, replace=False))
    env = BatchedDroneEnv(world, neighbors_num=10, action_decimals=2)
    sub = type(world)(world.waypoints[pick], world.n_points[pick], world.map_size, world.buildings)
    ref = orc.OracleEnv(sub.waypoints, sub.n_points, sub.map_size, sub.buildings, nm=10, threads=8)
    o0, c0 = env.observe(); r0, rc0 = ref.observe()
    tl = Tally(f"full_size/{N}x{E}", len(pick))
    tl.begin(ref.margin(), count=False)
    tl.check("obs0", eq_nan(o0.cpu().numpy()[pick], r0.astype(np.float32)))
    tl.end()
    for t in range(T):
        a = synthetic_actions(E, N, t)
        obs, cnt, rew, done, info, fin = env.step(torch.from_numpy(a.astype(np.float32)).cuda(), autoreset=True)
        ro, rcnt, rr, rd, ri, rf, rm = ref.step_autoreset(a[pick])
        tl.begin(ref.margin())
        tl.check(f"reset_mask t={t}", env.reset_mask.cpu().numpy()[pick] == rm)
        tl.check(f"done t={t}", done.cpu().numpy()[pick] == rd)
        tl.check(f"info t={t}", info.cpu().numpy()[pick] == ri)
        tl.check(f"finish t={t}", fin.cpu().numpy()[pick] == rf)
        tl.check(f"vo_count t={t}", cnt.cpu().numpy()[pick] == rcnt)
        tl.check(f"obs f32-exact t={t}", eq_nan(obs[pick].cpu().numpy(), ro.astype(np.float32)))
        tl.check(f"reward f32-exact t={t}", eq_nan(rew.cpu().numpy()[pick], rr.astype(np.float32)))
        tl.end()
    # whole-batch invariants of the padded observation: rows beyond vo_count are zero
    c = cnt.clamp(min=0)
    rows = obs[:, :, 12:].view(E, N, 10, 9)
    beyond = torch.arange(10, device=obs.device)[None, None, :] >= c[:, :, None]
    assert not bool((rows.abs().sum(dim=-1) * beyond).any())
    assert env.error_flags() == (1 if ref.nan_count else 0)
    env.close()
    print(tl.finish())


@pytest.mark.parametrize("N,E,radius", [(16, 128, 0.2), (24, 64, 0.5), (64, 48, 0.3), (100, 4, 0.4)])
def test_eval_mode_env_train_false_vs_oracle(N, E, radius):
    """env_train = False (the evaluator's env, train/policy_test.py:46; rvo_inter.py:144-150):
    collision at dis <= r - 0.2 + mr, evaluated from each side; pairs in the shell below
    r + mr that approach are where the reference raises "math domain error" - device and
    oracle (pinned by tests/golden/eval*.npz and calls_circle2.npz) both report the event and
    treat the pair as no VO.  Crowded worlds so that the shell is visited."""
    L = 5 + 2.2 * np.sqrt(N)
    world = synthetic_world(E, N, (L, L, 5.0), n_points=2, min_sep=2 * radius + 0.3, seed=17)
    for autoreset in (False, True):
        st = run_vs_oracle(world, T=30, nm=10, autoreset=autoreset, radius=radius, vlike=not autoreset,
                           env_train=False, seed=7, name=f"eval_mode/{N}x{E}_r{radius}_ar{int(autoreset)}")
        assert st["done"] > 0, st
        if not autoreset:
            # reported wherever the oracle saw it; in autoreset mode an env that reset skips the rows sweep of its
            # post-move state, the event is then reported by the collision sweep
            assert st["domain_pairs"] > 0, st


def test_more_than_2pow24_observation_rows():
    """E * N = 16.8 M rows (> 2^24): the row writer's index arithmetic is relative to the workgroup,
    so the rows far into a 6.8 GB observation tensor are written like the first ones (the chunk-indexed
    writer of round 1 decoded row 16 777 221 as row 5).  One world of 4100 envs tiled 64 times; the
    envs at the very end of the batch are compared with the oracle, and with the first copy."""
    E0, reps, N = 4100, 64, 64
    E = E0 * reps
    assert E * N > (1 << 24)
    w0 = synthetic_world(E0, N, (50.0, 50.0, 10.0))
    world = type(w0)(np.tile(w0.waypoints, (reps, 1, 1, 1)), np.tile(w0.n_points, (reps, 1)), w0.map_size,
                     w0.buildings)
    env = BatchedDroneEnv(world, neighbors_num=10, action_decimals=2)
    pick = np.array([0, 1, E0 - 1, E - E0, E - 2, E - 1])          # first copy and last copy
    sub = type(w0)(world.waypoints[pick], world.n_points[pick], w0.map_size, w0.buildings)
    ref = orc.OracleEnv(sub.waypoints, sub.n_points, sub.map_size, sub.buildings, nm=10, threads=6)
    env.observe(); ref.observe()
    tl = Tally("rows_beyond_2pow24", len(pick))
    for t in range(3):
        a0 = synthetic_actions(E0, N, t).astype(np.float32)
        a = torch.from_numpy(a0).cuda().repeat(reps, 1, 1)
        obs, cnt, rew, done, info, fin = env.step(a, autoreset=True)
        ro, rcnt, rr, rd, ri, rf, rm = ref.step_autoreset(np.tile(a0, (reps, 1, 1))[pick].astype(np.float64))
        tl.begin(ref.margin())
        tl.check(f"done t={t}", done[pick].cpu().numpy() == rd)
        tl.check(f"vo_count t={t}", cnt[pick].cpu().numpy() == rcnt)
        tl.check(f"obs t={t}", eq_nan(obs[pick].cpu().numpy(), ro.astype(np.float32)))
        tl.check(f"reward t={t}", eq_nan(rew[pick].cpu().numpy(), rr.astype(np.float32)))
        tl.end()
        # every copy of the world behaves like the first (identical inputs): whole-tensor check on the device
        o = obs.view(reps, E0, N, -1)
        assert bool(torch.equal(torch.nan_to_num(o[0]), torch.nan_to_num(o[reps - 1])))
        assert bool(torch.equal(cnt.view(reps, E0, N)[0], cnt.view(reps, E0, N)[reps // 2]))
    assert env.error_flags() == (1 if ref.nan_count else 0)
    env.close()
    tl.finish(rows=E * N)


def test_env_checkpoint_resume_is_bit_exact(tmp_path):
    """state_dict -> torch.save -> load (weights_only) -> load_state_dict resumes the rollout
    exactly: every output of the following steps is identical."""
    E, N = 32, 24
    world = synthetic_world(E, N, (20.0, 20.0, 8.0), n_points=3, seed=3)
    acts = [torch.from_numpy(synthetic_actions(E, N, t, 5).astype(np.float32)).cuda() for t in range(30)]
    env = BatchedDroneEnv(world, neighbors_num=10, action_decimals=2)
    env.observe()
    for t in range(12):
        env.step(acts[t], autoreset=True)
    torch.save(env.state_dict(), tmp_path / "env.pt")
    outs_a = []
    for t in range(12, 30):
        outs_a.append([x.clone() for x in env.step(acts[t], autoreset=True)] + [env.reset_mask.clone()])
    final_a = {k: v.clone() for k, v in env.get_state().items()}
    env.close()
    env2 = BatchedDroneEnv(world, neighbors_num=10, action_decimals=2)
    env2.load_state_dict(torch.load(tmp_path / "env.pt", weights_only=True))
    for t in range(12, 30):
        outs_b = list(env2.step(acts[t], autoreset=True)) + [env2.reset_mask]
        for xa, xb in zip(outs_a[t - 12], outs_b):
            assert torch.equal(torch.nan_to_num(xa.float(), nan=-7.0), torch.nan_to_num(xb.float(), nan=-7.0)), t
    for k, v in env2.get_state().items():
        assert torch.equal(v, final_a[k]), k
    env2.close()


def test_observation_buffer_alignment_paths_agree():
    """The row writer has a 16-B path (W even, buffer 16-B aligned) and a generic one; a
    caller-provided buffer that is only 8-B aligned must give the very same observations."""
    E, N = 37, 24
    world = synthetic_world(E, N, (14.0, 14.0, 6.0), seed=21, min_sep=0.6)
    env = BatchedDroneEnv(world, neighbors_num=10, action_decimals=2)
    flat = torch.full((E * N * env.W + 2,), 7.0, dtype=torch.float32, device="cuda")
    view = flat[2:].view(E, N, env.W)          # 8 bytes off a 16-B boundary
    assert view.data_ptr() % 16 == 8 and view.is_contiguous()
    cnt2 = torch.zeros((E, N), dtype=torch.int32, device="cuda")
    o1, c1 = env.observe()
    o1, c1 = o1.clone(), c1.clone()
    o2, c2 = env.observe(obs_out=view, cnt_out=cnt2)
    assert torch.equal(o1, o2) and torch.equal(c1, c2)
    for t in range(12):
        a = torch.from_numpy(synthetic_actions(E, N, t, 3).astype(np.float32)).cuda()
        st = env.state_dict()
        oa = [x.clone() for x in env.step_policy(a, autoreset=True)]
        env.load_state_dict(st)
        ob = env.step_policy(a, autoreset=True, obs_out=view, cnt_out=cnt2)
        for xa, xb in zip(oa, ob):
            assert torch.equal(torch.nan_to_num(xa.float(), nan=-7.0), torch.nan_to_num(xb.float(), nan=-7.0)), t
    assert float(flat[0]) == 7.0 and float(flat[1]) == 7.0  # nothing written in front of the view
    env.close()


def test_zz_knife_edge_share_over_all_tests():
    """Runs last: over every tally this session recorded, exempt samples stay below 1 % of the
    compared samples and the exempt samples that really differed below 0.1 %."""
    if not os.path.exists(_TALLY_FILE):
        pytest.skip("no tally file (tests ran in another order)")
    with open(_TALLY_FILE) as f:
        allr = json.load(f)
    allr.pop("TOTAL", None)
    n = sum(r["samples"] for r in allr.values())
    k = sum(r["knife"] for r in allr.values())
    km = sum(r["knife_mismatch"] for r in allr.values())
    dropped = sum(r["dropped_envs"] for r in allr.values())
    resyncs = sum(r.get("resyncs", 0) for r in allr.values())
    golden = {k: r for k, r in allr.items() if k.startswith("golden/")}
    assert all(r["steps_compared"] == r["steps"] for r in golden.values())
    tot = dict(tests=len(allr), samples=n, knife=k, knife_mismatch=km, dropped_envs=dropped, resyncs=resyncs,
               golden_scenarios=len(golden), golden_steps=sum(r["steps"] for r in golden.values()),
               golden_steps_compared=sum(r["steps_compared"] for r in golden.values()),
               knife_share=k / max(n, 1), mismatch_share=km / max(n, 1))
    _record_tally("TOTAL", tot)
    print(tot)
    assert n > 1_000_000 or len(allr) < 50  # the full suite compares > 1e6 drone-steps
    assert k <= KNIFE_SHARE * n and km <= 0.001 * n, tot
