"""ctypes wrapper around the CPU ORACLE (oracle/rvo3d_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (rvo3d_amd) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librvo3d_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds). Returns the .so path."""
    src = os.path.join(_HERE, "rvo3d_oracle.c")
    hdr = os.path.join(_HERE, "rvo3d_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(f) > os.path.getmtime(_SO) for f in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp, ip, bp, vp = (C.POINTER(C.c_double), C.POINTER(C.c_int32),
                          C.POINTER(C.c_uint8), C.c_void_p)
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.c_int] * 6 + [dp]
        L.orc_destroy.argtypes = [vp]
        L.orc_load_world.argtypes = [vp, dp, ip, dp, dp, dp]
        L.orc_reset.argtypes = [vp, bp]
        L.orc_reset_drones.argtypes = [vp, bp]
        L.orc_observe.argtypes = [vp, dp, ip]
        L.orc_step.argtypes = [vp, dp, dp, ip, dp, bp, bp, bp]
        L.orc_step_autoreset.argtypes = [vp, dp, dp, ip, dp, bp, bp, bp, bp]
        L.orc_get_state.argtypes = [vp] + [dp] * 7 + [ip, bp, bp]
        L.orc_set_state.argtypes = [vp] + [dp] * 7 + [ip, bp, bp]
        L.orc_des_vel.argtypes = [vp, dp]
        L.orc_rvo_vel.argtypes = [vp, dp, C.c_double, dp]
        L.orc_vo_inf.argtypes = [vp, C.c_int, C.c_int, dp, dp, ip, ip, dp, ip]
        L.orc_get_margin.argtypes = [vp, dp, ip]
        L.orc_nan_count.restype = C.c_int64
        L.orc_nan_count.argtypes = [vp]
        L.orc_domain_count.restype = C.c_int64
        L.orc_domain_count.argtypes = [vp]
        L.orc_vo_circle2.argtypes = [C.c_int, dp, dp, dp, dp, ip, dp, ip, dp, ip]
        L.orc_set_threads.argtypes = [vp, C.c_int]
        L.orc_py_round2.restype = C.c_double
        L.orc_py_round2.argtypes = [C.c_double]
        L.orc_np_round.restype = C.c_double
        L.orc_np_round.argtypes = [C.c_double, C.c_int]
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _bp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint8))


STATE_FIELDS = ("pos", "vel", "yaw", "pitch", "real_len", "max_dev", "extra_len",
                "wp_idx", "arrive", "dest")


class OracleEnv:
    """E x N batched CPU oracle. Arrays in/out are numpy, shapes [E, N, ...]."""

    def __init__(self, waypoints, n_points, map_size, buildings=None, nm=10,
                 env_train=True, radius=None, priority=None, threads=1):
        wp = np.ascontiguousarray(waypoints, dtype=np.float64)
        assert wp.ndim == 4 and wp.shape[-1] == 3, "waypoints must be [E, N, P, 3]"
        self.E, self.N, self.P, _ = wp.shape
        npts = np.ascontiguousarray(n_points, dtype=np.int32).reshape(self.E, self.N)
        bld = np.zeros((0, 4)) if buildings is None else np.asarray(buildings, dtype=np.float64)
        bld = np.ascontiguousarray(bld.reshape(-1, 4))
        self.nb, self.nm = bld.shape[0], int(nm)
        self.W = 12 + 9 * self.nm
        ms = np.ascontiguousarray(map_size, dtype=np.float64)
        self._h = lib().orc_create(self.E, self.N, self.P, self.nb, self.nm,
                                   int(bool(env_train)), _dp(ms))
        if not self._h:
            raise ValueError("orc_create rejected the configuration")
        rad = None if radius is None else np.ascontiguousarray(radius, dtype=np.float64)
        pri = None if priority is None else np.ascontiguousarray(priority, dtype=np.float64)
        lib().orc_load_world(self._h, _dp(wp), _ip(npts), _dp(bld), _dp(rad), _dp(pri))
        lib().orc_set_threads(self._h, int(threads))
        self._keep = (wp, npts, bld, ms, rad, pri)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.orc_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def _outs(self):
        E, N = self.E, self.N
        return (np.empty((E, N, self.W)), np.empty((E, N), np.int32), np.empty((E, N)),
                np.empty((E, N), np.uint8), np.empty((E, N), np.uint8),
                np.empty((E, N), np.uint8))

    def reset(self, env_mask=None):
        m = None if env_mask is None else np.ascontiguousarray(env_mask, dtype=np.uint8)
        lib().orc_reset(self._h, _bp(m))

    def reset_drones(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.E, self.N)
        lib().orc_reset_drones(self._h, _bp(m))

    def observe(self):
        obs, cnt = np.empty((self.E, self.N, self.W)), np.empty((self.E, self.N), np.int32)
        lib().orc_observe(self._h, _dp(obs), _ip(cnt))
        return obs, cnt

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.E, self.N, 3)
        obs, cnt, rew, done, info, fin = self._outs()
        lib().orc_step(self._h, _dp(a), _dp(obs), _ip(cnt), _dp(rew), _bp(done), _bp(info),
                       _bp(fin))
        return obs, cnt, rew, done, info, fin

    def step_autoreset(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.E, self.N, 3)
        obs, cnt, rew, done, info, fin = self._outs()
        rm = np.empty((self.E, self.N), np.uint8)
        lib().orc_step_autoreset(self._h, _dp(a), _dp(obs), _ip(cnt), _dp(rew), _bp(done),
                                 _bp(info), _bp(fin), _bp(rm))
        return obs, cnt, rew, done, info, fin, rm

    def get_state(self):
        E, N = self.E, self.N
        s = dict(pos=np.empty((E, N, 3)), vel=np.empty((E, N, 3)), yaw=np.empty((E, N)),
                 pitch=np.empty((E, N)), real_len=np.empty((E, N)), max_dev=np.empty((E, N)),
                 extra_len=np.empty((E, N)), wp_idx=np.empty((E, N), np.int32),
                 arrive=np.empty((E, N), np.uint8), dest=np.empty((E, N), np.uint8))
        lib().orc_get_state(self._h, _dp(s["pos"]), _dp(s["vel"]), _dp(s["yaw"]),
                            _dp(s["pitch"]), _dp(s["real_len"]), _dp(s["max_dev"]),
                            _dp(s["extra_len"]), _ip(s["wp_idx"]), _bp(s["arrive"]),
                            _bp(s["dest"]))
        return s

    def set_state(self, **kw):
        c = {}
        for k in STATE_FIELDS:
            v = kw.get(k)
            if v is None:
                c[k] = None
            elif k == "wp_idx":
                c[k] = np.ascontiguousarray(v, dtype=np.int32)
            elif k in ("arrive", "dest"):
                c[k] = np.ascontiguousarray(v, dtype=np.uint8)
            else:
                c[k] = np.ascontiguousarray(v, dtype=np.float64)
        lib().orc_set_state(self._h, _dp(c["pos"]), _dp(c["vel"]), _dp(c["yaw"]),
                            _dp(c["pitch"]), _dp(c["real_len"]), _dp(c["max_dev"]),
                            _dp(c["extra_len"]), _ip(c["wp_idx"]), _bp(c["arrive"]),
                            _bp(c["dest"]))

    def des_vel(self):
        out = np.empty((self.E, self.N, 3))
        lib().orc_des_vel(self._h, _dp(out))
        return out

    def rvo_vel(self, vmax=(2.0, 2.0, 2.0), acceler=0.5):
        """Classical RVO velocity selection (reciprocal_vel_obs.py as intended): [E, N, 3]."""
        out = np.empty((self.E, self.N, 3))
        vm = np.ascontiguousarray(vmax, dtype=np.float64)
        lib().orc_rvo_vel(self._h, _dp(vm), float(acceler), _dp(out))
        return out

    def vo_inf(self, e, i, action):
        """rvo_inter.config_vo_inf for drone (e, i): (rows[k,9], flag, tmin, collision)."""
        a = np.ascontiguousarray(action, dtype=np.float64)
        rows = np.zeros((max(self.nm, 1), 9))
        cnt, flag, col = (np.zeros(1, np.int32) for _ in range(3))
        tmin = np.zeros(1)
        lib().orc_vo_inf(self._h, int(e), int(i), _dp(a), _dp(rows), _ip(cnt), _ip(flag),
                         _dp(tmin), _ip(col))
        return rows[:cnt[0]].copy(), bool(flag[0]), float(tmin[0]), bool(col[0])

    def margin(self):
        """Per-drone minimum decision margin of the last observe/step call."""
        out = np.empty((self.E, self.N))
        lib().orc_get_margin(self._h, _dp(out), None)
        return out

    def margin_sites(self):
        """(margin, source line in rvo3d_oracle.c of the decision that set it)."""
        out, site = np.empty((self.E, self.N)), np.zeros((self.E, self.N), np.int32)
        lib().orc_get_margin(self._h, _dp(out), _ip(site))
        return out, site

    @property
    def nan_count(self):
        return int(lib().orc_nan_count(self._h))

    @property
    def domain_count(self):
        """Pairs on which the reference raises "math domain error" (env_train=False shell)."""
        return int(lib().orc_domain_count(self._h))

    def set_threads(self, n):
        lib().orc_set_threads(self._h, int(n))


def vo_circle2(env_train, self8, other8, action):
    """rvo_inter.config_vo_circle2 for one pair: (obs9, flag, exp_time, collision, min_dis,
    domain_error) - the reference's five return values plus "the reference raises here"."""
    s8 = np.ascontiguousarray(self8, dtype=np.float64)[:8].copy()
    o8 = np.ascontiguousarray(other8, dtype=np.float64)[:8].copy()
    a = np.ascontiguousarray(action, dtype=np.float64)
    obs = np.zeros(9)
    flag, col, dom = (np.zeros(1, np.int32) for _ in range(3))
    t, md = np.zeros(1), np.zeros(1)
    lib().orc_vo_circle2(int(bool(env_train)), _dp(s8), _dp(o8), _dp(a), _dp(obs), _ip(flag),
                         _dp(t), _ip(col), _dp(md), _ip(dom))
    return obs, bool(flag[0]), float(t[0]), bool(col[0]), float(md[0]), bool(dom[0])


# ---- reciprocal_vel_obs, call by call (the methods of the class that run on an instance) ----
def rvo_distance(p1, p2):
    L = lib()
    L.orc_rvo_distance.restype = C.c_double
    a, b = (np.ascontiguousarray(x, dtype=np.float64) for x in (p1, p2))
    return float(L.orc_rvo_distance(_dp(a), _dp(b)))


def rvo_preprocess(agent3, drones, buildings):
    """reciprocal_vel_obs.preprocess: (keep_drone[n], keep_building[nb]) as bool arrays."""
    a = np.ascontiguousarray(agent3, dtype=np.float64)
    d = np.ascontiguousarray(np.asarray(drones, dtype=np.float64).reshape(-1, 3))
    b = np.ascontiguousarray(np.asarray(buildings, dtype=np.float64).reshape(-1, 4))
    kd, kb = np.zeros(len(d), np.uint8), np.zeros(len(b), np.uint8)
    lib().orc_rvo_preprocess(_dp(a), _dp(d), len(d), _dp(b), len(b), _bp(kd), _bp(kb))
    return kd.astype(bool), kb.astype(bool)


def rvo_penalty(vel, vel_des, agent8, odro8, factor=1.0):
    L = lib()
    L.orc_rvo_penalty.restype = C.c_double
    L.orc_rvo_penalty.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_double]
    v, vd, a = (np.ascontiguousarray(x, dtype=np.float64) for x in (vel, vel_des, agent8))
    o = np.ascontiguousarray(np.asarray(odro8, dtype=np.float64).reshape(-1, 8))
    return float(L.orc_rvo_penalty(_dp(v), _dp(vd), _dp(a), _dp(o), len(o), float(factor)))


def rvo_candidates(vel3, vmax, acceler):
    L = lib()
    L.orc_rvo_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int]
    v, vm = (np.ascontiguousarray(x, dtype=np.float64) for x in (vel3, vmax))
    out = np.zeros((512, 3))
    n = L.orc_rvo_candidates(_dp(v), _dp(vm), float(acceler), _dp(out), 512)
    return out[:n].copy()


def rvo_select_inside(inside, agent11, odro8):
    L = lib()
    L.orc_rvo_select_inside.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    ins = np.ascontiguousarray(np.asarray(inside, dtype=np.float64).reshape(-1, 3))
    a = np.ascontiguousarray(agent11, dtype=np.float64)
    o = np.ascontiguousarray(np.asarray(odro8, dtype=np.float64).reshape(-1, 8))
    return int(L.orc_rvo_select_inside(_dp(ins), len(ins), _dp(a), _dp(o), len(o)))
