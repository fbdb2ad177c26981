"""BatchedDroneEnv: E environments x N drones stepped by one HIP launch.

Thin host object over the C-ABI (include/rvo3d.h).  Tensors live on the GPU
(PyTorch-ROCm is only the allocator / stream provider); every call enqueues on
torch's current stream and returns without synchronising.

Batched counterpart of the reference façade `mdin` (mdin.py:7-48):
    drone_step      -> step(actions)            (mdin.py:19-30)
    drone_reset     -> reset() ; observe()      (mdin.py:38, ir_gym.py:360)
    drone_reset_one -> reset_drones(mask)       (mdin.py:43)
    ir_gym.env_observation -> observe()         (ir_gym.py:372)
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .worlds import World

_F64 = {"pos": 3, "vel": 3, "yaw": 0, "pitch": 0, "real_len": 0, "max_dev": 0, "extra_len": 0}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedDroneEnv:
    def __init__(self, world: World, neighbors_num: int = 10, env_train: bool = True,
                 device="cuda:0", action_decimals: int = -1, radius=None, priority=None,
                 acceler: float = 0.5, reward_f64: bool = False):
        if not torch.cuda.is_available():
            raise RuntimeError("rvo3d_amd needs a GPU: there is no CPU fallback "
                               "(the CPU oracle under oracle/ is test infrastructure only)")
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:  # "cuda" -> "cuda:<current>"
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.E, self.N, self.P = world.shape
        self.nm = int(neighbors_num)
        self.W = 12 + 9 * self.nm
        self.acceler = acceler  # ir_gym.acceler (ir_gym.py:34), used by the trainer glue
        self.env_train = bool(env_train)  # rvo_inter.env_train (rvo_inter.py:14)
        self.world = world
        L = _lib.lib()
        cfg = _lib.Config(self.E, self.N, self.P, int(world.buildings.shape[0]), self.nm,
                          int(bool(env_train)), self.device.index or 0, int(action_decimals),
                          (C.c_double * 3)(*[float(x) for x in world.map_size]))
        h = C.c_void_p()
        _lib.check(L.rvo3d_create(C.byref(cfg), C.byref(h)), "rvo3d_create")
        self._h = h
        wp = np.ascontiguousarray(world.waypoints, dtype=np.float64)
        npts = np.ascontiguousarray(world.n_points, dtype=np.int32)
        bld = np.ascontiguousarray(world.buildings, dtype=np.float64)
        rad = None if radius is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(radius, dtype=np.float64), (self.E, self.N)))
        pri = None if priority is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(priority, dtype=np.float64), (self.E, self.N)))
        hp = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
        with torch.cuda.device(self.device):
            _lib.check(L.rvo3d_load_world(h, hp(wp), hp(npts), hp(bld) if bld.size else None,
                                          hp(rad), hp(pri), self._stream()), "rvo3d_load_world")
        E, N, dev = self.E, self.N, self.device
        self.obs = torch.zeros((E, N, self.W), dtype=torch.float32, device=dev)
        self.vo_count = torch.zeros((E, N), dtype=torch.int32, device=dev)
        self.reward = torch.zeros((E, N), dtype=torch.float32, device=dev)
        self.done = torch.zeros((E, N), dtype=torch.uint8, device=dev)
        self.info = torch.zeros((E, N), dtype=torch.uint8, device=dev)
        self.finish = torch.zeros((E, N), dtype=torch.uint8, device=dev)
        self.reset_mask = torch.zeros((E, N), dtype=torch.uint8, device=dev)
        # optional float64 copy of the reward, as the reference returns it (mdin.py:28)
        self.reward64 = None
        if reward_f64:
            self.reward64 = torch.zeros((E, N), dtype=torch.float64, device=dev)
            _lib.check(L.rvo3d_set_reward_f64(h, _ptr(self.reward64)), "rvo3d_set_reward_f64")

    # -- plumbing ---------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().rvo3d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _actions(self, actions):
        a = actions
        if not torch.is_tensor(a):
            a = torch.as_tensor(np.asarray(a), device=self.device)
        if a.device != self.device:
            a = a.to(self.device)
        if a.dtype not in (torch.float32, torch.float64):
            a = a.to(torch.float64)
        if tuple(a.shape) != (self.E, self.N, 3):
            raise AssertionError(f"actions must have shape ({self.E}, {self.N}, 3)")  # drone.py:101
        return a.contiguous(), (1 if a.dtype == torch.float64 else 0)

    # -- reference surface ----------------------------------------------------------
    def step(self, actions, autoreset: bool = False):
        """mdin.drone_step for every env.  Returns views of the handle-owned
        output tensors (obs f32 [E,N,W], vo_count, reward f32, done, info, finish)."""
        a, dt = self._actions(actions)
        L = _lib.lib()
        if autoreset:
            rc = L.rvo3d_step_autoreset(self._h, _ptr(a), dt, _ptr(self.obs), _ptr(self.vo_count),
                                        _ptr(self.reward), _ptr(self.done), _ptr(self.info),
                                        _ptr(self.finish), _ptr(self.reset_mask), self._stream())
        else:
            rc = L.rvo3d_step(self._h, _ptr(a), dt, _ptr(self.obs), _ptr(self.vo_count),
                              _ptr(self.reward), _ptr(self.done), _ptr(self.info),
                              _ptr(self.finish), self._stream())
        _lib.check(rc, "rvo3d_step")
        self._last_actions = a  # keep the borrowed buffer alive until the launch ran
        return self.obs, self.vo_count, self.reward, self.done, self.info, self.finish

    def step_policy(self, a_inc, autoreset: bool = True, obs_out=None, cnt_out=None):
        """Step from raw policy samples: the trainer's glue (multi_ppo.py:196-210,
        a = round(a_inc, 2); action = round(acceler * a + vel, 2)) runs on the device.
        a_inc: float32 [E, N, 3]."""
        a = torch.as_tensor(a_inc, device=self.device).to(torch.float32).contiguous()
        if tuple(a.shape) != (self.E, self.N, 3):
            raise AssertionError(f"a_inc must have shape ({self.E}, {self.N}, 3)")
        o, c = self._outs(obs_out, cnt_out)
        rc = _lib.lib().rvo3d_step_policy(
            self._h, _ptr(a), C.c_float(self.acceler), _ptr(o), _ptr(c),
            _ptr(self.reward), _ptr(self.done), _ptr(self.info), _ptr(self.finish),
            _ptr(self.reset_mask), 1 if autoreset else 0, self._stream())
        _lib.check(rc, "rvo3d_step_policy")
        self._last_actions = a
        return o, c, self.reward, self.done, self.info, self.finish

    def reset(self, env_mask=None):
        m = None
        if env_mask is not None:
            m = torch.as_tensor(env_mask, device=self.device).to(torch.uint8).contiguous()
        _lib.check(_lib.lib().rvo3d_reset(self._h, _ptr(m), self._stream()), "rvo3d_reset")
        self._keep = m

    def reset_drones(self, mask):
        m = torch.as_tensor(mask, device=self.device).to(torch.uint8).reshape(self.E, self.N)
        m = m.contiguous()
        _lib.check(_lib.lib().rvo3d_reset_drones(self._h, _ptr(m), self._stream()),
                   "rvo3d_reset_drones")
        self._keep = m

    def _outs(self, obs_out, cnt_out):
        """Observation outputs: the env's own tensors, or caller-provided ones (e.g. the next
        slot of a rollout buffer: the kernel writes there directly, nothing is copied)."""
        if obs_out is None:
            return self.obs, self.vo_count
        if (obs_out.dtype != torch.float32 or cnt_out.dtype != torch.int32 or not obs_out.is_contiguous()
                or not cnt_out.is_contiguous() or tuple(obs_out.shape) != (self.E, self.N, self.W)
                or tuple(cnt_out.shape) != (self.E, self.N) or obs_out.device != self.device):
            raise AssertionError("obs_out / cnt_out must be contiguous float32 [E,N,W] / int32 [E,N] on the env's device")
        return obs_out, cnt_out

    def observe(self, obs_out=None, cnt_out=None):
        o, c = self._outs(obs_out, cnt_out)
        _lib.check(_lib.lib().rvo3d_observe(self._h, _ptr(o), _ptr(c), self._stream()), "rvo3d_observe")
        return o, c

    def des_vel(self):
        out = torch.empty((self.E, self.N, 3), dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().rvo3d_des_vel(self._h, _ptr(out), self._stream()), "rvo3d_des_vel")
        return out

    def rvo_vel(self, vmax=(2.0, 2.0, 2.0), acceler: float = 0.5):
        """reciprocal_vel_obs.cal_vel for every drone (reciprocal_vel_obs.py:19-31, as
        intended: see include/rvo3d.h): the classical RVO velocity, [E, N, 3] float64."""
        out = torch.empty((self.E, self.N, 3), dtype=torch.float64, device=self.device)
        vm = (C.c_double * 3)(*[float(x) for x in vmax])
        _lib.check(_lib.lib().rvo3d_rvo_vel(self._h, vm, float(acceler), _ptr(out), self._stream()),
                   "rvo3d_rvo_vel")
        return out

    # -- state ------------------------------------------------------------------------
    def get_state(self):
        E, N, dev = self.E, self.N, self.device
        s = {k: torch.empty((E, N, 3) if d else (E, N), dtype=torch.float64, device=dev)
             for k, d in _F64.items()}
        s["wp_idx"] = torch.empty((E, N), dtype=torch.int32, device=dev)
        s["arrive"] = torch.empty((E, N), dtype=torch.uint8, device=dev)
        s["dest"] = torch.empty((E, N), dtype=torch.uint8, device=dev)
        order = ("pos", "vel", "yaw", "pitch", "real_len", "max_dev", "extra_len", "wp_idx",
                 "arrive", "dest")
        _lib.check(_lib.lib().rvo3d_get_state(self._h, *[_ptr(s[k]) for k in order],
                                              self._stream()), "rvo3d_get_state")
        return s

    @property
    def vel(self):
        """drone_list[i].vel of every drone (the trainer's / evaluator's reach-through,
        multi_ppo.py:202, post_train.py:72): a fresh [E, N, 3] float64 copy."""
        v = torch.empty((self.E, self.N, 3), dtype=torch.float64, device=self.device)
        none = [None] * 8
        _lib.check(_lib.lib().rvo3d_get_state(self._h, None, _ptr(v), *none, self._stream()),
                   "rvo3d_get_state")
        return v

    def set_state(self, **kw):
        order = ("pos", "vel", "yaw", "pitch", "real_len", "max_dev", "extra_len", "wp_idx",
                 "arrive", "dest")
        t = {}
        for k in order:
            v = kw.get(k)
            if v is None:
                t[k] = None
                continue
            dt = torch.int32 if k == "wp_idx" else (torch.uint8 if k in ("arrive", "dest")
                                                    else torch.float64)
            t[k] = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v,
                                   device=self.device).to(dt).contiguous()
        _lib.check(_lib.lib().rvo3d_set_state(self._h, *[_ptr(t[k]) for k in order],
                                              self._stream()), "rvo3d_set_state")
        self._keep = t

    # -- checkpoint / resume -----------------------------------------------------------
    def state_dict(self):
        """The complete mutable state of every drone (drone.py:14-82) as CPU tensors, plus the
        shape it belongs to: enough to resume a rollout bit-exactly (the values the step keeps
        on file between calls are derived data and are rebuilt by load_state_dict)."""
        sd = {k: v.cpu() for k, v in self.get_state().items()}
        sd["shape"] = torch.tensor([self.E, self.N, self.P, self.nm])
        return sd

    def load_state_dict(self, sd):
        shape = [int(x) for x in sd["shape"]]
        if shape != [self.E, self.N, self.P, self.nm]:
            raise ValueError(f"checkpoint is for E,N,P,nm = {shape}, this env has "
                             f"{[self.E, self.N, self.P, self.nm]}")
        self.set_state(**{k: v for k, v in sd.items() if k != "shape"})

    def error_flags(self) -> int:
        """Reads and clears the device error word (synchronises)."""
        f = C.c_uint32(0)
        _lib.check(_lib.lib().rvo3d_error_flags(self._h, C.byref(f), self._stream()),
                   "rvo3d_error_flags")
        return int(f.value)

    def check_finite(self):
        """The reference raises ValueError when an observation holds NaN/Inf
        (ir_gym.py:232-239); opt-in here because it synchronises."""
        f = self.error_flags()
        if f & 2:  # RVO3D_FLAG_DOMAIN_ERROR: vel_obs3D.py:13 asin((r + mr) / norm), env_train=False
            raise ValueError("math domain error")
        if f & 1:
            raise ValueError("observation contains NaN/Inf")

    def kernel_name(self, mode="step_autoreset") -> str:
        """The env_kernel instantiation the library launches for `mode` (rvo3d_kernel_name)."""
        m = {"observe": 0, "step": 1, "step_autoreset": 2}[mode]
        buf = C.create_string_buffer(128)
        _lib.check(_lib.lib().rvo3d_kernel_name(self._h, m, buf, 128), "rvo3d_kernel_name")
        return buf.value.decode()

    def launch_info(self):
        v = [C.c_int32(0) for _ in range(4)]
        _lib.check(_lib.lib().rvo3d_launch_info(self._h, *[C.byref(x) for x in v]), "launch_info")
        return dict(threads=v[0].value, envs_per_block=v[1].value, blocks=v[2].value,
                    lds_bytes=v[3].value)
