"""`mdin`: the reference's Gym-style façade (uaisa_env/drone_envs/mdin.py:7-48)
over the HIP step.  Same constructor, same list-in / list-out methods, one env.

    env = mdin(base_dir=".../world_4")           # directory holding data_1.json
    obs_list = env.drone_reset(False)
    obs_list, reward_list, done_list, info_list, finish_list = env.drone_step(action_list)

`env.ir_gym` exposes what the reference trainer reaches into (multi_ppo.py:110,
202-212, 242-246): drone_num, drone_list[i].vel / .state / .rvo_vel, acceler,
env_observation(), cal_des_list(), indicators_deviation(), indicators_extra_len().
"""
from __future__ import annotations

import numpy as np
import torch

from ..batched_env import BatchedDroneEnv
from ..worlds import World, load_world_dir, world_from_dict


class _Box:
    """Stand-in for gym.spaces.Box (ir_gym.py:31-32): shape/low/high/dtype only."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.dtype = low, high, dtype
        self.shape = tuple(shape) if shape is not None else tuple(np.shape(low))


class _DroneView:
    """Read view of one drone (drone.py:14-82 attributes the trainer touches)."""

    def __init__(self, owner, i):
        self._o, self.id = owner, i
        self.rvo_vel = [0, 0, 0]  # drawing-only attribute the trainer writes (multi_ppo.py:207)

    def _s(self, k):
        return self._o._state()[k][0, self.id]

    vel = property(lambda self: self._s("vel"))
    state = property(lambda self: self._s("pos"))
    yaw = property(lambda self: float(self._s("yaw")))
    pitch = property(lambda self: float(self._s("pitch")))
    i = property(lambda self: int(self._s("wp_idx")))
    real_route_len = property(lambda self: float(self._s("real_len")))
    max_deviation = property(lambda self: float(self._s("max_dev")))
    extra_len = property(lambda self: float(self._s("extra_len")))
    arrive_flag = property(lambda self: bool(self._s("arrive")))
    destination_arrive_flag = property(lambda self: bool(self._s("dest")))


class _IrGym:
    """The slice of `ir_gym` (ir_gym.py:17) reachable through `mdin.ir_gym`."""

    def __init__(self, env: BatchedDroneEnv, acceler, neighbors_region, neighbors_num, env_train):
        self._env = env
        self.drone_num = env.N
        self.acceler = acceler
        self.nr, self.nm, self.env_train = neighbors_region, neighbors_num, env_train
        self.map_size = list(env.world.map_size)
        self.building_list = env.world.buildings.tolist()
        self.observation_space = _Box(-np.inf, np.inf, shape=(21,), dtype=np.float32)
        self.action_space = _Box(np.array([-1, -1, -1]), np.array([1, 1, 1]), dtype=np.float32)
        self.drone_list = [_DroneView(self, i) for i in range(env.N)]
        self._cache = None

    def _state(self):
        """Host copy of the whole state, fetched once per step / reset (every attribute read of
        every `drone_list[i]` between two steps is served from it)."""
        if self._cache is None:
            self._cache = {k: v.cpu().numpy() for k, v in self._env.get_state().items()}
        return self._cache

    def _ragged(self, obs, cnt):
        o = obs[0].cpu().numpy().astype(np.float64)
        c = cnt[0].cpu().numpy()
        # k = 0 still yields one zero VO row (ir_gym.py:211-214)
        return [o[i, :12 + 9 * max(int(c[i]), 1)].copy() for i in range(len(c))]

    def _raise_domain_error(self):
        # env_train=False only: get_alpha raises out of the reference's step / observation when
        # a pair inside r + mr approaches (vel_obs3D.py:13); the device reports it in its error word
        if not self.env_train and (self._env.error_flags() & 2):
            raise ValueError("math domain error")

    def env_observation(self):                      # ir_gym.py:372-383
        obs, cnt = self._env.observe()
        self._cache = None
        self._raise_domain_error()
        return self._ragged(obs, cnt)

    def env_reset(self):                            # ir_gym.py:360-367
        self._env.reset()
        return self.env_observation()

    def cal_des_list(self):                         # ir_gym.py:44-46
        return list(self._env.des_vel()[0].cpu().numpy())

    def indicators_deviation(self):                 # ir_gym.py:414-416
        return list(self._state()["max_dev"][0])

    def indicators_extra_len(self):                 # ir_gym.py:418-420
        return list(self._state()["extra_len"][0])

    def render(self, *a, **k):                      # plotting is out of scope (env_plot.py)
        return None


class mdin:
    def __init__(self, world_name=None, neighbors_region=5, neighbors_num=10, vxmax=2, vymax=2,
                 vzmax=2, env_train=True, acceler=0.5, base_dir=None, world=None,
                 device="cuda:0", **kwargs):
        if world is None:
            if base_dir is None:
                raise ValueError("mdin needs base_dir=<directory with data_1.json> "
                                 "(env_base.py:15) or world=<dict/World>")
            world = load_world_dir(base_dir)
        elif isinstance(world, dict):
            world = world_from_dict(world)
        assert isinstance(world, World) and world.shape[0] == 1, "mdin is the one-env façade"
        self._env = BatchedDroneEnv(world, neighbors_num=neighbors_num, env_train=env_train,
                                    device=device, acceler=acceler, reward_f64=True)
        self.ir_gym = _IrGym(self._env, acceler, neighbors_region, neighbors_num, env_train)
        self.observation_space = self.ir_gym.observation_space
        self.action_space = self.ir_gym.action_space
        self.neighbors_region = neighbors_region
        self.rvo_observation_list = []
        self.drow_rvo_flag = True

    def drone_step(self, action, **kwargs):         # mdin.py:19-30
        if not isinstance(action, list):
            action = [action]
        for a in action:                            # drone.py:98-101 (lists are rejected there)
            assert isinstance(a, np.ndarray) and a.shape == (3,)
        act = torch.from_numpy(np.asarray(action, dtype=np.float64)[None])
        obs, cnt, rew, done, info, fin = self._env.step(act)
        g = self.ir_gym
        g._cache = None
        g._raise_domain_error()
        obs_list = g._ragged(obs, cnt)
        if not all(np.isfinite(o).all() for o in obs_list):
            raise ValueError("observation contains NaN/Inf")   # ir_gym.py:232-239
        return (obs_list, list(self._env.reward64[0].cpu().numpy()),   # float64, as mdin.py:28 returns it
                [bool(x) for x in done[0].cpu().numpy()], [bool(x) for x in info[0].cpu().numpy()],
                [bool(x) for x in fin[0].cpu().numpy()])

    def drone_reset(self, ifrender):                # mdin.py:38-41
        return self.ir_gym.env_reset()

    def drone_reset_one(self, ifrender, id):        # mdin.py:43-46
        m = np.zeros((1, self._env.N), np.uint8)
        m[0, id] = 1
        self._env.reset_drones(m)
        self.ir_gym._cache = None

    def drone_render(self, *a, **k):                # mdin.py:32 (rendering out of scope)
        return None

    def drone_show(self):
        return None

    def close(self):
        self._env.close()
