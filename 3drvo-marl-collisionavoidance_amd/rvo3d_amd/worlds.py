"""World data: the reference's `data_1.json` format (env_base.py:26-47) and the
synthetic generator of SURVEY.md section 8(d) used by bench.py and the tests."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field

import numpy as np


@dataclass
class World:
    """Routes for E envs x N drones plus buildings shared by all envs."""
    waypoints: np.ndarray          # [E, N, P, 3] float64 (padded with the destination)
    n_points: np.ndarray           # [E, N] int32
    map_size: np.ndarray           # [3]
    buildings: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))  # [nb, 4] x,y,h,r

    @property
    def shape(self):
        return self.waypoints.shape[:3]


def world_from_dict(d: dict, num_envs: int = 1) -> World:
    """`data_1.json` keys: drone_num, map_size, waypoints_list, n_points_list,
    building_list (env_base.py:35-39).  The one scenario is replicated E times."""
    wl = d["waypoints_list"]
    N = int(d.get("drone_num", len(wl)))
    P = max(len(w) for w in wl)
    wp = np.zeros((N, P, 3))
    for i, w in enumerate(wl[:N]):
        w = np.asarray(w, dtype=np.float64)
        wp[i, :len(w)] = w
        wp[i, len(w):] = w[-1]
    npts = np.asarray(d["n_points_list"][:N], dtype=np.int32)
    bld = np.asarray(d.get("building_list", []), dtype=np.float64).reshape(-1, 4)
    return World(np.repeat(wp[None], num_envs, 0).copy(), np.repeat(npts[None], num_envs, 0).copy(),
                 np.asarray(d["map_size"], dtype=np.float64), bld)


def load_world_dir(base_dir: str, num_envs: int = 1) -> World:
    """env_base.load_data (env_base.py:26-47).  E3d.npy / E3d_safe.npy are
    optional: the reference loads them but never uses them numerically."""
    with open(os.path.join(base_dir, "data_1.json")) as f:
        return world_from_dict(json.load(f), num_envs)


def synthetic_world(E: int, N: int, map_size, n_points: int = 2, nb: int = 0, seed: int = 1234,
                    min_sep: float = 1.0) -> World:
    """per env e, rng = Philox(seed, stream e); starts/ends
    ~U([1, L-1]^2 x [1, H-1]) rounded to 2 decimals, starts at least min_sep
    apart; buildings [x,y ~U(2, L-2), h ~U(3, H), r ~U(0.5, 1.5)] shared."""
    L, Wd, H = map_size
    lo = np.array([1.0, 1.0, 1.0])
    hi = np.array([L - 1.0, Wd - 1.0, H - 1.0])
    wp = np.empty((E, N, n_points, 3))
    for e in range(E):
        rng = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, e]))
        pts = np.round(rng.uniform(lo, hi, (N, n_points, 3)), 2)
        starts = pts[:, 0]
        for i in range(1, N):  # rejection: keep starts min_sep apart
            tries = 0
            while tries < 200 and np.min(np.linalg.norm(starts[:i] - starts[i], axis=1)) < min_sep:
                starts[i] = np.round(rng.uniform(lo, hi), 2)
                tries += 1
        wp[e] = pts
    brng = np.random.Generator(np.random.Philox(key=seed + 1))
    bld = np.round(np.stack([brng.uniform(2, L - 2, nb), brng.uniform(2, Wd - 2, nb),
                             brng.uniform(3, H, nb), brng.uniform(0.5, 1.5, nb)], axis=1), 2) \
        if nb else np.zeros((0, 4))
    return World(wp, np.full((E, N), n_points, np.int32), np.asarray(map_size, dtype=np.float64), bld)


def synthetic_actions(E: int, N: int, step: int, seed: int = 1234) -> np.ndarray:
    """round(U(-1,1)^3 * [1, 0.3, 0.15], 2) from Philox(seed, step)."""
    rng = np.random.Generator(np.random.Philox(key=seed + 7, counter=[0, 0, 0, step]))
    return np.round(rng.uniform(-1, 1, (E, N, 3)) * np.array([1.0, 0.3, 0.15]), 2)
