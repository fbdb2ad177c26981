"""rvo3d_amd -- MI355X-native batched 3D-RVO multi-drone environment.

Host-side Python over the C-ABI in include/rvo3d.h (librvo3d_hip.so, HIP kernels
for gfx950).  Mirrors the reference's `uaisa_env` surface:

    from rvo3d_amd.drone_envs.mdin import mdin          # list API, one env
    from rvo3d_amd import BatchedDroneEnv                # [E, N] tensors

There is no CPU fallback: importing the native library fails loudly when the
HIP extension is missing, and every compute call needs a GPU.
"""
from ._lib import build_hip, lib, lib_path, RVO3DError  # noqa: F401
from .batched_env import BatchedDroneEnv  # noqa: F401
from .worlds import World, load_world_dir, synthetic_world, synthetic_actions  # noqa: F401
from . import sharding  # noqa: F401

__all__ = ["BatchedDroneEnv", "World", "load_world_dir", "synthetic_world",
           "synthetic_actions", "build_hip", "lib", "lib_path", "RVO3DError"]
