"""Loader for librvo3d_hip.so (the C-ABI of include/rvo3d.h) via ctypes.

The library is built in-tree (next to this file) by `build_hip()`:
    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared ...
No fallback exists: a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_SO = os.path.join(_HERE, "librvo3d_hip.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
               "-std=c++17"]


class RVO3DError(RuntimeError):
    """A C-ABI call returned a negative status."""


def lib_path() -> str:
    return _SO


def sources():
    hpp = sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".hpp"))
    return [os.path.join(_CSRC, "rvo3d_capi.hip")] + hpp + [os.path.join(_ROOT, "include", "rvo3d.h")]


def build_hip(force: bool = False, verbose: bool = False, out: str | None = None,
              extra_flags=()) -> str:
    """Compile the HIP extension for gfx950 (cross-compiles without a GPU).  `out` /
    `extra_flags` are for tools/diaglib.py (the -DRVO3D_DIAG build, kept outside the package)."""
    src = sources()
    so = out or _SO
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(f) > os.path.getmtime(so) for f in src)
    if force or stale:
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + ["-o", so, src[0]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return so


def use_library(path: str) -> None:
    """tools/ only: load another build of the library (the diagnostics build) instead of the
    product one.  Must be called before the first lib()."""
    global _SO, _lib
    if _lib is not None:
        raise RuntimeError("the library is already loaded")
    _SO = path


class Config(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_drones", C.c_int32), ("max_points", C.c_int32),
                ("num_buildings", C.c_int32), ("neighbors_num", C.c_int32),
                ("env_train", C.c_int32), ("device", C.c_int32), ("action_decimals", C.c_int32),
                ("map_size", C.c_double * 3)]


class PolicyHeads(C.Structure):
    _fields_ = [("h_pi", C.c_void_p), ("h_v", C.c_void_p), ("ld_pi", C.c_int64), ("ld_v", C.c_int64),
                ("dtype", C.c_int32), ("hidden", C.c_int32), ("tanh_out", C.c_int32), ("reserved", C.c_int32),
                ("w_pi", C.c_void_p), ("b_pi", C.c_void_p), ("w_v", C.c_void_p), ("b_v", C.c_void_p),
                ("log_std", C.c_void_p)]


class GruReader(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_ih_f", "b_ih_f", "b_hh_f", "w_ih_r", "b_ih_r", "b_hh_r", "ln_w", "ln_b")] + \
               [("hidden", C.c_int32), ("in_dim", C.c_int32), ("state_dim", C.c_int32), ("ln_eps", C.c_float)]


class MlpWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2", "w3", "b3")]


class RnnPolicy(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_ih_f", "w_hh_f", "b_ih_f", "b_hh_f", "w_ih_r", "w_hh_r", "b_ih_r", "b_hh_r",
                                          "ln_w", "ln_b")] + \
               [("hidden", C.c_int32), ("in_dim", C.c_int32), ("state_dim", C.c_int32), ("slots", C.c_int32),
                ("ln_eps", C.c_float), ("reserved", C.c_int32), ("pi", MlpWeights), ("v", MlpWeights)]


RVO3D_F32, RVO3D_F64, RVO3D_BF16 = 0, 1, 2


class StateView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("px", "py", "pz", "vx", "vy", "vz", "yaw", "pitch", "real_len", "max_dev",
                 "extra_len", "wp_idx", "arrive", "dest")]


# every symbol include/rvo3d.h declares (tests check the library exports them all)
SYMBOLS = ("rvo3d_create", "rvo3d_destroy", "rvo3d_load_world", "rvo3d_reset",
           "rvo3d_reset_drones", "rvo3d_observe", "rvo3d_step", "rvo3d_step_autoreset",
           "rvo3d_step_policy", "rvo3d_policy_sample", "rvo3d_policy_mlp_blob_bytes", "rvo3d_policy_mlp_pack",
           "rvo3d_policy_mlp_sample", "rvo3d_reader_zero_features", "rvo3d_policy_rows", "rvo3d_reader_first_step", "rvo3d_rollout_account", "rvo3d_rollout_set_step_counter", "rvo3d_set_reward_f64",
           "rvo3d_des_vel", "rvo3d_rvo_vel", "rvo3d_state_ptrs", "rvo3d_get_state", "rvo3d_set_state",
           "rvo3d_error_flags", "rvo3d_launch_info", "rvo3d_kernel_name", "rvo3d_version", "rvo3d_last_error")

_lib = None


def lib():
    """The loaded C-ABI library (ctypes.CDLL) with argtypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(
            f"{_SO} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    L = C.CDLL(_SO)
    vp, i32 = C.c_void_p, C.c_int32
    L.rvo3d_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.rvo3d_destroy.argtypes = [vp]
    L.rvo3d_load_world.argtypes = [vp] * 7
    L.rvo3d_reset.argtypes = [vp, vp, vp]
    L.rvo3d_reset_drones.argtypes = [vp, vp, vp]
    L.rvo3d_observe.argtypes = [vp, vp, vp, vp]
    L.rvo3d_step.argtypes = [vp, vp, i32] + [vp] * 7
    L.rvo3d_step_autoreset.argtypes = [vp, vp, i32] + [vp] * 8
    L.rvo3d_step_policy.argtypes = [vp, vp, C.c_float] + [vp] * 7 + [i32, vp]
    L.rvo3d_set_reward_f64.argtypes = [vp, vp]
    L.rvo3d_policy_sample.argtypes = [C.POINTER(PolicyHeads), C.c_int64, C.c_float, C.c_uint64, C.c_uint64] + [vp] * 6
    L.rvo3d_policy_mlp_blob_bytes.argtypes = [i32]
    L.rvo3d_policy_mlp_blob_bytes.restype = C.c_int64
    L.rvo3d_policy_mlp_pack.argtypes = [C.POINTER(MlpWeights), C.POINTER(MlpWeights), i32, vp, vp]
    L.rvo3d_policy_mlp_sample.argtypes = [vp, i32, vp, C.c_int64, C.c_int64, vp, i32, i32, i32, vp, C.c_float,
                                          C.c_uint64, C.c_uint64] + [vp] * 6
    L.rvo3d_reader_zero_features.argtypes = [vp, C.c_int64, C.c_int64, i32, i32, vp, vp, C.c_float, C.c_float, C.c_float,
                                             vp, C.c_int64, vp, vp, vp, vp]
    L.rvo3d_policy_rows.argtypes = [C.POINTER(RnnPolicy), vp, C.c_int64, vp, vp, vp, vp, i32, vp, C.c_float, C.c_uint64,
                                    C.c_uint64, vp, vp, vp, vp]
    L.rvo3d_reader_first_step.argtypes = [C.POINTER(GruReader), vp, C.c_int64, C.c_int64, vp, i32, C.c_int64, vp]
    L.rvo3d_rollout_set_step_counter.argtypes = [vp]
    L.rvo3d_rollout_account.argtypes = [i32, i32, vp, vp, vp, i32, i32, i32] + [vp] * 8
    L.rvo3d_des_vel.argtypes = [vp, vp, vp]
    L.rvo3d_rvo_vel.argtypes = [vp, C.POINTER(C.c_double), C.c_double, vp, vp]
    L.rvo3d_state_ptrs.argtypes = [vp, C.POINTER(StateView)]
    L.rvo3d_get_state.argtypes = [vp] * 12
    L.rvo3d_set_state.argtypes = [vp] * 12
    L.rvo3d_error_flags.argtypes = [vp, C.POINTER(C.c_uint32), vp]
    L.rvo3d_launch_info.argtypes = [vp] + [C.POINTER(i32)] * 4
    L.rvo3d_kernel_name.argtypes = [vp, i32, C.c_char_p, i32]
    if hasattr(L, "rvo3d_debug_stamps"):  # the diagnostics build (tools/diaglib.py) only
        L.rvo3d_debug_stamps.argtypes = [vp, vp]
        L.rvo3d_debug_stamps.restype = i32
    L.rvo3d_version.restype = i32
    L.rvo3d_last_error.restype = C.c_char_p
    for s in SYMBOLS:
        if s not in ("rvo3d_version", "rvo3d_last_error"):
            getattr(L, s).restype = i32
    _lib = L
    return L


def is_diag_build() -> bool:
    """True when the loaded library is the -DRVO3D_DIAG build (never the case for the product)."""
    return hasattr(lib(), "rvo3d_debug_stamps")


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().rvo3d_last_error().decode("utf-8", "replace")
        raise RVO3DError(f"{what} failed ({rc}): {msg}")
