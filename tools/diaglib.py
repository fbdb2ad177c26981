"""tools/ only: build and load the DIAGNOSTICS build of the library (csrc/rvo3d_capi.hip with
-DRVO3D_DIAG -> tools/_build/librvo3d_hip_diag.so; include/rvo3d_diag.h).  It has the phase
stamps (rvo3d_debug_stamps) and honours RVO3D_ABLATE / RVO3D_LDS_PAD; the product library
has none of it.  Import this module BEFORE anything creates an env:

    import diaglib            # builds if stale, then points rvo3d_amd at the diag build
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd"))

from rvo3d_amd import _lib  # noqa: E402

SO = os.path.join(ROOT, "tools", "_build", "librvo3d_hip_diag.so")


def build(force=False):
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    return _lib.build_hip(force=force, out=SO, extra_flags=["-DRVO3D_DIAG"])


build()
_lib.use_library(SO)
assert _lib.is_diag_build()
