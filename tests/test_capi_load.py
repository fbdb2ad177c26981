"""CPU-side checks of the drop-in boundary: the C-ABI library builds for
gfx950, loads, and exports every symbol include/rvo3d.h declares.  No compute
calls (those need a GPU and live in the `-m gpu` tests)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from rvo3d_amd import _lib


@pytest.fixture(scope="module")
def so():
    return _lib.build_hip()


def test_header_symbols_all_exported(so):
    hdr = open(os.path.join(ROOT, "include", "rvo3d.h")).read()
    declared = set(re.findall(r"\b(rvo3d_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = C.CDLL(so)
    for s in declared:
        assert hasattr(L, s), f"{s} not exported"


def test_product_library_has_no_diagnostics(so):
    """Phase ablation / LDS padding / phase stamps exist only in the -DRVO3D_DIAG build that
    tools/diaglib.py makes for itself: the product library exports no rvo3d_debug_stamps and
    does not even contain the names of the diagnostic environment variables."""
    L = C.CDLL(so)
    assert not hasattr(L, "rvo3d_debug_stamps")
    blob = open(so, "rb").read()
    for name in (b"RVO3D_ABLATE", b"RVO3D_LDS_PAD"):
        assert name not in blob
    assert b"getenv" not in blob


def test_version_and_error_string(so):
    L = _lib.lib()
    assert L.rvo3d_version() == 1
    assert isinstance(L.rvo3d_last_error(), bytes)


def test_bad_arguments_return_status_not_crash(so):
    L = _lib.lib()
    h = C.c_void_p()
    assert L.rvo3d_create(None, C.byref(h)) == -1          # RVO3D_ERR_INVALID
    cfg = _lib.Config(0, 4, 2, 0, 10, 1, 0, -1, (C.c_double * 3)(10, 10, 5))
    assert L.rvo3d_create(C.byref(cfg), C.byref(h)) == -1   # num_envs < 1
    assert b"num_envs" in L.rvo3d_last_error()
    cfg = _lib.Config(1, 4096, 2, 0, 10, 1, 0, -1, (C.c_double * 3)(10, 10, 5))
    assert L.rvo3d_create(C.byref(cfg), C.byref(h)) == -1   # N > 512
    assert L.rvo3d_step(None, None, 0, None, None, None, None, None, None, None) == -1
    assert L.rvo3d_destroy(None) == 0


def test_product_does_not_import_oracle():
    """The product package must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in txt and "rvo3d_oracle" not in txt, f
                assert "orc_" not in txt, f


def test_absurd_sizes_return_a_status(so):
    """'Nothing throws' (include/rvo3d.h): an allocation that cannot succeed - here a route of 2^30
    waypoints for a million drones, i.e. petabytes - comes back as a negative status with a message,
    whether it fails in HIP (no device in this container / hipMalloc) or in a host std::vector."""
    L = _lib.lib()
    h = C.c_void_p()
    cfg = _lib.Config(4096, 256, 1 << 30, 0, 10, 1, 0, -1, (C.c_double * 3)(10, 10, 5))
    rc = L.rvo3d_create(C.byref(cfg), C.byref(h))
    assert rc < 0 and not h.value
    assert len(L.rvo3d_last_error()) > 0


def test_every_entry_point_is_exception_guarded():
    """Each extern "C" function with a status runs between RVO3D_API_BEGIN / RVO3D_API_END
    (try / catch -> RVO3D_ERR_INVALID)."""
    src = open(os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd", "csrc", "rvo3d_capi.hip")).read()
    body = src[src.index('extern "C" {'):]
    fns = re.findall(r"^int (rvo3d_\w+)\(", body, flags=re.M)
    assert len(fns) >= 18
    for f in fns:
        if f == "rvo3d_version":
            continue
        start = body.index("int " + f + "(")
        end = body.index("\n}\n", start)
        chunk = body[start:end]
        assert "RVO3D_API_BEGIN" in chunk and "RVO3D_API_END" in chunk, f
