#!/usr/bin/env python3
"""bench.py on the DIAGNOSTICS build of the library (tools/diaglib.py): the only way to time
the step with phases ablated (RVO3D_ABLATE) or occupancy capped (RVO3D_LDS_PAD).  The JSON line
carries "diag_build": true; such numbers are never the headline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import diaglib  # noqa: E402,F401
import bench  # noqa: E402

bench.ALLOW_DIAG = True
bench.main()
