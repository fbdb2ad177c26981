#!/usr/bin/env python3
"""usage: tools/bench_variant.py <library.so> [bench.py args]
bench.py on another build of the library (a variant compiled with different flags into
tools/_build/): for compiler-flag / macro experiments.  Not the headline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd")]
from rvo3d_amd import _lib  # noqa: E402

_lib.use_library(os.path.abspath(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402

bench.ALLOW_DIAG = True
bench.main()
