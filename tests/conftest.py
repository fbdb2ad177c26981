"""pytest configuration: markers + import paths.

`gpu` tests need a real MI355X (run with `-m gpu` on the GPU box); everything
else runs on CPU in the build container.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd")
for p in (ROOT, PKG_DIR, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X; runs the HIP path through the C-ABI")


def pytest_sessionstart(session):
    # the GPU parity tests append their knife-edge tallies to this file; start each session clean
    f = os.path.join(ROOT, "gpurun_out", "parity_tally.json")
    if os.path.exists(f):
        try:
            os.remove(f)
        except OSError:
            pass
