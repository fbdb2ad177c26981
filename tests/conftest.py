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
    # the GPU parity tests append their knife-edge tallies to this file; a session that runs them starts
    # clean (a CPU-only session - `-m "not gpu"` - leaves the last GPU run's record alone)
    f = os.path.join(ROOT, "gpurun_out", "parity_tally.json")
    if "not gpu" in (session.config.getoption("-m") or ""):
        return
    if os.path.exists(f):
        try:
            os.remove(f)
        except OSError:
            pass


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The knife-edge accounting of a `-m gpu` run in the run's own output (also under -q), so that
    whoever ran the suite - the driver at round end included - has the tally of exactly that run:
    the TOTAL line test_zz_knife_edge_share_over_all_tests computed over every test's record."""
    import json
    f = os.path.join(ROOT, "gpurun_out", "parity_tally.json")
    if "not gpu" in (config.getoption("-m") or ""):
        return
    try:
        with open(f) as fh:
            tot = json.load(fh).get("TOTAL")
    except (OSError, ValueError):
        return
    if tot:
        terminalreporter.write_line("parity tally TOTAL: " + json.dumps(tot, sort_keys=True))
