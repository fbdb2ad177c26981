"""Randomised parity soak (GPU box): longer rollouts of the HIP step against the oracle over
a spread of shapes, seeds, action styles and reset protocols.  Not part of the test suite
(minutes of oracle time); prints one line per case and fails on the first mismatch outside
the decision-margin exemptions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"),
                os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd")]
import numpy as np
import test_gpu_parity as T
from rvo3d_amd import synthetic_world

cases = [
    dict(E=64, N=64, size=(50, 50, 10), T=120, kw=dict(autoreset=True, f32_actions=True)),
    dict(E=256, N=16, size=(20, 20, 8), T=150, kw=dict(autoreset=True)),
    dict(E=48, N=33, size=(30, 30, 10), T=100, kw=dict(autoreset=True, vlike=True)),
    dict(E=32, N=64, size=(25, 25, 8), T=80, kw=dict(autoreset=True, nm=3), nb=12),
    dict(E=24, N=64, size=(40, 40, 10), T=80, kw=dict(autoreset=False)),
    dict(E=8, N=200, size=(80, 80, 10), T=40, kw=dict(autoreset=True), nb=30),
    dict(E=16, N=128, size=(60, 60, 10), T=60, kw=dict(autoreset=True, vlike=True, nm=5)),
    dict(E=40, N=7, size=(10, 10, 6), T=150, kw=dict(autoreset=True, nm=10)),
    # dense scenes: many flagged pairs, nm truncation, velocity-like actions
    dict(E=64, N=32, size=(12, 12, 5), T=40, kw=dict(autoreset=False, vlike=True, nm=2, radius=0.3), min_sep=0.8),
    dict(E=48, N=48, size=(14, 14, 5), T=40, kw=dict(autoreset=True, vlike=True, nm=10, radius=0.3), min_sep=0.8),
    dict(E=32, N=64, size=(16, 16, 5), T=40, kw=dict(autoreset=False, vlike=True, nm=4, radius=0.25), min_sep=0.7),
    # round 2: the compile-time-N kernels at 128 / 256 drones, env_train=False, 96 drones (generic kernel)
    dict(E=12, N=256, size=(100, 100, 10), T=40, kw=dict(autoreset=True), nb=50),
    dict(E=12, N=256, size=(60, 60, 10), T=30, kw=dict(autoreset=False, vlike=True, nm=6)),
    dict(E=24, N=128, size=(50, 50, 10), T=50, kw=dict(autoreset=True, vlike=True)),
    dict(E=24, N=96, size=(40, 40, 8), T=50, kw=dict(autoreset=True, nm=10), nb=8),
    dict(E=64, N=32, size=(20, 20, 6), T=60, kw=dict(autoreset=True, env_train=False, radius=0.4), min_sep=1.2),
    dict(E=32, N=64, size=(30, 30, 8), T=60, kw=dict(autoreset=False, env_train=False, vlike=True, radius=0.3), min_sep=1.0),
]
base = int(os.environ.get("SOAK_SEED", "0"))
for i, c in enumerate(cases):
    seeds = [1234, 99 + i] if base == 0 else [base + 10 * i, base + 10 * i + 1]
    for sd in seeds:
        w = synthetic_world(c["E"], c["N"], c["size"], nb=c.get("nb", 0), seed=sd, min_sep=c.get("min_sep", 1.0))
        t0 = time.time()
        st = T.run_vs_oracle(w, T=c["T"], seed=sd, **c["kw"])
        print(f"case {i} seed {sd}: {c['E']}x{c['N']} T={c['T']} {c['kw']} -> {st}  ({time.time()-t0:.1f}s)", flush=True)
print("soak ok")
