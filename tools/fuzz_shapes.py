"""Randomised SHAPE fuzz (GPU box): short rollouts of the HIP step against the oracle over random
drone counts, env counts, neighbour caps, building counts, radii, both env_train modes and both
reset protocols - every instantiation / writer path / partial workgroup the launch geometry can
produce.  Not part of the test suite; prints one line per case, fails on the first mismatch outside
the decision-margin exemptions.   usage: FUZZ_SEED=1 FUZZ_CASES=150 python tools/fuzz_shapes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"),
                os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd")]
import numpy as np
import test_gpu_parity as T
from rvo3d_amd import synthetic_world

seed = int(os.environ.get("FUZZ_SEED", "1"))
cases = int(os.environ.get("FUZZ_CASES", "120"))
rng = np.random.default_rng(seed)


def _finish(self, **extra):  # the share bound and the tally file are for the suite's fixed cases
    rec = dict(samples=self.samples, knife=self.knife, knife_mismatch=self.knife_mismatch,
               dropped_envs=int(self.dropped.sum()), envs=len(self.dropped), **extra)
    assert self.knife_mismatch <= self.knife, rec
    return rec


T.Tally.finish = _finish
t_all = time.time()
for c in range(cases):
    kind = rng.integers(6)
    N = int([rng.integers(2, 17), rng.integers(17, 65), rng.choice([16, 32, 64]), rng.integers(65, 129),
             rng.choice([128, 192, 256]), rng.integers(129, 301)][kind])
    budget = 6000 if N <= 64 else 3000
    E = int(max(1, min(rng.integers(1, 40), budget // N)))
    nm = int(rng.choice([0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 14]))
    L = float(np.round(4 + rng.uniform(1.2, 3.5) * np.sqrt(N), 1))
    nb = int(rng.choice([0, 0, 3, 12, 40]))
    radius = float(rng.choice([0.2, 0.2, 0.3, 0.45]))
    kw = dict(nm=nm, autoreset=bool(rng.integers(2)), f32_actions=bool(rng.integers(2)), vlike=bool(rng.integers(2)),
              env_train=bool(rng.integers(4) > 0), radius=radius)
    if kw["f32_actions"] and kw["vlike"]:
        kw["f32_actions"] = False
    w = synthetic_world(E, N, (L, L, float(rng.choice([5.0, 8.0, 10.0]))), nb=nb, seed=int(rng.integers(1 << 30)),
                        min_sep=max(0.6, 2.2 * radius + 0.1), n_points=int(rng.integers(2, 5)))
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != c:
        rng.integers(8, 25); rng.integers(1 << 30)  # (the draws of the skipped run)
        continue
    t0 = time.time()
    st = T.run_vs_oracle(w, T=int(rng.integers(8, 25)), seed=int(rng.integers(1 << 30)), name=f"fuzz/{seed}/{c}", **kw)
    print(f"case {c}: N={N} E={E} map={L} nb={nb} {kw} -> steps {st['steps']} done {st['done']} vo_rows {st['vo_rows']} "
          f"resets {st['resets']} knife {st.get('knife')} mism {st.get('knife_mismatch')} ({time.time() - t0:.1f}s)", flush=True)
print(f"fuzz ok: {cases} cases, seed {seed}, {time.time() - t_all:.0f}s")
