"""Shared helpers for replaying tests/golden/*.npz (vectors produced by the
Python reference; generator: oracle/gen_golden.py)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def scenario_files():
    skip = ("calls_", "policy_", "ppo_", "rvo_", "post_train_")  # call-level and policy/PPO vectors have their own tests
    return sorted(f for f in glob.glob(os.path.join(GOLDEN, "*.npz"))
                  if not os.path.basename(f).startswith(skip))


def load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def eq_nan(a, b):
    """Exact equality where NaN == NaN and +-inf compare by sign (Q9 rewards)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return (a == b) | (np.isnan(a) & np.isnan(b))
