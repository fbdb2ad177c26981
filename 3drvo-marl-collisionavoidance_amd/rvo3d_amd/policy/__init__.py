"""Counterpart of the reference package `train/policy` (actor-critic + PPO trainer).
Stays in PyTorch-ROCm by design (BASELINE.json north_star): the GEMMs are large
(E*N rows) and go to hipBLASLt / MIOpen; no custom kernel here."""
from .policy_rnn_ac import rnn_ac, mlp_ac  # noqa: F401
from .multi_ppo import multi_ppo, RolloutBuffer, gae_scan, gae_scan_loop  # noqa: F401
from .post_train import post_train  # noqa: F401
