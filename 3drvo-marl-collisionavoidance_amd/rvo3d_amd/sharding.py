"""Multi-GPU plumbing: environments are independent (no cross-env term anywhere in
the step), so the path shards with NO data-path collective.

One process per GPU owns `envs_per_rank` environments with its own seed stream.
The env step itself never communicates; the collectives of the design are
  * the PPO update's ONE all-reduce of the flattened gradient bucket per optimizer
    step, the KL estimate in its last slot (policy/multi_ppo.py: _allreduce_grads),
    which bench.py --gpus N > 1 also issues inside every timed step;
  * bench.py's timing barrier, the MAX over the ranks' elapsed time and the
    all-reduce of ones that counts the ranks (`ranks_seen`).
These helpers are backend-agnostic (nccl = RCCL on the GPU node, gloo in the CPU
tests).
"""
from __future__ import annotations

import os

import torch


def rank_info():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


SEED_STRIDE = 1_000_003  # > any per-world key offset (synthetic_world draws buildings from key + 1)


def shard_seed(base_seed: int, rank: int) -> int:
    """Seed of rank's shard.  worlds.synthetic_world uses Philox key `seed` (counter = env) for
    the routes and key `seed + 1` for the buildings, synthetic_actions key `seed + 7`: a stride
    of 1 between ranks would give rank r's buildings the key and counter of rank r+1's env 0.
    Ranks are a large prime stride apart, so no two keys of different shards coincide."""
    return int(base_seed) + SEED_STRIDE * int(rank)


def shard_env_range(total_envs: int, rank: int, world: int):
    """Contiguous block partition of a fixed total (strong scaling): [lo, hi)."""
    base, rem = divmod(int(total_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_process_group(backend: str, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return dist


def max_over_ranks(dist, value: float, device="cpu") -> float:
    """MAX all-reduce of a python float (the job's wall time is its slowest rank's)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value: float, device="cpu") -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
