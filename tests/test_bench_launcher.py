"""bench.py --gpus N without a launcher: the script starts its N ranks itself, the
line carries n_gpus = the ranks an all-reduce saw, and anything else is a non-zero exit - never a
single-GPU number labelled as N.  Driven here on the CPU (gloo, world 2) through --launcher-selftest,
which runs the whole launcher path (child processes, rendezvous, all-reduce, JSON relay, exit status)
and leaves out only the env step, which needs a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items()
         if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def _run(args, **kw):
    return subprocess.run([sys.executable, BENCH] + args, env=_env(**kw), capture_output=True, text=True,
                          timeout=600)


def test_launcherless_gpus_2_starts_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--backend", "gloo", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["collective"]["ranks_seen"] == 2
    assert "launcher" in rec


def test_launcherless_gpus_2_without_two_gpus_is_refused():
    """No GPU in the build container: `bench.py --gpus 2` must not print a 1-GPU line; it exits non-zero."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this host has two GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert not any(l.startswith("{") for l in r.stdout.splitlines())
    assert "refusing" in r.stderr


def test_world_size_that_contradicts_gpus_is_an_error():
    r = _run(["--gpus", "2", "--backend", "gloo", "--launcher-selftest"], WORLD_SIZE="1", RANK="0",
             LOCAL_RANK="0")
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_under_a_launcher_the_ranks_are_not_started_twice():
    """torch.distributed.run sets WORLD_SIZE: bench.py then is one rank (here: world 1 of 1)."""
    r = _run(["--gpus", "1", "--backend", "gloo", "--launcher-selftest"], WORLD_SIZE="1", RANK="0",
             LOCAL_RANK="0")
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and "launcher" not in rec
