"""bench.py as its own launcher: ``python bench.py --gpus N`` (no torch.distributed.run around it) starts N ranks.

On this CPU box the ranks get as far as "no GPU": the parent must relay that loudly and exit non-zero without printing a
record -- an 8-GPU run must never end in a silent or half-written line (bench contract)."""
import os
import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent


def _run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)


def test_self_launch_two_ranks_fails_loudly_without_gpus():
    if torch.cuda.device_count() >= 2:  # not this container; on a multi-GPU box the launcher would really run
        import pytest
        pytest.skip("two GPUs present")
    r = _run("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode != 0
    assert r.stdout.strip() == "", "no record may be printed by a failed run"
    assert "[rank 0]" in r.stderr and "[rank 1]" in r.stderr, r.stderr          # both ranks were started and relayed
    assert "needs GPU 1" in r.stderr and "no CPU fallback" in r.stderr, r.stderr  # the reason, per rank
    assert "--gpus 2 FAILED" in r.stderr


def test_mismatched_world_size_is_rejected():
    r = _run("--gpus", "4", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_rank_dying_mid_run_ends_the_job_non_zero():
    """A rank that dies MID-RUN (here: rank 5 of 8 exits with status 3 after the others have entered a long wait, as ranks stuck in
    a collective would): the launcher ends the other ranks, returns that status promptly and relays no result line."""
    import importlib.util
    import time

    spec = importlib.util.spec_from_file_location("_yl_launch", str(ROOT / "yolo-lite_amd" / "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    child = ("import os, sys, time\n"
             "r = int(os.environ['RANK']); assert os.environ['WORLD_SIZE'] == '8' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
             "print('started', r, flush=True)\n"
             "time.sleep(0.5)\n"
             "if r == 5: sys.exit(3)\n"
             "time.sleep(120)\n"
             "print('{\"value\": 1}', flush=True)\n")
    t0 = time.monotonic()
    rc, lines = mod.launch_ranks(8, [sys.executable, "-c", child])
    assert rc == 3 and time.monotonic() - t0 < 30
    assert not any(ln.lstrip().startswith("{") for ln in lines), lines  # rank 0 never got to print a record


def test_all_ranks_succeeding_returns_rank0_stdout():
    import importlib.util

    spec = importlib.util.spec_from_file_location("_yl_launch", str(ROOT / "yolo-lite_amd" / "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rc, lines = mod.launch_ranks(3, [sys.executable, "-c", "import os; print('{\"rank\": %s}' % os.environ['RANK'])"])
    assert rc == 0 and [ln.strip() for ln in lines] == ['{"rank": 0}']
