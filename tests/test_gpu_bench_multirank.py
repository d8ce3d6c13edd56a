"""GPU: bench.py's N > 1 path end to end on a one-GPU box.  A box with one GPU cannot hold two RCCL ranks, so
THJ_BENCH_INPROC=1 runs the two ranks as threads of one process on device 0, joined by the library's loopback transport:
the same step function, the same thj_events_allgather_async / thj_segjuncs_finish calls as the driver's one-process-per-GPU
RCCL runs.  THJ_BENCH_VERIFY makes every rank check that the merged sets are identical across ranks and contain its own.
A second case runs the RCCL transport itself at one rank (THJ_FORCE_COLLECTIVE)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, args):
    env = dict(os.environ, THJ_BENCH_VERIFY="1", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stderr.splitlines() if l.startswith("[verify]")]
    out = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return lines, json.loads(out[-1])


def test_two_ranks_exchange_and_agree():
    lines, d = _run({"THJ_BENCH_INPROC": "1"}, ["--gpus", "2", "--pairs", "500000", "--steps", "2", "--warmup", "1"])
    assert len(lines) == 2 and all("sets identical across 2 ranks" in l and "loopback" in l for l in lines), lines
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" in d and d["cpu_baseline"] is None


def test_two_ranks_with_the_coverage_search():
    lines, d = _run({"THJ_BENCH_INPROC": "1"}, ["--gpus", "2", "--pairs", "300000", "--read-len", "50", "--coverage-search", "0.2",
                                                "--steps", "1", "--warmup", "1"])
    assert len(lines) == 2 and all("sets identical across 2 ranks" in l for l in lines), lines
    assert "coverage junctions" in d["config"]["workload"]


def test_rccl_at_one_rank():
    lines, d = _run({"THJ_FORCE_COLLECTIVE": "1"}, ["--gpus", "1", "--pairs", "500000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert len(lines) == 1 and "rccl" in lines[0] and "identical" in lines[0], lines
    assert d["n_gpus"] == 1 and d["exchange"]["transport"] == "rccl"
