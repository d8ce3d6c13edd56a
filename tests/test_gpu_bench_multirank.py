"""GPU: bench.py's N > 1 path end to end on a one-GPU box -- two ranks share device 0 and exchange through gloo
(THJ_BENCH_BACKEND / THJ_BENCH_DEVICE; the driver's real runs are one rank per GPU over RCCL).  THJ_BENCH_VERIFY makes every
rank check that the merged junction sets are identical across ranks and contain its own.  (This test found a stream race
in the exchange step: a fill kernel on torch's stream zeroing keys already copied on the context's stream.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_exchange_and_agree():
    env = dict(os.environ, THJ_BENCH_BACKEND="gloo", THJ_BENCH_DEVICE="0", THJ_BENCH_VERIFY="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "500000",
           "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("[verify]")]
    assert len(lines) == 2 and all("sets identical across 2 ranks" in l for l in lines), lines
    out = [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = json.loads(out[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" in d and d["cpu_baseline"] is None
