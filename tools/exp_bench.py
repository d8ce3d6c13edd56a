#!/usr/bin/env python3
"""Developer tool (GPU box): bench.py against the THJ_EXP build of the library, once per THJ_EXP_FLAGS value, to see
what each part of a kernel costs.  Build first: tools/build_exp.sh.  Usage: [THJ_EXP_ARGS="--multihit-frac 0.1"] python tools/exp_bench.py 0 1 2 4 ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.path.insert(0, %r); import tophat_amd.host as h; "
        "h.LIB_PATH = %r; import bench; sys.argv = ['bench.py', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--e2e-pairs', '0', '--no-pmc', '--detail', '/tmp/exp_bench_detail.json'] + %r; bench.main()"
        % (ROOT, os.path.join(ROOT, "tophat_amd", "csrc", "libthj_exp.so"), os.environ.get("THJ_EXP_ARGS", "").split()))
for f in sys.argv[1:]:
    env = dict(os.environ, THJ_EXP_FLAGS=f)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(f, "FAILED", out.stderr[-400:])
        continue
    d = json.load(open("/tmp/exp_bench_detail.json"))
    print("flags=%s step=%.3f ms  " % (f, d["ms_per_step"]) + "  ".join("%s=%.3f/%s" % (k["kernel"][6:28], k["avg_kernel_ms"], ("%.3f" % k["avg_kernel_ms_alone"]) if k.get("avg_kernel_ms_alone") else "-") for k in d["kernels"]), flush=True)
