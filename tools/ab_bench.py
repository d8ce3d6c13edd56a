#!/usr/bin/env python3
"""Developer tool (GPU box): bench.py against several builds of the library in one gpurun call (boxes differ by a few
percent, so A/B in the same call).  Usage: [THJ_AB_ARGS="--read-len 50 ..."] python tools/ab_bench.py libthj_a.so libthj_b.so ... (files in tophat_amd/csrc)"""
import sys, json, subprocess, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sys.argv[1:]:
    code = ("import sys; sys.path.insert(0, %r); import tophat_amd.host as h; h.LIB_PATH = %r; import bench; "
            "sys.argv = ['bench.py', '--no-cpu-baseline', '--e2e-pairs', '0', '--no-pmc'] + %r; bench.main()" % (ROOT, os.path.join(ROOT, "tophat_amd", "csrc", lib), os.environ.get("THJ_AB_ARGS", "").split()))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line: print(lib, "FAILED", out.stderr[-300:]); continue
    d = json.loads(line[-1])
    print(lib, "step=%.3f" % d["ms_per_step"], "  ".join("%s=%.3f" % (k["kernel"][6:], k["avg_kernel_ms"]) for k in d["kernels"]), flush=True)
