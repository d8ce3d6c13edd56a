#!/usr/bin/env python3
"""Developer probe (GPU box, THJ_EXP build: tools/build_exp.sh): where thj_k_huffp's waves spend their clocks.  python tools/scratch/huffp_timing.py file.bam"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(ROOT, "tophat_amd", "csrc", "libthj_exp.so")
env = dict(os.environ, THJ_LIB=lib)
code = r"""
import ctypes as C, sys, os, runpy
sys.argv = ["inflate_bench.py", sys.argv[1], "3"]
runpy.run_path(os.path.join(%r, "tools", "inflate_bench.py"), run_name="__main__")
from tophat_amd import host
l = host.load_lib()
out = (C.c_ulonglong * 16)()
assert l.thj_huffp_dbg_read(out, 0) == 0
v = list(out)
mem = max(1, v[12]); names = ["compressed bytes to LDS", "header: tables built by the wave", "warm-up pass", "agreement passes", "last pass (tokens out) + scans"]
tot = sum(v[:5]) + v[7]
print("  %%-36s %%6.1f %%%%  %%.0f clocks per member" %% ("header: code lengths read by lane 0", 100.0 * v[7] / max(1, tot), v[7] / mem))
print("members", v[12], "blocks/member %%.2f  passes/block %%.2f  tokens/member %%.0f  clocks/member %%.0f" %% (v[5] / mem, v[6] / max(1, v[5]), v[13] / mem, tot / mem))
for i, nm in enumerate(names): print("  %%-36s %%6.1f %%%%  %%.0f clocks per member" %% (nm, 100.0 * v[i] / max(1, tot), v[i] / mem))
""" % ROOT
subprocess.run([sys.executable, "-c", code, sys.argv[1]], env=env)
