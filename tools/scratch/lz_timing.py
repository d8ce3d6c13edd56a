#!/usr/bin/env python3
"""Developer probe (GPU box, THJ_EXP build: tools/build_exp.sh): where thj_k_lz's waves spend their clocks.  python tools/scratch/lz_timing.py file.bam"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(ROOT, "tophat_amd", "csrc", "libthj_exp.so")
env = dict(os.environ, THJ_LIB=lib)
code = r'''
import ctypes as C, sys, os, runpy
sys.argv = ["inflate_bench.py", sys.argv[1], "3"]
runpy.run_path(os.path.join(%r, "tools", "inflate_bench.py"), run_name="__main__")
from tophat_amd import host
l = host.load_lib()
out = (C.c_ulonglong * 16)()
assert l.thj_lz_dbg_read(out, 0) == 0
v = list(out)
waves = max(1, v[12]); names = ["top: scan, ballots, literals", "small matches", "big matches + round end", "bottom: next tokens", "tail flush", "slides"]
tot = sum(v[:6])
print("waves", v[12], "batches/wave %%.1f rounds/batch %%.2f big matches/batch %%.2f" %% (v[6] / waves, v[7] / max(1, v[6]), v[8] / max(1, v[6])))
for i, nm in enumerate(names): print("  %%-32s %%6.1f %%%%  %%.0f clocks per batch" %% (nm, 100.0 * v[i] / max(1, tot), v[i] / max(1, v[6])))
''' % ROOT
subprocess.run([sys.executable, "-c", code, sys.argv[1]], env=env)
