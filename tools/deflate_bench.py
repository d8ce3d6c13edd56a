#!/usr/bin/env python3
"""GPU box: thj_bgzf_deflate on a synthetic stream of BAM records (the shape long_spanning_reads writes) -- GB/s of uncompressed
bytes per call, and with THJ_DEFLATE_TIMING=1 the kernel's phase times.  python tools/deflate_bench.py [records] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_bamout_sim_cpu import bam_like_stream  # noqa: E402
from tophat_amd.host import Context, bgzf_plan_cuts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(3)
base = bam_like_stream(rng, 3000)
recs = [base[i % len(base)][:40] + (b"%07d" % i) + base[i % len(base)][47:] for i in range(n)]      # distinct names, same shape
stream = b"".join(recs)
ends = bgzf_plan_cuts([len(r) for r in recs])
ctx = Context()
ctx.bam_stream_upload(stream)
for r in range(reps):
    t = time.time()
    comp, crcs = ctx.bgzf_deflate(ends)
    dt = time.time() - t
    print("%d members, %.1f MB -> %.1f MB in %.2f ms (host call incl. copies): %.1f GB/s" % (len(ends), len(stream) / 1e6, sum(len(c) for c in comp) / 1e6, dt * 1e3, len(stream) / dt / 1e9))
