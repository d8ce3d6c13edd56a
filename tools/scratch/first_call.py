#!/usr/bin/env python3
"""Developer probe (GPU box): what the FIRST call of each kind costs in a fresh process (runtime start-up, code objects loaded on first
launch, first allocations) against the second."""
import os, sys, time
t00 = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tophat_amd import host
from tophat_amd.params import Params
from tophat_amd.synth import make_case
from tophat_amd.batch import build_span_batch, events_to_span_inputs
from util import case_batches
import orc
t_imp = time.time()
case = make_case(seed=11, paired=True, read_len=100, seg_len=25, n_reads=2000)
seqs = [orc.fold_genome_char(s) for s in case.seqs]
pg = host.pack_genome(seqs)
def lap(name, t0): print("%-46s %.3f s" % (name, time.time() - t0), flush=True)
lap("imports", t00)
t = time.time(); lib = host.load_lib(); lap("load_lib (dlopen libthj_hip.so + libamdhip64)", t)
t = time.time(); ctx = host.Context(0); lap("thj_ctx_create (runtime start-up, stream, tables)", t)
t = time.time(); ctx.upload_genome(pg); ctx.sync(); lap("genome upload", t)
runs = []
for side, b in case_batches(case, True):
    runs.append((Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20), ctx.upload_batch(b)))
for k in range(3):
    t = time.time(); ev = ctx.segjuncs(runs); lap("segjuncs pass %d (reset, 2 runs, finish, download)" % k, t)
juncs, ins = events_to_span_inputs(ev)
t = time.time(); ctx.upload_span_sets(juncs, ins); ctx.sync(); lap("span sets upload", t)
sb = build_span_batch(case.seg_recs["left"], case.reads["left"], case.quals["left"])
db = ctx.upload_span_batch(sb)
for k in range(3):
    t = time.time(); ctx.spanning(Params(), [db]); lap("spanning pass %d" % k, t)
