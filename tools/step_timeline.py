#!/usr/bin/env python3
"""Developer tool (GPU box): start / end of every kernel of the LAST bench step, relative to the step's first thj_k_sj_flat, from a
rocprofv3 --kernel-trace results .db -- which kernels ran beside which, and what the step waited for.
Usage: python tools/step_timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "thj_k_sj_flat" in r[0]]
if len(idx) < 2:
    print("no step found")
    sys.exit(0)
lo = idx[-2]
t0 = rows[lo][1]
for name, s, e in rows[lo:]:
    short = name.replace("void ", "").split("(")[0]
    print("%9.1f .. %9.1f us  (%7.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, short[:70]))
