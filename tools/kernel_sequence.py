#!/usr/bin/env python3
"""Developer tool (GPU box): the kernel launch sequence of the last bench step from a rocprofv3 results .db.
Usage: python tools/kernel_sequence.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel_dispatch" in t and t.startswith("rocpd_")][:1]
rows = cur.execute("select name, start, end from kernels order by start").fetchall() if "kernels" in tabs else []
if not rows:
    print("tables:", tabs)
    sys.exit(0)
idx = [i for i, r in enumerate(rows) if "thj_k_segjuncs<" in r[0] or r[0].startswith("thj_k_segjuncs(")]
lo = idx[-2] if len(idx) >= 2 else 0
t0 = rows[lo][1]
for name, s, e in rows[lo:]:
    print("%10.1f us  %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name[:90]))
