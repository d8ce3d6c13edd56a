#!/usr/bin/env python3
"""Developer tool: long_spanning_reads' THJ_TRACE lines (stderr) -> where the shards' time goes.
usage: THJ_TRACE=1 long_spanning_reads ... 2> log; python tools/lsr_trace.py log"""
import sys
ev = {}
for l in open(sys.argv[1]):
    if l.startswith("[trace]"):
        _, k, what, t = l.split()
        ev.setdefault(int(k), {})[what] = float(t)
order = ["start", "ingest_begin", "ingest_end", "stitch_begin", "stitch_end", "encoded", "planned", "deflated", "written"]
print("shard " + " ".join("%12s" % o for o in order))
for k in sorted(ev):
    print("%5d " % k + " ".join("%12.4f" % ev[k].get(o, float("nan")) for o in order))
tot = {}
for k, e in ev.items():
    for a, b in zip(order, order[1:]):
        if a in e and b in e:
            tot[a + "->" + b] = tot.get(a + "->" + b, 0.0) + e[b] - e[a]
print("sums over shards (seconds):")
for k, v in tot.items():
    print("  %-28s %.3f" % (k, v))
