#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (rocpd sqlite) as a per-kernel stats table
(the `--stats` view) and, when PMC rows exist, per-kernel counter sums.
Usage: python tools/rocpd_summary.py <results.db> [name-filter]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = db.cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("%-78s %8s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows:
        if flt and flt not in name:
            continue
        print("%-78s %8d %14d %14.1f %7.2f" % (name[:78], calls, tot, avg, pct))
    try:
        pm = cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events "
                         "group by name, counter_name").fetchall()
    except sqlite3.Error:   # no PMC rows in this capture
        pm = []
    if pm:
        print()
        print("%-60s %-22s %8s %20s %16s" % ("kernel", "counter", "disp", "sum", "per_dispatch"))
        for name, ctr, n, s in pm:
            if flt and flt not in name:
                continue
            print("%-60s %-22s %8d %20.1f %16.1f" % (name[:60], ctr, n, s, s / max(1, n)))


if __name__ == "__main__":
    main()
