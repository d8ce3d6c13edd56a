#!/usr/bin/env python3
"""Per-dispatch durations of the kernels whose name contains a filter, from a rocprofv3 results .db (rocpd sqlite).
Usage: python tools/rocpd_dispatches.py <results.db> <name-filter>"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2]
    cur = db.cursor()
    try:
        rows = cur.execute("select name, start, end, grid_x, workgroup_x from kernels where name like ? order by start", ("%" + flt + "%",)).fetchall()
    except sqlite3.Error as e:
        print("kernels view not usable:", e)
        for (n,) in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall():
            print("  ", n)
        return
    for name, s, e, gx, wx in rows:
        print("%-60s %10.1f us  grid %s wg %s" % (name[:60], (e - s) / 1000.0, gx, wx))


if __name__ == "__main__":
    main()
