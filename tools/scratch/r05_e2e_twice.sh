#!/bin/bash
# GPU box: the files-in -> files-out leg at 10 M pairs, three times on the same files
cd "$(dirname "$0")/../.."; d=/dev/shm/e10; rm -rf $d
timeout 600 python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
for i in 1 2 3; do timeout 300 python tools/e2e_bench.py --pairs 10000000 --keep $d 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); r = json.loads(t[t.index('{'):])
print(r['segment_juncs_s'], r['long_spanning_reads_left_s'], r['long_spanning_reads_right_s'], 'sum %.3f' % r['both_stages_s'])"; done
rm -rf $d
