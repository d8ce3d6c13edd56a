#!/bin/bash
# GPU box: contexts per GPU in the executables (THJ_CTX_PER_GPU), 10 M pairs, alternating
cd "$(dirname "$0")/../.."; d=/dev/shm/e10; rm -rf $d
timeout 600 python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
run() { timeout 300 python tools/e2e_bench.py --pairs 10000000 --keep $d "$@" 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); r = json.loads(t[t.index('{'):])
print('$*', r['segment_juncs_s'], r['long_spanning_reads_left_s'], r['long_spanning_reads_right_s'], 'sum %.3f' % r['both_stages_s'])"; }
for i in 1 2 3; do run; run --env THJ_CTX_PER_GPU=2; run --env THJ_CTX_PER_GPU=3; done
rm -rf $d
