#!/bin/bash
# GPU box: the executables (three contexts a GPU) with more hardware queues than HIP's default four
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; d=/dev/shm/e10
python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
for q in "" 8 16 "" 8; do
  python tools/e2e_bench.py --pairs 10000000 --keep $d ${q:+--env GPU_MAX_HW_QUEUES=$q} 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); r = json.loads(t[t.index('{'):])
print('GPU_MAX_HW_QUEUES=${q:-default}', r['segment_juncs_s'], r['long_spanning_reads_left_s'], r['long_spanning_reads_right_s'], 'sum %.3f' % r['both_stages_s'])"
done | tee gpurun_out/r05_e2e_hwq.txt
rm -rf $d
