#!/bin/bash
# Developer probe (GPU box): what the executables spend after their report (teardown) under different runtime settings
d=/dev/shm/thj_exitprobe; rm -rf $d; mkdir -p $d
python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
for e in "X=1" "GPU_PINNED_MIN_XFER_SIZE=1000000" "GPU_PINNED_XFER_SIZE=4"; do
  for i in 1 2; do
  python tools/e2e_bench.py --pairs 10000000 --keep $d --env $e 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read(); d=json.loads(t[t.index('{'):])
print('$e', {k:v for k,v in d.items() if k in ('segment_juncs_s','long_spanning_reads_left_s','long_spanning_reads_right_s','both_stages_s','outside_main_s')}, d['segment_juncs_before_main_after_report_s'], d['long_spanning_reads_left_before_main_after_report_s'])"
  done
done
rm -rf $d
