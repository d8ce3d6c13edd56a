#!/bin/bash
d=/dev/shm/thj_ab; rm -rf $d; mkdir -p $d
python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
for e in "X=1" "THJ_NO_WARM=1" "X=2" "THJ_NO_WARM=1"; do
  python tools/e2e_bench.py --pairs 10000000 --keep $d --env $e 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read(); d=json.loads(t[t.index('{'):])
print('$e', {k:v for k,v in d.items() if k in ('segment_juncs_s','long_spanning_reads_left_s','long_spanning_reads_right_s','both_stages_s','outside_main_s')})
for st in ('segment_juncs','long_spanning_reads_left'):
    print('   ', st, [l.replace('[timing] ','') for l in d[st+'_log_tail'] if 'timing' in l][:5])"
done
rm -rf $d
