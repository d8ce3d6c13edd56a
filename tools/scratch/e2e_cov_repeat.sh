#!/bin/bash
# GPU box: the 2x50 bp coverage-search case three times on one set of files
d=/dev/shm/thj_cov_rep
for i in 1 2 3; do
python tools/e2e_bench.py --pairs 10000000 --read-len 50 --coverage-search --plain --keep $d | python -c "
import json,sys
d=json.load(sys.stdin)
print({k:d[k] for k in ('segment_juncs_s','long_spanning_reads_left_s','long_spanning_reads_right_s','both_stages_s','junctions')})
print('\n'.join(l for l in d['segment_juncs_log_tail'] if 'timing' in l and 'unix' not in l))
print('\n'.join(l for l in d['long_spanning_reads_left_log_tail'] if 'timing' in l and 'unix' not in l))
"
done
