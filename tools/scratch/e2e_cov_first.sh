#!/bin/bash
# GPU box: the 2x50 bp coverage-search case as the first thing on a box, twice, logs kept
for i in 1 2; do
python tools/e2e_bench.py --pairs 10000000 --read-len 50 --coverage-search --plain | python -c "
import json,sys
d=json.load(sys.stdin)
print({k:d[k] for k in ('segment_juncs_s','long_spanning_reads_left_s','long_spanning_reads_right_s','both_stages_s','junctions')})
for st in ('segment_juncs','long_spanning_reads_left','long_spanning_reads_right'):
    print('  ==',st); print('\n'.join('     '+l for l in d[st+'_log_tail'] if 'unix' not in l))
"
done
