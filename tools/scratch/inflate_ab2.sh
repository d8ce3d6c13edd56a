#!/bin/bash
# GPU box: correctness of the inflater, then its rate under a few settings on generated BAMs (zlib-compressed, as tophat.py hands them over)
set -u
d=/dev/shm/thj_infl_ab
rm -rf $d; mkdir -p $d
tools/bin/thj_gen --out $d --pairs ${1:-4000000} > /dev/null
shift
for cfg in "$@"; do
  for f in left_seg1 left_map left_reads; do
    echo -n "$cfg $f: "; env $cfg timeout 300 python tools/inflate_bench.py $d/$f.bam 5 2>/dev/null | tail -1
  done
done
rm -rf $d
