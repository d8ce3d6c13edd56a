#!/bin/bash
# GPU box: segment_juncs' own timeline at 2 M pairs (THJ_TRACE / THJ_TIMING)
cd "$(dirname "$0")/../.."; d=/dev/shm/e2; rm -rf $d
timeout 300 python tools/e2e_bench.py --pairs 2000000 --keep $d > /dev/null 2>&1
f() { echo $d/$1; }
segsL=$d/left_seg1.bam,$d/left_seg2.bam,$d/left_seg3.bam,$d/left_seg4.bam; segsR=$d/right_seg1.bam,$d/right_seg2.bam,$d/right_seg3.bam,$d/right_seg4.bam
for i in 1 2; do
THJ_TRACE=1 THJ_TIMING=1 timeout 120 tophat_amd/bin/segment_juncs --no-coverage-search --no-microexon-search --segment-length 25 --sam-header $d/hdr.sam --inner-dist-mean 50 --inner-dist-std-dev 20 \
  $d/ref.fa $d/x.juncs $d/x.ins $d/x.del $d/x.fus $d/left_reads.bam $d/left_map.bam $segsL $d/right_reads.bam $d/right_map.bam $segsR 2>&1 | grep -vE "^\[trace\] [0-9]+ " | head -60
echo ----
done
rm -rf $d
