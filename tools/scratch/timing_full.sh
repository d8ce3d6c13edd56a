#!/bin/bash
d=/dev/shm/thj_tf; rm -rf $d; mkdir -p $d
python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
segsL=$d/left_seg1.bam,$d/left_seg2.bam,$d/left_seg3.bam,$d/left_seg4.bam; segsR=$d/right_seg1.bam,$d/right_seg2.bam,$d/right_seg3.bam,$d/right_seg4.bam
for i in 1 2; do
echo "== segment_juncs"; THJ_TIMING=1 tophat_amd/bin/segment_juncs --no-coverage-search --no-microexon-search --segment-length 25 --sam-header $d/hdr.sam --inner-dist-mean 50 --inner-dist-std-dev 20 $d/ref.fa $d/o.juncs $d/o.ins $d/o.del $d/o.fus $d/left_reads.bam $d/left_map.bam $segsL $d/right_reads.bam $d/right_map.bam $segsR 2>&1 | grep -E "timing|worker|ingest"
echo "== long_spanning_reads left"; THJ_TIMING=1 tophat_amd/bin/long_spanning_reads --segment-length 25 --sam-header $d/hdr.sam $d/ref.fa $d/left_reads.bam $d/o.juncs $d/o.ins $d/o.del /dev/null $d/span_l.bam $segsL 2>&1 | grep -E "timing|worker|ingest"
done
wc -l $d/o.juncs $d/o.del
rm -rf $d
