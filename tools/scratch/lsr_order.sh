#!/bin/bash
d=/dev/shm/thj_ord; rm -rf $d; mkdir -p $d
python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
run() { sd=$1; segs=$d/${sd}_seg1.bam,$d/${sd}_seg2.bam,$d/${sd}_seg3.bam,$d/${sd}_seg4.bam
  t0=$(date +%s.%N)
  THJ_TIMING=1 tophat_amd/bin/long_spanning_reads --segment-length 25 --sam-header $d/hdr.sam $d/ref.fa $d/${sd}_reads.bam $d/out.juncs $d/out.insertions $d/out.deletions /dev/null $d/span_$sd.bam $segs 2>&1 | grep -E "all shards|device calls|GPU's lock" | tr '\n' ' '
  t1=$(date +%s.%N); python3 -c "print('  $sd wall %.3f' % ($t1 - $t0))"; }
for sd in right left right left left right right; do run $sd; done
ls -la $d/left_seg1.bam $d/right_seg1.bam $d/left_reads.bam $d/right_reads.bam | awk '{print $5, $9}'
rm -rf $d
