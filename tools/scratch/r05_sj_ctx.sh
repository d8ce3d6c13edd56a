#!/bin/bash
# GPU box: segment_juncs alone at 10 M pairs with 1 / 2 / 3 contexts a GPU (wall clock of the process)
cd "$(dirname "$0")/../.."; d=/dev/shm/e10; rm -rf $d
timeout 600 python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
segsL=$d/left_seg1.bam,$d/left_seg2.bam,$d/left_seg3.bam,$d/left_seg4.bam; segsR=$d/right_seg1.bam,$d/right_seg2.bam,$d/right_seg3.bam,$d/right_seg4.bam
for i in 1 2; do for k in 2 1 3; do
s=$(date +%s.%N)
THJ_CTX_PER_GPU=$k timeout 120 tophat_amd/bin/segment_juncs --no-coverage-search --no-microexon-search --segment-length 25 --sam-header $d/hdr.sam --inner-dist-mean 50 --inner-dist-std-dev 20 \
  $d/ref.fa $d/x.juncs $d/x.ins $d/x.del $d/x.fus $d/left_reads.bam $d/left_map.bam $segsL $d/right_reads.bam $d/right_map.bam $segsR > /dev/null 2>&1
e=$(date +%s.%N); echo "contexts $k: $(python3 -c "print('%.3f' % ($e - $s))") s"; sleep 0.5
done; done
rm -rf $d
