#!/bin/bash
d=/dev/shm/thj_tr; rm -rf $d; mkdir -p $d
python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
sd=left; segs=$d/${sd}_seg1.bam,$d/${sd}_seg2.bam,$d/${sd}_seg3.bam,$d/${sd}_seg4.bam
for i in 1 2; do
THJ_TRACE=1 THJ_TIMING=1 tophat_amd/bin/long_spanning_reads --segment-length 25 --sam-header $d/hdr.sam $d/ref.fa $d/${sd}_reads.bam $d/out.juncs $d/out.insertions $d/out.deletions /dev/null $d/span_$sd.bam $segs 2> /tmp/lsr.log
python tools/lsr_trace.py /tmp/lsr.log | head -14; python tools/lsr_trace.py /tmp/lsr.log | tail -10; grep timing /tmp/lsr.log
done
rm -rf $d
