#!/bin/bash
# scratch (GPU box): both inflate kernels on generated BAM files of several sizes
d=/dev/shm/thj_infl; rm -rf $d; mkdir -p $d
tools/bin/thj_gen --out $d --pairs ${1:-2000000} --read-len 100 --genome-len 64444167 --introns 20000 > /dev/null
for f in ${2:-left_seg1.bam left_reads.bam}; do
  for mode in wave lanes; do THJ_INFLATE=$mode python tools/inflate_bench.py $d/$f 3 2>&1 | tail -1 | sed "s/^/$mode /"; done
done
rm -rf $d
