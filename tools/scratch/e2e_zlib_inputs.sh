#!/bin/bash
# scratch (GPU box): the e2e run on inputs compressed by zlib at its default level (what samtools / tophat.py's own writers produce)
# instead of this build's fast DEFLATE, which thj_gen's BamWriter uses by default.  usage: tools/e2e_zlib_inputs.sh [pairs]
pairs=${1:-10000000}
for lvl in fast -1; do
  d=/dev/shm/thj_e2e_z; rm -rf $d; mkdir -p $d
  if [ $lvl = fast ]; then tools/bin/thj_gen --out $d --pairs $pairs --read-len 100 --genome-len 64444167 --introns 20000 > /dev/null
  else THJ_BGZF_LEVEL=$lvl tools/bin/thj_gen --out $d --pairs $pairs --read-len 100 --genome-len 64444167 --introns 20000 > /dev/null; fi
  du -sh $d | sed "s/^/inputs ($lvl): /"
  for rep in 1 2; do
    python tools/e2e_bench.py --pairs $pairs --keep $d > gpurun_out/x.json 2>/dev/null
    python - "$lvl" <<PY
import json,sys; d=json.load(open("gpurun_out/x.json")); print(sys.argv[1], d["input_bytes"], d["segment_juncs_s"], d["long_spanning_reads_left_s"], d["long_spanning_reads_right_s"], d["pairs_per_s_both_stages"], d["junctions"])
PY
  done
  THJ_LIB= python tools/inflate_bench.py $d/left_seg1.bam 3 2>&1 | tail -1
done
rm -rf /dev/shm/thj_e2e_z
