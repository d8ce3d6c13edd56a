#!/bin/bash
# GPU box: the inflater on the e2e leg's generated files (rate, then thj_k_huffp's phase clocks from the THJ_EXP build) + the ingest tests.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; d=/tmp/e2e_in
python tools/e2e_bench.py --pairs ${1:-2000000} --keep $d > /tmp/e2e_gen.log 2>&1 || tail -5 /tmp/e2e_gen.log
ls $d | head -20
{ for f in left_seg1.bam left_reads.bam left_map.bam; do [ -f $d/$f ] && { echo -n "$f: "; python tools/inflate_bench.py $d/$f 5 2>/dev/null | tail -1; }; done
  [ -f tophat_amd/csrc/libthj_exp.so ] && python tools/scratch/huffp_timing.py $d/left_seg1.bam 2>&1 | tail -7; } | tee gpurun_out/r05_inflate.txt
timeout 900 python -m pytest tests/test_gpu_ingest.py -x -q 2>&1 | tail -3 | tee -a gpurun_out/r05_inflate.txt
