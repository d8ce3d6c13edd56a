#!/bin/bash
# scratch (GPU box): inflate rate and e2e wall clock for several builds of the library (directories under tophat_amd/csrc holding a libthj_hip.so)
d=/dev/shm/thj_e2e_sweep; rm -rf $d; mkdir -p $d
pairs=${1:-8000000}; shift
python tools/e2e_bench.py --pairs $pairs --keep $d > gpurun_out/sweep_base.json 2>/dev/null
for v in "$@"; do
  L=$PWD/tophat_amd/csrc/$v
  for f in left_seg1.bam left_map.bam; do THJ_LIB=$L/libthj_hip.so python tools/inflate_bench.py $d/$f 3 2>&1 | tail -1 | sed "s/^/$v /"; done
  for rep in 1 2; do
  python tools/e2e_bench.py --pairs $pairs --keep $d --env LD_LIBRARY_PATH=$L > gpurun_out/sweep_x.json 2>gpurun_out/sweep_x.err || { echo "$v FAILED"; tail -5 gpurun_out/sweep_x.err; continue; }
  python - "$v" <<PY
import json,sys; d=json.load(open("gpurun_out/sweep_x.json")); print(sys.argv[1], d["segment_juncs_s"], d["long_spanning_reads_left_s"], d["long_spanning_reads_right_s"], d["pairs_per_s_both_stages"], d["junctions"])
PY
  done
done
rm -rf $d
