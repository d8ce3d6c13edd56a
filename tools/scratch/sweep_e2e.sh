#!/bin/bash
# scratch: e2e timing of the executables under a few environment settings (GPU box); one generated data set, reused
d=/dev/shm/thj_e2e_sweep
rm -rf $d; mkdir -p $d
pairs=${1:-8000000}
python tools/e2e_bench.py --pairs $pairs --keep $d > gpurun_out/sweep_base.json 2>/dev/null
python - <<PY
import json; d=json.load(open("gpurun_out/sweep_base.json")); print("base", d["segment_juncs_s"], d["long_spanning_reads_left_s"], d["long_spanning_reads_right_s"], d["pairs_per_s_both_stages"])
PY
shift
for cfg in "$@"; do
  python tools/e2e_bench.py --pairs $pairs --keep $d --env $cfg > gpurun_out/sweep_x.json 2>gpurun_out/sweep_x.err || { echo "$cfg FAILED"; tail -5 gpurun_out/sweep_x.err; continue; }
  python - "$cfg" <<PY
import json,sys; d=json.load(open("gpurun_out/sweep_x.json")); print(sys.argv[1], d["segment_juncs_s"], d["long_spanning_reads_left_s"], d["long_spanning_reads_right_s"], d["pairs_per_s_both_stages"], d["junctions"])
PY
done
rm -rf $d
