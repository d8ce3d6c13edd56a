#!/bin/bash
# GPU box: stage times of the e2e run with and without the output hand-off, several runs each (teardown collisions show as a slow right side)
pairs=${1:-10000000}
d=/dev/shm/e2e_probe
mkdir -p gpurun_out/probe
python tools/e2e_bench.py --pairs $pairs --keep $d > /dev/null 2>&1
for mode in dflt nohandoff linger300 ctx1; do
  case $mode in dflt) envs="";; nohandoff) envs="--env THJ_NO_HANDOFF=1";; linger300) envs="--env THJ_HANDOFF_LINGER_MS=300";; ctx1) envs="--env THJ_CTX_PER_GPU=1";; esac
  for i in 1 2 3 4; do
    python tools/e2e_bench.py --pairs $pairs --keep $d $envs > gpurun_out/probe/${mode}_$i.json 2> /dev/null
    python - <<P
import json
d=json.load(open("gpurun_out/probe/${mode}_$i.json"))
print("$mode $i", d["segment_juncs_s"], d["long_spanning_reads_left_s"], d["long_spanning_reads_right_s"], d["both_stages_s"])
P
  done
done
