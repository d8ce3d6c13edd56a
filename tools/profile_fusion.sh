#!/bin/bash
# Run on the GPU box (via gpurun): the fusion-search configuration (the shape of configs[3] at 100 bp: --fusion-min-dist 100000, 2 % chimeric
# left reads) -- bench line and rocprofv3 kernel stats.   tools/profile_fusion.sh TAG  ->  gpurun_out/TAG_fusion_bench10M.json, TAG_fusion_kernel_stats_bench10M.txt
set -u
tag=$1
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
args="--plain --fusion-search --fusion-frac 0.02 --e2e-pairs 0"
timeout 900 python bench.py $args --steps 10 --warmup 3 > $out/${tag}_fusion_bench10M.json 2> $out/${tag}_fusion_bench10M.err
rm -rf /tmp/prof_fu
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_fu -o res -- python $root/bench.py $args --steps 5 --warmup 1 --no-cpu-baseline > /tmp/prof_fu.log 2>&1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py $args --steps 5 --warmup 1 --no-cpu-baseline   (MI355X, $tag)";
  echo "# durations in microseconds"; python tools/rocpd_summary.py $(find /tmp/prof_fu -name '*.db' | head -1);
  echo; echo "# per dispatch (left side, right side, ...)"; python tools/rocpd_dispatches.py $(find /tmp/prof_fu -name '*.db' | head -1) usion\( | tail -8; } > $out/${tag}_fusion_kernel_stats_bench10M.txt
tail -1 $out/${tag}_fusion_bench10M.json | cut -c1-300
