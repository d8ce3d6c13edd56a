#!/bin/bash
# Developer tool (GPU box): SQ instruction counters of the THJ_EXP build under each ablation flag value.
# Usage: tools/exp_pmc.sh "CTR1 CTR2 ..." FLAG [FLAG ...]  ->  gpurun_out/exp_pmc_<flag>.txt   (build first: tools/build_exp.sh)
set -u
ctrs=$1; shift
export TMPDIR=/tmp
root=$(pwd)
cat > /tmp/exp_run.py <<PY
import sys
sys.path.insert(0, "$root")
import tophat_amd.host as h
h.LIB_PATH = "$root/tophat_amd/csrc/libthj_exp.so"
import bench
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
bench.main()
PY
for f in "$@"; do
  out=/tmp/exp_pmc_$f
  rm -rf $out
  (cd /tmp && THJ_EXP_FLAGS=$f timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d $out -o res -- python /tmp/exp_run.py > $out.log 2>&1 </dev/null)
  db=$(find $out -name '*.db' 2>/dev/null | head -1)
  echo "# THJ_EXP_FLAGS=$f counters: $ctrs" > $root/gpurun_out/exp_pmc_$f.txt
  if [ -n "$db" ]; then python $root/tools/rocpd_summary.py $db thj_k_seg >> $root/gpurun_out/exp_pmc_$f.txt; python $root/tools/rocpd_summary.py $db thj_k_stitch >> $root/gpurun_out/exp_pmc_$f.txt; else tail -5 $out.log >> $root/gpurun_out/exp_pmc_$f.txt; fi
done
