#!/bin/bash
# Run on the GPU box: one rocprofv3 counter pass per argument (each a space-separated counter list) over a short
# bench run (extra bench.py flags via BENCH_ARGS); per-kernel sums land in gpurun_out/pmc_<tag>_<i>.txt.  Usage: tools/pmc_pass.sh TAG "CTR1 CTR2" "CTR3" ...
set -u
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
i=0
for ctrs in "$@"; do
  out=/tmp/pmc_${tag}_$i
  rm -rf $out
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d $out -o res -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-pairs 0 --no-pmc ${BENCH_ARGS:-} > $out.log 2>&1)
  db=$(find $out -name '*.db' | head -1)
  echo "# counters: $ctrs" > $root/gpurun_out/pmc_${tag}_$i.txt
  if [ -n "$db" ]; then python $root/tools/rocpd_summary.py $db thj_k >> $root/gpurun_out/pmc_${tag}_$i.txt; else tail -5 $out.log >> $root/gpurun_out/pmc_${tag}_$i.txt; fi
  i=$((i+1))
done
