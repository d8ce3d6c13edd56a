#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of a short bench run, the thj kernels only.  Usage: tools/kstats.sh TAG [bench.py flags]
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
rm -rf /tmp/prof_$tag
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o res -- python $root/bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline --e2e-pairs 0 --no-pmc --detail /tmp/bench_detail_$tag.json > /tmp/prof_$tag.log 2>&1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py $* --steps 5 --warmup 1 --no-cpu-baseline --e2e-pairs 0 --no-pmc   (MI355X, $tag; microseconds)";
  python tools/rocpd_summary.py $(find /tmp/prof_$tag -name '*.db' | head -1) thj_k; } > $root/gpurun_out/${tag}_kernel_stats.txt
cat $root/gpurun_out/${tag}_kernel_stats.txt | cut -c1-70,82-130
