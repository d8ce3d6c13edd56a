#!/bin/bash
# Run on the GPU box: the kernel timeline of the last step of a short bench run.  Usage: tools/timeline.sh TAG [bench.py flags]
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
rm -rf /tmp/prof_$tag
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o res -- python $root/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --e2e-pairs 0 --no-pmc > /tmp/prof_$tag.log 2>&1)
python tools/step_timeline.py $(find /tmp/prof_$tag -name '*.db' | head -1) > $root/gpurun_out/${tag}_timeline.txt
cat $root/gpurun_out/${tag}_timeline.txt
