#!/bin/bash
# GPU box: per-kernel times of the inflater on generated BAMs (rocprofv3 kernel trace of tools/inflate_bench.py)
set -u
export TMPDIR=/tmp
root=$(pwd)
d=/dev/shm/thj_infl_prof
rm -rf $d; mkdir -p $d
tools/bin/thj_gen --out $d --pairs ${1:-4000000} > /dev/null
for f in left_seg1 left_reads; do
  rm -rf /tmp/pi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pi -o res -- python $root/tools/inflate_bench.py $d/$f.bam 5 > /tmp/pi.log 2>&1)
  echo "== $f"; tail -1 /tmp/pi.log
  python tools/rocpd_summary.py $(find /tmp/pi -name '*.db' | head -1) thj_k
done
rm -rf $d
