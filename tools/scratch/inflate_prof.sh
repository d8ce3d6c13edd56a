#!/bin/bash
# GPU box: per-kernel times of the inflater on generated BAMs (rocprofv3 kernel trace of tools/inflate_bench.py)
#   tools/inflate_prof.sh PAIRS "file[:members] ..."
set -u
export TMPDIR=/tmp
root=$(pwd)
d=/dev/shm/thj_infl_prof
rm -rf $d; mkdir -p $d
tools/bin/thj_gen --out $d --pairs ${1:-4000000} > /dev/null
for spec in ${2:-left_seg1 left_reads}; do
  f=${spec%%:*}; n=""; [ "$spec" != "$f" ] && n=${spec##*:}
  rm -rf /tmp/pi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pi -o res -- python $root/tools/inflate_bench.py $d/$f.bam 5 $n > /tmp/pi.log 2>&1)
  echo "== $spec"; grep "^{" /tmp/pi.log | tail -1
  python tools/rocpd_summary.py $(find /tmp/pi -name '*.db' | head -1) thj_k | cut -c1-50,79-
done
rm -rf $d
