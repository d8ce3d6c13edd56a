#!/bin/bash
# GPU box: (1) time of the inflater against the number of members in a launch; (2) rocprofv3 counter passes (one list per pass,
# kernel trace only) over tools/inflate_bench.py.   tools/inflate_pmc.sh PAIRS FILE "CTR CTR" "CTR" ...
set -u
export TMPDIR=/tmp
root=$(pwd)
d=/dev/shm/thj_infl_pmc
rm -rf $d; mkdir -p $d
tools/bin/thj_gen --out $d --pairs $1 > /dev/null
f=$2; shift; shift
for n in 250 500 1000 2000 4000 8000 16000; do echo -n "members<=$n: "; python tools/inflate_bench.py $d/$f.bam 5 $n 2>/dev/null | tail -1; done
i=0
for ctrs in "$@"; do
  rm -rf /tmp/pi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pi -o res -- python $root/tools/inflate_bench.py $d/$f.bam 2 > /tmp/pi.log 2>&1)
  echo "# counters: $ctrs"
  db=$(find /tmp/pi -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db thj_k | grep -v "^kernel  *calls\|^ *$"; else tail -3 /tmp/pi.log; fi
  i=$((i+1))
done
rm -rf $d
