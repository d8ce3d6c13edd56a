#!/bin/bash
# round 5: thj_k_join / thj_k_finish variants on one box (serial = kernels' own durations)
for jw in 4 3; do for fw in 4 3; do
  echo "== THJ_JOIN_WPE=$jw THJ_FIN_WPE=$fw (serial)"
  THJ_SPAN_SERIAL=1 THJ_JOIN_WPE=$jw THJ_FIN_WPE=$fw python bench.py --steps 6 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null > /tmp/x.json; python tools/show_bench.py /tmp/x.json | grep -E "ms/step|stitch|join|finish"
done; done
