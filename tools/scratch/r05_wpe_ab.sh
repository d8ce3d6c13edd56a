#!/bin/bash
for v in "THJ_ABUT_WPE=4 THJ_FIN_WPE=4" "THJ_ABUT_WPE=6 THJ_FIN_WPE=4" "THJ_ABUT_WPE=6 THJ_FIN_WPE=5" "THJ_ABUT_WPE=6 THJ_FIN_WPE=6"; do
  echo "== $v"; env $v THJ_BENCH_NO_REPLAY=1 THJ_SPAN_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null > /tmp/x.json; python tools/show_bench.py /tmp/x.json | grep -E "ms/step|thj_k_join |finish"
  env $v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null > /tmp/y.json; python tools/show_bench.py /tmp/y.json | grep -E "ms/step"
done
