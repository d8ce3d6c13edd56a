#!/bin/bash
# round 5: evidence for DESIGN.md section 3 -- the kernels' own durations (one stream), the overlapped line, rocprofv3 kernel stats of the default bench command
export TMPDIR=/tmp
root=$(pwd)
{ echo "# THJ_SPAN_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc  (stage 2 on one stream: the kernels' own durations; stage 1 as the product runs it)"
  THJ_SPAN_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null > /tmp/x.json; python tools/show_bench.py /tmp/x.json
  echo "# the same as the product runs it (two sides beside each other)"
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null > /tmp/y.json; python tools/show_bench.py /tmp/y.json
  echo "# --plain (no multihit family, no deletion reads)"
  python bench.py --plain --steps 10 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null > /tmp/z.json; python tools/show_bench.py /tmp/z.json
} > gpurun_out/r05_b_serial_kernel_ms.txt 2>&1
rm -rf /tmp/kt; (cd /tmp && THJ_BENCH_NO_REPLAY=1 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline --e2e-pairs 0 --no-pmc > /dev/null 2>&1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --e2e-pairs 0 --no-pmc   (MI355X, round 5); durations in microseconds"; python tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) thj_k; } > gpurun_out/r05_b_kernel_stats_bench10M.txt
cat gpurun_out/r05_b_serial_kernel_ms.txt
