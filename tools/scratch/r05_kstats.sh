#!/bin/bash
# GPU box: rocprofv3 kernel stats of the timed region's kind of steps (no serial replay), beside the bench's own HIP-event figures of the same run
export TMPDIR=/tmp; root=$(pwd); mkdir -p gpurun_out; rm -rf /tmp/kt
(cd /tmp && THJ_BENCH_NO_REPLAY=1 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc > /tmp/kt.json 2>/dev/null)
{ echo "# THJ_BENCH_NO_REPLAY=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc   (MI355X, end of round 5); durations in microseconds"
  python tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) thj_k
  echo "# the same run's own figures (HIP events around each kernel on its stream, bench.py's avg_kernel_ms):"
  python tools/show_bench.py /tmp/kt.json | head -18; } > gpurun_out/r05_z_kernel_stats_bench10M.txt
cat gpurun_out/r05_z_kernel_stats_bench10M.txt | cut -c1-160
