#!/bin/bash
# GPU box: does the number of hardware queues HIP maps the streams onto (GPU_MAX_HW_QUEUES, default 4) serialise the side chains?
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
for q in "" 2 8 12 16; do
  echo -n "GPU_MAX_HW_QUEUES=${q:-default}: "
  env ${q:+GPU_MAX_HW_QUEUES=$q} THJ_BENCH_NO_REPLAY=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms/step %.3f' % d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r05_hwq.txt
