#!/bin/bash
# GPU box: the e2e leg at 10 M pairs of the mix (twice: the second with warm page cache), the inflater on its files, then the GPU test-suite
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; d=/dev/shm/e10
{ python tools/e2e_bench.py --pairs 10000000 --keep $d 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
r = json.loads(t[t.index('{'):])
print({k: r[k] for k in r if k.endswith('_s') or k in ('pairs', 'both_stages_s', 'value')})
for k in ('stage_timing',):
    for n, v in (r.get(k) or {}).items():
        print(n); [print('   ', x) for x in v]
"
  for f in left_seg1.bam left_reads.bam left_map.bam; do echo -n "$f: "; python tools/inflate_bench.py $d/$f 5 2>/dev/null | tail -1; done; } 2>&1 | tee gpurun_out/r05_e_e2e_inflate.txt
rm -rf $d
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r05_e_gpu_tests.txt
