#!/bin/bash
# round-2 evidence runs (GPU box): default bench + rocprof stats + PMC traffic, then the shards of configs[2] and configs[4]
set -u
out=gpurun_out
tag=${1:-r02_a}
bash tools/profile_round.sh $tag
cp $out/${tag}_pmc_traffic.json profiles/r02_pmc_traffic.json 2>/dev/null
timeout 600 python bench.py --genome grch38 --introns 300000 --pairs 12500000 --no-cpu-baseline --e2e-pairs 0 > $out/r02_bench_config3_shard_grch38_12.5Mpairs.json 2>/dev/null
timeout 600 python bench.py --read-len 50 --genome grch38 --introns 300000 --intron-max 499999 --pairs 12500000 --no-cpu-baseline --e2e-pairs 0 > $out/r02_bench_config5_shard_2x50bp_intron500k.json 2>/dev/null
timeout 900 python bench.py --read-len 50 --genome grch38 --introns 300000 --intron-max 499999 --pairs 12500000 --coverage-search 0.2 --steps 3 --warmup 1 --no-cpu-baseline --e2e-pairs 0 > $out/r02_bench_config5_shard_coverage_search.json 2>/dev/null
timeout 600 python bench.py --read-len 150 --genome grch38 --introns 300000 --pairs 6250000 --no-cpu-baseline --e2e-pairs 0 > $out/r02_bench_config4_shape_2x150bp_no_fusions.json 2>/dev/null
for f in $out/r02_bench_config*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  value %.3g ms/step %.2f frac %.3f all-kernels %.0f GB/s events %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_all_kernels"]["achieved"], d["events"]))
except Exception as e: print("  FAILED", e)
PY
done
python - $tag <<'PY'
import json, sys
d=json.loads(open('gpurun_out/%s_bench10M.json' % sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ("value","ms_per_step","roofline","roofline_all_kernels","e2e","cpu_baseline")}, indent=1)[:3500])
PY
