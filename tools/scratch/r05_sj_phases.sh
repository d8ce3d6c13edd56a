#!/bin/bash
# GPU box: where segment_juncs' time inside the GPU's lock goes (10 M pairs of the mix): the ingest's phases with a synchronisation at every mark
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; d=/dev/shm/e10
python tools/e2e_bench.py --pairs ${1:-10000000} --keep $d > /tmp/e0.txt 2>&1
python tools/e2e_bench.py --pairs ${1:-10000000} --keep $d --env THJ_INGEST_TIMING=1 > /tmp/e1.txt 2>&1
python - <<'PY' | tee gpurun_out/r05_sj_phases.txt
import ast, json
for f, what in (("/tmp/e0.txt", "as the product runs"), ("/tmp/e1.txt", "THJ_INGEST_TIMING=1 (a stream synchronisation at every mark)")):
    t = open(f).read()
    r = json.loads(t[t.index("{"):])
    print("#", what, {k: r[k] for k in ("segment_juncs_s", "long_spanning_reads_left_s", "long_spanning_reads_right_s")})
    for st in ("segment_juncs", "long_spanning_reads_left"):
        print(st)
        for x in r.get(st + "_log_tail", []): print("   ", x)
PY
rm -rf $d
