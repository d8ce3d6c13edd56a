#!/bin/bash
# A/B on one box: tier 1 and the packed tier at three workgroups of four waves per CU (168 VGPRs, no spills) against the product's
# four waves per SIMD (128 VGPRs).  THJ_LEAN_SMALL / THJ_PACK_SMALL are developer switches of thj_span_run_async.
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --e2e-pairs 0 --no-pmc"
for rep in 1 2; do
  $B > gpurun_out/ab_base_$rep.json 2>/dev/null
  THJ_PACK_SMALL=1 $B > gpurun_out/ab_packsmall_$rep.json 2>/dev/null
  THJ_LEAN_SMALL=1 $B > gpurun_out/ab_leansmall_$rep.json 2>/dev/null
  THJ_PACK_SMALL=1 THJ_LEAN_SMALL=1 $B > gpurun_out/ab_both_$rep.json 2>/dev/null
done
THJ_PACK_SMALL=1 THJ_LEAN_SMALL=1 timeout 600 python -m pytest tests/test_gpu_spanning.py tests/test_ref_regression_gpu.py tests/test_golden_gpu.py tests/test_gpu_fullsize_properties.py -x -q > gpurun_out/ab_small_tests.log 2>&1
tail -2 gpurun_out/ab_small_tests.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        ks = {k["kernel"]: round(k["avg_kernel_ms"], 3) for k in d["kernels"]}
        print(f, round(d["ms_per_step"], 3), ks.get("thj_k_stitch_contig"), ks.get("thj_k_stitch"), ks.get("thj_k_stitch_pack"), d["events"]["spanning_records_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
