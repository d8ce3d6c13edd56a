#!/bin/bash
# second A/B on one box, the product now at three workgroups per CU for tier 1 and the packed tier (reads of up to four segments):
# two and four workgroups' worth of registers against it
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --e2e-pairs 0 --no-pmc"
for rep in 1 2; do
  $B > gpurun_out/ab2_default_$rep.json 2>/dev/null
  THJ_LEAN_WPE=2 THJ_PACK_WPE=2 $B > gpurun_out/ab2_wpe2_$rep.json 2>/dev/null
  THJ_LEAN_WPE=4 THJ_PACK_WPE=4 $B > gpurun_out/ab2_wpe4_$rep.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab2_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        ks = {k["kernel"]: round(k["avg_kernel_ms"], 3) for k in d["kernels"]}
        print(f, round(d["ms_per_step"], 3), ks.get("thj_k_stitch_contig"), ks.get("thj_k_stitch"), ks.get("thj_k_stitch_pack"), d["events"]["spanning_records_per_step"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
