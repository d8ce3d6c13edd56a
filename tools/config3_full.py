#!/usr/bin/env python3
"""GPU box: BASELINE configs[2] at full size through the executables (bench.e2e_config3_full) -> gpurun_out/<tag>_config3_full.json
   python tools/config3_full.py TAG [PAIRS]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
tag = sys.argv[1]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
sys.argv = ["bench.py"]
args = bench.parse_args()
try:
    res = bench.e2e_config3_full(args, pairs)
except Exception as e:      # noqa: BLE001
    res = {"error": repr(e)[-3000:], "ok": False}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "%s_config3_full.json" % tag), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not isinstance(v, (dict, list))}))
