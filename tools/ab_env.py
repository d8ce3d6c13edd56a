#!/usr/bin/env python3
"""Developer tool (GPU box): the resident-data bench once per environment setting, in one gpurun call (boxes differ by a few percent).
   python tools/ab_env.py "" "THJ_FIN_WPE=3" "THJ_FIN_WPE=3 THJ_JOIN_WPE=4" [-- extra bench.py flags]
Prints ms per step and every kernel's duration as run / alone."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--"); extra = args[k + 1:]; args = args[:k]
for i, setting in enumerate(args or [""]):
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=", 1); env[k] = v
    det = "/tmp/ab_env_%d.json" % i
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--e2e-pairs", "0", "--no-pmc", "--detail", det] + extra,
                       env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print("[%s] FAILED rc %d: %s" % (setting, r.returncode, r.stderr[-1500:])); continue
    d = json.load(open(det))
    print("[%s] ms/step %.3f  frac_step %.3f  streams_independent %s" % (setting or "default", d["ms_per_step"], d["roofline"]["frac_step"], d["roofline"].get("streams_independent")))
    print("   " + "  ".join("%s %.3f/%s" % (k["kernel"].replace("thj_k_", "").split("<")[0][:22], k["avg_kernel_ms"], ("%.3f" % k["avg_kernel_ms_alone"]) if k.get("avg_kernel_ms_alone") else "-") for k in d["kernels"]))
