import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["ms_per_step"], [(k["kernel"], round(k["avg_kernel_ms"],3)) for k in d["kernels"]])
