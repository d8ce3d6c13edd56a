import json,os,sys
# the per-kernel table lives in the detail file (bench.py --detail; stdout carries the contract's line only)
p=sys.argv[1] if len(sys.argv)>1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),"bench_detail.json")
d=json.load(open(p))
print(d["ms_per_step"], [(k["kernel"], round(k["avg_kernel_ms"],3)) for k in d["kernels"]])
