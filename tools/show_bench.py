import json,sys
d=json.load(open(sys.argv[1]))
print("ms/step", round(d["ms_per_step"],3), "value %.3e"%d["value"], d["config"]["reads_to_tiers_1_2or_fusion_3"])
for k in d["kernels"]: print("%-55s %.4f ms (alone %s)  frac %.3f (alone %s)  traffic/alg %s"%(k["kernel"], k["avg_kernel_ms"], ("%.4f" % k["avg_kernel_ms_alone"]) if k.get("avg_kernel_ms_alone") else "-", k["frac"], ("%.3f" % k["frac_alone"]) if k.get("frac_alone") is not None else "-", ("%.2f"%(k["traffic"]/k["algorithmic_bytes_8d_per_launch"])) if k.get("traffic") and k["algorithmic_bytes_8d_per_launch"]>0 else "-"))
print(d["roofline_all_kernels"])
print("roofline", {k:d["roofline"].get(k) for k in ("kernel","frac","frac_alone","frac_step","avg_kernel_ms","avg_kernel_ms_alone")})
