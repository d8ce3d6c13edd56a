#!/bin/bash
# round 5, VERDICT 4(d): the second long_spanning_reads that is sometimes slow -- three e2e runs on one set of files, every process's timing lines
D=/dev/shm/thj_slow
rm -rf $D
for k in 1 2 3; do
  python tools/e2e_bench.py --pairs 10000000 --keep $D > /tmp/s_$k.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("/tmp/s_$k.json"))
print("run $k:", {k:d[k] for k in ("segment_juncs_s","long_spanning_reads_left_s","long_spanning_reads_right_s","both_stages_s")}, "outside main", d.get("long_spanning_reads_left_before_main_after_report_s"), d.get("long_spanning_reads_right_before_main_after_report_s"))
for st in ("long_spanning_reads_left","long_spanning_reads_right"):
    print("  ", st, [l.replace("[timing] ","") for l in d[st+"_log_tail"] if "timing" in l or "worker-seconds" in l][:9])
PY
done
rm -rf $D
