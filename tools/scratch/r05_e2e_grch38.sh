#!/bin/bash
# round 5: files in -> files out on configs[2]'s genome (25 contigs, 3.09 Gb), 12.5 M pairs = its per-GPU shard at 8 GPUs; first run
# packs the reference and writes the cache, second run maps it; then the chr20-sized genome at the same pairs for comparison
D=/dev/shm/thj_g38
rm -rf $D
for k in 1 2; do
  echo "== GRCh38-sized, run $k"
  timeout 1500 python tools/e2e_bench.py --grch38 --introns 300000 --pairs ${PAIRS:-12500000} --keep $D > /tmp/g38_$k.json 2> /tmp/g38_$k.err || tail -5 /tmp/g38_$k.err
  python - <<PY
import json
d=json.load(open("/tmp/g38_$k.json"))
print({k:d[k] for k in ("pairs","gen_seconds","segment_juncs_s","long_spanning_reads_left_s","long_spanning_reads_right_s","both_stages_s","pairs_per_s_both_stages","outside_main_s") if k in d})
for st in ("segment_juncs","long_spanning_reads_left"):
    for l in d[st+"_log_tail"] + d.get(st+"_log_all", []):
        if "timing" in l or "reference" in l: print("   ", st, l)
PY
done
echo "== no cache"
python tools/e2e_bench.py --grch38 --introns 300000 --pairs ${PAIRS:-12500000} --keep $D --env THJ_GENOME_CACHE=0 > /tmp/g38_n.json 2>/tmp/g38_n.err; python -c "
import json; d=json.load(open('/tmp/g38_n.json')); print({k:d[k] for k in ('segment_juncs_s','long_spanning_reads_left_s','long_spanning_reads_right_s','both_stages_s','pairs_per_s_both_stages')})"
rm -rf $D
echo "== chr20-sized, same pairs"
python tools/e2e_bench.py --pairs ${PAIRS:-12500000} > /tmp/c20.json 2>/tmp/c20.err; python -c "
import json; d=json.load(open('/tmp/c20.json')); print({k:d[k] for k in ('gen_seconds','segment_juncs_s','long_spanning_reads_left_s','long_spanning_reads_right_s','both_stages_s','pairs_per_s_both_stages')})"
