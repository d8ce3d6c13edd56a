d=/dev/shm/e10; mkdir -p $d
python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
for e in ${SJ_ENVS:-"THJ_SHARDS=48" "THJ_SHARDS=32" "THJ_SHARDS=24" "THJ_SHARDS=16" "THJ_SHARDS=12" "THJ_SHARDS=24 THJ_CTX_PER_GPU=3" "THJ_SHARDS=48 THJ_CTX_PER_GPU=3"}; do
python tools/e2e_bench.py --pairs 10000000 --keep $d --env $e > gpurun_out/x.json 2>/dev/null
python - "$e" <<PY
import json,sys; d=json.load(open("gpurun_out/x.json")); print(sys.argv[1], d["segment_juncs_s"], d["long_spanning_reads_left_s"], d["long_spanning_reads_right_s"]); 
if "TIMING" in sys.argv[1]: print("\n".join(d["segment_juncs_log_tail"]))
PY
done
rm -rf $d
