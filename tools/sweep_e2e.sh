D=/dev/shm/e2e4m
tools/bin/thj_gen --out $D --pairs 4000000 > /dev/null
python tools/e2e_bench.py --pairs 4000000 --keep $D 2>&1 | grep -E "_s\"|pairs_per_s|worker-seconds|shards|timing"
S=$(for k in 1 2 3 4; do printf "$D/left_seg$k.bam,"; done); S=${S%,}
for W in 32; do
THJ_WORKERS=$W THJ_TIMING=1 tophat_amd/bin/long_spanning_reads --segment-length 25 --sam-header $D/hdr.sam $D/ref.fa $D/left_reads.bam $D/out.juncs $D/out.insertions $D/out.deletions /dev/null $D/x.bam $S; echo "rc=$?"
done
dmesg 2>/dev/null | tail -5
timeout 600 python bench.py > gpurun_out/r02_b_bench.json 2> gpurun_out/r02_b_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_b_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['e2e'], indent=1))
PY
tail -3 gpurun_out/r02_b_bench.err
