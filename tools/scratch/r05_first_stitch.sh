#!/bin/bash
# GPU box: what a long_spanning_reads context's first stitch spends before and in its kernels (THJ_TRACE)
cd "$(dirname "$0")/../.."; d=/dev/shm/e2; rm -rf $d
timeout 300 python tools/e2e_bench.py --pairs 2000000 --keep $d --env THJ_TRACE=1 > /tmp/e.txt 2>&1
python - <<'PY'
import json
t = open("/tmp/e.txt").read(); r = json.loads(t[t.index("{"):])
print({k: r[k] for k in ("segment_juncs_s", "long_spanning_reads_left_s", "long_spanning_reads_right_s")})
PY
sd=left; segs=$d/${sd}_seg1.bam,$d/${sd}_seg2.bam,$d/${sd}_seg3.bam,$d/${sd}_seg4.bam
THJ_TRACE=1 THJ_TIMING=1 timeout 120 tophat_amd/bin/long_spanning_reads --segment-length 25 --sam-header $d/hdr.sam $d/ref.fa $d/${sd}_reads.bam $d/out.juncs $d/out.insertions $d/out.deletions /dev/null $d/span_x.bam $segs 2>&1 | grep -E "first run|context ready|streams|\[timing\]|trace\] [0-2] " | head -60
rm -rf $d
