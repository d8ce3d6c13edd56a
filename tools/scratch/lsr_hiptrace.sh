#!/bin/bash
# HIP API calls of one long_spanning_reads run, longest first (where the first shard's time goes)
d=/dev/shm/thj_tr; rm -rf $d; mkdir -p $d
python tools/e2e_bench.py --pairs 10000000 --keep $d > /dev/null 2>&1
sd=left; segs=$d/${sd}_seg1.bam,$d/${sd}_seg2.bam,$d/${sd}_seg3.bam,$d/${sd}_seg4.bam
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --hip-runtime-trace --stats -d /tmp/ht -o lsr -- $R/tophat_amd/bin/long_spanning_reads --segment-length 25 --sam-header $d/hdr.sam $d/ref.fa $d/${sd}_reads.bam $d/out.juncs $d/out.insertions $d/out.deletions /dev/null $d/span_$sd.bam $segs > /dev/null 2>&1
f=$(find /tmp/ht -name '*hip_api_stats.csv' | head -1)
head -25 $f
t=$(find /tmp/ht -name '*hip_api_trace.csv' | head -1)
echo; echo "longest single calls:"
python3 - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
t0=min(int(r['Start_Timestamp']) for r in rows)
rows.sort(key=lambda r:int(r['End_Timestamp'])-int(r['Start_Timestamp']),reverse=True)
for r in rows[:30]:
    print(r['Function'], r['Thread_Id'], '%.4f'%((int(r['Start_Timestamp'])-t0)/1e9), '%.4f s'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e9))
PY
rm -rf $d
