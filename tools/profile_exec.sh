#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 evidence for the EXECUTABLES (the files-in -> files-out metric), per round.
#   tools/profile_exec.sh TAG [PAIRS]  ->  gpurun_out/TAG_exec_kernel_stats_<stage>.txt (+ TAG_exec_e2e_plain.json: the same run unprofiled)
# Kernel trace only; counter passes are separate (tools/pmc_pass.sh).  The executables run with THJ_NO_HANDOFF=1 under the profiler
# (one process, so that the tool sees the kernels).
set -u
tag=$1
pairs=${2:-10000000}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
d=/dev/shm/thj_prof_exec
rm -rf $d; mkdir -p $d
python tools/e2e_bench.py --pairs $pairs --keep $d > $out/${tag}_exec_e2e_plain.json 2> $out/${tag}_exec_e2e_plain.err
rm -rf /tmp/p_segment_juncs /tmp/p_lsr_left /tmp/p_lsr_right
(cd /tmp && python $root/tools/e2e_bench.py --pairs $pairs --keep $d --env THJ_NO_HANDOFF=1 THJ_EXIT_HANDLERS=1 "THJ_EXEC_PREFIX=rocprofv3 --kernel-trace --stats -d /tmp/p_{stage} -o res --" > $out/${tag}_exec_e2e_profiled.json 2> $out/${tag}_exec_e2e_profiled.err)
for st in segment_juncs lsr_left; do
  db=$(find /tmp/p_$st -name '*.db' | head -1)
  [ -n "$db" ] && { echo "# rocprofv3 --kernel-trace --stats -- $st (THJ_NO_HANDOFF=1), $pairs pairs of 2x100 bp, MI355X, $tag; durations in microseconds";
    python tools/rocpd_summary.py $db; } > $out/${tag}_exec_kernel_stats_$st.txt
done
for f in left_seg1 left_reads; do python tools/inflate_bench.py $d/$f.bam 5; done > $out/${tag}_inflate_bench.txt 2>&1
rm -rf $d
head -c 1500 $out/${tag}_exec_e2e_plain.json
