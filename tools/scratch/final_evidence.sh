#!/bin/bash
# end-of-round evidence on one box: the inflater's rate on the e2e leg's files (HEAD) and the --plain workload's profile
mkdir -p gpurun_out
d=/dev/shm/thj_final; rm -rf $d; mkdir -p $d
python tools/e2e_bench.py --pairs 10000000 --keep $d > gpurun_out/r04_final_e2e_10M_mix.json 2>/dev/null
ls $d | head -30 > gpurun_out/r04_final_e2e_files.txt
{ echo "# python tools/inflate_bench.py <file> 5 -- members of the e2e leg's files, input and output resident (HEAD of round 4)";
  for f in left_seg1.bam left_reads.bam left_map.bam; do [ -f $d/$f ] && python tools/inflate_bench.py $d/$f 5 2>/dev/null | tail -1; done; } > gpurun_out/r04_final_inflate_bench.txt
cat gpurun_out/r04_final_inflate_bench.txt
rm -rf $d
bash tools/profile_round.sh r04_final_plain plain
