mkdir -p gpurun_out; export TMPDIR=/tmp; root=$(pwd)
args="--plain --pairs 6250000 --read-len 150 --genome grch38 --introns 300000 --e2e-pairs 0 --no-cpu-baseline"
timeout 1200 python bench.py $args --fusion-search --fusion-frac 0.02 --steps 5 --warmup 2 > gpurun_out/r03_j_bench_config4_shape_2x150bp_fusion_search_2pct_chimeric.json 2> gpurun_out/c4a.err; tail -2 gpurun_out/c4a.err
timeout 1200 python bench.py $args --steps 5 --warmup 2 > gpurun_out/r03_j_bench_config4_shape_2x150bp_no_fusion_search.json 2> gpurun_out/c4b.err; tail -2 gpurun_out/c4b.err
rm -rf /tmp/pf; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pf -o res -- python $root/bench.py $args --fusion-search --fusion-frac 0.02 --steps 2 --warmup 1 > /tmp/pf.log 2>&1)
python tools/rocpd_summary.py $(find /tmp/pf -name "*.db" | head -1) thj_k > gpurun_out/r03_j_config4_fusion_kernel_stats.txt
python tools/rocpd_dispatches.py $(find /tmp/pf -name "*.db" | head -1) usion\( | tail -4 >> gpurun_out/r03_j_config4_fusion_kernel_stats.txt
