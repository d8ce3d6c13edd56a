mkdir -p gpurun_out/fus; export TMPDIR=/tmp; root=$(pwd)
timeout 900 python -m pytest tests/test_gpu_fusions.py tests/test_gpu_fullsize_fusion.py -m gpu -x -q 2>&1 | tail -3
for gsz in 8192; do
rm -rf /tmp/pf; (cd /tmp && THJ_FUSION_GRID=$gsz timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pf -o res -- python $root/bench.py --plain --fusion-search --fusion-frac 0.02 --steps 2 --warmup 1 --no-cpu-baseline --e2e-pairs 0 > /tmp/pf.log 2>&1); echo grid $gsz; python tools/rocpd_dispatches.py $(find /tmp/pf -name "*.db" | head -1) usion\( | tail -4
done
