python -m pytest tests/test_gpu_binaries.py tests/test_gpu_ingest.py -x -q 2>&1 | tail -15
D=/dev/shm/e2e4m
tools/bin/thj_gen --out $D --pairs 4000000 > /dev/null
python tools/e2e_bench.py --pairs 4000000 --keep $D --env THJ_BGZF_LEVEL=1 2>&1 | grep -E "_s\"|pairs_per_s|worker-seconds|timing|shards"
