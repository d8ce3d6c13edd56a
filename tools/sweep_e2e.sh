D=/dev/shm/e2e8m
timeout 200 tools/bin/thj_gen --out $D --pairs 8000000 > /dev/null
timeout 300 python tools/e2e_bench.py --pairs 8000000 --keep $D --env THJ_BGZF_LEVEL=1 2>&1 | tail -32
