#!/bin/bash
# GPU box: files in -> files out on the shapes of configs[3] and configs[4] (one GPU), next to configs[1]'s line in bench.py
out=gpurun_out/$1_e2e_shapes.txt
{
echo "# python tools/e2e_bench.py --pairs 10000000 --read-len 50 --coverage-search --plain   (configs[4]'s shape: 2x50 bp, two segments, coverage search on as tophat.py runs short reads)"
python tools/e2e_bench.py --pairs 10000000 --read-len 50 --coverage-search --plain | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps({k:d[k] for k in d if not k.endswith('_log_tail') and not k.endswith('_log_all')}))"
echo "# python tools/e2e_bench.py --pairs 10000000 --read-len 50 --plain   (the same without the coverage search)"
python tools/e2e_bench.py --pairs 10000000 --read-len 50 --plain | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps({k:d[k] for k in d if not k.endswith('_log_tail') and not k.endswith('_log_all')}))"
echo "# python tools/e2e_bench.py --pairs 5000000 --read-len 150 --fusion-search --plain   (configs[3]'s shape: 2x150 bp, six segments, --fusion-search in both executables; no planted fusions)"
python tools/e2e_bench.py --pairs 5000000 --read-len 150 --fusion-search --plain | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps({k:d[k] for k in d if not k.endswith('_log_tail') and not k.endswith('_log_all')}))"
echo "# python tools/e2e_bench.py --pairs 5000000 --read-len 150 --plain"
python tools/e2e_bench.py --pairs 5000000 --read-len 150 --plain | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps({k:d[k] for k in d if not k.endswith('_log_tail') and not k.endswith('_log_all')}))"
} > $out 2>&1
cut -c1-400 $out
