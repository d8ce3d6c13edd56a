#!/bin/bash
# round 5: the kernels' own durations (one stream), then the product's overlapped line
THJ_SPAN_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null > /tmp/x.json; python tools/show_bench.py /tmp/x.json | grep -E "ms/step|stitch|join|finish|chains"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null > /tmp/y.json; python tools/show_bench.py /tmp/y.json | grep -E "ms/step"
