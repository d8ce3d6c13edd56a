#!/bin/bash
for dr in 32 16 8 4 2; do echo "== THJ_PACK_DRAW=$dr"; THJ_PACK_DRAW=$dr THJ_PACK_TIMING=1 THJ_SPAN_SERIAL=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/tmp/e.txt >/tmp/x.json; grep "packed tier" /tmp/e.txt | tail -1 | cut -c1-110; python tools/show_bench.py /tmp/x.json | grep -E "stitch_pack"; done
