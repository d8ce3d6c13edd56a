#!/bin/bash
# GPU box: which leg of the default bench.py run makes its timed steps slower than a bare run's (6.5 against 5.6 ms)?
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
ms() { python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f ms/step; contig in the timed region %.3f ms' % (d['ms_per_step'], [k for k in d['kernels'] if k['kernel']=='thj_k_stitch_contig'][0]['avg_kernel_ms']))"; }
{ echo -n "bare (--no-cpu-baseline --e2e-pairs 0 --no-pmc): "; python bench.py --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null | ms
  echo -n "+ cpu baseline: "; python bench.py --e2e-pairs 0 --no-pmc 2>/dev/null | ms
  echo -n "+ pmc passes: "; python bench.py --no-cpu-baseline --e2e-pairs 0 2>/dev/null | ms
  echo -n "+ e2e 1 M pairs only: "; python bench.py --no-cpu-baseline --no-pmc --e2e-pairs 1000000 --e2e-pairs-large 0 --e2e-grch38-pairs 0 2>/dev/null | ms
  echo -n "bare again: "; python bench.py --no-cpu-baseline --e2e-pairs 0 --no-pmc 2>/dev/null | ms
} 2>&1 | tee gpurun_out/r05_default_slow.txt
