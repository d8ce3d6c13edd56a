#!/bin/bash
# Registers, spills, LDS and occupancy of the thj_k_* kernels of one .hip file (compile only; no GPU needed).
#   tools/resource_usage.sh tophat_amd/csrc/thj_span.hip [name filter]
f=$1; pat=${2:-thj_k_}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 | python3 -c "
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r'remark: (.*)', line)
    if not m: continue
    t = m.group(1).replace('[-Rpass-analysis=kernel-resource-usage]', '').strip()
    if t.startswith('Function Name:'):
        if cur: print('\t'.join('%s: %s' % kv if kv[0] != 'name' else kv[1] for kv in cur.items()))
        cur = {'name': t.split(':', 1)[1].strip()}
    else:
        for key in ('VGPRs', 'AGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'SGPRs Spill', 'VGPRs Spill', 'LDS Size [bytes/block]'):
            if t.startswith(key + ':'): cur[key] = t.split(':', 1)[1].strip()
if cur: print('\t'.join('%s: %s' % kv if kv[0] != 'name' else kv[1] for kv in cur.items()))
" | grep -- "$pat"
