#!/usr/bin/env python3
"""Developer tool: static instruction mix per kernel of a `hipcc -S --cuda-device-only` listing.  Usage: isa_mix.py file.s pattern..."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pats and not any(p in name for p in pats):
        continue
    c = collections.Counter()
    n = 0
    for l in body.split('\n'):
        l = l.strip()
        if not l or l[0] in '.;' or l.endswith(':'):
            continue
        op = l.split()[0]
        n += 1
        if op.startswith('v_'):
            c['valu'] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif op.startswith('s_cbranch') or op.startswith('s_branch'):
            c['branch'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_')):
            c['vmem'] += 1
        elif op.startswith('scratch_'):
            c['scratch'] += 1
        else:
            c['other'] += 1
    print("%-70s %6d  %s" % (name[:70], n, dict(c)))
