#!/usr/bin/env python3
"""Static instruction mix of one kernel instance in a hipcc -S dump.
usage: tools/isa_mix.py file.s <mangled-name-regex> [--dump out.s]"""
import re, sys, collections
L = open(sys.argv[1]).read().split('\n')
pat = re.compile(sys.argv[2])
start = next(i for i, l in enumerate(L) if pat.search(l) and l.rstrip().split(';')[0].rstrip().endswith(':'))
end = next(i for i in range(start, len(L)) if 's_endpgm' in L[i])
body = L[start:end + 1]
cnt, ops = collections.Counter(), collections.Counter()
for l in body:
    t = l.strip()
    if not t or t.startswith((';', '.')) or t.split()[0].endswith(':'):
        continue
    op = t.split()[0]
    cls = ('salu' if op.startswith('s_') else 'mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_')
           else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')) else 'other')
    cnt[cls] += 1
    ops[op] += 1
print(dict(cnt))
print(ops.most_common(40))
if '--dump' in sys.argv:
    open(sys.argv[sys.argv.index('--dump') + 1], 'w').write('\n'.join(body))
