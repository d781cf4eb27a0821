#!/usr/bin/env python3
"""Instruction mix of ONE kernel of a `hipcc -S` listing: tools/asm_kernel.py build_dbg/dl_kernels.s <substring of the kernel's mangled name>
(build the listing with the product flags + --cuda-device-only -S)."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].startswith('_Z') and ': ' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('\t.amdhsa_kernel') or lines[i].startswith('.Lfunc_end'))
body = lines[start + 1:end]
c = collections.Counter()
for l in body:
    t = l.strip().split(' ')[0].split('\t')[0]
    if t and not t.startswith(('.', ';', 'BB', '_Z')) and not t.endswith(':'):
        c[t] += 1
groups = collections.OrderedDict([('scratch', r'^scratch_'), ('flat', r'^flat_'), ('global', r'^global_'), ('buffer', r'^buffer_'), ('ds', r'^ds_'), ('mfma', r'^v_mfma'), ('readlane/writelane', r'^v_(read|write)lane'),
                                  ('dpp', r'_dpp$'), ('s_nop', r'^s_nop'), ('salu', r'^s_'), ('valu', r'^v_')])
print('instructions', sum(c.values()))
for name, pat in groups.items():
    print(f'{name:20s}', sum(v for k, v in c.items() if re.search(pat, k)))
if len(sys.argv) > 3:
    for k, v in c.most_common(40):
        print(f'  {k:40s} {v}')
