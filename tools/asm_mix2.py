#!/usr/bin/env python3
"""Static instruction classes of ONE kernel of a device listing BY LOOP NEST: LLVM annotates every basic block with `in Loop: Header=BBx_y Depth=d`, so the straight-line cost of
the body of each loop (excluding inner loops) can be told apart -- the control-step loop, the mj_step loop, the RK4 stage loop (= one forward evaluation per trip), the Newton
loop, the line search, the contact loops.  With the trip counts the product's diagnostics report (iterations / rows per evaluation: bench.py --solver-stats) this is the
per-section weighting of the dynamic mix that the PMC pass gives in total (profiles/r06_summary.txt).
usage: tools/asm_mix2.py <listing.s> <substring of the kernel's mangled name> [--blocks]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
show_blocks = '--blocks' in sys.argv
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(tuple('0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ_')) or (l.startswith('_Z') and key in l and ':' in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))

CLASSES = [
    ('lane spill (v_readlane / v_writelane)', r'^v_(read|write)lane'),
    ('readfirstlane', r'^v_readfirstlane'),
    ('fp32 arithmetic, DPP operand', r'^v_(add|sub|mul|fma|fmac|mac|max|min)\w*_f32_dpp'),
    ('move through DPP (v_mov_b32_dpp)', r'^v_mov_b32_dpp'),
    ('fp32 fma / fmac', r'^v_(fma|fmac|mac|mad)\w*_f32'),
    ('fp32 mul / add / sub', r'^v_(mul|add|sub|subrev)\w*_f32'),
    ('fp32 transcendental / division helpers', r'^v_(rcp|rsq|sqrt|exp|log|sin|cos|div_|frexp|ldexp|trig|rndne|fract|floor|ceil)'),
    ('fp32 min / max / med / clamp', r'^v_(max|min|med3)\w*_f32'),
    ('compare', r'^v_cmp'),
    ('select (v_cndmask)', r'^v_cndmask'),
    ('move (v_mov / v_accvgpr)', r'^v_(mov|accvgpr|swap)'),
    ('convert', r'^v_cvt'),
    ('integer / address VALU', r'^v_'),
    ('s_nop', r'^s_nop'),
    ('s_waitcnt', r'^s_waitcnt'),
    ('branch', r'^s_(c?branch|setpc|endpgm|call)'),
    ('scalar memory', r'^s_(load|buffer_load|store)'),
    ('SALU other', r'^s_'),
    ('LDS', r'^ds_'),
    ('global / flat memory', r'^(global|flat|buffer|scratch)_'),
]


def classify(op):
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return 'other'


nest = {}            # header -> (depth, parent chain as found)
cur = ('(no loop)', 0)
per = collections.defaultdict(collections.Counter)
order = []
label = None
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', l)
    mh = re.search(r'=>This (?:Inner )?Loop Header: Depth=(\d+)', l)
    if re.match(r'^\.?L?BB\d+_\d+:', t) or t.startswith('; %bb.'):
        cur = None
    if mh and re.match(r'^\.LBB\d+_\d+:', t):
        cur = (t.split(':')[0].lstrip('.L'), int(mh.group(1)))
    elif m:
        cur = (m.group(1), int(m.group(2)))
    if cur is None:
        cur = ('(no loop)', 0)
    op = t.split(' ')[0].split('\t')[0]
    if not op or op.startswith(('.', ';', 'BB', '_Z')) or op.endswith(':'):
        continue
    if cur not in per:
        order.append(cur)
    per[cur][classify(op)] += 1

names = [c[0] for c in CLASSES]
print(f'{"loop header":14s} {"depth":>5s} {"instr":>6s} | ' + ' '.join(f'{n[:11]:>11s}' for n in names[:12]) + ' | ' + ' '.join(f'{n[:9]:>9s}' for n in names[12:]))
tot = collections.Counter()
for k in order:
    c = per[k]
    tot.update(c)
    print(f'{k[0]:14s} {k[1]:5d} {sum(c.values()):6d} | ' + ' '.join(f'{c[n]:11d}' for n in names[:12]) + ' | ' + ' '.join(f'{c[n]:9d}' for n in names[12:]))
print(f'{"total":14s} {"":5s} {sum(tot.values()):6d} | ' + ' '.join(f'{tot[n]:11d}' for n in names[:12]) + ' | ' + ' '.join(f'{tot[n]:9d}' for n in names[12:]))
print()
for i, n in enumerate(names):
    print(f'  column {i + 1:2d}: {n}')
