#!/usr/bin/env python3
"""survey_snop.py [listing.s] [--check]: the COMPILER's s_nop padding (outside inline asm) per kernel of the device listing, by count, and what stands around every
`s_nop N` with N >= 2.  Why: an s_wakeup of another wave of the workgroup ends an s_nop after one wait state (tools/ubench/snop_wakeup.hip), so in kernels
whose workgroups hand over with s_wakeup a software-managed hazard that needs more than ONE state must not rest on a single s_nop.  Product build (round 5):
the split / rollout kernels carry `s_nop 0` / `s_nop 1` only (the VALU -> DPP hazard: one state is what gfx950 needs).  Round 5 exempted hipcc's `s_nop 3` in front of the
fault word's `global_atomic_or` on the time-out paths (VALU writes the SGPR base -> VMEM reads it: 5 states); round 6 found that exemption to be a crash -- cut short by a
partner's s_wakeup the atomic goes out with a stale SGPR pair (a memory access fault in one of five runs of the forced-time-out test) -- and made the statement hand-written
with v_nop (dl_group.hpp, dl_fault_or): NO exemption is left.  --check: exit code 1 if a kernel whose workgroups use s_wakeup (name contains `split` or `k_rollout_`) has an
`s_nop N`, N >= 2, anywhere outside inline asm (tests/test_dpp_hazards.py)."""
import collections
import os
import re
import sys

check = '--check' in sys.argv
args = [a for a in sys.argv[1:] if a != '--check']
path = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build_dbg', 'listing', 'dl_kernels-hip-amdgcn-amd-amdhsa-gfx950.s')
L = open(path).read().split('\n')
kern, inasm = None, False
stats, ctx = collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)


def real(i, step):
    while 0 < i < len(L) and (L[i].strip().startswith((';', 's_nop', 's_waitcnt')) or not L[i].strip()):
        i += step
    return L[i].strip().split(' ')[0] if 0 <= i < len(L) else ''


for i, l in enumerate(L):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        kern = m.group(1)
    if l.startswith('.Lfunc_end'):
        kern = None
    t = l.strip()
    inasm = True if t.startswith(';;#ASMSTART') else (False if t.startswith(';;#ASMEND') else inasm)
    if kern and not inasm and t.startswith('s_nop'):
        n = int(t.split()[1])
        stats[kern][n] += 1
        if n >= 2:
            ctx[kern][(n, real(i - 1, -1), real(i + 1, 1))] += 1
bad = 0
for k in sorted(stats):
    print(f'{k[:90]:90s}', ' '.join(f's_nop {n}: {c}' for n, c in sorted(stats[k].items())))
    for (n, a, b), c in ctx[k].most_common():
        wake = 'split' in k or 'k_rollout_' in k
        flag = wake
        bad += c if flag else 0
        print(f'        {c:4d} x  {a}  |  s_nop {n}  |  {b}' + ('     <-- a multi-state wait in a kernel with s_wakeup' if flag else ''))
print(f'{bad} multi-state s_nop in the kernels that use s_wakeup')
if check and bad:
    sys.exit(1)
