#!/usr/bin/env python3
"""Static instruction mix of the 16-lane step kernels from the compiler's assembly (hipcc -S): a quick look at what the
instruction stream is made of (VALU / packed VALU / AGPR copies / LDS / waits).  usage: tools/asm_mix.py [k.s]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else '/tmp/dl_kernels.s'
if not os.path.exists(path):
    sys.path.insert(0, ROOT)
    from drloco_amd import lib            # the product build's flags
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'drloco_amd', 'csrc')] + lib.EXTRA_FLAGS +
                          ['--cuda-device-only', '-S', os.path.join(ROOT, 'drloco_amd', 'csrc', 'dl_kernels.hip'), '-o', path])
lines = open(path).read().split('\n')
cur, bodies = None, collections.defaultdict(list)
for ln in lines:
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        cur = m.group(1)
        continue
    if ln.startswith('.Lfunc_end'):
        cur = None
    if cur:
        bodies[cur].append(ln)
for name, body in bodies.items():
    if 'k_env_step_g16If' not in name or 'ELb1' in name:
        continue
    ops = collections.Counter()
    for ln in body:
        ln = ln.strip()
        if not ln or ln[0] in '.;/' or ln.endswith(':'):
            continue
        ops[ln.split()[0]] += 1
    grp = collections.Counter()
    for op, c in ops.items():
        key = ('v_pk' if op.startswith('v_pk_') else 'accvgpr' if op.startswith('v_accvgpr') else 'v_mov' if op.startswith('v_mov') else 'dpp' if op.endswith('_dpp') else
               'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 's_waitcnt' if op.startswith('s_waitcnt') else 's_nop' if op.startswith('s_nop') else
               'salu' if op.startswith('s_') else 'vmem' if op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')) else 'other')
        grp[key] += c
    print(('TopoWalker165' if 'Walker165' in name else 'TopoStraight'), 'static instructions:', sum(ops.values()))
    print('  ', dict(grp))
    print('  ', ops.most_common(18))
