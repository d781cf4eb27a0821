#!/usr/bin/env python3
"""asm_patch_mfma.py <mode> <in.s> <out.s> [kernel-name substring]
Insert (or remove) wait states behind the v_mfma_f32_4x4x1 instructions of one kernel of a device listing -- the bisection that found the round-4
stale-row defect (EXPERIMENTS.md, "the 4x4x1 defect, found"; driver: tools/asm_bisect.sh).  Default kernel: k_rollout_pairs<TopoStraight>.

modes (what is padded with 12 wait states unless said otherwise):
  none                 the listing unchanged (the control: rebuilds the failing library from its own assembly)
  all3                 `s_nop 3` behind every 4x4x1
  reloc                `s_nop 1` behind every 4x4x1 whose destination is not its source C (round 4's suspect)
  tail                 every 4x4x1 whose next instruction is not a 4x4x1
  tail_valu / tail_ds / tail_vmem / tail_other   ... and is a VALU / LDS / memory / other instruction
  war_c / war_ab       a 4x4x1 followed by an LDS load that overwrites its source C / its A or B register
  ds_not_site          every LDS successor except the heads' store
  site<N>              N extra wait states in front of the heads' store only (`ds_write2_b32 v0, v8, v9` behind the single-accumulator chain)
  site-1               one wait state FEWER there (s_nop 3 -> s_nop 2)
  site_nop7            hipcc's `s_nop 3` there made ONE `s_nop 7` (8 states in one instruction: what an s_wakeup can end)
  site_vnop4           ... made four `v_nop` (the same 4 states, which nothing can end)
  list                 print every non-MFMA successor of a 4x4x1 with its class, change nothing
Prints the number of sites touched."""
import re
import sys

mode, src, dst = sys.argv[1:4]
kernel = sys.argv[4] if len(sys.argv) > 4 else '_Z15k_rollout_pairsIN2dl12TopoStraight'
lines = open(src).read().split('\n')
MF = re.compile(r'^\tv_mfma_f32_4x4x1_16b_f32 v\[(\d+):(\d+)\], v(\d+), v(\d+), (v\[(\d+):(\d+)\]|0)')
PAD12 = ['\ts_nop 7', '\ts_nop 3']


def successor(i):
    """index and text of the next real instruction behind line i (skipping s_nop / s_waitcnt / comments)"""
    k = i + 1
    while k < len(lines) and (lines[k].strip().startswith(('s_nop', ';', 's_waitcnt')) or not lines[k].strip()):
        k += 1
    return k, (lines[k].strip() if k < len(lines) else '')


def kind(t):
    if 'v_mfma_f32_4x4x1' in t:
        return 'mfma'
    if t.startswith('ds_'):
        return 'ds'
    if t.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    return 'valu' if t.startswith('v_') else 'other'


def is_site(t):
    return bool(re.match(r'ds_write2_b32 v\d+, v(\d+), v(\d+) ', t))


out, n, inside = [], 0, False
for i, l in enumerate(lines):
    if l.startswith(kernel):
        inside = True
    if inside and l.startswith('.Lfunc_end'):
        inside = False
    m = MF.match(l) if inside else None
    if not m:
        out.append(l)
        continue
    d0, a, b = int(m.group(1)), int(m.group(3)), int(m.group(4))
    c0 = int(m.group(6)) if m.group(6) else None
    k, nx = successor(i)
    kd = kind(nx)
    out.append(l)
    if mode == 'all3':
        out.append('\ts_nop 3'); n += 1
    elif mode == 'reloc' and (c0 is None or c0 != d0):
        out.append('\ts_nop 1'); n += 1
    elif mode == 'tail' and kd != 'mfma':
        out += PAD12; n += 1
    elif mode in ('tail_valu', 'tail_ds', 'tail_vmem', 'tail_other') and kd == mode[5:]:
        out += PAD12; n += 1
    elif mode in ('war_c', 'war_ab'):
        mm = re.match(r'ds_read_b\d+ v\[(\d+):(\d+)\]', nx)
        if mm:
            lo, hi = int(mm.group(1)), int(mm.group(2))
            hits_c = c0 is not None and c0 != d0 and not (hi < c0 or lo > c0 + 3)
            if (mode == 'war_c' and hits_c) or (mode == 'war_ab' and not hits_c and (lo <= a <= hi or lo <= b <= hi)):
                out += PAD12; n += 1
    elif mode == 'ds_not_site' and kd == 'ds' and not is_site(nx):
        out += PAD12; n += 1
    elif mode in ('site_nop7', 'site_vnop4') and is_site(nx):
        assert lines[i + 1].strip() == 's_nop 3', lines[i + 1]
        lines[i + 1] = '\ts_nop 7' if mode == 'site_nop7' else '\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop'
        n += 1
    elif mode.startswith('site') and is_site(nx):
        extra = int(mode[4:])
        if extra < 0:
            assert lines[i + 1].strip() == 's_nop 3', lines[i + 1]
            lines[i + 1] = '\ts_nop %d' % (3 + extra)
        while extra > 0:
            out.append('\ts_nop %d' % (min(extra, 8) - 1)); extra -= min(extra, 8)
        n += 1
        print('   line', i + 1, '|', l.strip(), '|', lines[i + 1].strip(), '|', nx, file=sys.stderr)
    elif mode == 'list' and kd != 'mfma':
        print(i + 1, kd, '|', l.strip(), '|', nx)
        n += 1
open(dst, 'w').write('\n'.join(out))
print(mode, 'sites:', n)
