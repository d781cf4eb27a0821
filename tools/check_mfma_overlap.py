#!/usr/bin/env python3
"""Static check of a `hipcc -S` listing for the MFMA forms the policy kernels rely on.

Background (drloco_amd/csrc/dl_policy.hpp, EXPERIMENTS.md "the 4x4x1 defect, found"): written as builtins, the per-rollout kernel stored a stale row of a
v_mfma_f32_4x4x1 result about once in a thousand rows.  Round 4 blamed the relocated accumulators the listing showed and tied every MFMA as inline asm; round 5
bisected the builtin build by patching its assembly to ONE site -- hipcc's `s_nop 3` between the last MFMA of a chain and the `ds_write2_b32` of its rows -- and
then found the mechanism in a micro test (tools/ubench/snop_wakeup.hip): an s_wakeup executed by another wave of the workgroup ENDS THE s_nop THIS WAVE IS IN
after one wait state.  The split workgroups hand over with s_sleep / s_wakeup, so in the rollout kernels no software-managed hazard may rest on an s_nop.
hipcc pads nothing around inline asm: the tied form keeps its s_nop out, the hand-written waits are v_nop, and this tool proves on the listing of the product
build that they are there -- counting EVERY instruction, an `s_nop N` too, as one wait state.  Measured need of a reader (tools/ubench/mfma_ds_store.hip):
3 (LDS store) / 4 (VALU) states behind a 4x4x1, 9 / 10 behind a 16x16x4.

  R1  vdst != srcC and vdst overlaps srcA or srcB                                   (what round 4 blamed; any MFMA shape)        
  R2  vdst != srcC for a 4x4x1                                                      (no relocation at all in the small shape)                 
  R3  a VALU write of an A / B / C register less than 2 wait states before an MFMA  (hipcc pads nothing in front of inline asm)
  R4  an MFMA's D read or written by anything but an MFMA taking it whole as srcC less than passes + 6 wait states later
                                                                                    (2-pass 4x4x1: 8, 8-pass 16x16x4: 14 = the measured need of a VALU
                                                                                     reader, passes + 2, and 4 of margin; asm MFMAs only)
  R5  dependent 4x4x1 on the same accumulator less than 2 wait states apart         (asm MFMAs only; the larger shapes interlock)

usage: tools/check_mfma_overlap.py <listing.s> [substring of a kernel name]      exit code 1 if a rule is violated.
A wait state = one issued instruction (`s_nop N`: ONE, see above).  R4 / R5 follow every path: the fall-through and the target of each branch inside the
window (the layer loops end with MFMAs right in front of their back edge); R3 looks back inside the basic block only."""
import re
import sys

REG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')
MFMA = re.compile(r'^v_(s?mfma\w*)\s+(.*)$')
SHAPE = re.compile(r'_(\d+)x(\d+)x(\d+)')


def regset(tok):
    """register file + indices named by one operand token ('v[4:7]', 'v100', 'a[0:3]'); empty for literals / SGPRs"""
    tok = tok.strip().lstrip('-').strip('|')
    m = re.fullmatch(r'([va])(\d+)', tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


def all_regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def passes(op):
    m = SHAPE.search(op)
    if not m:
        return 8
    mm = int(m.group(1))
    return {4: 2, 16: 8, 32: 16}.get(mm, 8)


def parse(lines):
    """-> {kernel: [(line number, kind, text, in_asm)]}, kind in 'label' / 'ins'"""
    kernels, cur, in_asm = {}, None, False
    for ln, raw in enumerate(lines, 1):
        if raw.startswith('_Z') and ':' in raw and not raw.startswith('\t'):
            cur = kernels.setdefault(raw.split(':')[0], [])
            in_asm = False
            continue
        if cur is None:
            continue
        if raw.startswith('.Lfunc_end'):
            cur = None
            continue
        if ';;#ASMSTART' in raw:
            in_asm = True
            continue
        if ';;#ASMEND' in raw:
            in_asm = False
            continue
        l = raw.split(';')[0].strip()
        if not l:
            continue
        if l.endswith(':'):
            cur.append((ln, 'label', l[:-1], False))
        elif not l.startswith('.'):
            cur.append((ln, 'ins', l, in_asm))
    return kernels


def states(l):
    """wait states an instruction is worth: ONE, also for `s_nop N` -- an s_wakeup of another wave of the workgroup ends an s_nop after one state
    (tools/ubench/snop_wakeup.hip), and the rollout kernels' split workgroups hand over with s_wakeup"""
    return 1


def valu_write(l):
    """VGPRs written by a VALU (non-MFMA) instruction"""
    op, _, rest = l.partition(' ')
    if not op.startswith('v_') or MFMA.match(l) or op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane', 'v_nop')):
        return set()
    ops = rest.split(',')
    w = regset(ops[0].split(' ')[0]) if ops else set()
    if op.startswith(('v_swap', 'v_permlane')) and len(ops) > 1:          # both operands are written
        w |= regset(ops[1].split(' ')[0])
    return w


def check(path, name, items, report):
    counts = {'R1': 0, 'R2': 0, 'R3': 0, 'R4': 0, 'R5': 0}
    n_mfma = n_asm = 0
    labels = {l: i for i, (_, kind, l, _) in enumerate(items) if kind == 'label'}
    for i, (ln, kind, l, in_asm) in enumerate(items):
        if kind != 'ins':
            continue
        m = MFMA.match(l)
        if not m:
            continue
        op = 'v_' + m.group(1)
        ops = [o.strip() for o in m.group(2).split(',')]
        if len(ops) < 4:
            continue
        n_mfma += 1
        n_asm += in_asm
        d, a, b, c = regset(ops[0]), regset(ops[1]), regset(ops[2]), regset(ops[3].split(' ')[0])
        small = '_4x4x' in op
        tag = f'{path}:{ln}: {name[:56]}: `{l}`'
        if d != c and (d & (a | b)):
            counts['R1'] += 1
            report(f'{tag} R1: destination != source C and the destination overlaps an A / B operand')
        if small and d != c:
            counts['R2'] += 1
            report(f'{tag} R2: 4x4x1 with a relocated accumulator (destination != source C)')
        # R3: look back over < 2 wait states
        need, j = 2, i - 1
        while need > 0 and j >= 0 and items[j][1] == 'ins':
            pl = items[j][2]
            if pl.split(' ')[0].startswith(('s_cbranch', 's_branch', 's_setpc', 's_endpgm')):
                break
            w = valu_write(pl)
            if w & (a | b | c):
                counts['R3'] += 1
                report(f'{tag} R3: operand written by `{pl}` (line {items[j][0]}) only {2 - need} wait state(s) earlier')
                break
            need -= states(pl)
            j -= 1
        if not in_asm:
            continue
        # R4 / R5: look ahead along every path (fall-through and branch targets) until D has settled
        need4, need5 = passes(op) + 6, 2
        seen, work, hit = set(), [(i + 1, 0)], None
        while work and hit is None:
            j, gone = work.pop()
            while gone < need4 and j < len(items):
                if (j, gone) in seen:
                    break
                seen.add((j, gone))
                lnj, kj, lj, _ = items[j]
                if kj == 'label':
                    j += 1
                    continue          # fall-through: the window continues
                opj = lj.split(' ')[0]
                mj = MFMA.match(lj)
                if mj:
                    oj = [o.strip() for o in mj.group(2).split(',')]
                    dj, aj, bj, cj = regset(oj[0]), regset(oj[1]), regset(oj[2]), regset(oj[3].split(' ')[0])
                    if cj == d and dj == d:
                        if small and gone < need5:
                            hit = ('R5', f'`{lj}` (line {lnj}) accumulates into the same registers {gone} wait state(s) later (2 needed)')
                        break             # the chain goes on: this MFMA's own window takes over
                    if d & (aj | bj | cj | dj):
                        hit = ('R4', f'D touched {gone} wait state(s) later by `{lj}` (line {lnj}); {need4} needed')
                        break
                elif opj.startswith(('s_cbranch', 's_branch')):
                    tgt = lj.split(' ')[-1].strip()
                    if tgt in labels:
                        work.append((labels[tgt], gone + 1))
                    else:
                        hit = ('R4', f'a branch to an unknown target (`{lj}`, line {lnj}) {gone} wait state(s) after the MFMA')
                        break
                    if opj == 's_branch':
                        break
                elif opj.startswith('s_setpc'):
                    hit = ('R4', f'an indirect branch (`{lj}`, line {lnj}) {gone} wait state(s) after the MFMA, before D has settled ({need4} needed)')
                    break
                elif opj == 's_endpgm':
                    break
                elif not opj.startswith(('s_nop', 's_waitcnt', 's_sleep', 's_barrier')) and (all_regs(lj) & d):
                    hit = ('R4', f'D touched {gone} wait state(s) later by `{lj}` (line {lnj}); {need4} needed')
                    break
                gone += states(lj)
                j += 1
        if hit:
            counts[hit[0]] += 1
            report(f'{tag} {hit[0]}: {hit[1]}')
    return counts, n_mfma, n_asm


def main():
    path = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else ''
    kernels = parse(open(path).read().split('\n'))
    total = {'R1': 0, 'R2': 0, 'R3': 0, 'R4': 0, 'R5': 0}
    tot_mfma = tot_asm = 0
    for name, items in kernels.items():
        if 'hwprobe' in name:          # dl::hwprobe::k_snop (dl_hwprobe.hpp) reads an MFMA result too early ON PURPOSE: it measures what that does beside an s_wakeup
            continue
        if key in name:
            c, n, na = check(path, name, items, print)
            tot_mfma += n
            tot_asm += na
            for k in total:
                total[k] += c[k]
            if n:
                print(f'    {name[:90]}: {n} MFMA ({na} inline asm)' + ''.join(f' {k}={v}' for k, v in c.items() if v))
    bad = sum(total.values())
    print(f'{bad} MFMA operand-overlap / wait-state violation(s) in {tot_mfma} MFMA instructions ({tot_asm} inline asm): ' + ' '.join(f'{k}={v}' for k, v in total.items()))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
