#!/usr/bin/env python3
"""Static check of a `hipcc -S` listing for the gfx9 hazard the hand-written DPP statements have to respect themselves: a VGPR written by a VALU
instruction may be read through DPP (src0 of a *_dpp instruction) only NEED wait states later.  Inline asm is opaque to the compiler's hazard
recogniser, and the register allocator may put a copy right in front of an asm statement -- behind the s_nop that was meant to cover it.
NEED = 2 (default): the ISA manual's figure and what the default library pads with (dl_group.hpp, DL_DPP_WAIT = 2: `s_nop 0` twice); `--need 1` for the
-DDL_DPP_WAIT=1 code object (libdrloco_hip_dpp1.so: what gfx950 was measured to need -- stale with no wait, never with one state, 11 G lane-reads,
tools/ubench/dpp_wait.hip -- selected at run time only on a device that passes dl_hw_probe).  Every instruction is one wait state; an `s_nop N` counts
  --snop spec   N + 1 states, the manual's definition (default with --need 2: hipcc's own `s_nop 1` in front of its DPP instructions is conformant), or
  --snop one    ONE state whatever N: an s_wakeup of another wave of the workgroup ends an s_nop after one state (tools/ubench/snop_wakeup.hip) -- the default with
                --need 1, where it is what makes the check meaningful: one state is then what every path really has.
usage: tools/check_dpp_hazards.py <listing.s> [substring of a kernel name] [--need N] [--snop spec|one]      (exit code 1 if a violation is found)
Paths: the straight-line window inside a basic block, carried over a label on the fall-through path and -- two passes -- from the tail of every
block that branches to the label (s_branch / s_cbranch_* with a label operand).  Only the last instructions of a predecessor block are looked at
(a predecessor shorter than the hazard window does not inherit from its own predecessors)."""
import re
import sys

argv = list(sys.argv[1:])
NEED = 2            # wait states between the VALU write and the DPP read
if '--need' in argv:
    i = argv.index('--need')
    NEED = int(argv[i + 1])
    del argv[i:i + 2]
SNOP_SPEC = NEED >= 2
if '--snop' in argv:
    i = argv.index('--snop')
    SNOP_SPEC = argv[i + 1] == 'spec'
    del argv[i:i + 2]
path = argv[0]
key = argv[1] if len(argv) > 1 else ''
REG = re.compile(r'^v(\d+)$|^v\[(\d+):(\d+)\]$')


def regs(tok):
    m = REG.match(tok.strip().lstrip('-').strip('|'))
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def parse(lines):
    """-> {kernel: [(line number, kind, payload)]}, kind in 'label' / 'ins'"""
    kernels, cur, name = {}, None, None
    for ln, raw in enumerate(lines, 1):
        if raw.startswith('_Z') and ':' in raw and not raw.startswith('\t'):
            name = raw.split(':')[0]
            cur = kernels.setdefault(name, [])
            continue
        if cur is None:
            continue
        if raw.startswith('.Lfunc_end'):
            cur = None
            continue
        l = raw.split(';')[0].strip()
        if not l:
            continue
        if l.endswith(':'):
            cur.append((ln, 'label', l[:-1]))
            continue
        if l.startswith('.'):
            continue
        cur.append((ln, 'ins', l))
    return kernels


def effect(l):
    """(wait states, VGPRs written by a VALU instruction) of one instruction"""
    op, _, rest = l.partition(' ')
    ops = [o.strip() for o in rest.split(',')] if rest else []
    if op == 's_nop':
        return (int(ops[0], 0) + 1 if SNOP_SPEC else 1), set()          # (see the module docstring)
    if op.startswith('v_') and ops and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
        return 1, regs(ops[0].split(' ')[0])
    return 1, set()


def dpp_src(l):
    op, _, rest = l.partition(' ')
    if '_dpp' not in op and not re.search(r'\b(row_|quad_perm|wave_)', l):
        return None
    ops = [o.strip() for o in rest.split(',')]
    return regs(ops[1].split(' ')[0]) if len(ops) >= 2 else None


def check(name, items, report):
    bad = 0
    tails = {}                      # label -> [window at a branch to it]
    for final in (False, True):
        window, fall = [], True     # window: [(wait states, written VGPRs, text)] most recent last
        entry = [[]]                # the windows a block may start with
        for ln, kind, l in items:
            if kind == 'label':
                entry = ([list(window)] if fall else []) + [list(w) for w in tails.get(l, [])]
                if not entry:
                    entry = [[]]
                window, fall = [], True
                continue
            src = dpp_src(l)
            if final and src:
                for pre in entry:
                    need = NEED
                    for ws, w, text in reversed(pre + window):
                        if need <= 0:
                            break
                        if src & w:
                            report(f'{path}:{ln}: {name[:60]}: `{l}` reads v{sorted(src & w)} through DPP {NEED - need} wait state(s) after `{text}`')
                            bad += 1
                            need = -1
                            break
                        need -= ws
                    if need == -1:
                        break
            ws, w = effect(l)
            window = (window + [(ws, w, l)])[-(NEED + 1):]
            op = l.split(' ')[0]
            if op.startswith(('s_cbranch', 's_branch')):
                tgt = l.split(' ')[-1].strip()
                if not final:
                    tails.setdefault(tgt, []).append(list(window))
                if op == 's_branch':
                    fall = False
            elif op.startswith(('s_setpc', 's_endpgm')):
                fall = False
            if sum(x[0] for x in window) >= NEED and len(window) >= NEED:
                entry = [[]]        # the block is long enough: its predecessors no longer matter
    return bad


kernels = parse(open(path).read().split('\n'))
total = 0
for name, items in kernels.items():
    if 'hwprobe' in name:          # dl::hwprobe::k_dpp / k_snop (dl_hwprobe.hpp) violate the rule ON PURPOSE: they are the kernels that measure what a violation does
        continue
    if key in name:
        total += check(name, items, print)
print(f'{total} DPP read-after-write hazard(s)')
sys.exit(1 if total else 0)
