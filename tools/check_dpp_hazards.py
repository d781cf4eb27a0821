#!/usr/bin/env python3
"""Static check of a `hipcc -S` listing for the gfx9 hazard the hand-written DPP statements have to respect themselves: a VGPR written by a VALU
instruction may be read through DPP (src0 of a *_dpp instruction) only two wait states later.  Inline asm is opaque to the compiler's hazard
recogniser, and the register allocator may put a copy right in front of an asm statement -- behind the s_nop that was meant to cover it.
usage: tools/check_dpp_hazards.py build_dbg/dl_kernels.s [substring of a kernel name]      (exit code 1 if a violation is found)
Straight-line check per basic block (a label resets the window: a branch target's predecessors are not followed, so this can miss a
hazard across a branch but reports no false ones)."""
import re
import sys

path = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else ''
REG = re.compile(r'^v(\d+)$|^v\[(\d+):(\d+)\]$')


def regs(tok):
    m = REG.match(tok.strip().lstrip('-').strip('|'))
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


bad = 0
kernel = None
window = []          # [(wait states this instruction occupies, set of VGPRs it writes, text)]
for ln, raw in enumerate(open(path), 1):
    l = raw.split(';')[0].strip()
    if raw.startswith('_Z') and raw.rstrip().endswith(':') or (raw.startswith('_Z') and ': ' in raw):
        kernel = raw.split(':')[0]
        window = []
        continue
    if not l or l.startswith('.') or l.startswith(';'):
        if l.endswith(':') or l.startswith('.LBB'):
            window = []
        continue
    if l.endswith(':'):
        window = []
        continue
    if kernel is None or key not in kernel:
        continue
    op, _, rest = l.partition(' ')
    ops = [o.strip() for o in rest.split(',')] if rest else []
    if '_dpp' in op or re.search(r'\b(row_|quad_perm|wave_)', l):
        # src0 is operand 1 (dst, src0[, src1]); modifiers trail the last operand separated by spaces
        if len(ops) >= 2:
            src = regs(ops[1].split(' ')[0])
            need = 2
            for ws, w, text in reversed(window):
                if need <= 0:
                    break
                if src & w:
                    print(f'{path}:{ln}: {kernel[:60]}: `{l}` reads v{sorted(src & w)} through DPP {2 - need} wait state(s) after `{text}`')
                    bad += 1
                    break
                need -= ws
    if op == 's_nop':
        window.append((int(ops[0], 0) + 1, set(), l))
    elif op.startswith('v_') and ops and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
        window.append((1, regs(ops[0].split(' ')[0]), l))
    elif op.startswith(('s_cbranch', 's_branch', 's_setpc', 's_endpgm')):
        window = []
    else:
        window.append((1, set(), l))
    window = window[-4:]
print(f'{bad} DPP read-after-write hazard(s)')
sys.exit(1 if bad else 0)
