#!/usr/bin/env python3
"""Rebuild the library from a PATCHED device listing -- how the round-4 stale-row defect was bisected to one instruction (EXPERIMENTS.md, "the 4x4x1 defect, found").

  tools/asm_bisect.py prepare                 build_variants/asm_bisect/: the product sources with -DDL_EXP_POLP_BUILTIN (the round-4 builtin form of the 4x4x1 chains),
                                              `hipcc -save-temps` (orig.s = its device listing) and the `hipcc -###` command list of the same build
  tools/asm_bisect.py variant <mode> [...]    tools/asm_patch_mfma.py <mode> on orig.s, then hipcc's own pipeline replayed from the assembler on:
                                              build_variants/libpolp_<mode>.so
  on the GPU box:  DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/libpolp_<mode>.so python3 tools/diag_polp_rows.py      counts the wrong policy rows of the per-rollout kernel

CPU only (hipcc cross-compiles); nothing here is part of the product or the tests."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W = os.path.join(ROOT, 'build_variants', 'asm_bisect')
DEV_S = 'dl_kernels-hip-amdgcn-amd-amdhsa-gfx950.s'


def hipcc_cmd():
    from drloco_amd import lib
    return [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-I' + lib.INCLUDE, '-I' + lib.CSRC] + lib.EXTRA_FLAGS + \
           ['-DDL_EXP_POLP_BUILTIN', '-save-temps', os.path.join(lib.CSRC, 'dl_kernels.hip'), '-o', os.path.join(W, 'out.so')]


def prepare():
    os.makedirs(W, exist_ok=True)
    cmd = hipcc_cmd()
    subprocess.run(cmd, cwd=W, check=True)
    os.replace(os.path.join(W, DEV_S), os.path.join(W, 'orig.s'))
    r = subprocess.run(cmd + ['-###'], cwd=W, capture_output=True, text=True)
    cmds = [l for l in r.stderr.split('\n') if l.startswith(' "')]
    open(os.path.join(W, 'cmds.txt'), 'w').write('\n'.join(cmds) + '\n')
    print('prepared', W, ':', len(cmds), 'pipeline commands; orig.s', os.path.getsize(os.path.join(W, 'orig.s')) >> 20, 'MiB')


def variant(mode):
    cmds = open(os.path.join(W, 'cmds.txt')).read().strip().split('\n')
    first = next(i for i, c in enumerate(cmds) if '"-S"' in c and '"amdgcn-amd-amdhsa"' in c) + 1        # everything behind the step that wrote the device listing ...
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'asm_patch_mfma.py'), mode, os.path.join(W, 'orig.s'), os.path.join(W, DEV_S)], check=True)
    out = os.path.join(ROOT, 'build_variants', f'libpolp_{mode}.so')
    for i in range(first, len(cmds)):
        if '"-E"' in cmds[i]:
            continue                 # ... except the host side's preprocessing (the fat binary is embedded by the host compile step behind it, which runs)
        c = cmds[i].replace(f'"-o" "{os.path.join(W, "out.so")}"', f'"-o" "{out}"')
        subprocess.run(c, shell=True, cwd=W, check=True)
    print(out)


if __name__ == '__main__':
    if sys.argv[1] == 'prepare':
        prepare()
    else:
        for m in sys.argv[2:]:
            variant(m)
