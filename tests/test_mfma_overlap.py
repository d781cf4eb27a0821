"""The policy kernels write every MFMA as inline asm with the accumulator tied to the destination (drloco_amd/csrc/dl_policy.hpp, dl_policy_pair.hpp):
as builtins, accumulators of a dense v_mfma_f32_4x4x1 chain were relocated by the register allocator onto dying A / B operands and about one row in
a thousand came out wrong inside the per-rollout kernel (mechanism unknown; the isolated instruction is exact).  Inline asm is opaque to hipcc's
hazard recogniser, so the wait states are hand-written.  tools/check_mfma_overlap.py checks a device listing for both; here:
  * the checker itself on synthetic listings, one per rule (R1 .. R5), each with its corrected twin;
  * the listing of the product build (the -save-temps output of the very compilation that makes the library): zero findings, all MFMAs inline asm."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, 'tools', 'check_mfma_overlap.py')

HEAD = "_Z3kerPf: ; @_Z3kerPf\n"
TAIL = "\ts_endpgm\n"


def _run(body, tmp_path, name):
    f = tmp_path / name
    f.write_text(HEAD + body + TAIL)
    p = subprocess.run([sys.executable, TOOL, str(f)], capture_output=True, text=True)
    return p.returncode, p.stdout


def _asm(ins):
    return f"\t;;#ASMSTART\n\t{ins}\n\t;;#ASMEND\n"


VN16 = '\tv_nop\n' * 16

CASES = {
    # rule: (violating body, corrected body)
    'R1': ("\tv_mfma_f32_16x16x4_f32 v[70:73], v73, v41, v[78:81]\n",                      # the builtin form the product used to contain
           "\tv_mfma_f32_16x16x4_f32 v[70:73], v74, v41, v[78:81]\n"),                      # relocated but not over A / B: allowed for 16x16x4 ...
    'R2': ("\tv_mfma_f32_4x4x1_16b_f32 v[4:7], v100, v32, v[8:11]\n",                      # ... never for 4x4x1
           "\tv_mfma_f32_4x4x1_16b_f32 v[4:7], v100, v32, v[4:7]\n"),
    'R3': ("\tv_mov_b32_e32 v22, 0\n" + _asm("v_mfma_f32_16x16x4_f32 v[22:25], v68, v10, v[22:25]") + VN16,
           "\tv_mov_b32_e32 v22, 0\n" + _asm("v_nop\n\tv_nop\n\tv_mfma_f32_16x16x4_f32 v[22:25], v68, v10, v[22:25]") + VN16),
    'R4': (_asm("v_mfma_f32_16x16x4_f32 v[22:25], v68, v10, v[22:25]") + "\ts_nop 7\n\ts_nop 7\n\ts_nop 3\n\tv_add_f32_e32 v1, v22, v2\n",          # 20 states on paper, three once an s_wakeup arrives
           _asm("v_mfma_f32_16x16x4_f32 v[22:25], v68, v10, v[22:25]") + VN16 + "\tv_add_f32_e32 v1, v22, v2\n"),
    'R5': (_asm("v_mfma_f32_4x4x1_16b_f32 v[4:7], v100, v32, v[4:7]") + _asm("v_mfma_f32_4x4x1_16b_f32 v[4:7], v101, v33, v[4:7]") + VN16,
           _asm("v_mfma_f32_4x4x1_16b_f32 v[4:7], v100, v32, v[4:7]") + _asm("v_nop\n\tv_nop\n\tv_mfma_f32_4x4x1_16b_f32 v[4:7], v101, v33, v[4:7]") + VN16),
}


@pytest.mark.parametrize('rule', sorted(CASES))
def test_checker_on_synthetic_listings(tmp_path, rule):
    bad, good = CASES[rule]
    rc, out = _run(bad, tmp_path, 'bad.s')
    assert rc == 1 and f' {rule}:' in out and f'{rule}=1' in out.splitlines()[-1], out
    rc, out = _run(good, tmp_path, 'good.s')
    assert rc == 0 and out.splitlines()[-1].startswith('0 MFMA'), out


def test_checker_follows_the_back_edge_of_a_loop(tmp_path):
    # an accumulate chain that ends in front of a loop's back edge: the window continues at the branch target (chain goes on: fine) and on the
    # fall-through path (a reader too early: R4)
    loop = ".LBB0_1:\n" + _asm("v_mfma_f32_4x4x1_16b_f32 v[4:7], v100, v32, v[4:7]") + "\ts_add_i32 s0, s0, 1\n\ts_cmp_lt_i32 s0, 8\n\ts_cbranch_scc1 .LBB0_1\n"
    rc, out = _run(loop + "\tv_add_f32_e32 v1, v4, v2\n", tmp_path, 'loop_bad.s')
    assert rc == 1 and ' R4:' in out, out
    rc, out = _run(loop + VN16 + "\tv_add_f32_e32 v1, v4, v2\n", tmp_path, 'loop_ok.s')
    assert rc == 0, out


@pytest.mark.timeout(900)
def test_product_listing_has_only_tied_mfma():
    from drloco_amd import lib
    last = lib.check_mfma_overlap()
    assert last.startswith('0 MFMA'), last
    n_mfma, n_asm = (int(x) for x in __import__('re').search(r'in (\d+) MFMA instructions \((\d+) inline asm\)', last).groups())
    assert n_mfma == n_asm > 1500          # no builtin MFMA is left in the product
    text = open(lib.LISTING).read()
    assert 'v_mfma_f32_16x16x4_f32' in text and 'v_mfma_f32_4x4x1_16b_f32' in text and 'k_rollout_pairs' in text
    src = ''.join(open(os.path.join(lib.CSRC, f)).read() for f in os.listdir(lib.CSRC) if f.endswith(('.hpp', '.hip')))
    src = __import__('re').sub(r'#ifdef DL_EXP_POLP_BUILTIN.*?#else', '', src, flags=__import__('re').S)          # the round-4 form kept for tools/asm_bisect.py: never compiled into the product (n_mfma == n_asm above)
    assert '__builtin_amdgcn_mfma' not in src
