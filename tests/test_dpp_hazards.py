"""The hand-written DPP statements of the step kernels (v_fmac_f32_dpp / v_max_f32_dpp with row broadcasts and row shifts) have to keep the two wait
states between a VALU write of a register and its DPP read themselves: inline asm is opaque to the compiler's hazard recogniser, and the register
allocator may put a copy right in front of an asm statement.  tools/check_dpp_hazards.py checks a device listing for that; here:
  * the checker itself on two synthetic listings (a violation, and the same code with the wait states in place);
  * the listing of the product build (drloco_amd.lib.build(listing=True): the -save-temps output of the very compilation that makes the library)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, 'tools', 'check_dpp_hazards.py')

BAD = """_Z3kerPf: ; @_Z3kerPf
\tv_mul_f32_e32 v3, v1, v2
\tv_mov_b32_e32 v5, v3
\tv_fmac_f32_dpp v5, v5, v4 row_newbcast:3 row_mask:0xf bank_mask:0xf
\ts_endpgm
"""
GOOD = BAD.replace("\tv_fmac_f32_dpp", "\ts_nop 0\n\tv_fmac_f32_dpp")          # one wait state: what gfx950 was measured to need -- the opt-in code object (-DDL_DPP_WAIT=1), NOT the default
ONE_NOP1 = BAD.replace("\tv_fmac_f32_dpp", "\ts_nop 1\n\tv_fmac_f32_dpp")          # two states by the manual's count (hipcc's own padding); an s_wakeup can cut it to one
TWO = BAD.replace("\tv_fmac_f32_dpp", "\ts_nop 0\n\ts_nop 0\n\tv_fmac_f32_dpp")
OTHER_REG = BAD.replace("v_mov_b32_e32 v5, v3", "v_mov_b32_e32 v6, v3")


def _run(text, tmp_path, name, *extra):
    f = tmp_path / name
    f.write_text(text)
    return subprocess.run([sys.executable, TOOL, str(f), *extra], capture_output=True, text=True)


def test_checker_on_synthetic_listings(tmp_path):
    p = _run(BAD, tmp_path, 'bad.s')
    assert p.returncode == 1 and '1 DPP read-after-write hazard(s)' in p.stdout and 'v_mov_b32_e32 v5, v3' in p.stdout
    # the DEFAULT is the ISA manual's two wait states (the default library, DL_DPP_WAIT = 2): one state is a violation at the default setting ...
    p = _run(GOOD, tmp_path, 'one_state.s')
    assert p.returncode == 1 and '1 DPP read-after-write hazard(s)' in p.stdout
    # ... two `s_nop 0`, hipcc's own `s_nop 1` (two states by the manual's count) and an unrelated register are fine
    for text, name in ((TWO, 'two.s'), (OTHER_REG, 'other.s'), (ONE_NOP1, 'nop1.s')):
        p = _run(text, tmp_path, name)
        assert p.returncode == 0 and '0 DPP read-after-write hazard(s)' in p.stdout, p.stdout
    # counted as what it is worth beside an s_wakeup (--snop one), a single s_nop of any count is ONE state
    assert _run(ONE_NOP1, tmp_path, 'nop1_one.s', '--need', '2', '--snop', 'one').returncode == 1
    assert _run(TWO, tmp_path, 'two_one.s', '--need', '2', '--snop', 'one').returncode == 0
    # the opt-in one-state code object is checked with --need 1 (every s_nop one state): one `s_nop 0` passes, none fails
    assert _run(GOOD, tmp_path, 'good1.s', '--need', '1').returncode == 0 and _run(ONE_NOP1, tmp_path, 'nop1_1.s', '--need', '1').returncode == 0
    assert _run(BAD, tmp_path, 'bad1.s', '--need', '1').returncode == 1


@pytest.mark.timeout(900)
@pytest.mark.parametrize('variant', ['w2', 'w1'])
def test_product_listing_has_no_dpp_hazard(variant):
    """Both code objects of the product: the default (two wait states, the manual's figure) and the one-state build a device has to earn (drloco_amd/lib.py)."""
    from drloco_amd import lib
    assert lib.check_dpp_hazards(variant).startswith('0 DPP')
    text = open(lib.VARIANTS[variant]['listing']).read()
    # the two builds differ where they should: the hand-written wait in front of a DPP chain step
    assert ('s_nop 0\n\ts_nop 0\n\tv_fmac_f32_dpp' in text) == (variant == 'w2')
    # the listing is the product's: the step kernels and their hand-written DPP forms are in it
    assert 'k_env_step_g16_split' in text and 'k_rollout_persistent' in text and text.count('v_fmac_f32_dpp') > 1000 and 'v_max_f32_dpp' in text


ACROSS_BRANCH = """_Z3kerPf: ; @_Z3kerPf
\tv_mul_f32_e32 v3, v1, v2
\tv_mov_b32_e32 v5, v3
\ts_cbranch_vccz .LBB0_2
\tv_add_f32_e32 v7, v1, v2
\tv_add_f32_e32 v8, v1, v2
\tv_add_f32_e32 v9, v1, v2
.LBB0_2:
\tv_fmac_f32_dpp v6, v5, v4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
\ts_endpgm
"""


def test_checker_follows_branches(tmp_path):
    # the write is two instructions (one wait state: the branch) in front of the read on the TAKEN path, four on the fall-through path
    # (the default, the manual's two states; with --need 1 the branch itself is the wait)
    p = _run(ACROSS_BRANCH, tmp_path, 'branch.s', '--need', '2')
    assert p.returncode == 1 and 'v_mov_b32_e32 v5, v3' in p.stdout, p.stdout
    assert _run(ACROSS_BRANCH, tmp_path, 'branch1.s', '--need', '1').returncode == 0
    ok = ACROSS_BRANCH.replace("\ts_cbranch_vccz", "\ts_nop 0\n\ts_cbranch_vccz")
    assert _run(ok, tmp_path, 'branch_ok.s', '--need', '2').returncode == 0
    direct = ACROSS_BRANCH.replace("\ts_cbranch_vccz .LBB0_2\n", "").replace("\tv_add_f32_e32 v7, v1, v2\n\tv_add_f32_e32 v8, v1, v2\n\tv_add_f32_e32 v9, v1, v2\n", "")
    assert _run(direct, tmp_path, 'direct.s').returncode == 1          # no instruction between the write and the DPP read: a hazard at any setting


@pytest.mark.timeout(900)
@pytest.mark.parametrize('variant', ['w2', 'w1'])
def test_no_multi_state_s_nop_in_the_kernels_that_use_s_wakeup(variant):
    """An s_wakeup of another wave of the workgroup ends an s_nop after ONE wait state (tools/ubench/snop_wakeup.hip), so hipcc's own padding may be trusted in the split /
    rollout kernels only where one state is enough: `s_nop 0 / 1` (the DPP class).  Anything longer -- v_div_fmas behind a VCC write, an SGPR address behind a VALU write, a
    builtin MFMA's reader -- must not appear ANYWHERE: round 5 had exempted the `s_nop 3` in front of the fault word's global_atomic_or on the time-out paths, and round 6 saw exactly
    that one cut short (a stale SGPR base, a memory access fault in one of five runs of test_split_handover_timeout_raises); the atomic is hand-written with v_nop since."""
    from drloco_amd import lib
    lib.check_dpp_hazards(variant)          # (builds the listing if needed)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'survey_snop.py'), lib.VARIANTS[variant]['listing'], '--check'], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip().splitlines()[-1].startswith('0 multi-state'), p.stdout[-2000:]
    assert 'k_env_step_g16_split' in p.stdout and 'k_rollout_pairs' in p.stdout
