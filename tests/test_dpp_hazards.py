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
GOOD = BAD.replace("\tv_fmac_f32_dpp", "\ts_nop 1\n\tv_fmac_f32_dpp")
ONE_SHORT = BAD.replace("\tv_fmac_f32_dpp", "\ts_nop 0\n\tv_fmac_f32_dpp")
OTHER_REG = BAD.replace("v_mov_b32_e32 v5, v3", "v_mov_b32_e32 v6, v3")


def _run(text, tmp_path, name):
    f = tmp_path / name
    f.write_text(text)
    return subprocess.run([sys.executable, TOOL, str(f)], capture_output=True, text=True)


def test_checker_on_synthetic_listings(tmp_path):
    p = _run(BAD, tmp_path, 'bad.s')
    assert p.returncode == 1 and '1 DPP read-after-write hazard(s)' in p.stdout and 'v_mov_b32_e32 v5, v3' in p.stdout
    assert _run(ONE_SHORT, tmp_path, 'short.s').returncode == 1
    for text, name in ((GOOD, 'good.s'), (OTHER_REG, 'other.s')):
        p = _run(text, tmp_path, name)
        assert p.returncode == 0 and '0 DPP read-after-write hazard(s)' in p.stdout, p.stdout


@pytest.mark.timeout(900)
def test_product_listing_has_no_dpp_hazard():
    from drloco_amd import lib
    assert lib.check_dpp_hazards().startswith('0 DPP')
    text = open(lib.LISTING).read()
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
    p = _run(ACROSS_BRANCH, tmp_path, 'branch.s')
    assert p.returncode == 1 and 'v_mov_b32_e32 v5, v3' in p.stdout, p.stdout
    ok = ACROSS_BRANCH.replace("\ts_cbranch_vccz", "\ts_nop 0\n\ts_cbranch_vccz")
    assert _run(ok, tmp_path, 'branch_ok.s').returncode == 0
