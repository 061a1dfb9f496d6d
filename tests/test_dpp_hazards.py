"""Static check of the hand-written DPP blocks (csrc/mppi_quad.hpp: rotations folded into v_fmac_f32_dpp).  On gfx9 a VGPR read
through a DPP lane permutation must not have been written during the two preceding wait states; the compiler pads its own DPP
instructions but does not look into inline assembly, so the blocks are spaced by construction - and this test disassembles
every gfx950 code object of the built library and checks every DPP instruction in it (tools/check_dpp_hazards.py) - and, in the same pass, that no ordinary VALU instruction reads the result of a
transcendental one in the next issue slot (the second hazard inline assembly can run into on gfx940+)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_dpp_hazards as chk  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def test_checker_sees_a_hazard_and_accepts_padding(tmp_path):
    src = tmp_path / "h.s"
    src.write_text("\tv_mul_f32_e32 v1, v2, v3\n\tv_add_f32_e32 v4, v5, v6\n"
                   "\tv_fmac_f32_dpp v7, v1, v8 quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"   # v1: one wait state ago
                   "\tv_mul_f32_e32 v9, v2, v3\n\ts_nop 1\n"
                   "\tv_add_f32_dpp v4, v9, v6 quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xf\n")             # padded: fine
    assert chk.check(str(src)) == 1


def test_checker_sees_a_transcendental_read_too_early(tmp_path):
    """gfx940+: the result of v_sin / v_cos / v_rcp ... must not be read by an ordinary VALU instruction in the next issue slot
    (the packed kinematics block of quad_fk takes (cos q, sin q): it is an operand of the whole block, read six instructions in)"""
    src = tmp_path / "t.s"
    src.write_text("\tv_sin_f32_e32 v1, v2\n\tv_pk_mul_f32 v[4:5], v[6:7], v[0:1] op_sel_hi:[1,0]\n"      # hazard
                   "\tv_cos_f32_e32 v8, v2\n\ts_nop 0\n\tv_mul_f32_e32 v9, v8, v8\n"                      # padded: fine
                   "\tv_rcp_f32_e32 v10, v2\n\tv_sqrt_f32_e32 v11, v10\n")                                 # trans after trans: fine
    assert chk.check(str(src)) == 1


def test_checker_flags_hand_written_dpp_right_behind_a_label(tmp_path):
    """the write history ends at a label (control flow may arrive from a block that has just written the operand).  The
    compiler's own DPP instructions are trusted there; the forms only the inline-assembly blocks emit (v_fmac_f32_dpp, row_ror:8
    exchanges) are not - within two wait states of a label they are reported unless padded"""
    src = tmp_path / "l.s"
    src.write_text(".LBB0_1:\n\tv_fmac_f32_dpp v7, v1, v8 quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"     # hazard
                   ".LBB0_2:\n\tv_mul_f32_e32 v9, v2, v3\n\tv_mov_b32_dpp v4, v5 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"   # 1 slot: hazard
                   ".LBB0_3:\n\ts_nop 1\n\tv_fmac_f32_dpp v7, v1, v8 quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"   # padded: fine
                   ".LBB0_4:\n\tv_add_f32_dpp v4, v9, v6 quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xf\n")                              # compiler form: trusted
    assert chk.check(str(src)) == 2


def test_built_library_has_no_dpp_hazard():
    lib = os.path.join(ROOT, "mppi-isaac_amd", "csrc", "libmppi_hip.so")
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    assert os.path.exists(lib), "libmppi_hip.so is not built (__graft_entry__.build())"
    bad, seen = chk.check_library(lib, OBJDUMP)
    assert seen > 10000          # the quad / octet kernels are full of DPP instructions: the disassembly was really read
    assert bad == 0
