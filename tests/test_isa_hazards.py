"""The build gate tools/isa_hazards.py: it must flag the instruction pair that faulted in round 1
(an SGPR restored by a VALU instruction right in front of an inline-asm LDS-DMA that reads it as its
base) and accept the same pair with five wait states between, or with a scalar-ALU producer."""
import os
import sys
import textwrap

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import isa_hazards  # noqa: E402

HEAD = "\t.type\tk_test,@function\nk_test:\n"


def _scan(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text(HEAD + textwrap.dedent(body))
    return isa_hazards.scan(str(p))


def test_spill_restore_in_front_of_asm_dma_is_flagged(tmp_path):
    found = _scan(tmp_path, """\
        \tv_readlane_b32 s2, v255, 13
        \tv_readlane_b32 s3, v255, 14
        \t;;#ASMSTART
        \tglobal_load_lds_dwordx4 v252, s[2:3]
        \t;;#ASMEND
        \tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]
        \tv_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[8:11], a[16:31]
        \t;;#ASMSTART
        \tglobal_load_lds_dwordx4 v252, s[2:3] offset:1024
        \t;;#ASMEND
        """)
    assert [f[0] for f in found] == ["H1", "H1"]   # 0 and 4 wait states: both short of 5
    assert "v_readlane_b32 s3" in found[0][4]


def test_five_wait_states_or_a_scalar_producer_pass(tmp_path):
    assert _scan(tmp_path, """\
        \tv_readlane_b32 s2, v255, 13
        \tv_readlane_b32 s3, v255, 14
        \ts_nop 4
        \t;;#ASMSTART
        \tglobal_load_lds_dwordx4 v252, s[2:3]
        \t;;#ASMEND
        """) == []
    assert _scan(tmp_path, """\
        \t;;#ASMSTART
        \ts_add_u32 s2, s16, 0x4000
        \ts_addc_u32 s3, s17, 0
        \t;;#ASMEND
        \t;;#ASMSTART
        \tglobal_load_lds_dwordx4 v252, s[2:3]
        \t;;#ASMEND
        """) == []


def test_compiler_loads_are_the_compilers_business(tmp_path):
    # outside an asm block hipcc pads the pair itself; the gate looks at asm blocks only
    assert _scan(tmp_path, """\
        \tv_readfirstlane_b32 s4, v1
        \tglobal_load_dword v6, v3, s[4:5]
        """) == []


def test_m0_written_by_the_compiler_or_without_a_wait_state(tmp_path):
    found = _scan(tmp_path, """\
        \t;;#ASMSTART
        \ts_mov_b32 m0, s53
        \ts_nop 0
        \t;;#ASMEND
        \ts_mov_b32 m0, s9
        \t;;#ASMSTART
        \tglobal_load_lds_dwordx4 v252, s[2:3]
        \t;;#ASMEND
        """)
    assert sorted(f[0] for f in found) == ["H2", "H3"]
