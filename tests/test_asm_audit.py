"""CPU: the register hazards of the compositing kernels' asm statements, checked in the generated gfx950 code
(tools/audit_touch.py): nothing writes the sink register of a touch-ahead load between the load and its drain, and
every asm statement whose memory instruction takes a scalar base opens with the wait states the hardware needs after
v_readfirstlane."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_compositing_asm_statements_have_no_register_hazard():
    import audit_touch
    touches, problems = audit_touch.audit(audit_touch.assembly())
    assert touches >= 8, "the audit found no touch-ahead loads: its patterns no longer match the generated code"
    assert not problems, "\n".join(problems)


def test_the_audit_sees_a_hazard_when_there_is_one():
    import audit_touch
    text = """_Zkernel:
.LBB0_1:
	;;#ASMSTART
	s_nop 4
	global_load_dword v7, v3, s[4:5]
	;;#ASMEND
	v_mul_f32_e32 v7, v1, v2
	s_cbranch_scc1 .LBB0_1
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_endpgm
.Lfunc_end0:
"""
    touches, problems = audit_touch.audit(text)
    assert touches == 1 and len(problems) == 1 and "writes v7" in problems[0]
    clean = text.replace("	v_mul_f32_e32 v7, v1, v2\n", "	v_mul_f32_e32 v8, v1, v2\n")
    assert audit_touch.audit(clean) == (1, [])
    unpadded = clean.replace("	s_nop 4\n", "")
    assert any("s_nop 4" in p for p in audit_touch.audit(unpadded)[1])
    undrained = clean.replace("	s_waitcnt vmcnt(0)\n", "")
    assert any("without a drain" in p for p in audit_touch.audit(undrained)[1])
