"""Invariants of the generated gfx950 code that the source cannot guarantee by itself (CPU suite: hipcc -S cross-compiles without a GPU).

1. No merged conditional store in the state machine.  Round 4 root-caused a wrong result (profiles/r04_flat_anomaly.txt) to LLVM sinking
   the `int` stores of complementary branches into ONE store through a SELECTED ADDRESS -- `st.trial[b] = 0` became `st.status[b] = 0` in one
   build.  The fix is a source pattern (store_settled / arm_line_search in kernels_common.h: every word stored unconditionally with a selected
   VALUE, no complementary branches), which nothing enforced: a compiler bump or an edit could bring the merged store back silently.  This test
   disassembles the library's device code and fails if, in any backward kernel (they contain the settle step and the arming of the line
   search), a 32-bit global / flat store takes its address from registers written by v_cndmask in the same basic block.  (The gain stores
   select between a trajectory's slot and a dump slot on purpose; they are 64- and 128-bit stores and are not what this looks at.)
2. The general backward kernel keeps nothing in scratch memory (round 5: its right-hand side lived there, six round trips per knot on the
   dependent chain, because three conditional exchanges were merged into one indexed access).
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "quadrotorilqr_amd", "csrc")
ASM = os.path.join(ROOT, "tests", "_device_code.s")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm():
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc", ".hip"))] + [os.path.join(ROOT, "include", "quadrotor_ilqr.h")]
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    if not os.path.exists(ASM) or any(os.path.getmtime(ASM) < os.path.getmtime(f) for f in srcs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S",
                               "--cuda-device-only", "-o", ASM, os.path.join(CSRC, "ilqr_capi.hip")], stderr=subprocess.DEVNULL)
    return open(ASM).read().split("\n")


def functions(lines):
    """name -> body lines of every kernel / function in the assembly"""
    out, name, start = {}, None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, start = m.group(1), i
        elif l.startswith(".Lfunc_end") and name:
            out[name] = lines[start:i]
            name = None
    return out


def regs(tok):
    """register numbers named by an operand like v12 or v[12:13]"""
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def selected_address_stores(body):
    """32-bit global / flat stores of a function body whose address registers were written by v_cndmask in the same basic block"""
    bad, selected = [], set()
    for l in body:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if t.endswith(":") or t.startswith(".LBB"):
            selected = set()
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        if op.startswith("v_cndmask_b32"):
            selected |= regs(ops[0])
        elif op in ("global_store_dword", "flat_store_dword"):
            if regs(ops[0]) & selected:
                bad.append(t)
        elif op.startswith("v_") and ops:
            selected -= regs(ops[0])  # overwritten by something else
    return bad


def test_the_detector_sees_a_merged_store():
    merged = """
	v_cndmask_b32_e32 v4, v10, v12, vcc
	v_cndmask_b32_e32 v5, v11, v13, vcc
	v_mov_b32_e32 v6, 0
	global_store_dword v[4:5], v6, off
""".split("\n")
    assert selected_address_stores(merged) == ["global_store_dword v[4:5], v6, off"]
    plain = """
	v_cndmask_b32_e32 v6, v10, v12, vcc
	global_store_dword v[4:5], v6, off
.LBB0_2:
	v_cndmask_b32_e32 v4, v10, v12, vcc
	v_add_u32_e32 v4, 1, v9
	global_store_dword v[4:5], v6, off
""".split("\n")
    assert selected_address_stores(plain) == []  # a selected VALUE is the pattern of the fix; an overwritten register is not selected any more


def test_no_int_store_through_a_selected_address_in_the_backward_kernels(asm):
    fns = functions(asm)
    backward = {n: b for n, b in fns.items() if re.search(r"k_backward|k_round|k_accept", n)}
    assert len(backward) >= 8, sorted(backward)  # k_backward<..>, k_backward4<..>, k_backward_rollout<..>, k_round<..>, k_accept
    bad = [(name[:60], t) for name, body in backward.items() for t in selected_address_stores(body)]
    assert not bad, bad


def test_the_general_backward_kernel_uses_no_scratch_memory(asm):
    text = "\n".join(asm)
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (_ZN5qilqr10k_backwardILb0E\w+)(.*?)\.end_amdhsa_kernel", text, re.S):
        seen += 1
        size = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2)).group(1))
        assert size == 0, (m.group(1), size)
    assert seen == 2  # double and float storage
