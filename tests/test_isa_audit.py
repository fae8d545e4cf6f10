"""The constant-time contract, machine-checked: tools/isa_audit.py compiles the index-independent kernels to gfx950 ISA
(hipcc -S cross-compiles here, no GPU) and runs a taint analysis from the secret arguments: no branch condition and no
memory address may depend on them (/root/reference README.md:92-97, src/include/constant_time.h:134-183).

Cold, the five translation units take about three minutes to compile (cached in build/isa/ afterwards)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_audit  # noqa: E402

pytestmark = pytest.mark.skipif(not (os.path.exists(isa_audit.HIPCC) or shutil.which("hipcc")), reason="needs hipcc")


@pytest.fixture(scope="module")
def isa():
    from concurrent.futures import ThreadPoolExecutor
    tus = sorted({v[0] for v in list(isa_audit.AUDIT.values()) + list(isa_audit.CONTROLS.values())})
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(isa_audit.compile_isa, tus))
    return True


def _kernel(text, secret_args, nargs=3):
    """a synthetic kernel body for the analyser: `nargs` pointer arguments at kernarg offsets 0, 8, 16, ..."""
    body = [l for l in text.strip().split("\n")]
    offs = [(8 * k, 8, True) for k in range(nargs)]
    return isa_audit.Audit("synthetic", body, 1, offs, secret_args)


def test_the_analyser_on_hand_written_snippets():
    # arg 0: output, arg 1: a public table, arg 2: the secret
    prologue = """
	s_load_dwordx4 s[4:7], s[0:1], 0x0
	s_load_dwordx2 s[8:9], s[0:1], 0x10
	v_lshlrev_b32_e32 v1, 2, v0
	global_load_dword v2, v1, s[8:9]
"""
    # (1) a table read at a secret index
    bad = _kernel(prologue + """
	v_lshlrev_b32_e32 v3, 2, v2
	global_load_dword v4, v3, s[6:7]
	s_endpgm
""", [2]).run()
    assert [v.kind for v in bad] == ["address"] and bad[0].prov == (1,)
    # (2) a branch on a secret comparison, and the clean version of the same choice (a select)
    bad = _kernel(prologue + """
	v_cmp_eq_u32_e32 vcc, 0, v2
	s_cbranch_vccz .LBB0_2
	v_mov_b32_e32 v5, 1
.LBB0_2:
	s_endpgm
""", [2]).run()
    assert [v.kind for v in bad] == ["branch"]
    bad = _kernel(prologue + """
	v_cmp_eq_u32_e32 vcc, 0, v2
	v_cndmask_b32_e32 v5, v6, v7, vcc
	global_store_dword v1, v5, s[4:5]
	s_endpgm
""", [2]).run()
    assert bad == []
    # (3) divergence on a secret: the skipped branch and the memory access under the secret EXEC are both reported
    bad = _kernel(prologue + """
	v_cmp_eq_u32_e32 vcc, 0, v2
	s_and_saveexec_b64 s[10:11], vcc
	s_cbranch_execz .LBB0_2
	global_load_dword v8, v1, s[6:7]
.LBB0_2:
	s_or_b64 exec, exec, s[10:11]
	s_endpgm
""", [2]).run()
    assert sorted(v.kind for v in bad) == ["branch", "exec"]
    # (4) the secret through LDS, a spill slot and an SGPR lane spill; a wavefront shuffle by a secret lane number is a
    #     register move, a v_readlane with a secret lane select is not
    a = _kernel(prologue + """
	ds_write_b32 v1, v2
	ds_read_b32 v9, v1
	scratch_store_dword off, v9, off offset:16
	scratch_load_dword v10, off, off offset:16
	scratch_load_dword v12, off, off offset:32
	v_readfirstlane_b32 s12, v10
	v_writelane_b32 v40, s12, 3
	v_readlane_b32 s13, v40, 3
	v_readlane_b32 s14, v40, 4
	ds_bpermute_b32 v11, v10, v1
	v_lshlrev_b32_e32 v13, 2, v11
	global_load_dword v14, v13, s[6:7]
	v_lshlrev_b32_e32 v15, 2, v12
	global_load_dword v16, v15, s[6:7]
	s_lshl_b32 s15, s14, 2
	s_load_dword s16, s[6:7], s15
	s_lshl_b32 s17, s13, 2
	s_load_dword s18, s[6:7], s17
	v_readlane_b32 s19, v1, s13
	s_endpgm
""", [2])
    bad = a.run()
    assert a.lds_tainted
    got = [(v.kind, v.ins.text.split()[0], v.ins.ops[0]) for v in bad]
    assert got == [("address", "global_load_dword", "v14"), ("address", "s_load_dword", "s18"), ("address", "v_readlane_b32", "s19")], got
    # (5) results stored through another argument make that argument secret (a later reload is secret data)
    a = _kernel(prologue + """
	global_store_dword v1, v2, s[4:5]
	global_load_dword v20, v1, s[4:5]
	v_lshlrev_b32_e32 v21, 2, v20
	global_load_dword v22, v21, s[6:7]
	s_endpgm
""", [2])
    bad = a.run()
    assert a.secret == {0, 2} and [v.kind for v in bad] == ["address"]
    # (6) a loop whose trip count is public is fine, one whose exit depends on the secret is not
    bad = _kernel(prologue + """
	s_mov_b32 s20, 4
.LBB0_1:
	v_add_u32_e32 v2, v2, v2
	s_add_i32 s20, s20, -1
	s_cmp_lg_u32 s20, 0
	s_cbranch_scc1 .LBB0_1
	v_readfirstlane_b32 s21, v2
.LBB0_3:
	s_add_i32 s21, s21, -1
	s_cmp_lg_u32 s21, 0
	s_cbranch_scc1 .LBB0_3
	s_endpgm
""", [2]).run()
    assert [(v.kind, v.ins.ops[0]) for v in bad] == [("branch", ".LBB0_3")]


def test_every_index_independent_kernel_is_clean(isa):
    report = []
    for kernel in isa_audit.AUDIT:
        bad, allowed, stats = isa_audit.audit_kernel(kernel)
        report.append((kernel, len(bad), len(allowed), stats))
        assert stats["unhandled"] == [], (kernel, stats["unhandled"])
        assert stats["tainted_loads"] > 0, "%s: no load returned secret data -- the secret arguments are mis-declared" % kernel
        assert stats["reached"] > 0.9 * stats["instructions"], (kernel, "unreachable code: the CFG is wrong")
        assert bad == [], "%s: %d secret-dependent branches / addresses, first: %s" % (kernel, len(bad), bad[0])
        for v in allowed:                      # an exception is a documented reason, never a branch
            assert v.kind == "address"
    # only the kernels that serve a public-by-contract entry point through the same code have exceptions
    assert {k for k, _, n, _ in report if n} <= {"k_x448", "k_double_scalarmul_wave"}


def test_the_audit_flags_the_digit_addressed_kernels(isa):
    """negative control: the opt-in `fast` kernels read their tables by the digit, and the audit must see it"""
    for kernel in isa_audit.CONTROLS:
        bad, _, stats = isa_audit.audit_kernel(kernel)
        kinds = {v.kind for v in bad}
        assert "address" in kinds, (kernel, kinds)
        assert all(v.kind in ("address", "exec", "branch") for v in bad)
        # ... and what it flags are loads from the window table (argument 1 of k_base_scalarmul, 10 of k_ed448_sign)
        table_arg = {"k_base_scalarmul": 1, "k_ed448_sign": 10}[kernel]
        assert any(table_arg in v.prov for v in bad if v.kind == "address"), (kernel, [v.prov for v in bad][:5])
