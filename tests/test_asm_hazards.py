"""The library's inline-asm vector-memory statements against the hazards the compiler cannot see (tools/asm_hazard_audit.py): an asm load's
destination may not be touched before its wait; an asm store of more than 64 bits must be followed by its wait inside the same statement.
r04: the k split of th_mlp2_xent stored with one statement per store, the compiler reused the first store's data registers, and with other
processes on the GPU 5 - 8 of 30 captured runs had a wrong step.  hipcc cross-compiles without a GPU: a CPU test."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


def test_inline_asm_memory_statements_are_hazard_free():
    import asm_hazard_audit as A
    if not A.have_hipcc():
        pytest.skip("needs hipcc ($HIPCC, PATH or /opt/rocm/bin)")
    srcs = [p for p in sorted((ROOT / "taper_amd" / "csrc").glob("*.hip")) if "asm volatile" in p.read_text()
            and any(k in p.read_text() for k in ("global_load_dword", "global_store_dword", "buffer_load_dword", "buffer_store_dword"))]
    assert srcs, "no inline-asm memory statements found: the audit has nothing to look at"
    for s in srcs:
        findings = A.audit(A.isa(s))
        assert not findings, f"{s.name}: " + "; ".join(findings)


def test_the_audit_sees_a_store_without_its_wait():
    import asm_hazard_audit as A
    listing = """_Zkernel:
\t;;#ASMSTART
\tglobal_store_dwordx4 v[8:9], v[4:7], off sc0 sc1
\t;;#ASMEND
\tv_accvgpr_read_b32 v7, a3
\t;;#ASMSTART
\tglobal_load_dwordx4 v[20:23], v[2:3], off sc0 sc1
\t;;#ASMEND
\tv_mov_b32_e32 v30, v21
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
.Lfunc_end0:""".split("\n")
    f = A.audit(listing)
    assert len(f) == 2 and "asm store" in f[0] and "touches the destination" in f[1], f


def test_the_audit_sees_a_narrow_store_without_its_wait():
    """<= 64-bit asm stores read their data after issue as well: flagged like the wide ones"""
    import asm_hazard_audit as A
    listing = """_Zkernel:
\t;;#ASMSTART
\tglobal_store_dwordx2 v[8:9], v[4:5], off sc0 sc1
\t;;#ASMEND
\tv_mov_b32_e32 v4, 0
\t;;#ASMSTART
\tglobal_store_dword v[8:9], v6, off sc1
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
.Lfunc_end0:""".split("\n")
    f = A.audit(listing)
    assert len(f) == 1 and "global_store_dwordx2" in f[0], f


def test_the_audit_follows_branches():
    """a wait in ANOTHER basic block only counts on the paths that pass it: the taken branch below reaches a reader of the destination
    without a wait (a linear scan of the file finds the wait first and sees nothing); with the wait on both paths the listing is clean"""
    import asm_hazard_audit as A
    bad = """_Zkernel:
\t;;#ASMSTART
\tglobal_load_dwordx4 v[20:23], v[2:3], off sc0 sc1
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_2
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tv_add_f32_e32 v1, v20, v21
\ts_endpgm
.LBB0_2:
\tv_add_f32_e32 v1, v22, v23
\ts_endpgm
.Lfunc_end0:""".split("\n")
    f = A.audit(bad)
    assert len(f) == 1 and "v22, v23" in f[0], f
    good = [l for l in bad]
    k = good.index(".LBB0_2:")
    good[k + 1:k + 1] = ["\t;;#ASMSTART", "\ts_waitcnt vmcnt(0) lgkmcnt(0)", "\t;;#ASMEND"]
    assert A.audit(good) == []
    # a loop back edge: the load's destination is read at the top of the next iteration, before the wait at the loop's end
    loop = """_Zkernel:
.LBB0_1:
\tv_mov_b32_e32 v30, v21
\t;;#ASMSTART
\tglobal_load_dwordx4 v[20:23], v[2:3], off sc0 sc1
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_1
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\ts_endpgm
.Lfunc_end0:""".split("\n")
    f = A.audit(loop)
    assert len(f) == 1 and "v_mov_b32_e32 v30, v21" in f[0], f
