"""The device listing of the shipped library passes scripts/isa_audit.py (no GPU needed: hipcc cross-compiles).

Why this is a test: hipcc treats an inline-asm statement as one opaque instruction (no hazard padding across its boundary), and the register
allocator of this toolchain can place a live-range split copy in front of the `s_or_b64 exec` that re-opens the lanes at an if / else join --
the copy then runs under one side's mask (possibly no lane) and the value is lost for the other lanes.  Round 2 met that as a build variant
whose rti_solve_kernel<10, 64, 3> stored status / iterations / cost to wrong addresses.  Both defects are visible in
the listing and invisible to every functional test that happens not to touch the damaged lanes.  Round 3 met a second register-allocator defect (rule P2, a
kernel-argument load re-materialised over the live part of another one: rti_solve_kernel<3, 32, 2> added its iteration count to the episode's step counter,
DESIGN.md section 8.5b); its dynamic counterpart is tests/test_gpu_every_kernel.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUDIT = os.path.join(ROOT, "scripts", "isa_audit.py")

BAD_JOIN = """
kern:
	s_and_saveexec_b64 s[0:1], s[2:3]
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_add_f64 v[0:1], v[0:1], v[2:3]
.LBB0_2:
	v_accvgpr_write_b32 a0, v106
	s_mov_b32 s7, s68
	s_or_b64 exec, exec, s[0:1]
	s_endpgm
"""
GOOD_JOIN = BAD_JOIN.replace("\tv_accvgpr_write_b32 a0, v106\n\ts_mov_b32 s7, s68\n\ts_or_b64 exec, exec, s[0:1]\n",
                             "\tv_writelane_b32 v255, s4, 3\n\ts_or_b64 exec, exec, s[0:1]\n\tv_accvgpr_write_b32 a0, v106\n")
BAD_DPP = """
kern:
	v_add_f64 v[4:5], v[0:1], v[2:3]
	;;#ASMSTART
	v_fmac_f64_dpp v[6:7], v[4:5], v[8:9] row_newbcast:0 row_mask:0xf bank_mask:0xf
	;;#ASMEND
	s_endpgm
"""

BAD_TORN = """
kern:
	s_load_dwordx8 s[8:15], s[6:7], 0x248
	s_waitcnt lgkmcnt(0)
	v_writelane_b32 v253, s12, 18
	v_writelane_b32 v253, s13, 19
	s_load_dwordx16 s[12:27], s[6:7], 0x1c8
	s_waitcnt lgkmcnt(0)
	s_load_dwordx8 s[8:15], s[6:7], 0x248
	s_waitcnt lgkmcnt(0)
	s_cmp_lg_u64 s[10:11], 0
	v_writelane_b32 v254, s12, 12
	v_writelane_b32 v254, s13, 13
	v_writelane_b32 v254, s14, 14
	v_writelane_b32 v254, s15, 15
	v_writelane_b32 v254, s16, 16
	v_writelane_b32 v254, s17, 17
	v_writelane_b32 v254, s18, 18
	v_writelane_b32 v254, s19, 19
	s_endpgm
"""
# the same, but the first four registers of the wide load were consumed before the narrow load took them over: legitimate reuse
GOOD_TORN = BAD_TORN.replace("\ts_load_dwordx16 s[12:27], s[6:7], 0x1c8\n\ts_waitcnt lgkmcnt(0)\n",
                             "\ts_load_dwordx16 s[12:27], s[6:7], 0x1c8\n\ts_waitcnt lgkmcnt(0)\n\ts_cmp_eq_u64 s[12:13], 0\n\ts_add_u32 s30, s14, s15\n")


def run(path):
    return subprocess.run([sys.executable, AUDIT, path], capture_output=True, text=True)


def test_audit_rules_fire_on_minimal_listings(tmp_path):
    for name, text, clean in (("bad_join", BAD_JOIN, False), ("good_join", GOOD_JOIN, True), ("bad_dpp", BAD_DPP, False),
                              ("good_dpp", BAD_DPP.replace("\t;;#ASMSTART\n", "\t;;#ASMSTART\n\ts_nop 1\n"), True), ("bad_torn", BAD_TORN, False),
                              ("good_torn", GOOD_TORN, True)):
        f = tmp_path / f"{name}.s"
        f.write_text(text)
        r = run(str(f))
        assert (r.returncode == 0) == clean, (name, r.stdout)
    assert "P1" in run(str(tmp_path / "bad_join.s")).stdout and "R1" in run(str(tmp_path / "bad_dpp.s")).stdout and "P2" in run(str(tmp_path / "bad_torn.s")).stdout


def test_shipped_library_listing_is_clean(built):
    """the listing build() wrote beside the library it built (same flags, same sources)"""
    from mpc_gpu import _lib
    if not os.path.exists(_lib.ISA_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_lib.ISA_PATH) for s in _lib.sources()):
        _lib.build(force=True)
    assert os.path.exists(_lib.ISA_PATH)
    r = run(_lib.ISA_PATH)
    assert r.returncode == 0, r.stdout[-3000:]
    n = int(r.stdout.strip().split("\n")[-1].split()[0])
    assert n > 300000, n         # every kernel instantiation of the library was scanned, not a stub
