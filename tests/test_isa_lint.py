"""The shipped code objects carry no store-data hazard (gfx950, found in round 5).

A buffer store of more than 64 bits reads its data registers after issue: a vector instruction that rewrites one of them
needs one wait state behind the store when the store's soffset is an SGPR (the compiler inserts none: the ISA manuals exempt
that form) and two when it is a literal (the compiler pads those).  tools/mb/store_hazard_scan.py reads the disassembly of
every code object in the library; `make` runs it after the link, this test runs it on what is about to be loaded.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "mb"))
import store_hazard_scan as lint  # noqa: E402

ROUND4 = """
0000000000001000 <cn_minsum_rec_kernel_first_long>:
	s_waitcnt vmcnt(0)
	buffer_store_dwordx4 v[0:3], v58, s[56:59], s0 offen nt
	v_and_b32_e32 v2, 63, v53
	s_endpgm
"""


def hits(asm, **kw):
    return lint.scan(asm, check=True, report=lambda *_: None, **kw)[0]


def test_scanner_flags_the_pattern_round_4_shipped_around():
    assert hits(ROUND4) == 1
    # one wait state is enough for the SGPR-soffset form, whatever provides it
    assert hits(ROUND4.replace("\tv_and_b32_e32 v2", "\ts_nop 0\n\tv_and_b32_e32 v2")) == 0
    assert hits(ROUND4.replace("\tv_and_b32_e32 v2", "\ts_add_u32 s1, s1, 4\n\tv_and_b32_e32 v2")) == 0
    assert hits(ROUND4.replace("v_and_b32_e32 v2,", "v_and_b32_e32 v4,")) == 0          # another register
    assert hits(ROUND4.replace("v_and_b32_e32 v2, 63, v53", "buffer_load_dword v2, v58, s[56:59], 0 offen")) == 0   # not VALU
    # the literal-soffset form needs two (the compiler pads these itself; the lint checks it did)
    lit = ROUND4.replace("s0 offen", "0 offen")
    assert hits(lit.replace("\tv_and_b32_e32 v2", "\ts_nop 0\n\tv_and_b32_e32 v2")) == 1
    assert hits(lit.replace("\tv_and_b32_e32 v2", "\ts_nop 1\n\tv_and_b32_e32 v2")) == 0
    # 64-bit stores are not subject to it; 96-bit ones are; global stores need two wait states
    assert hits(ROUND4.replace("dwordx4 v[0:3]", "dwordx2 v[2:3]")) == 0
    assert hits(ROUND4.replace("dwordx4 v[0:3]", "dwordx3 v[0:2]")) == 1
    glob = ROUND4.replace("buffer_store_dwordx4 v[0:3], v58, s[56:59], s0 offen nt", "global_store_dwordx4 v[10:11], v[0:3], off")
    assert hits(glob) == 1 and hits(glob.replace("\tv_and_b32_e32 v2", "\ts_nop 1\n\tv_and_b32_e32 v2")) == 0


def test_product_library_has_no_store_data_hazard():
    from ldpc_toolbox_amd import _capi
    text = lint.disassemble(_capi.LIB_PATH)
    found = []
    n, stores, per = lint.scan(text, check=True, report=found.append)
    assert stores > 500, "no wide stores found: is this the product library?"
    assert n == 0, "\n".join(found[:5])


def test_a_check_that_saw_nothing_does_not_pass(tmp_path):
    """the build's lint (`make`: --check --min-stores 500) must not pass vacuously: a file without a gfx code object (here the
    host-only oracle library), or a disassembly with fewer wide stores than the library is known to have, exits 2"""
    import subprocess
    tool = os.path.join(ROOT, "tools", "mb", "store_hazard_scan.py")
    host_only = os.path.join(ROOT, "oracle", "libldpc_oracle.so")
    if not os.path.exists(host_only):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    r = subprocess.run([sys.executable, tool, host_only, "--check", "--quiet"], capture_output=True, text=True)
    assert r.returncode == 2 and "nothing was checked" in r.stderr, (r.returncode, r.stderr)
    asm = tmp_path / "one.s"
    asm.write_text(ROUND4.replace("\tv_and_b32_e32 v2", "\ts_nop 0\n\tv_and_b32_e32 v2"))
    ok = subprocess.run([sys.executable, tool, str(asm), "--check", "--quiet"], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    few = subprocess.run([sys.executable, tool, str(asm), "--check", "--quiet", "--min-stores", "500"], capture_output=True, text=True)
    assert few.returncode == 2 and "nothing was checked" in few.stderr
    bad = subprocess.run([sys.executable, tool, str(asm).replace("one.s", "bad.s"), "--check", "--quiet"], capture_output=True, text=True) \
        if (tmp_path / "bad.s").write_text(ROUND4) or True else None
    assert bad.returncode == 1
    # the tool names its objdump from the toolchain, not from one fixed path
    assert lint.find_objdump().endswith("llvm-objdump")
