#!/usr/bin/env python3
"""Scans a gfx950 disassembly (llvm-objdump -d --no-show-raw-insn) for the pattern behind round 4's run-to-run
differences in cn_minsum_rec_kernel: a buffer/global/flat STORE of more than 64 bits whose data registers are
overwritten by a VALU instruction issued within the next `--window` instructions.  The GCN3/Vega ISA manuals list
"VMEM store of more than 64 bits followed by a VALU write of the VGPRs holding the write data: 1 wait state" and exempt
MUBUF stores that use an SGPR soffset; LLVM's GCNHazardRecognizer (createsVALUHazard) follows that exemption, so for the
`buffer_store_dwordx4 v[..], v, s[..], sN offen` form the library's buf_store emits, nothing is inserted.

Measured on MI355X (tools/mb/store_hazard_repro.hip, profiles/r05_store_hazard.txt): the SGPR-soffset form still needs ONE
wait state (0.5 % of the stores took the rewritten register in lanes 12-15 of every 16 with none), the literal-soffset form two.

  store_hazard_scan.py <file.s | library.so | binary> [--kernel substring] [--window N] [--check]
A .so / binary is taken apart here (every gfx code object in it, llvm-objdump).  Default: list every store followed within
--window issue slots by a VALU write of its data registers.  --check: exit 1 if any store has FEWER wait states than the
hardware needs (1 for MUBUF with an SGPR soffset, 2 otherwise) -- the build's lint (`make lint`, tests/test_isa_lint.py)."""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

def find_objdump():
    """llvm-objdump of the toolchain that built the library: $LLVM_OBJDUMP, next to $HIPCC / hipcc's clang, the ROCm
    default, or PATH"""
    import shutil
    cands = [os.environ.get("LLVM_OBJDUMP")]
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
    cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), "/opt/rocm/lib/llvm/bin/llvm-objdump",
              shutil.which("llvm-objdump")]
    for c in cands:
        if c and os.path.exists(c):
            return c
    sys.exit("store_hazard_scan.py: no llvm-objdump found (set LLVM_OBJDUMP=/path/to/llvm-objdump); the library was NOT checked")


OBJDUMP = None
STORE = re.compile(r"\b(buffer_store_dwordx[34]|buffer_store_format_xyzw?|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)")
VRANGE = re.compile(r"v\[(\d+):(\d+)\]")
VSINGLE = re.compile(r"\bv(\d+)\b")


def disassemble(path):
    """text of every gfx code object inside a host binary / shared library (or the file itself if it is one)"""
    global OBJDUMP
    OBJDUMP = OBJDUMP or find_objdump()
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(b"\x7fELF", pos)
        if i < 0:
            break
        if struct.unpack_from("<H", data, i + 18)[0] == 224:          # EM_AMDGPU
            shoff = struct.unpack_from("<Q", data, i + 40)[0]
            shentsize, shnum = struct.unpack_from("<HH", data, i + 58)
            with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
                f.write(data[i:i + shoff + shentsize * shnum])
            try:
                out.append(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], check=True, capture_output=True,
                                          text=True).stdout)
            finally:
                os.unlink(f.name)
        pos = i + 4
    return "\n".join(out)


def regs(tok):
    m = VRANGE.match(tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = VSINGLE.match(tok)
    return {int(m.group(1))} if m else set()


def dest_regs(ins):
    """VGPRs a VALU instruction writes (first operand of v_*; memory instructions are not VALU and do not count)"""
    parts = ins.split(None, 1)
    if len(parts) < 2 or not parts[0].startswith("v_") or parts[0].startswith(("v_cmp", "v_cmpx", "v_nop")):
        return set()
    return regs(parts[1].split(",")[0].strip())


def scan(text, kernel="", window=2, check=False, report=print):
    """-> (hits, stores seen).  A hit: VALU write of the store's data registers with fewer than `window` wait states
    (check: fewer than the hardware needs for that store's form) behind the store."""
    hits, stores, per = 0, 0, {}
    name, body = None, []

    def flush():
        nonlocal hits, stores
        if name is None or kernel not in name:
            return
        for i, ins in enumerate(body):
            m = STORE.search(ins)
            if not m:
                continue
            stores += 1
            ops = [o.strip() for o in m.group(2).split(",")]
            mubuf = m.group(1).startswith("buffer")
            data = regs(ops[0]) if mubuf else regs(ops[1])
            soffset_is_reg = mubuf and len(ops) >= 4 and re.match(r"(s\d+|s\[|m0|ttmp)", ops[3].split()[0]) is not None
            need = (1 if soffset_is_reg else 2) if check else window
            waited = 0
            for j in range(i + 1, min(len(body), i + 12)):
                nxt = body[j]
                if waited >= need:
                    break
                if nxt.startswith("s_nop"):
                    arg = nxt.split()[1] if len(nxt.split()) > 1 else "0"
                    waited += 1 + (int(arg, 0) if re.match(r"^(0x)?[0-9a-fA-F]+$", arg) else 0)
                    continue
                if nxt.startswith(("s_branch", "s_cbranch", "s_setpc", "s_endpgm")):
                    break                                   # (a taken branch costs more than the wait states in question)
                clobber = dest_regs(nxt) & data
                if clobber:
                    hits += 1
                    per[name] = per.get(name, 0) + 1
                    report(f"{name[:140]}\n   [{i}] {ins}\n   +{j - i} after {waited} wait state(s), needs {need}: {nxt}\n"
                           f"   rewrites v{sorted(clobber)}  (soffset in an SGPR: {soffset_is_reg})")
                    break
                waited += 1

    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            flush()
            name, body = m.group(1), []
        elif name is not None:
            t = ln.split("//")[0].strip()
            if t:
                body.append(t)
    flush()
    return hits, stores, per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--kernel", default="")
    ap.add_argument("--window", type=int, default=2)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--min-stores", type=int, default=1,
                    help="--check fails (exit 2) when fewer wide stores than this were seen: a lint that found nothing to look "
                         "at must not pass (the library's Makefile asks for 500)")
    a = ap.parse_args()
    text = open(a.path).read() if a.path.endswith(".s") else disassemble(a.path)
    if a.check and not a.path.endswith(".s") and not text.strip():
        # a check that saw nothing must not pass: a compressed offload bundle, another bundle layout, a host-only file
        print(f"store_hazard_scan.py: no gfx code object (EM_AMDGPU ELF) found in {a.path}: nothing was checked", file=sys.stderr)
        return 2
    hits, stores, per = scan(text, a.kernel, a.window, a.check, report=(lambda *_: None) if a.quiet else print)
    for k, v in per.items():
        print(f"{v:4d}  {k[:150]}")
    what = "with fewer wait states than gfx950 needs" if a.check else f"within {a.window} issue slot(s)"
    print(f"{stores} store(s) of more than 64 bits; {hits} whose data registers a VALU instruction rewrites {what}")
    if a.check and stores < a.min_stores and not a.kernel:
        print(f"store_hazard_scan.py: only {stores} wide store(s) seen, expected at least {a.min_stores} (--min-stores): the "
              "disassembly is not the library's kernels; nothing was checked", file=sys.stderr)
        return 2
    return 1 if (a.check and hits) else 0


if __name__ == "__main__":
    sys.exit(main())
