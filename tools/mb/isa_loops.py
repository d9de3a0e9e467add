#!/usr/bin/env python3
"""Static instruction mix of one kernel's loops and straight-line regions, from `llvm-objdump -d --symbolize-operands`.
  isa_loops.py <library.so> <mangled-name substring>
Lists every backward branch (a loop) with the VALU / SALU / LDS / VMEM instructions of its body, and the totals between the
kernel entry and its first loop -- what a wavefront pays once (profiles/r05_config3_salu.txt)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import store_hazard_scan as sh  # noqa: E402  (llvm-objdump's path)


def kernel_text(lib, pat):
    data = open(lib, "rb").read()
    pos, best = 0, None
    while True:
        i = data.find(b"\x7fELF", pos)
        if i < 0:
            break
        if struct.unpack_from("<H", data, i + 18)[0] == 224:
            shoff = struct.unpack_from("<Q", data, i + 40)[0]
            shentsize, shnum = struct.unpack_from("<HH", data, i + 58)
            img = data[i:i + shoff + shentsize * shnum]
            if pat.encode() in img:
                best = img
        pos = i + 4
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
        f.write(best)
    try:
        txt = subprocess.run([(sh.OBJDUMP or sh.find_objdump()), "-d", "--no-show-raw-insn", "--symbolize-operands", f.name], check=True,
                             capture_output=True, text=True).stdout
    finally:
        os.unlink(f.name)
    out, on = [], False
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m and not re.match(r"L\d+$", m.group(1)):
            if on:
                break
            on = pat in m.group(1)
            continue
        if on:
            out.append(ln)
    return out


def classify(t):
    op = t.split()[0]
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_barrier")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    return "vmem"


def main():
    lines = kernel_text(sys.argv[1], sys.argv[2])
    ins, labels = [], {}
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(L\d+)>:$", ln.strip())
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = ln.split("//")[0].strip()
        if t:
            ins.append(t)
    print(f"{len(ins)} instructions, {len(labels)} labels")

    def mix(a, b):
        c = {}
        for t in ins[a:b]:
            k = classify(t)
            c[k] = c.get(k, 0) + 1
        return " ".join(f"{k} {v}" for k, v in sorted(c.items()))

    loops = []
    for i, t in enumerate(ins):
        m = re.match(r"(s_cbranch_\w+|s_branch)\s+(L\d+)", t)
        if m and m.group(2) in labels and labels[m.group(2)] <= i:
            loops.append((labels[m.group(2)], i))
    first = min((a for a, _ in loops), default=len(ins))
    print(f"entry .. first loop [0:{first}]: {mix(0, first)}")
    for a, b in sorted(loops):
        print(f"loop [{a}:{b}] {b - a + 1} instructions: {mix(a, b + 1)}")
    hist = {}
    for t in ins:
        if classify(t) == "salu":
            hist[t.split()[0]] = hist.get(t.split()[0], 0) + 1
    print("static SALU opcodes:", ", ".join(f"{k} {v}" for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:14]))


if __name__ == "__main__":
    main()
