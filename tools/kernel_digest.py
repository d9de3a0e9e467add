#!/usr/bin/env python3
"""Per-kernel digest of a built library's gfx950 code: {demangled kernel name: sha1 of its disassembly without addresses}.
Used to check that a refactoring of the host code / a split into translation units left every kernel's instructions as
they were:  tools/kernel_digest.py old.so new.so  -> kernels only in one, kernels whose code differs."""
import hashlib
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def digests(path):
    data = open(path, "rb").read()
    out, pos = {}, 0
    while True:
        i = data.find(b"\x7fELF", pos)
        if i < 0:
            break
        if struct.unpack_from("<H", data, i + 18)[0] == 224:
            shoff = struct.unpack_from("<Q", data, i + 40)[0]
            shentsize, shnum = struct.unpack_from("<HH", data, i + 58)
            with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
                f.write(data[i:i + shoff + shentsize * shnum])
            try:
                text = subprocess.run([OBJDUMP, "-d", "-C", "--no-show-raw-insn", "--no-leading-addr", f.name], check=True,
                                      capture_output=True, text=True).stdout
            finally:
                os.unlink(f.name)
            name, body = None, []
            for ln in text.splitlines() + ["<end>:"]:
                m = re.match(r"^(?:[0-9a-f]+ )?<(.*)>:$", ln)
                if m:
                    if name is not None and not name.startswith("L") :
                        # branch targets carry absolute addresses in comments: drop everything after "//"
                        lines = [re.sub(r"\s*//.*", "", b) for b in body]
                        # pc-relative distances to the constant tables (s_getpc_b64, then s_add_u32 sN, sN, <distance>)
                        # depend on where the linker put the kernel: not the kernel's code
                        for k in range(1, len(lines)):
                            if lines[k - 1].startswith("s_getpc_b64") and lines[k].startswith("s_add_u32"):
                                lines[k] = re.sub(r"0x[0-9a-f]+$", "<pc-relative>", lines[k])
                        while lines and lines[-1] in ("", "...", "s_nop 0", "s_code_end"):   # padding behind the kernel's last instruction
                            lines.pop()
                        h = hashlib.sha1("\n".join(lines).encode()).hexdigest()
                        out.setdefault(name, []).append(h)
                    name, body = m.group(1), []
                else:
                    body.append(ln.strip())
        pos = i + 4
    return out


if __name__ == "__main__":
    a, b = digests(sys.argv[1]), digests(sys.argv[2])
    only_a, only_b = sorted(set(a) - set(b)), sorted(set(b) - set(a))
    diff = sorted(k for k in set(a) & set(b) if sorted(a[k]) != sorted(b[k]))
    print(f"{sum(len(v) for v in a.values())} kernel bodies in {sys.argv[1]}, {sum(len(v) for v in b.values())} in {sys.argv[2]}")
    print(f"only in the first: {len(only_a)}; only in the second: {len(only_b)}; same name, different code: {len(diff)}")
    for k in (only_a + only_b + diff)[:40]:
        print("  ", k[:200])
    sys.exit(1 if (only_a or only_b or diff) else 0)
