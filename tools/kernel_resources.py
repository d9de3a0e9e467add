#!/usr/bin/env python3
"""Register / LDS / scratch use of the kernels in the built product library, read from the gfx950 code objects'
metadata notes (no GPU needed).  usage: tools/kernel_resources.py [substring of the demangled kernel name ...]"""
import re
import struct
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ldpc_toolbox_amd", "lib", "libldpc_toolbox.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "/usr/bin/c++filt"


def images(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        i = data.find(b"\x7fELF", pos)
        if i < 0:
            return
        if struct.unpack_from("<H", data, i + 18)[0] == 224:  # EM_AMDGPU
            shoff = struct.unpack_from("<Q", data, i + 40)[0]
            shentsize, shnum = struct.unpack_from("<HH", data, i + 58)
            yield data[i:i + shoff + shentsize * shnum]
        pos = i + 4


def kernels(lib=None):
    """-> [(demangled name, vgpr, agpr, sgpr, spilled vgprs, scratch bytes, static lds bytes, max workgroup size)]"""
    lib = lib or os.environ.get("LDPC_LIB", LIB)
    rows = []
    for k, img in enumerate(images(lib)):
        tmp = f"/tmp/_kres_{os.getpid()}_{k}.elf"
        open(tmp, "wb").write(img)
        notes = subprocess.run([READELF, "--notes", tmp], capture_output=True, text=True).stdout
        os.unlink(tmp)
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            blk = ".agpr_count:" + blk
            get = lambda key: (re.search(r"\." + key + r":\s+(\S+)", blk) or [None, "?"])[1]
            rows.append((get("name"), get("vgpr_count"), get("agpr_count"), get("sgpr_count"), get("vgpr_spill_count"),
                         get("private_segment_fixed_size"), get("group_segment_fixed_size"), get("max_flat_workgroup_size")))
    names = subprocess.run([FILT], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    return [(nm,) + tuple(int(x) if x.isdigit() else -1 for x in r[1:]) for r, nm in zip(rows, names)]


def main():
    want = sys.argv[1:]
    rows = [(None,) + tuple(str(x) for x in k[1:]) + (k[0],) for k in kernels()]
    names = [r[-1] for r in rows]
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'spill':>5} {'scratch':>7} {'lds':>6} {'wgmax':>5}  kernel")
    for r, nm in zip(rows, names):
        if want and not all(w in nm for w in want):
            continue
        print(f"{r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>5} {r[5]:>7} {r[6]:>6} {r[7]:>5}  {nm[:150]}")


if __name__ == "__main__":
    main()
