#!/bin/bash
# disassembles the gfx950 kernels of a hipcc-built binary into /tmp/co/<name>.s:  tools/mb/disasm.sh <binary> [kernel-name substring]
mkdir -p /tmp/co && cd /tmp/co
python3 - "$1" <<'PY'
import struct, sys
data = open(sys.argv[1], 'rb').read()
pos = 0
while True:
    i = data.find(b'\x7fELF', pos)
    if i < 0: break
    if struct.unpack_from('<H', data, i + 18)[0] == 224:
        shoff = struct.unpack_from('<Q', data, i + 40)[0]; shentsize, shnum = struct.unpack_from('<HH', data, i + 58)
        open('/tmp/co/_img.elf', 'wb').write(data[i:i + shoff + shentsize * shnum])
    pos = i + 4
PY
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn _img.elf | sed 's#//.*##' > all.s
if [ -n "$2" ]; then awk -v pat="$2" '/^[0-9a-f]+ </{f = index($0, pat) > 0} f' all.s > kernel.s; wc -l kernel.s; fi
