"""The libm contract of the bit-exact claim (include/ldpc_toolbox.h).

The transcendental decoder rules (/root/reference/src/decoder/arithmetic.rs:184, 357, 376, 510, 965-966) call
Rust's f32/f64::{exp, ln, ln_1p, tanh}, which on linux-gnu are the platform libm's expf / logf / log1pf / tanhf
(and the double versions).  csrc/exact_math.h reproduces, operation by operation, the glibc generation that
ships these as the Szabolcs-Nagy table routines (expf, logf: glibc 2.28+) and the fdlibm routines (log1pf,
expm1f, tanhf) -- glibc 2.28 ... 2.40 as far as we know.  Newer glibc releases replace some of them by
correctly rounded CORE-MATH routines; a reference (or this repo's oracle, which links the host libm) built
there differs from the GPU path in rare last ulps.  This test makes that situation say its name instead of
surfacing as a bare parity mismatch: it evaluates the HOST libm on committed arguments where the two
generations part (tests/golden/libm_sensitive_args.json: arguments on which the reproduced generation does
not round correctly, with its results) and fails with the host's glibc version when they disagree."""
import ctypes
import ctypes.util
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "libm_sensitive_args.json")


def _glibc_version():
    libc = ctypes.CDLL(None)
    try:
        libc.gnu_get_libc_version.restype = ctypes.c_char_p
        return libc.gnu_get_libc_version().decode()
    except AttributeError:
        return "not glibc"


def test_host_libm_is_the_generation_the_kernels_reproduce():
    data = json.load(open(GOLDEN))
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    wrong = []
    for name, rows in data["f32"].items():
        fn = getattr(libm, name)
        fn.restype, fn.argtypes = ctypes.c_float, [ctypes.c_float]
        for arg, want in rows:
            x = np.uint32(arg).view(np.float32)
            got = int(np.float32(fn(float(x))).view(np.uint32))
            if got != want:
                wrong.append(f"{name}({float(x)!r}) = 0x{got:08x}, the kernels reproduce 0x{want:08x}")
    for name, rows in data["f64"].items():
        fn = getattr(libm, name)
        fn.restype, fn.argtypes = ctypes.c_double, [ctypes.c_double]
        for arg, want in rows:
            x = np.uint64(arg).view(np.float64)
            got = int(np.float64(fn(float(x))).view(np.uint64))
            if got != want:
                wrong.append(f"{name}({float(x)!r}) = 0x{got:016x}, the kernels reproduce 0x{want:016x}")
    assert not wrong, (
        f"this host's libm (glibc {_glibc_version()}) is not the generation csrc/exact_math.h reproduces (glibc "
        f"{data['generated_with_glibc']}: Nagy expf/logf + fdlibm log1pf/expm1f/tanhf, glibc 2.28 .. 2.40): "
        f"{len(wrong)} of the committed sensitive arguments differ, e.g. {wrong[:3]}.  The GPU path stays "
        "bit-identical to a reference linked against that generation; on THIS host the oracle (and a Rust build of "
        "the reference) would differ from it in rare last ulps of the Phi / Tanh / Minstarapprox / Aminstar rules.")


def test_exact_math_header_returns_the_committed_results(tmp_path):
    """the same arguments through csrc/exact_math.h (host build): the committed results are its results"""
    data = json.load(open(GOLDEN))
    lines = []
    for prec, suffix, ctype, fmt in (("f32", "f", "uint32_t", "%u"), ("f64", "", "uint64_t", "%llu")):
        for name, rows in data[prec].items():
            for arg, want in rows:
                if prec == "f32":
                    lines.append(f"  n++; if (as_u32(ldpc::em::{name}(as_f32({arg}u))) != {want}u) bad++;")
                else:
                    lines.append(f"  n++; if (as_u64(ldpc::em::{name}(as_f64({arg}ull))) != {want}ull) bad++;")
    src = tmp_path / "c.cpp"
    src.write_text('#include <math.h>\n#include <stdio.h>\n#include <stdint.h>\n#include "%s/ldpc_toolbox_amd/csrc/exact_math.h"\n'
                   "using namespace ldpc::em;\nint main() {\n  unsigned n = 0, bad = 0;\n%s\n"
                   '  printf("%%u %%u\\n", bad, n);\n  return 0;\n}\n' % (ROOT, "\n".join(lines)))
    exe = tmp_path / "c"
    subprocess.run(["g++", "-O1", "-std=c++17", "-mfma", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"], check=True)
    bad, n = (int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    assert n >= 5 * 48 and bad == 0, f"{bad} of {n} committed results differ from csrc/exact_math.h"
