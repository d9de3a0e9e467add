"""CPU: csrc/exact_math.h (the glibc-identical f32 exp/log/log1p/expm1/tanh the HIP kernels use)
against the host libm.  The exhaustive 2^32 sweep is tools/check_exact_math.cpp (25 s on 8
cores); here a dense sample + the special values keep the default CPU suite short."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include "%s/ldpc_toolbox_amd/csrc/exact_math.h"
using namespace ldpc::em;
int main() {
  // every 97th bit pattern (44 M arguments per function) plus all patterns near the special
  // exponents; all NaNs count as equal
  unsigned long long bad = 0, n = 0;
  for (unsigned long long i = 0; i < (1ull << 32); i += 97) {
    float x = as_f32((uint32_t)i);
    float a[5] = {ldpc::em::expf(x), ldpc::em::logf(x), ldpc::em::log1pf(x), ldpc::em::expm1f(x), ldpc::em::tanhf(x)};
    float b[5] = {::expf(x), ::logf(x), ::log1pf(x), ::expm1f(x), ::tanhf(x)};
    for (int k = 0; k < 5; k++) { n++; if (as_u32(a[k]) != as_u32(b[k]) && !(a[k] != a[k] && b[k] != b[k])) bad++; }
    // the Tanh rule's clamped form and Rust's atanh on their domains
    if (fabsf(x) <= 9.0f) { n++; if (as_u32(ldpc::em::tanhf_c9(x)) != as_u32(::tanhf(x))) bad++; }
    if (fabsf(x) < 1.0f) { n++; if (as_u32(ldpc::em::atanh_rs(x)) != as_u32(0.5f * ::log1pf((2.0f * x) / (1.0f - x)))) bad++; }
    { float a = fabsf(x), m = ldpc::em::corrf(a), w = ::log1pf(::expf(-a)); n++; if (as_u32(m) != as_u32(w) && !(m != m && w != w)) bad++; }
    { float m = ldpc::em::phif(x), w = -(::logf(::tanhf(0.5f * ::fmaxf(x, 1e-30f)))); n++; if (as_u32(m) != as_u32(w) && !(m != m && w != w)) bad++; }
  }
  printf("%%llu %%llu\n", bad, n);
  return 0;
}
"""


SRC64 = r"""
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include "%s/ldpc_toolbox_amd/csrc/exact_math.h"
using namespace ldpc::em;
static uint64_t s = 88172645463325252ull;
static uint64_t nxt() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main() {
  unsigned long long bad = 0, n = 0;
  for (long i = 0; i < 6000000; i++) {
    uint64_t r = nxt();
    double xs[4] = {as_f64(r), (double)(int64_t)(r >> 11) * 0x1p-53 * 64.0 - 32.0, -(double)(r >> 11) * 0x1p-53 * 800.0,
                    1.0 + ((double)(int64_t)(r >> 11) * 0x1p-53 - 0.5) * 0.25};
    for (int j = 0; j < 4; j++) {
      double x = xs[j];
      double a[5] = {ldpc::em::exp(x), ldpc::em::log(x), ldpc::em::log1p(x), ldpc::em::expm1(x), ldpc::em::tanh(x)};
      double b[5] = {::exp(x), ::log(x), ::log1p(x), ::expm1(x), ::tanh(x)};
      for (int k = 0; k < 5; k++) { n++; if (as_u64(a[k]) != as_u64(b[k]) && !(a[k] != a[k] && b[k] != b[k])) bad++; }
    }
  }
  printf("%%llu %%llu\n", bad, n);
  return 0;
}
"""


def test_exact_math_f64_matches_host_libm(tmp_path):
    """double precision: 120 M sampled results (tools/check_exact_math64.cpp ran 2e9 per function)"""
    src = tmp_path / "t64.cpp"
    src.write_text(SRC64 % ROOT)
    exe = tmp_path / "t64"
    subprocess.run(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    bad, n = int(out[0]), int(out[1])
    assert n == 120_000_000 and bad == 0, f"{bad} of {n} results differ from the host libm"


def test_exact_math_matches_host_libm(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC % ROOT)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    bad, n = int(out[0]), int(out[1])
    assert n > 200_000_000
    assert bad == 0, f"{bad} of {n} results differ from the host libm"


@pytest.mark.gpu
def test_device_functions_equal_glibc_on_every_float():
    """tools/check_exact_math_device.hip: the DEVICE build of exact_math.h against this box's glibc over
    all 2^32 float arguments, every function (about 6 s each on the GPU box).  This is what licenses the
    short division sequences of exact_math.h (fdiv_v) and the branch-free tanhf_c9."""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    exe = os.path.join(root, "tools", "mb", "check_device")
    src = os.path.join(root, "tools", "check_exact_math_device.hip")
    hdr = os.path.join(root, "ldpc_toolbox_amd", "csrc", "exact_math.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-pthread",
                        os.path.join(root, "tools", "check_exact_math_device.hip"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("2^32 arguments: 0 mismatches") == 11 and "mismatches   e.g." not in r.stdout, r.stdout
