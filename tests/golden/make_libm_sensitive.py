#!/usr/bin/env python3
"""Generates tests/golden/libm_sensitive_args.json: arguments on which the libm generation the HIP kernels
reproduce (glibc 2.35, the one in this image; csrc/exact_math.h) does NOT return the correctly rounded result.
A libm that rounds correctly (the CORE-MATH routines newer glibc releases adopt for tanhf / log1pf / expm1f /
expf / logf) returns a different value on exactly such arguments, so these are the inputs on which a reference
built against another libm generation would part from the GPU path in the last ulp.  tests/test_libm_contract.py
evaluates the HOST libm on them and says so by name when it differs.

Method: random arguments in the ranges the decoder rules use; "correctly rounded" is taken as the double
(f32 functions) or long-double (f64 functions) result of the same libm rounded once to the narrower type --
double rounding can mislabel an argument only when the wide result lies within 2^-29 ulp of a rounding
boundary, which none of the chosen arguments does (checked below).  Run in the build image:
  python tests/golden/make_libm_sensitive.py
"""
import ctypes
import ctypes.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
libm = ctypes.CDLL(ctypes.util.find_library("m"))
libc = ctypes.CDLL(None)
libc.gnu_get_libc_version.restype = ctypes.c_char_p
PER_FUNCTION = 48

F32 = {"expf": (-20.0, 20.0), "logf": (1e-6, 40.0), "log1pf": (-0.99, 8.0), "expm1f": (-10.0, 10.0), "tanhf": (-9.0, 9.0)}
F64 = {"exp": (-40.0, 40.0), "log": (1e-12, 80.0), "log1p": (-0.99, 8.0), "expm1": (-20.0, 20.0), "tanh": (-18.0, 18.0)}


def main():
    rng = np.random.Generator(np.random.Philox(key=[2024, 0]))
    out = {"generated_with_glibc": libc.gnu_get_libc_version().decode(), "f32": {}, "f64": {}}
    for name, (lo, hi) in F32.items():
        fn = getattr(libm, name)
        fn.restype, fn.argtypes = ctypes.c_float, [ctypes.c_float]
        wide = getattr(np, name[:-1])
        rows = []
        while len(rows) < PER_FUNCTION:
            xs = rng.uniform(lo, hi, size=200000).astype(np.float32)
            got = np.array([fn(float(x)) for x in xs[:20000]], dtype=np.float32)
            w = wide(xs[:20000].astype(np.float64))
            cr = w.astype(np.float32)
            # distance of the double result from the midpoint between the two neighbouring floats, in float ulps
            ulp = np.spacing(np.abs(cr))
            mid = np.abs(np.abs(w - cr.astype(np.float64)) - 0.5 * ulp) / ulp
            for x, g, c, m in zip(xs[:20000], got, cr, mid):
                if g.view(np.uint32) != c.view(np.uint32) and m > 1e-6 and np.isfinite(g) and len(rows) < PER_FUNCTION:
                    rows.append([int(x.view(np.uint32)), int(g.view(np.uint32))])
        out["f32"][name] = rows
    for name, (lo, hi) in F64.items():
        fn = getattr(libm, name)
        fn.restype, fn.argtypes = ctypes.c_double, [ctypes.c_double]
        wide = getattr(np, name)
        rows = []
        tries = 0
        while len(rows) < PER_FUNCTION and tries < 40:
            tries += 1
            xs = rng.uniform(lo, hi, size=20000)
            got = np.array([fn(float(x)) for x in xs])
            w = wide(xs.astype(np.longdouble))
            cr = w.astype(np.float64)
            ulp = np.spacing(np.abs(cr))
            mid = np.abs(np.abs(w - cr.astype(np.longdouble)) - 0.5 * ulp) / ulp
            for x, g, c, m in zip(xs, got, cr, mid):
                if g.view(np.uint64) != c.view(np.uint64) and m > 1e-2 and np.isfinite(g) and len(rows) < PER_FUNCTION:
                    rows.append([int(np.float64(x).view(np.uint64)), int(np.float64(g).view(np.uint64))])
        out["f64"][name] = rows
    with open(os.path.join(HERE, "libm_sensitive_args.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print({k: {n: len(v) for n, v in out[k].items()} for k in ("f32", "f64")}, out["generated_with_glibc"])


if __name__ == "__main__":
    main()
