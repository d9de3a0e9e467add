#!/usr/bin/env python3
"""Prints how far the HIP path is from the CPU oracle for every implementation (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob
from frames import alist, awgn_frames

import ldpc_toolbox_amd as lt


def main():
    cases = [("ar4ja:1/2:1024", "1,1,1,1,0", 1.6), ("nr5g:2:24", "", 1.2)]
    batch, max_iter = 256, 20
    for spec, punct, ebn0 in cases:
        msgs, llrs, full = awgn_frames(spec, batch, ebn0, 5, punct)
        g = ob.Graph(alist(spec))
        for impl in lt.IMPLEMENTATIONS:
            dec = lt.LdpcDecoder(alist(spec), impl, punct)
            gin = llrs.astype(np.float64) if impl.endswith("f64") else llrs
            bits, its, post = dec.decode_batch(gin, max_iter, want_posterior=True)
            obits, oits, opost = ob.decode_batch(g, impl, full, max_iter, threads=8)
            # the reference's own f32-vs-f64 spread, for scale
            impl64 = impl[:-3] + "f64"
            _, its64, post64 = ob.decode_batch(g, impl64, full, max_iter, threads=8)
            same = its == oits
            conv = same & (its >= 0)
            scale = np.maximum(np.abs(opost[conv]), 1.0)
            err = np.abs(post[conv].astype(np.float64) - opost[conv]) / scale
            c64 = conv & (its64 == oits)
            e64 = np.abs(opost[c64] - post64[c64]) / np.maximum(np.abs(post64[c64]), 1.0)
            q = lambda e, p: float(np.quantile(e, p)) if e.size else float("nan")
            print(f"{spec:16s} {impl:20s} same-iter {same.mean():.3f} bitmis {(bits[same] != obits[same]).sum():4d} "
                  f"gpu-vs-oracle med {q(err, .5):.1e} p99 {q(err, .99):.1e} p99.9 {q(err, .999):.1e} max {q(err, 1):.1e} | "
                  f"oracle f32-vs-f64 med {q(e64, .5):.1e} p99.9 {q(e64, .999):.1e} max {q(e64, 1):.1e}", flush=True)


if __name__ == "__main__":
    main()
