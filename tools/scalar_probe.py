#!/usr/bin/env python3
"""Latency of the reference-style scalar call (one codeword per call) and of small batches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

MAXIT = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for spec, impl, ebn0 in (("dvbs2:R1_2", "Minsumf32", 2.0), ("dvbs2:R1_2", "Minsumf32", 0.0),
                         ("nr5g:1:384", "HLTanhf32", 1.5), ("ar4ja:1/2:1024", "Phif64", 2.0)):
    msgs, llrs, _ = awgn_frames(spec, 64, ebn0, 3)
    dec = lt.LdpcDecoder(alist(spec), impl)
    dec.decode(llrs[0].astype(np.float64), MAXIT)
    t0 = time.perf_counter()
    its = []
    for i in range(64):
        ok, out = dec.decode(llrs[i].astype(np.float64), MAXIT)
        its.append(out.iterations)
    dt = (time.perf_counter() - t0) / 64
    line = f"{spec} {impl} Eb/N0 {ebn0}: scalar call {dt*1e3:.3f} ms (avg iterations {np.mean(its):.1f})"
    for B in (8, 64):
        dec.decode_batch(llrs[:B], MAXIT)
        t0 = time.perf_counter()
        for _ in range(5):
            dec.decode_batch(llrs[:B], MAXIT)
        line += f"; batch {B}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms"
    print(line, flush=True)
