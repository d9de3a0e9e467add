#!/usr/bin/env python3
"""Bisecting aid: the small-batch path against the batched kernels on one code, many calls."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames
spec = sys.argv[1]
msgs, llrs, _ = awgn_frames(spec, 19, float(sys.argv[2]), 77)
dec = lt.LdpcDecoder(alist(spec), "Minsumf32")
dec.set("latency", 0)
want = dec.decode_batch(llrs, 25, want_posterior=True)
bad = 0
for latency, sizes in ((8, (1, 3, 8)), (32, (11, 19))):
    dec.set("latency", latency)
    for B in sizes:
        for rep in range(3):
            got = dec.decode_batch(llrs[:B], 25, want_posterior=True)
            ok = all(np.array_equal(a, b[:B]) for a, b in zip(got, want))
            bad += not ok
            print(spec, "latency", latency, "B", B, "rep", rep, "ok" if ok else f"MISMATCH its {got[1]} vs {want[1][:B]}", flush=True)
print("bad", bad)
