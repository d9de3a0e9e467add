#!/usr/bin/env python3
"""Latency of the reference-style scalar call (ldpc_toolbox_decoder_decode_f32: one codeword per call,
host buffers in and out) and of small batches, with the small-batch path (csrc/latency.hip.h) and with the
batched kernels ("latency" = 0).  The C entry is called directly with preallocated buffers.
  python3 tools/scalar_probe.py [max_iterations]"""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

MAXIT = int(sys.argv[1]) if len(sys.argv) > 1 else 50
L = lt._capi.lib()
for spec, impl, ebn0 in (("dvbs2:R1_2", "Minsumf32", 2.0), ("dvbs2:R1_2", "Minsumf32", 0.0),
                         ("nr5g:1:384", "Minsumf32", 1.5), ("ar4ja:1/2:1024", "Minsumf32", 2.0)):
    msgs, llrs, _ = awgn_frames(spec, 64, ebn0, 3)
    dec = lt.LdpcDecoder(alist(spec), impl)
    out = np.zeros(dec.k, dtype=np.uint8)
    line = f"{spec} {impl} Eb/N0 {ebn0}:"
    for latency in (64, 0):
        dec.set("latency", latency)
        L.ldpc_toolbox_decoder_decode_f32(dec._h, out.ctypes.data, dec.k, llrs[0].ctypes.data, llrs.shape[1], MAXIT)
        its, ts = [], []
        for i in range(64):
            row = llrs[i]
            t0 = time.perf_counter()
            it = L.ldpc_toolbox_decoder_decode_f32(dec._h, out.ctypes.data, dec.k, row.ctypes.data, llrs.shape[1], MAXIT)
            ts.append(time.perf_counter() - t0)
            its.append(MAXIT if it < 0 else it)
        ts = np.array(ts) * 1e3
        line += f"  [latency<={latency}] scalar call mean {ts.mean():.3f} ms, median {np.median(ts):.3f}, max {ts.max():.3f} (avg iterations {np.mean(its):.1f})"
        for B in (8, 16, 32, 64):
            dec.decode_batch(llrs[:B], MAXIT, output_len=dec.k)
            t0 = time.perf_counter()
            for _ in range(5):
                dec.decode_batch(llrs[:B], MAXIT, output_len=dec.k)
            line += f"; batch {B}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms"
    print(line, flush=True)
