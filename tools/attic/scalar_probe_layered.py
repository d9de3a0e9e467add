#!/usr/bin/env python3
"""Latency of the reference-style scalar call and of small batches for the LAYERED schedule, with the small-batch path
(csrc/latency_edge.hip.h) and with the batched kernels ("latency" = 0).
  python3 tools/scalar_probe_layered.py [max_iterations] [lat_grid] [i8]     (i8: the 8-bit implementations instead)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

MAXIT = int(sys.argv[1]) if len(sys.argv) > 1 else 50
GRIDS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
L = lt._capi.lib()
I8_CASES = (("dvbs2:R1_2", "Minstarapproxi8", 2.0), ("dvbs2:R1_2", "Aminstari8Jones", 2.0), ("nr5g:1:384", "HLMinstarapproxi8", 1.5),
            ("nr5g:1:384", "HLAminstari8PartialHardLimit", 1.5), ("nr5g:1:384", "Minstarapproxi8JonesPartialHardLimitDeg1Clip", 1.5),
            ("ar4ja:1/2:1024", "Aminstari8Deg1Clip", 2.0))
for spec, impl, ebn0 in I8_CASES if "i8" in sys.argv[3:] else (("nr5g:1:384", "HLTanhf32", 1.5), ("dvbs2:R1_2", "Phif64", 2.0), ("dvbs2:R1_2", "Tanhf32", 2.0), ("nr5g:1:384", "Tanhf32", 1.5), ("nr5g:1:384", "HLTanhf64", 1.5), ("nr5g:1:384", "HLMinsumf32", 1.5), ("nr5g:1:384", "HLAminstarf32", 1.5),
                         ("nr5g:1:384", "HLMinstarapproxf32", 1.5), ("nr5g:1:384", "HLPhif32", 1.5),
                         ("nr5g:2:24", "HLTanhf32", 2.0), ("ar4ja:1/2:1024", "HLMinsumf32", 2.0)):
    msgs, llrs, _ = awgn_frames(spec, 64, ebn0, 3)
    dec = lt.LdpcDecoder(alist(spec), impl)
    out = np.zeros(dec.k, dtype=np.uint8)
    line = f"{spec} {impl} Eb/N0 {ebn0}:"
    for latency, grid in [(int(os.environ.get("LDPC_PROBE_LATENCY", "32")), g) for g in GRIDS] + [(0, 0)]:
        dec.set("latency", latency)
        dec.set("lat_grid", grid)
        L.ldpc_toolbox_decoder_decode_f32(dec._h, out.ctypes.data, dec.k, llrs[0].ctypes.data, llrs.shape[1], MAXIT)
        its, ts = [], []
        for i in range(64):
            row = llrs[i]
            t0 = time.perf_counter()
            it = L.ldpc_toolbox_decoder_decode_f32(dec._h, out.ctypes.data, dec.k, row.ctypes.data, llrs.shape[1], MAXIT)
            ts.append(time.perf_counter() - t0)
            its.append(MAXIT if it < 0 else it)
        ts = np.array(ts) * 1e3
        line += f"\n   [latency<={latency} grid {grid}] scalar call mean {ts.mean():.3f} ms, median {np.median(ts):.3f}, max {ts.max():.3f} (avg iterations {np.mean(its):.1f})"
        for B in (8, 16, 32, 64):
            dec.decode_batch(llrs[:B], MAXIT, output_len=dec.k)
            t0 = time.perf_counter()
            for _ in range(5):
                dec.decode_batch(llrs[:B], MAXIT, output_len=dec.k)
            line += f"; batch {B}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms"
    print(line, flush=True)
