#!/usr/bin/env python3
"""Where the small-batch kernel's time goes (csrc/latency.hip.h): the scalar call on a frame that never
converges, 50 iterations, with parts of the kernel switched off ("lat_debug": results are wrong then)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames
L = lt._capi.lib()
spec = sys.argv[1] if len(sys.argv) > 1 else "dvbs2:R1_2"
msgs, llrs, _ = awgn_frames(spec, 8, 0.0, 3)
dec = lt.LdpcDecoder(alist(spec), "Minsumf32")
out = np.zeros(dec.k, dtype=np.uint8)
for name, dbg in (("full", 0), ("no CN", 1), ("no VN", 2), ("no CN, no VN (barriers only)", 3), ("CN without its stores", 4), ("VN stores to scratch", 8), ("CN without stores, no VN", 6)):
    dec.set("lat_debug", dbg)
    for maxit in (50, 10):
        ts = []
        for i in range(16):
            t0 = time.perf_counter()
            L.ldpc_toolbox_decoder_decode_f32(dec._h, out.ctypes.data, dec.k, llrs[i % 8].ctypes.data, llrs.shape[1], maxit)
            ts.append(time.perf_counter() - t0)
        print(f"{spec} {name:32s} max_iterations {maxit:3d}: median {np.median(ts[4:])*1e3:.3f} ms", flush=True)
