#!/usr/bin/env python3
"""Simulation throughput (frames generated, decoded and scored on the device) with and without straggler pooling, at
waterfall points where a few frames per chunk are slow or fail.  Same counters either way (asserted).
  python3 tools/pooling_probe.py [max_iterations]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist

MAXIT = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for spec, impl, frames, pts in (("nr5g:1:384", "HLMinsumf32", 65536, (1.6, 1.8, 2.0)), ("nr5g:1:384", "HLTanhf32", 65536, (1.2, 1.4, 1.6)),
                                ("dvbs2:R1_2", "Minsumf32", 32768, (1.3, 1.4, 1.6)), ("dvbs2:R3_4", "Minsumf32", 32768, (2.5, 2.7)),
                                ("ar4ja:1/2:1024", "Minsumf32", 524288, (2.2, 2.6))):
    s = lt.Simulator(alist(spec), impl, "1,1,1,1,0" if spec.startswith("ar4ja") else "", device=0, pool_size=64, pool_seed=1)
    for e in pts:
        row = f"{spec} {impl} Eb/N0 {e}:"
        ref = None
        for pooling in (0, 1):
            s.set("pooling", pooling)
            s.run(e, seed=3, first_frame=0, frames=frames, max_iterations=MAXIT)
            t0 = time.perf_counter()
            got = s.run(e, seed=3, first_frame=0, frames=frames, max_iterations=MAXIT)
            dt = time.perf_counter() - t0
            if ref is None:
                ref = got
            assert np.array_equal(got, ref)
            row += f"  pooling={pooling}: {frames / dt:9.0f} frames/s ({s.get('pooled_frames')} pooled)"
        print(row + f"   FER {ref[2] / ref[0]:.2e}, avg iterations {ref[4] / ref[0]:.1f}", flush=True)
