#!/usr/bin/env python3
"""Batches between the small-batch path's default limit (64) and the sizes where the batched kernels fill the chip:
time per call by default, through the lane-per-edge path ("latency_edge" raised: further rounds inside one launch) and
through the batched kernels.   python3 tools/midbatch_probe.py [max_iterations]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

MAXIT = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for spec, impl, ebn0 in (("nr5g:1:384", "HLTanhf32", 1.5), ("nr5g:1:384", "HLMinsumf32", 1.5), ("dvbs2:R1_2", "Phif64", 2.0),
                         ("nr5g:1:384", "HLMinstarapproxi8", 1.5), ("nr5g:2:96", "HLTanhf32", 2.0)):
    msgs, llrs, _ = awgn_frames(spec, 512, ebn0, 3)
    dec = lt.LdpcDecoder(alist(spec), impl)
    line = f"{spec} {impl} Eb/N0 {ebn0}:"
    for name, edge in (("default", -1), ("lane-per-edge forced", 512), ("batched", 0)):
        if edge >= 0:
            dec.set("latency", 64 if edge else 0)
            dec.set("latency_edge", edge)
        line += f"\n   [{name}]"
        for B in (64, 96, 128, 192, 256, 384, 512):
            dec.decode_batch(llrs[:B], MAXIT, output_len=dec.k)
            t0 = time.perf_counter()
            for _ in range(3):
                dec.decode_batch(llrs[:B], MAXIT, output_len=dec.k)
            line += f" {B}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms ({dec.get('last_group')})"
    print(line, flush=True)
