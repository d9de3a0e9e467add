#!/usr/bin/env python3
"""Experiment: chunks of 4096 frames decoded back to back on one handle, against the same chunks
alternated over two handles on two streams (each chunk starts when its handle's previous chunk is
done, so the two streams run half a period apart and one's shrinking tail overlaps the other's
full-width head)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

spec, B, chunks = "dvbs2:R1_2", 4096, 12
for ebn0 in (1.6, 2.0, 2.5, 0.0):
    msgs, llrs, _ = awgn_frames(spec, B, ebn0, 7)
    d = [torch.from_numpy(llrs).cuda(), torch.from_numpy(llrs[::-1].copy()).cuda()]
    decs = [lt.LdpcDecoder(alist(spec), "Minsumf32") for _ in range(2)]
    bits = [torch.zeros((B, decs[0].k), dtype=torch.uint8, device="cuda") for _ in range(2)]
    its = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def run(two):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for c in range(chunks):
            i = (c & 1) if two else 0
            decs[i].decode_batch_device(d[i].data_ptr(), False, B, 50, bits[i].data_ptr(), decs[i].k, its[i].data_ptr(), 0, streams[i].cuda_stream)
        torch.cuda.synchronize(); return time.perf_counter() - t0
    run(False); run(True)
    t1 = min(run(False) for _ in range(2)); t2 = min(run(True) for _ in range(2))
    print(f"Eb/N0 {ebn0}: one stream {t1*1e3/chunks:.1f} ms per chunk ({B*chunks/t1:.0f} cw/s); two staggered streams {t2*1e3/chunks:.1f} ms per chunk ({B*chunks/t2:.0f} cw/s)", flush=True)
