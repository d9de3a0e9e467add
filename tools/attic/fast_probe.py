#!/usr/bin/env python3
"""The opt-in "@fast" variants (native exp2 / log2 / rcp in the Tanh and Phi rules; NOT bit-identical, never the
default) against the exact implementations: throughput at fixed work on BASELINE config 3 (5G NR BG1 Zc=384, 8192
frames) and DVB-S2 1/2, and -- on the same frames at a waterfall point -- frame errors, bit errors, frames whose
iteration count or hard decisions differ, and the soft outputs' relative difference.
  python3 tools/fast_probe.py > profiles/r03_fast_variants.txt"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

def rate(dec, llrs, iters):
    d = torch.from_numpy(llrs).cuda(); B = len(llrs)
    bits = torch.zeros((B, dec.n), dtype=torch.uint8, device="cuda"); its = torch.zeros(B, dtype=torch.int32, device="cuda")
    post = torch.zeros((B, dec.n), dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream(); best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec.decode_batch_device(d.data_ptr(), False, B, iters, bits.data_ptr(), dec.n, its.data_ptr(), post.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return B / best, bits.cpu().numpy(), its.cpu().numpy(), post.cpu().numpy()

print("# exact vs @fast: fixed-work throughput (50 iterations, no frame converges) and agreement on a waterfall batch")
for spec, batch, p1, p2s, impls in (("nr5g:1:384", 8192, -2.0, (0.9, 0.7, 0.5, 0.3, 0.1), ("HLTanhf32", "HLPhif32", "Tanhf32", "Phif32")),
                                    ("dvbs2:R1_2", 4096, 0.0, (1.1, 1.0, 0.9, 0.8, 0.7), ("Tanhf32", "Phif32"))):
    E = None
    for impl in impls:
        de, df = lt.LdpcDecoder(alist(spec), impl), lt.LdpcDecoder(alist(spec), impl + "@fast")
        n, E = de.n, de.edges
        layered = impl.startswith("HL")
        bytes_it = ((4 * E + n) if layered else (4 * E + 2 * n)) * 4
        msgs, llrs, _ = awgn_frames(spec, batch, p1, 5)
        re_, _, ie, _ = rate(de, llrs, 50); rf_, _, if_, _ = rate(df, llrs, 50)
        assert (ie < 0).all() and (if_ < 0).all()
        for p2 in p2s:                                        # down to a waterfall point of this rule
            msgs, llrs, _ = awgn_frames(spec, batch, p2, 6)
            _, be, ie, pe = rate(de, llrs, 50)
            k = msgs.shape[1]
            fe_e = int((be[:, :k] != msgs).any(axis=1).sum())
            if fe_e >= 50:
                break
        _, bf, if_, pf = rate(df, llrs, 50)
        fe_f = int((bf[:, :k] != msgs).any(axis=1).sum())
        bit_e, bit_f = int((be[:, :k] != msgs).sum()), int((bf[:, :k] != msgs).sum())
        ok = (ie >= 0) & (if_ >= 0) & (ie == if_)
        rel = np.abs(pe[ok] - pf[ok]) / (np.abs(pe[ok]) + 1.0)
        print(f"{spec:11s} {impl:10s} exact {re_:8.0f} cw/s ({re_ * 50 * bytes_it / 8e12:.3f} of the roofline)   @fast {rf_:8.0f} cw/s "
              f"({rf_ * 50 * bytes_it / 8e12:.3f})  x{rf_ / re_:.2f} | Eb/N0 {p2} dB, {batch} frames: frame errors {fe_e} vs {fe_f}, bit errors {bit_e} vs {bit_f}, "
              f"frames with another iteration count {int((ie != if_).sum())}, with other hard decisions {int((be != bf).any(axis=1).sum())}, "
              f"soft outputs: median relative difference {np.median(rel):.1e}, 99th percentile {np.percentile(rel, 99):.1e}", flush=True)
        de.close(); df.close()
