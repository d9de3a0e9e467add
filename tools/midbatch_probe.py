#!/usr/bin/env python3
"""Batches between the small-batch paths and the sizes where the batched kernels fill the chip: time per call by
default, through the lane-per-edge path ("latency_edge" raised: further rounds inside one launch) and through the
batched kernels.  Round 4: the three modes are timed back to back per batch size, in rotating order, best of three
rounds of three calls (round 3 timed a whole "default" pass first, on a chip that had not reached its clocks and with
the staging buffers still growing: its default column was slower than both paths it chooses from).
   python3 tools/midbatch_probe.py [max_iterations]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

MAXIT = int(sys.argv[1]) if len(sys.argv) > 1 else 50
SIZES = (33, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024)
MODES = (("default", None), ("lane-per-edge forced", 1024), ("batched", 0))
worst = 0.0
for spec, impl, ebn0 in (("nr5g:1:384", "HLTanhf32", 1.5), ("nr5g:1:384", "HLMinsumf32", 1.5), ("dvbs2:R1_2", "Phif64", 2.0),
                         ("nr5g:1:384", "HLMinstarapproxi8", 1.5), ("nr5g:2:96", "HLTanhf32", 2.0), ("dvbs2:R1_2", "Tanhf32", 2.0),
                         ("ar4ja:1/2:1024", "HLPhif32", 2.0)):
    msgs, llrs, _ = awgn_frames(spec, max(SIZES), ebn0, 3)
    # ONE decoder handle for the three modes (two handles of one process differ by up to 20 % on these latency-bound
    # calls, whichever mode they run: where their state landed in memory), switched between the timed blocks
    dec = lt.LdpcDecoder(alist(spec), impl)
    dec.decode_batch(llrs[:max(SIZES)], MAXIT, output_len=dec.k)      # staging buffers at their final size, clocks up
    defaults = (32, 256)                                             # "latency", "latency_edge" as constructed

    def select(edge):
        if edge is None:
            dec.set("latency", defaults[0])
            dec.set("latency_edge", defaults[1])
        else:
            dec.set("latency", 64 if edge else 0)
            dec.set("latency_edge", edge)
    decs = {name: dec for name, _ in MODES}
    edge_of = dict(MODES)
    rows = {name: [] for name, _ in MODES}
    for B in SIZES:
        best = {name: float("inf") for name, _ in MODES}
        for rnd in range(3):
            order = [MODES[(i + rnd) % 3][0] for i in range(3)]
            for name in order:
                d = decs[name]
                select(edge_of[name])
                d.decode_batch(llrs[:B], MAXIT, output_len=d.k)
                t0 = time.perf_counter()
                for _ in range(3):
                    d.decode_batch(llrs[:B], MAXIT, output_len=d.k)
                best[name] = min(best[name], (time.perf_counter() - t0) / 3 * 1e3)
        for name, _ in MODES:
            rows[name].append((B, best[name], 0))
        ratio = best["default"] / min(best["lane-per-edge forced"], best["batched"])
        worst = max(worst, ratio)
    print(f"{spec} {impl} Eb/N0 {ebn0}:")
    for name, _ in MODES:
        print(f"   [{name}] " + " ".join(f"{B}: {ms:.2f} ms" for B, ms, g in rows[name]))
    print("   default / better of the two: " + " ".join(f"{B}: {rows['default'][i][1] / min(rows['lane-per-edge forced'][i][1], rows['batched'][i][1]):.2f}" for i, B in enumerate(SIZES)), flush=True)
print(f"# worst default / min(forced, batched) over all cells: {worst:.2f}")
