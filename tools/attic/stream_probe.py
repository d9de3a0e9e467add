#!/usr/bin/env python3
"""Continuous batching (DeviceDecoder::decode_stream through ldpc_toolbox_sim_run) against drained batches of 4096
frames: frames/s over Eb/N0 on one code, the average iterations, and the fraction of the iteration-proportional
bound (the 50-iteration rate x 50 / average iterations) each reaches.
  python3 tools/stream_probe.py [code] [frames] [harvest periods, comma separated]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import ldpc_toolbox_amd as lt

code = sys.argv[1] if len(sys.argv) > 1 else "dvbs2:R1_2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
periods = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2]
MAXIT = 50
s = lt.Simulator(lt.code_alist(code), "Minsumf32", "", device=0, pool_size=32, pool_seed=1)
r = s.rate
shannon = 10 * np.log10((2 ** (2 * r) - 1) / (2 * r))
points = [round(shannon + d, 2) for d in (0.0, 1.5, 1.6, 1.8, 2.0, 2.3)]
base = None
for ebn0 in points:
    line = f"{code} Eb/N0 {ebn0:5.2f}:"
    ref = None
    for mode in ["drained"] + [f"stream/{p}" for p in periods]:
        if mode == "drained":
            s.set("streaming", 0)
        else:
            s.set("streaming", 1)
            s.set("stream_harvest", int(mode.split("/")[1]))
        s.run(ebn0, 3, 0, 8192, MAXIT)
        t0 = time.perf_counter()
        c = s.run(ebn0, 3, 0, frames, MAXIT)
        dt = time.perf_counter() - t0
        avg = c[4] / c[0]
        rate = frames / dt
        if base is None:
            base = rate * avg / MAXIT          # first point, drained: nothing converges -> the fixed-work rate
        same = "" if ref is None else ("  same counters" if np.array_equal(ref, c) else "  COUNTERS DIFFER")
        ref = c if ref is None else ref
        line += f"\n    {mode:10s} {rate:9.0f} frames/s  avg-it {avg:5.1f}  frame errors {int(c[2]):6d}  = {rate * avg / MAXIT / base:5.3f} of the bound{same}"
    print(line, flush=True)
