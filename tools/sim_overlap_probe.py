#!/usr/bin/env python3
"""Does a second simulator on the same GPU (its own handle, streams and host thread) fill the tail of the
first one's batches?  With syndrome early termination the last codewords of a batch run on a mostly empty chip;
a second batch in flight can use it.  Prints frames/s for 1 and 2 concurrent simulators.
  python3 tools/sim_overlap_probe.py [code] [implementation]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch  # noqa: F401  (first: tests/conftest.py)
import ldpc_toolbox_amd as lt

code = sys.argv[1] if len(sys.argv) > 1 else "dvbs2:R1_2"
impl = sys.argv[2] if len(sys.argv) > 2 else "Minsumf32"
alist = lt.code_alist(code)
B, RUNS, MAXIT = 4096, 12, 50
sims = [lt.Simulator(alist, impl, device=0, pool_size=64, pool_seed=1) for _ in range(3)]
for ebn0 in (0.0, 1.2, 1.5, 2.0, 3.0):
    line = f"{code} {impl} Eb/N0 {ebn0:4.1f} dB:"
    ref = None
    for workers in (1, 2, 3):
        for s in sims[:workers]:
            s.run(ebn0, 7, 0, B, MAXIT)  # warm
        totals = [np.zeros(6, dtype=np.int64) for _ in range(workers)]

        def work(w):
            for r in range(w, RUNS, workers):
                totals[w] += sims[w].run(ebn0, 7, r * B, B, MAXIT)

        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(w,)) for w in range(workers)]
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        tot = sum(totals)
        if ref is None:
            ref = tot
        assert np.array_equal(tot, ref), (tot, ref)  # same frames, same counters
        line += f"  {workers} sim: {RUNS * B / dt:9.0f} frames/s"
        if workers == 1:
            line += f" (avg iterations {tot[4] / tot[0]:.1f}, frame errors {tot[2]})"
    print(line, flush=True)
