#!/usr/bin/env python3
"""A few Simulator batches at one Eb/N0 (for rocprofv3 --kernel-trace --stats): where a simulated frame's fixed
cost goes.  python3 tools/sim_profile_probe.py [ebn0_db] [runs]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401
import ldpc_toolbox_amd as lt

ebn0 = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sim = lt.Simulator(lt.code_alist("dvbs2:R1_2"), "Minsumf32", device=0, pool_size=64, pool_seed=1)
sim.run(ebn0, 7, 0, 4096, 50)
t0 = time.perf_counter()
for r in range(runs):
    c = sim.run(ebn0, 7, (r + 1) * 4096, 4096, 50)
dt = time.perf_counter() - t0
print(f"{runs * 4096 / dt:.0f} frames/s, {dt / runs * 1e3:.2f} ms per batch, last counters {c.tolist()}")
