#!/usr/bin/env python3
"""BASELINE.json config 5 on one GPU: every DVB-S2 normal-frame rate x 8 Eb/N0 points, flooding
min-sum f32, 50 iterations, syndrome early termination, frames generated and scored on the device.
Grid per SURVEY.md section 8(d): a coarse pre-scan (0.1 dB steps, one group of frames each) finds where the code's
frame error rate falls through 0.5; the 8 points are spaced 0.1 dB around that waterfall (three below, the crossing,
four above), each run to the reference's stop rule (100 frame errors, /root/reference/src/simulation/ber.rs:522-531)
or the frame cap."""
import argparse
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import ldpc_toolbox_amd as lt
from ldpc_toolbox_amd import ber, simulation as sim

ap = argparse.ArgumentParser()
ap.add_argument("--frame-errors", type=int, default=100, help="the reference's stop rule (src/cli/ber.rs:68-70)")
ap.add_argument("--max-frames", type=int, default=1 << 20, help="cap per Eb/N0 point (a point without errors ends here)")
ap.add_argument("--batch", type=int, default=16384, help="frames per simulator call")
ap.add_argument("--prescan", type=int, default=4096, help="frames per pre-scan point")
ap.add_argument("--rates", default="R1_4,R1_3,R2_5,R1_2,R3_5,R2_3,R3_4,R4_5,R5_6,R8_9,R9_10")
a = ap.parse_args()
RATES = a.rates.split(",")
print(f"# stop rule: {a.frame_errors} frame errors per point, at most {a.max_frames} frames per point; {a.batch} frames per call")
t_all = time.perf_counter()
frames_all = 0
points = []
for code in RATES:
    alist = lt.code_alist("dvbs2:" + code)
    s = lt.Simulator(alist, "Minsumf32", "", device=0, pool_size=32, pool_seed=1)
    r = s.rate
    shannon = 10 * np.log10((2 ** (2 * r) - 1) / (2 * r))
    # pre-scan: first 0.1 dB step above the Shannon limit at which fewer than half of one group's frames fail
    cross = None
    for i in range(40):
        e = round(shannon + 0.3 + 0.1 * i, 1)
        c = s.run(e, ber.point_seed(99, e), 0, a.prescan, 50)
        if c[2] * 2 < c[0]:
            cross = e
            break
    if cross is None:
        raise SystemExit(f"no waterfall found for {code}")
    grid = ber.ebn0_grid(cross - 0.3, cross + 0.4 + 1e-6, 0.1)
    points.append((code, cross, grid))
    print(f"# DVB-S2 {code}: n={s.n} k={s.k} rate {r:.4f} (BPSK Shannon limit {shannon:.2f} dB)")
    print(sim.format_header())
    res = ber.sweep(s, grid, max_iterations=50, max_frame_errors=a.frame_errors, max_frames=a.max_frames, frames_per_batch=a.batch, seed=7)
    for st in res:
        print(sim.format_progress(st), flush=True)
        frames_all += st.num_frames
    s.close()
print("# grid (Eb/N0 in dB; pre-scan crossing of FER = 0.5 in brackets):")
for code, cross, grid in points:
    print(f"#   {code} [{cross:.1f}]: " + " ".join(f"{e:.1f}" for e in grid))
print(f"# {frames_all} frames in {time.perf_counter() - t_all:.1f} s wall (incl. graph setup and encoder construction)")
