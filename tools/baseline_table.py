#!/usr/bin/env python3
"""The measured rows of BASELINE.md section 3 (GPU box): for BASELINE.json's configurations the CPU oracle
(C restatement of the reference decoder with its data structures, `oracle/`) with ONE thread and with the best
worker count, next to the GPU path on the same seeded frames.  Prints markdown rows.
  python3 tools/baseline_table.py > gpurun_out/r03_baseline_table.md"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import ldpc_toolbox_amd as lt
import oracle_binding as ob
from ldpc_toolbox_amd import sharding, simulation as sim

CORES = os.cpu_count() or 1


def cpu_rates(g, impl, llrs, iters, budget_s=12.0):
    """-> (single-thread cw/s, best multi-thread cw/s, its worker count, bits, its of the multi-thread run)"""
    _, _, dt1 = ob.decode_batch_timed(g, impl, llrs[:1], iters, threads=1)
    n1 = int(max(1, min(len(llrs), budget_s / 2 / max(dt1, 1e-4))))
    _, _, dt = ob.decode_batch_timed(g, impl, llrs[:n1], iters, threads=1)
    single = n1 / dt
    best = None
    for threads in sorted({max(1, CORES // 2), CORES}):
        count = min(len(llrs), threads * max(1, int(budget_s / 4 * single / 4)))   # a few seconds each
        count = max(count, min(len(llrs), threads))
        bits, its, dt = ob.decode_batch_timed(g, impl, llrs[:count], iters, threads=threads)
        if best is None or count / dt > best[0]:
            best = (count / dt, threads, bits, its, count)
    return single, best


def gpu_rate(dec, llrs, iters, reps=3):
    d = torch.from_numpy(llrs).cuda()
    B = len(llrs)
    bits = torch.zeros((B, dec.n), dtype=torch.uint8, device="cuda")
    its = torch.zeros(B, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    best = None
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec.decode_batch_device(d.data_ptr(), False, B, iters, bits.data_ptr(), dec.n, its.data_ptr(), 0, s.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return B / best, bits.cpu().numpy(), its.cpu().numpy()


def frames_for(spec, impl, batch, ebn0, seed, punct=""):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from frames import awgn_frames
    return awgn_frames(spec, batch, ebn0, seed, punct)


def main():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    h = bench.host_cpu_info()
    print(f"<!-- tools/baseline_table.py on a box with {h['sockets']} x {h['cpu_model']}: {h['physical_cores']} physical cores, "
          f"{h['hardware_threads']} hardware threads -->")
    print("| config (BASELINE.json) | CPU oracle, 1 thread | CPU oracle, best worker count | GPU (1 MI355X) | same output |")
    print("|---|---|---|---|---|")
    # config 1: AR4JA r=1/2 k=1024, BER at 2 dB through the simulation driver: identical counters CPU vs GPU
    spec, punct = "ar4ja:1/2:1024", "1,1,1,1,0"
    s = lt.Simulator(lt.code_alist(spec), "Minsumf32", punct, device=0, pool_size=64, pool_seed=1)
    frames = 1 << 17
    t0 = time.perf_counter()
    got = s.run(2.0, seed=7, first_frame=0, frames=frames, max_iterations=50)
    t_gpu = time.perf_counter() - t0
    msgs, tx = s.pool_data()
    g = ob.Graph(lt.code_alist(spec))
    llrs, idx = ob.generate_llrs(tx, s.rate, 2.0, 7, 0, frames)
    full = sim.depuncture(llrs, sim.parse_puncturing_pattern(punct))
    t0 = time.perf_counter()
    bits, its, _ = ob.decode_batch(g, "Minsumf32", full, 50, threads=CORES, want_posterior=False)
    t_cpu = time.perf_counter() - t0
    _, _, dt1 = ob.decode_batch_timed(g, "Minsumf32", full[:2048], 50, threads=1)
    st = sim.fold_statistics(2.0, s.k, msgs[idx], bits, its, 50, t_cpu)
    same = bool(np.array_equal(got, sharding.counters_from_statistics(st)))
    print(f"| 1. AR4JA r=1/2 k=1024 (puncturing `{punct}`), flooding min-sum, 50 it, BER @ Eb/N0 = 2 dB, {frames} frames | "
          f"{2048 / dt1:.0f} cw/s | {frames / t_cpu:.0f} cw/s ({CORES} workers); BER {st.ldpc.ber:.3e}, FER {st.ldpc.fer:.3e}, "
          f"{st.ldpc.frame_errors} frame errors, avg {st.average_iterations:.2f} it | {frames / t_gpu:.0f} cw/s (frames generated and scored "
          f"on the device); counters {[int(x) for x in got]} | {'identical counters' if same else 'DIFFERENT'} |")
    s.close()
    # configs 2 and 3: fixed work (P1) and realistic (P2)
    for label, spec, impl, batch, p1, p2 in (("2. DVB-S2 n=64800 r=1/2, flooding `Minsumf32`, 50 it, batch 4096", "dvbs2:R1_2", "Minsumf32", 4096, 0.0, 2.0),
                                             ("3. 5G NR BG1 Zc=384, `HLTanhf32`, 50 it, batch 8192", "nr5g:1:384", "HLTanhf32", 8192, -2.0, 2.0)):
        dec = lt.LdpcDecoder(lt.code_alist(spec), impl, device=0)
        dec.set("throttle", 1)     # (every call below is followed by a synchronisation: the calls may pace themselves)
        g = ob.Graph(lt.code_alist(spec))
        for pname, ebn0 in (("P1 fixed work", p1), ("P2 early termination", p2)):
            msgs, llrs, _ = frames_for(spec, impl, batch, ebn0, 11)
            rate, gbits, gits = gpu_rate(dec, llrs, 50)
            single, (multi, threads, obits, oits, count) = cpu_rates(g, impl, llrs, 50)
            same = bool(np.array_equal(obits, gbits[:count]) and np.array_equal(oits, gits[:count]))
            avg = (int((gits < 0).sum()) * 50 + int(gits[gits >= 0].sum())) / batch
            print(f"| {label}; {pname} (Eb/N0 = {ebn0} dB, avg {avg:.1f} it) | {single:.2f} cw/s | {multi:.1f} cw/s ({threads} workers) | "
                  f"{rate:.0f} cw/s = {rate * dec.k / 1e6:.0f} Mbit/s info | {'identical' if same else 'DIFFERENT'} on the {count} CPU-decoded frames |", flush=True)
        dec.close()


if __name__ == "__main__":
    main()
