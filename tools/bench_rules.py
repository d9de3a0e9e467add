#!/usr/bin/env python3
"""Secondary measurements (GPU box): every rule x schedule on BASELINE.json's configurations,
with the decode path's roofline fraction and the CPU oracle timed beside it.  Writes a table to
stdout (committed as profiles/r03_rules_table.txt; earlier rounds: r01_, r02_rules_table.txt).  Fixed work: Eb/N0 far below threshold so that
every frame runs all iterations (asserted)."""
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

CASES = [
    # (spec, batch, sigma, implementations, iterations)
    ("dvbs2:R1_2", 4096, 1.0, ["Minsumf32", "Minstarapproxf32", "Aminstarf32", "Phif32", "Tanhf32", "Minsumf64", "Phif64",
                                "Minstarapproxi8", "Aminstari8JonesPartialHardLimitDeg1Clip"], 10),
    ("nr5g:1:384", 8192, 1.8, ["HLTanhf32", "HLMinsumf32", "HLPhif32", "HLMinstarapproxf32", "HLAminstarf32",
                               "Tanhf32", "Minsumf32", "HLMinstarapproxi8", "HLAminstari8"], 10),
    ("ar4ja:1/2:1024", 65536, 1.3, ["Minsumf32", "HLMinsumf32", "Tanhf32"], 10),   # (a small graph: the default group is 65536)
]


def main():
    dev = torch.device("cuda:0")
    # one worker per physical core (bench.py's sweep: twice as many workers are slower on the 256-thread box);
    # decoders are built before the clock starts (oracle_decode_batch_timed_f32)
    threads = max(1, (os.cpu_count() or 2) // 2)
    print(f"{'code':16s} {'implementation':20s} {'batch':>6s} {'it':>3s} {'GPU cw/s':>11s} {'alg GB/s':>9s} {'frac':>6s} "
          f"{'kernel us (cn/vn | layer)':>28s} {'CPU cw/s':>9s} {'thr':>4s} {'GPU/CPU':>8s} {'same':>5s}")
    for spec, batch, sigma, impls, iters in CASES:
        alist = lt.code_alist(spec)
        g = ob.Graph(alist)
        for impl in impls:
            if os.environ.get("BENCH_RULES_ONLY", "") not in impl:  # e.g. BENCH_RULES_ONLY=i8: a substring filter
                continue
            dec = lt.LdpcDecoder(alist, impl, device=0)
            n, E = dec.n, dec.edges
            layered = impl.startswith("HL")
            f64 = impl.endswith("f64")
            elem = 8 if f64 else (1 if "i8" in impl else 4)
            bytes_cw_iter = ((4 * E + n) if layered else (4 * E + 2 * n)) * elem
            gen = torch.Generator(device=dev).manual_seed(1)
            llrs = (2.0 / sigma ** 2) * (1.0 + sigma * torch.randn((batch, n), generator=gen, device=dev))
            if f64:
                llrs = llrs.double()
            bits = torch.zeros((batch, n), dtype=torch.uint8, device=dev)
            its = torch.zeros(batch, dtype=torch.int32, device=dev)
            stream = torch.cuda.current_stream(dev)
            best = None
            for rep in range(4):
                # rep 1 is profiled (hipEvents around every launch: slows launch-bound sequences),
                # reps 2-3 are timed without events
                dec.set("profiling", 1 if rep == 1 else 0)
                dec.kernel_stats(0, reset=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dec.decode_batch_device(llrs.data_ptr(), f64, batch, iters, bits.data_ptr(), n, its.data_ptr(), 0,
                                        stream.cuda_stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if rep == 1:
                    stats = [dec.kernel_stats(k) for k in range(3)]
                if rep >= 2 and (best is None or dt < best):
                    best = dt
            dt = best
            it_np = its.cpu().numpy()
            assert (it_np == -1).all(), f"{spec} {impl}: not a fixed-work point"
            gbs = batch * iters * bytes_cw_iter / dt / 1e9
            kern = " / ".join(f"{ms / cnt * 1e3:.0f}" for cnt, ms in stats if cnt)
            if os.environ.get("BENCH_RULES_NO_CPU") == "1":   # A/B of two builds on one box: the GPU columns only
                print(f"{spec:16s} {impl:20s} {batch:6d} {iters:3d} {batch / dt:11.0f} {gbs:9.0f} {gbs / 8000:6.3f} {kern:>28s}", flush=True)
                dec.close()
                continue
            # CPU oracle on a bounded sample of the same frames
            sample = min(batch, threads)
            host = llrs[:sample].float().cpu().numpy()
            obits, oits, cdt = ob.decode_batch_timed(g, impl, host, iters, threads=threads)
            if cdt < 3.0 and sample == threads:      # enlarge the sample to a few seconds
                reps = int(min(max(1, 4.0 / cdt), batch // threads))
                host = llrs[:threads * reps].float().cpu().numpy()
                obits, oits, cdt = ob.decode_batch_timed(g, impl, host, iters, threads=threads)
            same = bool(np.array_equal(obits, bits[:len(obits)].cpu().numpy()) and np.array_equal(oits, it_np[:len(oits)]))
            cpu = len(host) / cdt
            print(f"{spec:16s} {impl:20s} {batch:6d} {iters:3d} {batch / dt:11.0f} {gbs:9.0f} {gbs / 8000:6.3f} "
                  f"{kern:>28s} {cpu:9.1f} {threads:4d} {batch / dt / cpu:8.0f} {str(same):>5s}", flush=True)
            dec.close()


if __name__ == "__main__":
    main()
