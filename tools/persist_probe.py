#!/usr/bin/env python3
"""Slice-persistent layered kernel (hl_slice_kernel) against the per-level launches: same outputs, then throughput of
BASELINE config 3 (5G NR BG1 Zc=384, HLTanhf32, 8192 frames, fixed work and +2 dB) for a list of execution choices.

  python tools/persist_probe.py [--parity-only] [--impl HLTanhf32] [--steps 3] [--iters 50]
"""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import ldpc_toolbox_amd as lt
from bench_config3 import frames


def parity(impls):
    from frames import awgn_frames
    for spec, nb, ebn0 in (("nr5g:1:16", 2304, 1.0), ("nr5g:2:24", 1100, 1.5), ("ar4ja:1/2:1024", 700, 1.8), ("nr5g:1:384", 640, 0.5)):
        for impl in impls:
            msgs, llrs, full = awgn_frames(spec, nb, ebn0, 777)
            dec = lt.LdpcDecoder(lt.code_alist(spec), impl)
            dec.set("latency", 0)
            x = llrs.astype(np.float64) if impl.endswith("f64") else llrs
            want = None
            for persist, width, lanes, group in ((0, 0, 1, 4096), (2, 32, 1, 4096), (2, 64, 1, 4096), (2, 32, 2, 512), (2, 64, 2, 1024)):
                dec.set("hl_persist", persist)
                dec.set("hl_slice", width)
                dec.set("lanes", lanes)
                dec.set("group_size", group)
                got = dec.decode_batch(x, 12, want_posterior=True)
                used = dec.get("last_persist")
                assert used == (width if persist and width == 32 else used) and used in (0, width), (spec, impl, persist, width, used)
                if want is None:
                    want = got
                else:
                    for a, b in zip(want, got):
                        assert np.array_equal(a, b), (spec, impl, persist, width, lanes, group)
            print(f"parity ok: {spec} {impl}: {int((want[1] >= 0).sum())}/{nb} decoded", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parity-only", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--impl", default="HLTanhf32")
    ap.add_argument("--spec", default="nr5g:1:384")
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--ebn0", type=float, default=-2.0)
    ap.add_argument("--configs", default="0:0:2:4096,2:32:1:8192,2:32:2:4096,2:64:1:8192,2:32:1:4096")
    a = ap.parse_args()
    if not a.no_parity:
        parity(["HLTanhf32", "HLTanhf32@fast"] if False else ["HLTanhf32"])
    if a.parity_only:
        return
    device = torch.device("cuda", 0)
    alist = lt.code_alist(a.spec)
    dec = lt.LdpcDecoder(alist, a.impl, device=0)
    enc = lt.Encoder(alist)
    B = a.batch
    bits = torch.zeros((B, dec.k), dtype=torch.uint8, device=device)
    its = torch.zeros(B, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream(device)
    E, n = dec.edges, dec.n
    bytes_cw_iter = (4 * E + n) * 4
    ref = {}
    for ebn0 in (a.ebn0, 2.0):
        msgs, llrs = frames(dec, enc, B, ebn0, 31, device)
        for cfg in a.configs.split(","):
            persist, width, lanes, group = (int(x) for x in cfg.split(":"))
            dec.set("hl_persist", persist)
            dec.set("hl_slice", width)
            dec.set("lanes", lanes)
            dec.set("group_size", group)
            dec.set("throttle", 1)

            def run():
                dec.decode_batch_device(llrs.data_ptr(), False, B, a.iters, bits.data_ptr(), dec.k, its.data_ptr(), 0,
                                        stream.cuda_stream)
            run()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                run()
            torch.cuda.synchronize(device)
            el = (time.perf_counter() - t0) / a.steps
            its_np = its.cpu().numpy()
            key = (ebn0,)
            sig = (its_np.copy(), bits.cpu().numpy().copy())
            if key in ref:
                assert np.array_equal(ref[key][0], sig[0]) and np.array_equal(ref[key][1], sig[1]), cfg
            else:
                ref[key] = sig
            avg = np.where(its_np < 0, a.iters, its_np).mean()
            cw_s = B / el
            print(f"Eb/N0 {ebn0:+.1f} dB  persist={persist} slice={width} lanes={lanes} group={group}: {el * 1e3:8.2f} ms  "
                  f"{cw_s:9.0f} cw/s  avg it {avg:5.2f}  frac(of {a.iters} it) {cw_s * avg * bytes_cw_iter / 8e12:.3f}  "
                  f"used slice {dec.get('last_persist')}", flush=True)


if __name__ == "__main__":
    main()
