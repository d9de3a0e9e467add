#!/usr/bin/env python3
"""A/B of the flooding min-sum check-node pass inside ONE process on one box: row records
(cn_minsum_rec_kernel, "records" = 1) against per-edge messages (cn_minsum_lfree_kernel, "records" = 0).
Same frames, same handle; prints ms per decode, codewords/s, the HIP-event averages of both launches, and
whether bits / iteration counts / posterior LLRs of the two forms are identical.
  python tools/records_ab.py [--spec dvbs2:R1_2] [--batch 4096] [--iters 50] [--ebn0 0.0] [--variants "records=1;records=0;..."]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import ldpc_toolbox_amd as lt
from ldpc_toolbox_amd import simulation as sim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", default="dvbs2:R1_2")
    ap.add_argument("--impl", default="Minsumf32")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--ebn0", type=float, default=0.0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variants", default="records=0;records=1")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    alist = lt.code_alist(a.spec)
    dec = lt.LdpcDecoder(alist, a.impl, device=0)
    enc = lt.Encoder(alist)
    rng = np.random.Generator(np.random.Philox(key=[5, 0]))
    pool = min(a.batch, 256)
    msgs = rng.integers(0, 2, size=(pool, dec.k), dtype=np.uint8)
    cws = np.stack([enc.encode(m, dec.n) for m in msgs])[np.arange(a.batch) % pool]
    sigma = sim.noise_sigma(dec.k / dec.n, a.ebn0)
    g = torch.Generator(device=dev).manual_seed(11)
    y = torch.from_numpy(cws).to(dev).to(torch.float32) * 2.0 - 1.0
    y = y + sigma * torch.randn(y.shape, generator=g, device=dev, dtype=torch.float32)
    f64 = a.impl.endswith("f64")
    llrs = ((-2.0 / sigma ** 2) * y).to(torch.float64 if f64 else torch.float32).contiguous()
    bits = torch.zeros((a.batch, dec.n), dtype=torch.uint8, device=dev)
    its = torch.zeros(a.batch, dtype=torch.int32, device=dev)
    post = torch.zeros((a.batch, dec.n), dtype=llrs.dtype, device=dev)
    stream = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    ref = None
    E, n = dec.edges, dec.n
    bytes_cw_iter = (4 * E + 2 * n) * (8 if f64 else 4)
    for variant in a.variants.split(";"):
        for kv in filter(None, variant.split(",")):
            k, v = kv.split("=")
            dec.set(k, int(v))
        dec.set("profiling", 1)
        best = None
        for rep in range(a.reps + 1):
            dec.kernel_stats(0, reset=True)
            dec.kernel_stats(1, reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.decode_batch_device(llrs.data_ptr(), f64, a.batch, a.iters, bits.data_ptr(), dec.n, its.data_ptr(),
                                    post.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep and (best is None or dt < best[0]):
                best = (dt, dec.kernel_stats(0), dec.kernel_stats(1))
        dec.set("profiling", 0)
        dt, (cn_n, cn_ms), (vn_n, vn_ms) = best
        out = (bits.cpu().numpy().copy(), its.cpu().numpy().copy(), post.cpu().numpy().copy())
        same = "reference"
        if ref is None:
            ref = out
        else:
            same = "identical" if all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(ref, out)) else "DIFFERENT"
        it_np = out[1]
        avg_it = (int((it_np < 0).sum()) * a.iters + int(it_np[it_np >= 0].sum())) / a.batch
        print(f"{variant:40s} {dt * 1e3:8.2f} ms  {a.batch / dt:9.1f} cw/s  alg {a.batch * avg_it * bytes_cw_iter / dt / 1e12:5.2f} TB/s"
              f"  cn {cn_ms / max(cn_n, 1) * 1e3:7.1f} us x{cn_n}  vn {vn_ms / max(vn_n, 1) * 1e3:6.1f} us  avg-it {avg_it:5.1f}"
              f"  failed {int((it_np < 0).sum())}  {same}", flush=True)


if __name__ == "__main__":
    main()
