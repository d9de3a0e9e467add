#!/usr/bin/env python3
"""GPU tuning probe (not part of the product): times the decode path for one code / rule over
a grid of group sizes, with random LLRs resident in HBM.  Usage on the GPU box:
  python tools/perf_probe.py --spec dvbs2:R1_2 --impl Minsumf32 --batch 4096 --iters 50 --groups 4096,1024,256
Environment knobs read by the library: LDPC_TOOLBOX_VEC, LDPC_TOOLBOX_UNROLL, LDPC_TOOLBOX_WAVES."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import ldpc_toolbox_amd as lt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", default="dvbs2:R1_2")
    ap.add_argument("--impl", default="Minsumf32")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--groups", default="4096")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--set", default="", help="decoder options, e.g. hl_reg=0,vec=2")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    alist = lt.code_alist(a.spec)
    dec = lt.LdpcDecoder(alist, a.impl, device=0)
    for kv in filter(None, a.set.split(",")):
        key, val = kv.split("=")
        dec.set(key, int(val))
    n, E = dec.n, dec.edges
    layered = a.impl.startswith("HL")
    elem = 8 if a.impl.endswith("f64") else (1 if "i8" in a.impl else 4)
    bytes_cw_iter = ((4 * E + n) if layered else (4 * E + 2 * n)) * elem
    g = torch.Generator(device=dev).manual_seed(0)
    # all-zero codeword over AWGN: LLR = 2(1 + sigma z)/sigma^2
    llrs = (2.0 / a.sigma ** 2) * (1.0 + a.sigma * torch.randn((a.batch, n), generator=g, device=dev))
    bits = torch.zeros((a.batch, dec.k), dtype=torch.uint8, device=dev)
    its = torch.zeros(a.batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    for grp in [int(x) for x in a.groups.split(",")]:
        dec.set("group_size", grp)
        dec.set("profiling", 1)
        best = None
        for rep in range(a.reps + 1):
            dec.kernel_stats(0, reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.decode_batch_device(llrs.data_ptr(), False, a.batch, a.iters, bits.data_ptr(), dec.k,
                                    its.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep and (best is None or dt < best[0]):
                stats = [dec.kernel_stats(k) for k in range(3)]
                best = (dt, stats)
        dt, stats = best
        it_np = its.cpu().numpy()
        iters_done = int((it_np < 0).sum()) * a.iters + int(it_np[it_np >= 0].sum())
        cw_s = a.batch / dt
        gbs = a.batch * a.iters * bytes_cw_iter / dt / 1e9
        line = f"group {grp:6d}  {dt*1e3:9.2f} ms  {cw_s:10.1f} cw/s  alg {gbs:8.1f} GB/s ({gbs/8000:.3f} of 8 TB/s)"
        line += f"  failed {int((it_np < 0).sum())}/{a.batch} avg-iter {iters_done / a.batch:.1f}"
        for name, (cnt, ms) in zip(("cn", "vn", "layer"), stats):
            if cnt:
                line += f"  {name}: {cnt} x {ms / cnt * 1e3:.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    main()
