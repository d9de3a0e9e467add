#!/usr/bin/env python3
"""Interleaved in-process A/B of launch tunables (cdna guide rule 24: one process, N variants x M
rounds, report median and min).  GPU box only.
  python tools/ab_tune.py --configs "waves=8192;waves=262144;waves=262144,unroll_vn=4" --rounds 5"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import ldpc_toolbox_amd as lt

DEFAULTS = {"waves": 0, "unroll_cn": 8, "unroll_vn": 8, "vec": 4, "block": 256, "group_size": 4096,
            "staged_minsum": 0, "pad_kb": 0, "tile": 0, "nt": 1, "nt_vn": 0, "waves_vn": 0, "lfree": 1, "lfree_unroll": 4, "lfree_nt_in": 0, "compact": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", default="dvbs2:R1_2")
    ap.add_argument("--impl", default="Minsumf32")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--configs", default="waves=8192;waves=262144")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dec = lt.LdpcDecoder(lt.code_alist(a.spec), a.impl, device=0)
    n = dec.n
    g = torch.Generator(device=dev).manual_seed(0)
    llrs = (2.0 / a.sigma ** 2) * (1.0 + a.sigma * torch.randn((a.batch, n), generator=g, device=dev))
    bits = torch.zeros((a.batch, dec.k), dtype=torch.uint8, device=dev)
    its = torch.zeros(a.batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    configs = []
    for c in a.configs.split(";"):
        d = dict(DEFAULTS)
        for kv in c.split(","):
            if kv.strip():
                k, v = kv.split("=")
                d[k.strip()] = int(v)
        configs.append((c, d))
    dec.set("profiling", 1)
    res = {c: {"t": [], "cn": [], "vn": [], "layer": []} for c, _ in configs}
    for rnd in range(a.rounds + 1):
        for name, d in configs:
            for k, v in d.items():
                dec.set(k, v)
            dec.kernel_stats(0, reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.decode_batch_device(llrs.data_ptr(), False, a.batch, a.iters, bits.data_ptr(), dec.k,
                                    its.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rnd == 0:
                continue
            r = res[name]
            r["t"].append(dt * 1e3)
            if a.verbose:
                print(f"round {rnd} {name}: cn {dec.kernel_stats(0)[1] / max(dec.kernel_stats(0)[0], 1) * 1e3:.1f} us "
                      f"vn {dec.kernel_stats(1)[1] / max(dec.kernel_stats(1)[0], 1) * 1e3:.1f} us", file=sys.stderr, flush=True)
            for key, kind in (("cn", 0), ("vn", 1), ("layer", 2)):
                cnt, ms = dec.kernel_stats(kind)
                if cnt:
                    r[key].append(ms / cnt * 1e3)
    for name, _ in configs:
        r = res[name]
        line = f"{name:45s} total ms med {statistics.median(r['t']):8.2f} min {min(r['t']):8.2f}"
        for key in ("cn", "vn", "layer"):
            if r[key]:
                line += f" | {key} us med {statistics.median(r[key]):8.1f} min {min(r[key]):8.1f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
