#!/usr/bin/env python3
"""Experiment: the same batch decoded by S independent handles on S streams concurrently."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import ldpc_toolbox_amd as lt

spec, impl, batch, iters, sigma = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
dev = torch.device("cuda:0")
alist = lt.code_alist(spec)
for S in (1, 2, 4):
    decs = [lt.LdpcDecoder(alist, impl, device=0) for _ in range(S)]
    n = decs[0].n
    g = torch.Generator(device=dev).manual_seed(0)
    llrs = (2.0 / sigma ** 2) * (1.0 + sigma * torch.randn((batch, n), generator=g, device=dev))
    bits = torch.zeros((batch, decs[0].k), dtype=torch.uint8, device=dev)
    its = torch.zeros(batch, dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    per = batch // S
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, (d, st) in enumerate(zip(decs, streams)):
            sl = slice(i * per, (i + 1) * per)
            d.decode_batch_device(llrs[sl].data_ptr(), False, per, iters, bits[sl].data_ptr(), d.k, its[sl].data_ptr(), 0,
                                  st.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep and (best is None or dt < best):
            best = dt
    print(f"{spec} {impl} batch {batch} streams {S}: {best*1e3:.2f} ms  {batch/best:.0f} cw/s", flush=True)
    for d in decs:
        d.close()
