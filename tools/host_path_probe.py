#!/usr/bin/env python3
"""Times the host-buffer entry (PCIe-inclusive) against the device-resident one."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import ldpc_toolbox_amd as lt

spec, impl, B, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
dec = lt.LdpcDecoder(lt.code_alist(spec), impl)
rng = np.random.default_rng(0)
llrs = (2.0 * (1.0 + 1.0 * rng.standard_normal((B, dec.n), dtype=np.float32))).astype(np.float32)
for lanes in (1, 2):
    dec.set("lanes", lanes)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        out = dec.decode_batch(llrs, iters)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{spec} {impl} host path lanes={lanes}: {best*1e3:.1f} ms for {B} frames = {B/best:.0f} cw/s", flush=True)
d = torch.from_numpy(llrs).cuda()
bits = torch.zeros((B, dec.k), dtype=torch.uint8, device="cuda"); its = torch.zeros(B, dtype=torch.int32, device="cuda")
for lanes in (1, 2):
    dec.set("lanes", lanes)
    best = None
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec.decode_batch_device(d.data_ptr(), False, B, iters, bits.data_ptr(), dec.k, its.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{spec} {impl} device path lanes={lanes}: {best*1e3:.1f} ms = {B/best:.0f} cw/s", flush=True)
