#!/usr/bin/env python3
"""Times the host-buffer entry (PCIe-inclusive: caller's pageable rows in, bits + iteration counts out)
against the device-resident one, same frames, same output length (k bits per frame), output buffers
reused across calls as a C caller would.
  python3 tools/host_path_probe.py dvbs2:R1_2 Minsumf32 16384 50"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import ldpc_toolbox_amd as lt

spec, impl, B, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
dec = lt.LdpcDecoder(lt.code_alist(spec), impl)
rng = np.random.default_rng(0)
llrs = (2.0 * (1.0 + 1.0 * rng.standard_normal((B, dec.n), dtype=np.float32))).astype(np.float32)
out = (np.ones((B, dec.k), dtype=np.uint8), np.ones(B, dtype=np.int32))      # touched: pages are mapped
res = {}
for lanes in (1, 2):
    dec.set("lanes", lanes)
    best = None
    for rep in range(4):
        t0 = time.perf_counter()
        dec.decode_batch(llrs, iters, output_len=dec.k, out=out)
        dt = time.perf_counter() - t0
        best = dt if best is None or rep == 1 else min(best, dt)            # rep 0 allocates the staging
    res["host", lanes] = best if ("host", lanes) not in res else min(best, res["host", lanes])
    print(f"{spec} {impl} host path lanes={lanes}: {best*1e3:.1f} ms for {B} frames = {B/best:.0f} cw/s", flush=True)
host_bits, host_its = out[0].copy(), out[1].copy()
d = torch.from_numpy(llrs).cuda()
bits = torch.zeros((B, dec.k), dtype=torch.uint8, device="cuda"); its = torch.zeros(B, dtype=torch.int32, device="cuda")
stream = torch.cuda.Stream()
torch.cuda.synchronize()
for lanes in (1, 2):
    dec.set("lanes", lanes)
    best = None
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec.decode_batch_device(d.data_ptr(), False, B, iters, bits.data_ptr(), dec.k, its.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res["dev", lanes] = best
    print(f"{spec} {impl} device path lanes={lanes}: {best*1e3:.1f} ms = {B/best:.0f} cw/s", flush=True)
same = bool(np.array_equal(bits.cpu().numpy(), host_bits) and np.array_equal(its.cpu().numpy(), host_its))
h, dv = min(res["host", 1], res["host", 2]), min(res["dev", 1], res["dev", 2])
print(f"host / device = {dv / h:.3f} of the device-resident rate (host {B/h:.0f} cw/s, device {B/dv:.0f} cw/s), same results: {same}")
