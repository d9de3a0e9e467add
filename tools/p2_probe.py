#!/usr/bin/env python3
"""An early-terminating batch through the device-resident entry, no event brackets: wall time per call.
  python3 tools/p2_probe.py <spec> <impl> <ebn0_db> <batch> [key=value ...]     (decoder options)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

spec, impl, ebn0, B = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
msgs, llrs, _ = awgn_frames(spec, min(B, int(os.environ.get("P2_DISTINCT", "1024"))), ebn0, 17)
llrs = np.concatenate([llrs] * ((B + len(llrs) - 1) // len(llrs)))[:B]
dec = lt.LdpcDecoder(alist(spec), impl)
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    dec.set(k, int(v))
d = torch.from_numpy(llrs).cuda()
bits = torch.zeros((B, dec.k), dtype=torch.uint8, device="cuda")
its = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run():
    dec.decode_batch_device(d.data_ptr(), False, B, 50, bits.data_ptr(), dec.k, its.data_ptr(), 0, s)
run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
it = its.cpu().numpy()
avg = np.where(it < 0, 50, it).mean()
print(f"{spec} {impl} Eb/N0 {ebn0} batch {B} {' '.join(sys.argv[5:])}: best {min(ts) * 1e3:.2f} ms, median {sorted(ts)[2] * 1e3:.2f} ms "
      f"= {B / sorted(ts)[2]:.0f} cw/s; average iterations {avg:.2f}, max {it.max()}, failed {(it < 0).sum()}; lanes {dec.get('last_lanes')} x group {dec.get('last_group')}")
