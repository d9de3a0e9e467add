#!/usr/bin/env python3
"""One decode of 4096 DVB-S2 1/2 frames at a waterfall Eb/N0 (wide spread of iteration counts):
the workload batch compaction exists for.  Run under rocprofv3 --kernel-trace to see what the
compaction kernels cost against the iterations they save."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

ebn0 = float(sys.argv[1]) if len(sys.argv) > 1 else 1.6
compact = int(sys.argv[2]) if len(sys.argv) > 2 else 1
spec, B = "dvbs2:R1_2", 4096
msgs, llrs, _ = awgn_frames(spec, B, ebn0, 7)
dec = lt.LdpcDecoder(alist(spec), "Minsumf32")
dec.set("compact", compact)
d = torch.from_numpy(llrs).cuda()
bits = torch.zeros((B, dec.k), dtype=torch.uint8, device="cuda"); its = torch.zeros(B, dtype=torch.int32, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec.decode_batch_device(d.data_ptr(), False, B, 50, bits.data_ptr(), dec.k, its.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
it = its.cpu().numpy(); it = np.where(it < 0, 50, it)
print(f"Eb/N0 {ebn0} compact={compact}: {dt*1e3:.1f} ms, mean iterations {it.mean():.1f}, no-waste bound {it.sum()/256:.0f} tile-iterations", flush=True)
