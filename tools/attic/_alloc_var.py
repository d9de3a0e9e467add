import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames
spec, impl, B = "dvbs2:R1_2", "Minsumf32", 4096
msgs, llrs, _ = awgn_frames(spec, 512, 0.0, 17)
llrs = np.concatenate([llrs] * 8)[:B]
d = torch.from_numpy(llrs).cuda()
bits = torch.zeros((B, 32400), dtype=torch.uint8, device="cuda"); its = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
junk = []
for trial in range(7):
    dec = lt.LdpcDecoder(alist(spec), impl)
    row = f"trial {trial}:"
    for lanes in (2, 1):
        dec.set("lanes", lanes)
        dec.decode_batch_device(d.data_ptr(), False, B, 50, bits.data_ptr(), dec.k, its.data_ptr(), 0, s); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); dec.decode_batch_device(d.data_ptr(), False, B, 50, bits.data_ptr(), dec.k, its.data_ptr(), 0, s); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row += f"  lanes={lanes}: {min(ts)*1e3:.2f} ms"
    print(row, flush=True)
    dec.close()
    junk.append(torch.empty(int(np.random.default_rng(trial).integers(1, 600)) * (1 << 20), dtype=torch.uint8, device="cuda"))
