#!/usr/bin/env python3
"""Does the two-lane call's time depend on where the allocator put the workspaces?  Seven decoders in one process, other
allocations in between; each timed with two lanes and with one.   python3 tools/lanes_placement.py [lane_pad_kb ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

spec, impl, B = os.environ.get("LP_SPEC", "dvbs2:R1_2"), os.environ.get("LP_IMPL", "Minsumf32"), int(os.environ.get("LP_BATCH", "4096"))
pads = [int(x) for x in sys.argv[1:]] or [0]
msgs, llrs, _ = awgn_frames(spec, 512, float(os.environ.get("LP_EBN0", "0.0")), 17)
llrs = np.concatenate([llrs] * ((B + 511) // 512))[:B]
d = torch.from_numpy(llrs).cuda()
s = torch.cuda.current_stream().cuda_stream
for pad in pads:
    junk, row2, row1 = [], [], []
    for trial in range(7):
        dec = lt.LdpcDecoder(alist(spec), impl)
        dec.set("lane_pad_kb", pad)
        dec.set("lane_align_mb", int(os.environ.get("LP_ALIGN_MB", "2")))
        bits = torch.zeros((B, dec.k), dtype=torch.uint8, device="cuda"); its = torch.zeros(B, dtype=torch.int32, device="cuda")
        for lanes, row in ((2, row2), (1, row1)):
            dec.set("lanes", lanes)
            dec.decode_batch_device(d.data_ptr(), False, B, 50, bits.data_ptr(), dec.k, its.data_ptr(), 0, s); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); dec.decode_batch_device(d.data_ptr(), False, B, 50, bits.data_ptr(), dec.k, its.data_ptr(), 0, s); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            row.append(min(ts) * 1e3)
        dec.close()
        junk.append(torch.empty(int(np.random.default_rng(trial).integers(1, 600)) * (1 << 20), dtype=torch.uint8, device="cuda"))
    print(f"{spec} {impl} {B} frames, lane_pad_kb={pad}: two lanes " + " ".join(f"{x:.2f}" for x in row2) + " ms;  one lane " + " ".join(f"{x:.2f}" for x in row1) + " ms", flush=True)
    del junk
