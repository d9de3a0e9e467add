#!/usr/bin/env python3
"""Decode time of 4096 DVB-S2 1/2 frames over Eb/N0 for several settings of the compaction rule."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

spec, B = "dvbs2:R1_2", 4096
VARIANTS = {
    "off": {"compact": 0},
    "default": {},
    "cost5": {"compact_cost_live": 5},
    "cost3": {"compact_cost_live": 3},
    "freed1": {"compact_min_freed_q": 1},
    "freed1cost5": {"compact_min_freed_q": 1, "compact_cost_live": 5},
    "freed1cost3": {"compact_min_freed_q": 1, "compact_cost_live": 3},
    "hor12": {"compact_horizon": 12},
    "first4": {"compact_first": 4},
    "every1": {"compact_every": 1},
    "tile128": {"tile": 128, "vec": 2},
    "t128freed1": {"tile": 128, "vec": 2, "compact_min_freed_q": 1, "compact_cost_live": 5},
    "t128every1": {"tile": 128, "vec": 2, "compact_every": 1, "compact_first": 4},
}
DEFAULTS = {"compact": 1, "compact_horizon": 8, "compact_cost_live": 9, "compact_cost_slots": 0, "compact_min_freed_q": 2,
            "compact_first": 6, "compact_every": 2, "retire_blocks": 256, "move_waves": 65536, "tile": 256, "vec": 4}
dec = lt.LdpcDecoder(alist(spec), "Minsumf32")
bits = torch.zeros((B, dec.k), dtype=torch.uint8, device="cuda"); its = torch.zeros(B, dtype=torch.int32, device="cuda")
print(f"{'Eb/N0':>6s} " + " ".join(f"{k:>12s}" for k in VARIANTS))
for ebn0 in (1.3, 1.5, 1.6, 1.8, 2.0, 2.5, 0.0):
    msgs, llrs, _ = awgn_frames(spec, B, ebn0, 7)
    d = torch.from_numpy(llrs).cuda()
    row = []
    for name, opts in VARIANTS.items():
        for k, v in {**DEFAULTS, **opts}.items():
            dec.set(k, v)
        best = None
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            dec.decode_batch_device(d.data_ptr(), False, B, 50, bits.data_ptr(), dec.k, its.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        row.append(best * 1e3)
    it_np = its.cpu().numpy()
    avg = (int((it_np < 0).sum()) * 50 + int(it_np[it_np >= 0].sum())) / B
    print(f"{ebn0:6.2f} " + " ".join(f"{x:12.1f}" for x in row) + f"   avg-it {avg:5.1f}", flush=True)
