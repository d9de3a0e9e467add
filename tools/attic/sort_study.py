#!/usr/bin/env python3
"""Would ordering the frames of a batch by a cheap difficulty estimate (the weight of the raw input's syndrome) make the
256-codeword tiles finish together?  Prints, for tiles of 256 / wavefront slices of 64, the lane-iterations spent
(slots x iterations until the tile's last frame stops) in arrival order, sorted by syndrome weight, and sorted by the
true iteration count (the unreachable ideal), relative to the sum of the frames' own iteration counts.
  python3 tools/sort_study.py [spec impl ebn0 frames]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

spec = sys.argv[1] if len(sys.argv) > 1 else "dvbs2:R1_2"
impl = sys.argv[2] if len(sys.argv) > 2 else "Minsumf32"
ebn0 = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
msgs, llrs, _ = awgn_frames(spec, B, ebn0, 5)
dec = lt.LdpcDecoder(alist(spec), impl)
bits, its, _ = dec.decode_batch(llrs, 50)
its = np.where(its < 0, 50, its).astype(np.int64)
hard = (llrs <= 0).astype(np.uint8)
_, weight = dec.syndrome(hard)
weight = np.asarray(weight).astype(np.int64)
print(f"{spec} {impl} Eb/N0 {ebn0}: {B} frames, average iterations {its.mean():.2f}, max {its.max()}, "
      f"correlation(syndrome weight, iterations) = {np.corrcoef(weight, its)[0, 1]:.3f}")
own = its.sum()
for unit in (256, 64):
    def spent(order):
        t = its[order].reshape(-1, unit)
        return (t.max(axis=1) * unit).sum()
    arrival = np.arange(B)
    print(f"  units of {unit}: arrival order {spent(arrival) / own:.3f} x the frames' own iterations, "
          f"sorted by syndrome weight {spent(np.argsort(weight, kind='stable')) / own:.3f}, "
          f"sorted by true iterations {spent(np.argsort(its, kind='stable')) / own:.3f}")
