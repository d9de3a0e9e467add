#!/usr/bin/env python3
"""Experiment: does a cheap statistic of the raw input predict a codeword's iteration count well
enough that placing similar codewords in the same 256-wide tile reduces the work of early
termination?  Prints, per Eb/N0, the tile-iterations of: caller order, order sorted by the raw
syndrome weight, the ideal (sorted by the true iteration count) and the no-waste bound."""
import sys, os
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

SPEC, B, TILE = "dvbs2:R1_2", 4096, 256
a = alist(SPEC)
lines = a.split("\n")
n, m = (int(x) for x in lines[0].split())
rows, cols = [], []
for c in range(n):
    for r in lines[4 + c].split():
        if int(r) > 0:
            rows.append(int(r) - 1); cols.append(c)
H = sp.csr_matrix((np.ones(len(rows), np.int32), (rows, cols)), shape=(m, n))
dec = lt.LdpcDecoder(a, "Minsumf32")
dec.set("compact", 0)

def cost(its):
    t = its.reshape(-1, TILE)
    return int(t.max(axis=1).sum())

for ebn0 in (1.4, 1.6, 2.0, 2.5):
    msgs, llrs, _ = awgn_frames(SPEC, B, ebn0, 7)
    bits, its = dec.decode_batch(llrs, 50)[:2]
    its = np.where(its < 0, 50, its).astype(np.int64)
    hard = (llrs <= 0).astype(np.int32)
    sw = (H @ hard.T % 2).sum(axis=0)               # raw syndrome weight per frame
    soft = np.abs(llrs).mean(axis=1)
    nerr = (hard[:, : msgs.shape[1]] != msgs).sum(axis=1)
    out = [f"Eb/N0 {ebn0}: mean it {its.mean():.2f} max {its.max()} | tile-iterations: caller {cost(its)}"]
    for name, key in (("syndrome-weight", sw), ("mean|llr|", -soft), ("raw-errors(genie)", nerr), ("true-iterations", its)):
        order = np.argsort(key, kind="stable")
        out.append(f"{name} {cost(its[order])} (corr {np.corrcoef(key, its)[0,1]:.2f})")
    out.append(f"no-waste {its.sum() / TILE:.0f}")
    print("; ".join(out), flush=True)
