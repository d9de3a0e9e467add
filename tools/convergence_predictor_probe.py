#!/usr/bin/env python3
"""Does the state of a codeword in the MIDDLE of a decode predict the iteration it will converge in?

VERDICT r4 item 1: convergence-aware re-packing -- order the live codewords by their unsatisfied-check count at a chosen
iteration k so that whole 64*VEC-codeword slices freeze together.  Before building it, the numbers that decide whether it can
pay: for k = 6 .. 14 the batch is decoded for k iterations (posterior out), then in full, and for several predictors taken from
the state at iteration k
  unsat     unsatisfied checks of hard(posterior_k)                     (what the check-node kernels' ballots could count)
  weak      variables with |posterior_k| < 2.0
  meanabs   -mean |posterior_k|
this prints Pearson / Spearman correlation with the final iteration count of the codewords still running at k, and the figure
that matters -- the TILE COST: with the live codewords sorted by the predictor and cut into tiles of 256, sum over tiles of the
tile's last iteration, against the same sum for the order as it is (what the kernels pay today without any re-packing), for a
perfect predictor (sorted by the true count) and the per-codeword sum (the iteration-proportional bound).

  python3 tools/convergence_predictor_probe.py <spec> <impl> <ebn0_db> <batch> [max_iterations]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ldpc_toolbox_amd as lt  # noqa: E402
from frames import alist, awgn_frames  # noqa: E402


def spearman(a, b):
    ra, rb = np.argsort(np.argsort(a)).astype(np.float64), np.argsort(np.argsort(b)).astype(np.float64)
    return float(np.corrcoef(ra, rb)[0, 1])


def tile_cost(order, final, tile=256):
    f = final[order]
    pad = (-len(f)) % tile
    if pad:
        f = np.concatenate([f, np.zeros(pad, dtype=f.dtype)])
    return int(f.reshape(-1, tile).max(axis=1).sum()) * tile


def main():
    spec, impl, ebn0, B = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
    max_it = int(sys.argv[5]) if len(sys.argv) > 5 else 50
    msgs, llrs, _ = awgn_frames(spec, B, ebn0, 23)
    dec = lt.LdpcDecoder(alist(spec), impl)
    _, final, _ = dec.decode_batch(llrs, max_it)
    final = np.where(final < 0, max_it, final).astype(np.int64)
    # a codeword that converges at iteration i is seen converged in pass i + 1 (fused parity ballots): it costs i + 1 passes
    cost = final + 1
    pct = np.percentile(final, [1, 10, 50, 90, 99]).astype(int)
    print(f"{spec} {impl} Eb/N0 {ebn0} dB, {B} frames: iterations mean {final.mean():.2f}, percentiles 1/10/50/90/99 = {list(pct)}, max {final.max()}")
    print(f"{'k':>3} {'live':>5} | {'predictor':>8} {'pearson':>8} {'spearman':>8} | tile cost of the live codewords' remaining passes, relative to the bound "
          f"(as-is order / sorted by predictor / perfect)")
    dump = {"final": final}
    for k in range(int(os.environ.get("K_FIRST", "6")), int(os.environ.get("K_LAST", "14")) + 1):
        bits, its_k, post = dec.decode_batch(llrs, k, want_posterior=True)
        if os.environ.get("DUMP"):      # the whole batch's trajectory for offline analysis (tools/convergence_predictor_fit.py)
            dump[f"unsat_{k}"] = dec.syndrome(bits)[1]
            dump[f"weak_{k}"] = (np.abs(post) < 2.0).sum(axis=1).astype(np.int32)
        live = np.nonzero(final > k)[0]
        if len(live) < 512:
            continue
        _, unsat = dec.syndrome(bits[live])
        p = np.abs(post[live])
        preds = {"unsat": unsat.astype(np.float64), "weak": (p < 2.0).sum(axis=1).astype(np.float64), "meanabs": -p.mean(axis=1).astype(np.float64)}
        rem = (cost[live] - k).astype(np.int64)            # passes still to pay after iteration k
        bound = int(rem.sum())
        asis = tile_cost(np.arange(len(live)), rem)
        perfect = tile_cost(np.argsort(rem, kind="stable"), rem)
        for name, x in preds.items():
            srt = tile_cost(np.argsort(x, kind="stable"), rem)
            print(f"{k:3d} {len(live):5d} | {name:>8} {np.corrcoef(x, rem)[0, 1]:8.3f} {spearman(x, rem):8.3f} | "
                  f"{asis / bound:6.3f} / {srt / bound:6.3f} / {perfect / bound:6.3f}")
    if os.environ.get("DUMP"):
        np.savez_compressed(os.environ["DUMP"], **dump)
    print("(tile cost 1.000 = every tile stops with its last codeword at no loss; as-is = no re-packing at all)")


if __name__ == "__main__":
    main()
