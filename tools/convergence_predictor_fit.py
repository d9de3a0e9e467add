#!/usr/bin/env python3
"""Offline companion of convergence_predictor_probe.py: given the trajectory it dumps (DUMP=file.npz: per codeword the
unsatisfied-check count and the count of weak posteriors after every iteration 3..16, and the final iteration count), how
well can ANY function of the trajectory up to iteration k order the codewords by the passes they still need?  A gradient-boosted
regressor is fitted on half of the batch and scored on the other half by the figure that matters for re-packing -- the tile
cost of the sorted order (sum over 256-codeword tiles of the tile's last pass) relative to the per-codeword bound.
  python3 tools/convergence_predictor_fit.py gpurun_out/r05_b/traj_config2.npz"""
import sys

import numpy as np
from sklearn.ensemble import GradientBoostingRegressor


def tile_cost(order, rem, tile=256):
    f = rem[order]
    pad = (-len(f)) % tile
    if pad:
        f = np.concatenate([f, np.zeros(pad, dtype=f.dtype)])
    return f.reshape(-1, tile).max(axis=1).sum() * tile


d = np.load(sys.argv[1])
final = d["final"]
B = len(final)
print(f"{B} frames, mean {final.mean():.2f} iterations")
idx = np.random.default_rng(0).permutation(B)
tr, te = idx[:B // 2], idx[B // 2:]
for k in (8, 10, 12, 13):
    ks = list(range(3, k + 1))
    X = np.stack([d[f"unsat_{j}"] for j in ks] + [d[f"weak_{j}"] for j in ks], axis=1).astype(np.float64)
    rem = (final + 1 - k).astype(np.int64)
    live = final > k
    m = GradientBoostingRegressor(n_estimators=300, max_depth=3, learning_rate=0.05, subsample=0.8, random_state=0)
    m.fit(X[tr][live[tr]], rem[tr][live[tr]])
    tl = te[live[te]]
    pred, r = m.predict(X[tl]), rem[tl]
    bound = r.sum()
    print(f"k={k:2d}: live {len(tl)}  corr(unsat_k) {np.corrcoef(d[f'unsat_{k}'][tl], r)[0, 1]:.3f}  corr(model on the trajectory 3..k) "
          f"{np.corrcoef(pred, r)[0, 1]:.3f}   tile cost / bound: as-is {tile_cost(np.arange(len(tl)), r) / bound:.3f}  by unsat_k "
          f"{tile_cost(np.argsort(d[f'unsat_{k}'][tl], kind='stable'), r) / bound:.3f}  by the model "
          f"{tile_cost(np.argsort(pred, kind='stable'), r) / bound:.3f}  perfect {tile_cost(np.argsort(r, kind='stable'), r) / bound:.3f}")
