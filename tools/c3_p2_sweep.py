#!/usr/bin/env python3
"""BASELINE config 3 with early termination (5G NR BG1 Zc=384, HLTanhf32, 8192 frames at +2 dB, 4.5 iterations on average):
the compaction schedule and the pacing knobs of the layered path, identical outputs asserted.
  python tools/c3_p2_sweep.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import ldpc_toolbox_amd as lt
from bench_config3 import frames

device = torch.device("cuda", 0)
alist = lt.code_alist("nr5g:1:384")
dec = lt.LdpcDecoder(alist, "HLTanhf32", device=0)
enc = lt.Encoder(alist)
B = 8192
msgs, llrs = frames(dec, enc, B, 2.0, 31, device)
bits = torch.zeros((B, dec.k), dtype=torch.uint8, device=device)
its = torch.zeros(B, dtype=torch.int32, device=device)
stream = torch.cuda.current_stream(device)
ref = None
configs = [dict(), dict(compact_first=6, compact_every=2)] + [dict(compact_first=f, compact_every=e) for f in (2, 3, 4) for e in (1, 2)] + [dict(compact=0)] + \
          [dict(compact_first=3, compact_every=1, group_size=8192, lanes=1), dict(compact_first=3, compact_every=1, group_size=2048)]
for cfg in configs:
    for k, v in dict(compact=1, compact_first=0, compact_every=0, group_size=0, lanes=0, throttle=1).items():
        dec.set(k, v)
    for k, v in cfg.items():
        dec.set(k, v)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec.decode_batch_device(llrs.data_ptr(), False, B, 50, bits.data_ptr(), dec.k, its.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    sig = (its.cpu().numpy().copy(), bits.cpu().numpy().copy())
    if ref is None:
        ref = sig
    assert np.array_equal(ref[0], sig[0]) and np.array_equal(ref[1], sig[1]), cfg
    print(f"{str(cfg):70s} {best * 1e3:7.2f} ms  {B / best:9.0f} cw/s  (avg it {np.where(sig[0] < 0, 50, sig[0]).mean():.2f}, max {sig[0].max()})", flush=True)
