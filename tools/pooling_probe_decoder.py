#!/usr/bin/env python3
"""Option "pooling" of the batch entries (csrc/device_decoder.h) in the waterfall: the device-resident entry on the
library's own stream, max_iterations = 100 (the reference CLI's default, src/cli/ber.rs:64-66), with and without, same
frames, identical outputs asserted.   python3 tools/pooling_probe_decoder.py [spec impl ebn0 frames]..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import ldpc_toolbox_amd as lt

CASES = [("dvbs2:R1_2", "Minsumf32", 1.55, 32768), ("dvbs2:R1_2", "Minsumf32", 1.7, 32768), ("dvbs2:R1_2", "Minsumf32", 1.3, 32768),
         ("nr5g:1:384", "HLMinsumf32", 1.8, 65536), ("nr5g:1:384", "HLTanhf32", 1.2, 65536), ("ar4ja:1/2:1024", "Minsumf32", 3.0, 524288)]
MAX_IT = 100
print(f"{'code':16s} {'implementation':14s} {'Eb/N0':>6s} {'frames':>7s} {'FER':>9s} {'avg it':>7s} {'plain cw/s':>11s} {'pooled cw/s':>11s} {'gain':>6s} {'second pass':>11s}")
for spec, impl, ebn0, frames in CASES:
    punct = "1,1,1,1,0" if spec.startswith("ar4ja") else ""
    a = lt.code_alist(spec)
    gen = lt.Simulator(a, impl, punct, device=0, pool_size=32, pool_seed=5)
    llrs = torch.empty((frames, gen.n_tx), dtype=torch.float32, device="cuda")
    gen.generate_into(llrs.data_ptr(), ebn0, 11, 0, frames)
    gen.close()
    dec = lt.LdpcDecoder(a, impl, punct)
    bits = torch.zeros((frames, dec.k), dtype=torch.uint8, device="cuda")
    its = torch.zeros(frames, dtype=torch.int32, device="cuda")
    res = {}
    for pooling in (0, 1, 0, 1):
        dec.set("pooling", pooling)
        best = None
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            dec.decode_batch_device(llrs.data_ptr(), False, frames, MAX_IT, bits.data_ptr(), dec.k, its.data_ptr(), 0, 0)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out = (bits.cpu().numpy().copy(), its.cpu().numpy().copy())
        if "ref" in res:
            assert np.array_equal(res["ref"][0], out[0]) and np.array_equal(res["ref"][1], out[1]), (spec, impl, pooling)
        else:
            res["ref"] = out
        res[pooling] = min(best, res.get(pooling, 1e9))
        if pooling:
            res["pooled"] = dec.get("last_pooled")
    i = res["ref"][1]
    print(f"{spec:16s} {impl:14s} {ebn0:6.2f} {frames:7d} {(i < 0).mean():9.2e} {np.where(i < 0, MAX_IT, i).mean():7.2f} {frames / res[0]:11.0f} "
          f"{frames / res[1]:11.0f} {res[0] / res[1]:6.2f} {res['pooled']:11d}", flush=True)
    dec.close()
