#!/usr/bin/env python3
"""One-off wide parity sweep: every implementation name x a set of codes (all families, short and
long rows, punctured), a few dozen frames each, GPU against the oracle: bits, iteration counts and
posterior LLRs must be identical."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import ldpc_toolbox_amd as lt
import oracle_binding as ob
from frames import alist, awgn_frames

CASES = [("ar4ja:1/2:1024", "1,1,1,1,0", 2.0), ("ar4ja:2/3:1024", "1,1,1,1,1,1,0", 2.8), ("ar4ja:4/5:1024", "", 3.4), ("c2", "", 3.9),
         ("nr5g:1:8", "", 1.0), ("nr5g:2:24", "", 1.2), ("nr5g:1:36", "", 0.8), ("dvbs2:R1_4short", "", 0.9), ("dvbs2:R1_2short", "", 1.6),
         ("dvbs2:R8_9short", "", 4.3), ("dvbs2:R3_5", "", 2.4)]
bad = 0
t0 = time.perf_counter()
for spec, punct, ebn0 in CASES:
    frames = 24 if spec == "dvbs2:R3_5" else 48
    try:
        msgs, llrs, full = awgn_frames(spec, frames, ebn0, 11, punct)
    except ValueError:                    # no systematic encoder (CCSDS C2's H is rank-deficient): all-zero codeword
        n = int(alist(spec).split()[0])
        sigma = 0.62
        llrs = ((2.0 / sigma ** 2) * (1.0 + sigma * np.random.default_rng(11).standard_normal((frames, n)))).astype(np.float32)
        full = llrs
    g = ob.Graph(alist(spec))
    fails = []
    for impl in lt.ALL_IMPLEMENTATIONS:
        if impl.startswith("HL") and spec == "dvbs2:R3_5":
            continue                      # row-serial on the normal frame: covered by the short codes
        try:
            dec = lt.LdpcDecoder(alist(spec), impl, punct)
        except Exception as e:
            obits = None
            try:
                ob.decode_batch(g, impl, full[:1], 1, threads=1)
                fails.append(impl + "(gpu refused: %s)" % str(e)[:40])
            except RuntimeError:
                pass                      # the reference panics on this combination as well
            continue
        f64 = impl.endswith("f64")
        bits, its, post = dec.decode_batch(llrs.astype(np.float64) if f64 else llrs, 10, want_posterior=True)
        try:
            obits, oits, opost = ob.decode_batch(g, impl, full, 10, threads=8)
        except RuntimeError:
            fails.append(impl + "(oracle panics, gpu decodes)")
            continue
        want = opost if f64 else opost.astype(np.float32)
        if not (np.array_equal(bits, obits) and np.array_equal(its, oits) and np.array_equal(post, want)):
            fails.append(impl)
        dec.close()
    print(f"{spec:18s} puncturing {punct or '-':14s} {len(lt.ALL_IMPLEMENTATIONS)} implementations: " + ("all identical" if not fails else "MISMATCH " + ", ".join(fails)), flush=True)
    bad += len(fails)
print(f"{'OK' if not bad else 'FAILED'} in {time.perf_counter() - t0:.0f} s")
sys.exit(1 if bad else 0)
