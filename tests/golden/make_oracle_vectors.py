#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.npz: seeded LLR frames and what the CPU ORACLE
(oracle/, this build's C restatement of the reference) returns for them, per implementation.
SELF-GENERATED: the reference (Rust) cannot be run in this image, so these vectors pin the
HIP path and the oracle against regressions; they are not reference outputs.

  python tests/golden/make_oracle_vectors.py        (needs only the CPU: oracle + host library)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle_binding as ob
from frames import alist, awgn_frames

import ldpc_toolbox_amd as lt

SPEC, BATCH, EBN0, MAX_ITER, SEED = "nr5g:2:8", 24, 2.0, 15, 2024


def main():
    msgs, llrs, full = awgn_frames(SPEC, BATCH, EBN0, SEED)
    llrs[0] = 1.0          # the all-zero codeword, clean: pre-check path, 0 iterations
    llrs[1, ::7] = 0.0     # erasures (0.0 LLRs decide bit 1, arithmetic.rs:199)
    g = ob.Graph(alist(SPEC))
    out = {"spec": np.array(SPEC), "max_iterations": np.array(MAX_ITER), "llrs": llrs.astype(np.float32)}
    for impl in lt.ALL_IMPLEMENTATIONS:
        bits, its, post = ob.decode_batch(g, impl, out["llrs"], MAX_ITER, threads=4)
        out[impl + "/bits"] = np.packbits(bits, axis=1)
        out[impl + "/iterations"] = its
        if "i8" in impl:
            out[impl + "/posterior"] = post.astype(np.int8)      # 8-bit LLRs (values of the converged/last pass)
        else:
            out[impl + "/posterior"] = post if impl.endswith("f64") else post.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    print("wrote oracle_vectors.npz:", {k: v.shape for k, v in out.items() if k.endswith("iterations")}.__len__(),
          "implementations")


if __name__ == "__main__":
    main()
