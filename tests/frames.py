"""Seeded AWGN frames shared by the parity tests (host side: C ABI encoder + numpy Philox)."""
import numpy as np

import ldpc_toolbox_amd as lt
from ldpc_toolbox_amd import simulation as sim

_ALIST = {}


def alist(spec):
    if spec not in _ALIST:
        _ALIST[spec] = lt.code_alist(spec)
    return _ALIST[spec]


def awgn_frames(spec, batch, ebn0_db, seed, puncturing="", dtype=np.float32):
    """-> (messages [B][k], llrs [B][n_tx] as handed to the decoder, full llrs [B][n])"""
    a = alist(spec)
    h_cols, h_rows = (int(x) for x in a.split("\n", 1)[0].split())
    n, k = h_cols, h_cols - h_rows
    enc = lt.Encoder(a)
    pattern = sim.parse_puncturing_pattern(puncturing) if puncturing else None
    n_tx = n if not pattern else n // len(pattern) * sum(pattern)
    sigma = sim.noise_sigma(k / n_tx, ebn0_db)
    msgs, llrs = sim.generate_frames(lambda m: enc.encode(m, n), k, batch, sigma, seed, pattern=pattern,
                                     dtype=dtype)
    full = sim.depuncture(llrs, pattern) if pattern else llrs
    return msgs, llrs, full
