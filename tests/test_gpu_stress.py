"""GPU: randomised configurations of the decode path against the oracle -- batch sizes that do not
fill a tile, several groups per call, every tunable that changes the launch structure (tile,
vector width, L-free on/off, compaction on/off, group size).  The tunables must never change a
result; every case is compared bit for bit (min-sum / f32 / i8 rules)."""
import os

import numpy as np
import pytest

import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

pytestmark = pytest.mark.gpu

CODES = [("ar4ja:1/2:1024", "1,1,1,1,0", 2.0), ("nr5g:2:24", "", 1.2), ("dvbs2:R1_2short", "", 1.5),
         ("ar4ja:4/5:1024", "", 3.6), ("nr5g:1:8", "", 1.5)]
EXACT = ["Minsumf32", "Minsumf64", "HLMinsumf32", "Tanhf32", "HLPhif32", "Minstarapproxf32", "HLAminstarf32",
         "Phif64", "HLTanhf64", "Aminstarf64", "HLMinstarapproxf64",
         "Aminstari8JonesPartialHardLimitDeg1Clip", "Minstarapproxi8", "HLAminstari8", "HLMinstarapproxi8PartialHardLimit"]


# (LDPC_STRESS_FIRST / LDPC_STRESS_SEEDS: a builder-side soak over another range of seeds; the driver runs 0..399)
_FIRST = int(os.environ.get("LDPC_STRESS_FIRST", "0"))


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + int(os.environ.get("LDPC_STRESS_SEEDS", "400"))))
def test_random_configuration(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    spec, punct, ebn0 = CODES[rng.integers(len(CODES))]
    impl = EXACT[rng.integers(len(EXACT))]
    # (layered decoding of the DVB-S2 staircase runs in the row-serial mode; "serial_levels" below
    # sometimes forces one launch per row instead)
    batch = int(rng.choice([1, 3, 64, 65, 129, 130, 190, 200, 256, 257, 449, 600, 1100]))
    max_iter = int(rng.choice([0, 1, 2, 7, 13, 24, 40]))
    if "i8" not in impl and ("Tanh" in impl or "Phi" in impl or "star" in impl):
        batch = min(batch, 300)
    msgs, llrs, full = awgn_frames(spec, batch, ebn0 + float(rng.uniform(-0.4, 0.6)), seed, punct)
    dec = lt.LdpcDecoder(alist(spec), impl, punct)
    knobs = {"group_size": int(rng.choice([0, 64, 128, 192, 256, 320, 512, 1024])),
             "tile": int(rng.choice([0, 64, 128, 192, 256, 512])), "lanes": int(rng.choice([0, 1, 2])),
             "hl_reg": int(rng.integers(2)), "cn_reg": int(rng.integers(2)), "serial_levels": int(rng.choice([512, 512, 10 ** 6])), "poll": int(rng.integers(2)),
             "vec": int(rng.choice([1, 2, 4])), "lfree": int(rng.integers(2)), "compact": int(rng.integers(2)),
             "waves": int(rng.choice([0, 256, 4096, 1 << 20])), "latency": int(rng.choice([0, 8, 32])),
             "compact_first": int(rng.choice([2, 6])), "compact_every": int(rng.choice([1, 2])),
             "vn_event": int(rng.integers(2)), "throttle": int(rng.integers(2)), "records": int(rng.choice([0, 1, 2])),
             "rec_quiet": int(rng.integers(2)), "rec_run": int(rng.choice([1, 3, 8, 64])), "hl_records": int(rng.integers(2)),
             "lane_threads": int(rng.integers(2)), "lane_pace": int(rng.integers(2)), "pooling": int(rng.integers(2))}
    for k, v in knobs.items():
        dec.set(k, v)
    gpu_in = llrs.astype(np.float64) if impl.endswith("f64") else llrs
    out_len = int(rng.choice([dec.n, dec.k, 1]))
    bits, its, post = dec.decode_batch(gpu_in, max_iter, output_len=out_len, want_posterior=True)
    g = oracle.Graph(alist(spec))
    obits, oits, opost = oracle.decode_batch(g, impl, full, max_iter, threads=8)
    ctx = (spec, impl, batch, max_iter, knobs)
    assert np.array_equal(its, oits), ctx
    assert np.array_equal(bits, obits[:, :out_len]), ctx
    if "i8" in impl:
        run = its != 0
        assert np.array_equal(post[run].astype(np.float64), opost[run]), ctx
    else:
        assert np.array_equal(post, opost.astype(post.dtype)), ctx
