"""GPU: randomised calls on the LARGE graphs -- DVB-S2 normal frames, 5G NR BG1 Zc=384 -- whose launches have far more
wavefronts than the chip holds at once.  tests/test_gpu_stress.py randomises everything on small codes, where every wave of a
launch has started before the first one finishes; what depends on the ORDER in which a launch's waves run (round 5's advisor
finding: the variable-node launch whose bookkeeping waves end the group while others have not started) only shows on graphs
of this size, with groups of a few codewords, tail groups and slices that converge whole.

Per seed: a handful of distinct frames (decoded by the oracle) replicated into a batch -- slice-homogeneous or mixed --,
random execution choices (row records or per-edge messages among them), host and device entries in turn, all n hard bits
and the posterior; against the oracle."""
import os

import numpy as np
import pytest

import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

pytestmark = pytest.mark.gpu

CASES = [("dvbs2:R1_2", "Minsumf32", 2.1, 50), ("dvbs2:R3_5", "Minsumf32", 3.0, 40), ("nr5g:1:384", "Minsumf32", 2.2, 30),
         ("dvbs2:R1_2", "Minsumf64", 2.1, 50), ("nr5g:1:384", "HLMinsumf32", 1.8, 20), ("dvbs2:R9_10", "Minsumf32", 4.6, 30)]
_FIRST = int(os.environ.get("LDPC_STRESS_FIRST", "0"))


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + int(os.environ.get("LDPC_STRESS_LARGE_SEEDS", "24"))))
def test_random_call_on_a_large_graph(oracle, seed):
    import torch
    rng = np.random.default_rng(77000 + seed)
    spec, impl, ebn0, max_it = CASES[seed % len(CASES)]
    f64 = impl.endswith("f64")
    distinct = int(rng.choice([1, 2, 3, 6]))
    msgs, llrs, full = awgn_frames(spec, distinct, ebn0 + float(rng.uniform(-0.1, 0.4)), 900 + seed)
    dec = lt.LdpcDecoder(alist(spec), impl)
    G = dec.get("preferred_group")
    batch = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 255, 257, G + 1, G + 1, 2 * G + 3, G - 1]))
    if f64:
        batch = min(batch, G + 1)
    pattern = rng.choice(["mixed", "slices", "one"])
    if pattern == "mixed":
        idx = rng.integers(0, distinct, size=batch)
    elif pattern == "slices":        # every 256-codeword slice holds one frame: a slice's first convergences are all of it
        idx = (np.arange(batch) // 256 + int(rng.integers(distinct))) % distinct
    else:
        idx = np.zeros(batch, dtype=np.int64)
    obits, oits, opost = oracle.decode_batch(oracle.Graph(alist(spec)), impl, full, max_it, threads=8)
    want = (obits[idx], oits[idx], opost[idx] if f64 else opost[idx].astype(np.float32))
    gin = (llrs.astype(np.float64) if f64 else llrs)[idx]
    knobs = {"latency": int(rng.choice([0, 0, 32])), "vn_event": int(rng.integers(2)), "rec_quiet": int(rng.integers(2)),
             "compact": int(rng.integers(2)), "throttle": int(rng.integers(2)), "lanes": int(rng.choice([0, 1, 2])),
             "records": int(rng.choice([1, 2, 2, 0])), "rec_run": int(rng.choice([1, 8, 8, 64])), "poll": int(rng.integers(2)),
             "compact_every": int(rng.choice([0, 1])), "hl_records": int(rng.integers(2)), "pooling": int(rng.integers(2))}
    for k, v in knobs.items():
        dec.set(k, v)
    ctx = (seed, spec, impl, batch, pattern, knobs)
    host = bool(rng.integers(2))
    for rep in range(3):
        if host:
            got = dec.decode_batch(gin, max_it, want_posterior=True, output_len=dec.n)
        else:
            d = torch.from_numpy(gin).cuda()
            bits = torch.zeros((batch, dec.n), dtype=torch.uint8, device="cuda")
            its = torch.full((batch,), -9, dtype=torch.int32, device="cuda")
            post = torch.zeros((batch, dec.n), dtype=torch.float64 if f64 else torch.float32, device="cuda")
            stream = torch.cuda.Stream()
            torch.cuda.synchronize()
            dec.decode_batch_device(d.data_ptr(), f64, batch, max_it, bits.data_ptr(), dec.n, its.data_ptr(), post.data_ptr(),
                                    stream.cuda_stream if rep % 2 else 0)
            torch.cuda.synchronize()
            got = (bits.cpu().numpy(), its.cpu().numpy(), post.cpu().numpy())
        for name, a, b in zip(("bits", "iterations", "posterior"), got, want):
            assert np.array_equal(a, b), (ctx, rep, "host" if host else "device", name,
                                          np.argwhere(np.atleast_2d(a != b).reshape(len(a), -1).any(axis=1)).ravel()[:8])
        host = not host
    assert (want[1] >= 0).any() or ebn0 < 0
