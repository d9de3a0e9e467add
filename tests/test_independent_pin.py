"""The oracle's second pin: tests/independent_restatement.py -- numpy + libm,
written from the reference's .rs files, sharing nothing with oracle/ -- must agree with the C oracle
bit for bit (hard decisions, iteration counts, final LLRs) on every one of the reference's 36
implementation names, both schedules, on small irregular codes with real iterations.  The reference
itself pins only Phif64 on a 4x6 matrix (flooding.rs:161-189); with this test a misreading of
arithmetic.rs / flooding.rs / horizontal_layered.rs would have to be made twice, independently."""
import zlib

import numpy as np
import pytest

import independent_restatement as ind
import oracle_binding as oracle

import ldpc_toolbox_amd as lt

# 5G NR base graphs at the smallest lifting sizes: irregular (check degrees 3..19 / 3..10), with
# degree-1 variables (the extension columns: Deg1Clip matters) and high-degree punctured columns
CODES = ("nr5g:1:2", "nr5g:2:3")


def _frames(alist, n, k, count, ebn0_db, seed):
    """random codewords of the code over BPSK/AWGN, f32 LLRs (exactly representable in f64)"""
    rng = np.random.default_rng(seed)
    enc = lt.Encoder(alist)
    msgs = rng.integers(0, 2, size=(count, k), dtype=np.uint8)
    cws = np.stack([enc.encode(m, n) for m in msgs])
    sigma = np.sqrt(0.5 / ((k / n) * 10.0 ** (ebn0_db / 10.0)))
    y = (cws.astype(np.float64) * 2.0 - 1.0) + sigma * rng.standard_normal(cws.shape)
    llrs = (-2.0 / sigma ** 2 * y).astype(np.float32)
    llrs[0] = np.where(cws[0] == 1, -3.5, 3.5)                  # a clean codeword: iterations 0
    llrs[1, : n // 2] = 0.0                                     # ties: LLR exactly zero -> bit 1, "positive" sign
    llrs[2] *= 40.0                                             # saturates the i8 quantiser, deep tanh clamp
    return llrs


@pytest.mark.parametrize("spec", CODES)
@pytest.mark.parametrize("name", ind.REFERENCE_NAMES)
def test_oracle_equals_the_independent_restatement(spec, name):
    alist = lt.code_alist(spec)
    g = oracle.Graph(alist)
    n, m = g.cols, g.rows
    llrs = _frames(alist, n, n - m, 48, 3.0, seed=zlib.crc32(f"{spec}/{name}".encode()) % 1000)
    max_it = 5
    obits, oits, opost = oracle.decode_batch(g, name, llrs, max_it, threads=2)
    ibits, iits, ipost = ind.decode(alist, name, llrs.astype(np.float64), max_it)
    assert np.array_equal(oits, iits), (name, oits, iits)
    assert np.array_equal(obits, ibits), name
    run = iits != 0                                            # iterations 0: no decoder state exists in the reference
    assert np.array_equal(opost[run], ipost[run]), name        # bit for bit (NaN-free at these inputs)
    # the sample exercises what it claims to: a pre-check hit, real successes after >= 1 iteration, failures
    assert iits[0] == 0 and (iits > 0).any() and (iits == -1).any()


def test_minus_zero_sum_identity_and_degree_zero_variable():
    """arithmetic.rs:146: the variable sum is Rust's float Sum, a fold from -0.0.  A variable with no
    checks (degree 0) therefore keeps input + (-0.0) = input, sign of zero included; and a variable whose
    messages are all -0.0 sums to -0.0.  Both restatements must agree on the sign bit."""
    # 3 checks x 5 variables; variable 4 has degree 0
    alist = "5 3\n2 3\n2 2 1 1 0\n3 2 1\n1 2\n1 3\n1\n2\n\n1 2 3\n1 2 4\n2\n"
    g = oracle.Graph(alist)
    llrs = np.array([[-0.0, 1.5, -2.0, 0.5, -0.0], [0.0, -1.0, 2.0, -0.5, 0.0], [1.0, 1.0, -1.0, 1.0, -0.0]], dtype=np.float32)
    for name in ("Phif32", "Phif64", "Tanhf32", "Tanhf64"):          # degree-1 checks: defined for Phi and Tanh only
        obits, oits, opost = oracle.decode_batch(g, name, llrs, 3, threads=1)
        ibits, iits, ipost = ind.decode(alist, name, llrs.astype(np.float64), 3)
        assert np.array_equal(oits, iits) and np.array_equal(obits, ibits), name
        run = iits != 0
        assert np.array_equal(opost[run], ipost[run]) and np.array_equal(np.signbit(opost[run]), np.signbit(ipost[run])), name
        # the degree-0 variable's final LLR is its input, -0.0 kept as -0.0
        for b in np.nonzero(run)[0]:
            assert opost[b, 4] == llrs[b, 4] and np.signbit(opost[b, 4]) == np.signbit(llrs[b, 4])


def test_degree_one_clip_in_i8():
    """arithmetic.rs:817-833: with ...Deg1Clip a degree-one variable's channel LLR is clipped to +-116
    before the sum; without it the full +-127 is used.  nr5g codes have degree-1 variables; saturate them."""
    alist = lt.code_alist("nr5g:2:2")
    g = oracle.Graph(alist)
    rng = np.random.default_rng(5)
    llrs = (rng.standard_normal((16, g.cols)) * 30.0).astype(np.float32)   # |8 * llr| >> 127: every input saturates
    differs = False
    for base in ("Minstarapproxi8", "Aminstari8Jones"):
        res = {}
        for name in (base, base + "Deg1Clip"):
            obits, oits, opost = oracle.decode_batch(g, name, llrs, 4, threads=1)
            ibits, iits, ipost = ind.decode(alist, name, llrs.astype(np.float64), 4)
            assert np.array_equal(oits, iits) and np.array_equal(obits, ibits), name
            assert np.array_equal(opost[iits != 0], ipost[iits != 0]), name
            res[name] = opost
        differs = differs or not np.array_equal(res[base], res[base + "Deg1Clip"])
    assert differs                                              # the option is live on this input
