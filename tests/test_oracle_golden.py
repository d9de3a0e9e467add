"""CPU: the oracle (oracle/, C restatement of the reference decoder) against every known
answer the reference's own tests hold for this path, and against the committed vectors."""
import json
import os

import numpy as np
import pytest

import ldpc_toolbox_amd as lt

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))


def toy_alist():
    t = KATS["decoder_toy"]
    h = lt.SparseMatrix(len(t["rows"]), t["ncols"])
    for r, cols in enumerate(t["rows"]):
        h.insert_row(r, cols)
    return h.alist()


def to_llrs(bits, mag):
    return np.array([mag if b == 0 else -mag for b in bits])


def test_reference_no_errors(oracle):
    """src/decoder/flooding.rs:161-172"""
    t = KATS["decoder_toy"]
    dec = oracle.Decoder(oracle.Graph(toy_alist()), t["implementation"])
    ok, bits, it, _ = dec.decode(to_llrs(t["codeword"], t["llr_magnitude"]), t["max_iterations"])
    assert ok and list(bits) == t["codeword"] and it == t["no_errors_iterations"]


def test_reference_single_error(oracle):
    """src/decoder/flooding.rs:174-189 -- the decoder object is reused across the six decodes"""
    t = KATS["decoder_toy"]
    dec = oracle.Decoder(oracle.Graph(toy_alist()), t["implementation"])
    for j in range(len(t["codeword"])):
        bad = list(t["codeword"])
        bad[j] ^= 1
        ok, bits, it, _ = dec.decode(to_llrs(bad, t["llr_magnitude"]), t["max_iterations"])
        assert ok and list(bits) == t["codeword"] and it == t["single_error_iterations"]


def test_reference_depuncture(oracle):
    """src/simulation/puncturing.rs:118-129"""
    p = KATS["puncturing"]
    assert list(oracle.depuncture(p["pattern"], p["llrs"])) == p["depunctured"]
    with pytest.raises(ValueError):
        oracle.depuncture(p["pattern"], p["llrs"][:5])


def test_names(oracle):
    g = oracle.Graph(toy_alist())
    for name in lt.ALL_IMPLEMENTATIONS:
        oracle.Decoder(g, name)
    assert len(lt.ALL_IMPLEMENTATIONS) == 40          # the reference's 36 + 4 Minsum names
    for bad in ("Phif16", "phif64", "HLPhi", "Minstarapproxi8Deg1ClipJones", "HLAminstari8Jones", ""):
        with pytest.raises(ValueError):
            oracle.Decoder(g, bad)


def test_graph_reads_column_section_only(oracle):
    """src/sparse.rs:352-389: padding zeros ignored, row section never read"""
    a = KATS["alist_irregular_padded"]
    g = oracle.Graph(a)
    assert (g.rows, g.cols, g.edges) == (4, 12, 10)
    truncated = "\n".join(a.split("\n")[:4 + 12]) + "\n"      # drop the row lists entirely
    g2 = oracle.Graph(truncated)
    assert (g2.rows, g2.cols, g2.edges) == (4, 12, 10)
    with pytest.raises(ValueError):
        oracle.Graph("\n".join(a.split("\n")[:10]))            # missing column lines


def test_degree_one_check_is_a_panic(oracle):
    """arithmetic.rs:513-514: Minstarapprox on a degree-1 check panics; Phi does not"""
    a = "3 2\n1 2\n1 1 1\n2 1\n1\n1\n2\n1 2\n3\n"
    g = oracle.Graph(a)
    llr = np.array([-1.0, 2.0, -0.5])
    with pytest.raises(RuntimeError):
        oracle.Decoder(g, "Minstarapproxf32").decode(llr, 3)
    oracle.Decoder(g, "Phif32").decode(llr, 3)


def test_max_iterations_zero(oracle):
    """flooding reports its never-written output_llrs (all bits 1 on a fresh decoder),
    layered reports hard(Qv = input) -- SURVEY.md Appendix A.0"""
    g = oracle.Graph(toy_alist())
    llr = to_llrs([1, 0, 1, 0, 1, 1], 1.0)
    ok, bits, it, _ = oracle.Decoder(g, "Phif32").decode(llr, 0)
    assert not ok and it == 0 and bits.all()
    ok, bits, it, _ = oracle.Decoder(g, "HLPhif32").decode(llr, 0)
    assert not ok and it == 0 and list(bits) == [1, 0, 1, 0, 1, 1]


def test_minsum_closed_form(oracle):
    """Minsum (new rule): fold form == min1/min2/first-argmin closed form"""
    rng = np.random.default_rng(1)
    h = lt.SparseMatrix(1, 7)
    h.insert_row(0, range(7))
    g = oracle.Graph(h.alist())
    dec = oracle.Decoder(g, "Minsumf64")
    for trial in range(50):
        x = np.round(rng.normal(0, 2, 7), 1)      # ties happen
        x[x == 0] = 0.5
        if (x < 0).sum() % 2 == 0:
            x[0] = -x[0]                          # odd parity: not a codeword, one iteration runs
        ok, bits, it, post = dec.decode(x, 1)
        a = np.abs(x)
        p = int(np.argmin(a))
        m1 = a[p]
        m2 = np.min(np.delete(a, p))
        tot = (x < 0).sum() % 2
        c2v = np.array([(m2 if i == p else m1) * (-1 if (tot ^ int(x[i] < 0)) else 1) for i in range(7)])
        assert np.array_equal(post, x + c2v)


def test_committed_oracle_vectors(oracle):
    v = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    from frames import alist
    g = oracle.Graph(alist(str(v["spec"])))
    for impl in lt.ALL_IMPLEMENTATIONS:
        bits, its, post = oracle.decode_batch(g, impl, v["llrs"], int(v["max_iterations"]), threads=4)
        assert np.array_equal(its, v[impl + "/iterations"]), impl
        assert np.array_equal(np.packbits(bits, axis=1), v[impl + "/bits"]), impl
        ref = v[impl + "/posterior"]
        assert np.array_equal(post.astype(ref.dtype), ref), impl
    assert v["Minsumf32/iterations"][0] == 0          # clean all-zero codeword: pre-check


def test_philox_known_answers(oracle):
    """Philox4x32-10 known-answer vectors of the Random123 distribution (kat_vectors)"""
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        assert tuple(int(x) for x in oracle.philox4x32_10(ctr, key)) == want


def test_frame_generator_statistics(oracle):
    """the generated LLRs are -2 y / sigma^2 with y = +-1 + N(0, sigma^2): check the noise moments"""
    rng = np.random.default_rng(0)
    tx = rng.integers(0, 2, (4, 4096), dtype=np.uint8)
    rate, ebn0 = 0.5, 1.0
    sigma = np.sqrt(0.5 / (rate * 10 ** 0.1))
    llrs, idx = oracle.generate_llrs(tx, rate, ebn0, seed=12345, first_frame=10, frames=64)
    assert set(idx.tolist()) <= {0, 1, 2, 3} and len(set(idx.tolist())) > 1
    y = llrs.astype(np.float64) * (-(sigma ** 2) / 2)
    z = (y - np.where(tx[idx] == 1, 1.0, -1.0)) / sigma
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs(np.mean(z ** 4) - 3.0) < 0.1                    # Gaussian kurtosis
    assert np.abs(z).max() > 4.0                               # tails are there
    again, _ = oracle.generate_llrs(tx, rate, ebn0, seed=12345, first_frame=10, frames=64)
    assert np.array_equal(llrs, again)
    shifted, _ = oracle.generate_llrs(tx, rate, ebn0, seed=12345, first_frame=12, frames=8)
    assert np.array_equal(shifted, llrs[2:10])                 # counter-based: frame = f(seed, index)


def test_oracle_syndrome_matches_matrix_product(oracle):
    """oracle_syndrome (decoder.rs:157-164 with the parities kept) = H . bits over GF(2): checked
    against an independent scipy product, and zero on the reference KAT's codeword."""
    import scipy.sparse as sp
    for spec in ("ar4ja:1/2:1024", "nr5g:2:24"):
        a = lt.code_alist(spec)
        g = oracle.Graph(a)
        lines = a.split("\n")
        n, m = (int(x) for x in lines[0].split())
        rows, cols = [], []
        for c in range(n):
            for r in lines[4 + c].split():
                if int(r) > 0:
                    rows.append(int(r) - 1)
                    cols.append(c)
        H = sp.csr_matrix((np.ones(len(rows), np.int64), (rows, cols)), shape=(m, n))
        bits = np.random.default_rng(5).integers(0, 2, size=(7, n)).astype(np.uint8)
        bits[0] = 0
        syn, weight = oracle.syndrome(g, bits)
        want = (H @ bits.T.astype(np.int64) % 2).T.astype(np.uint8)
        assert np.array_equal(syn, want) and np.array_equal(weight, want.sum(axis=1))
        assert weight[0] == 0 and weight[1:].min() > 0
    t = KATS["decoder_toy"]                                       # src/decoder/flooding.rs:161-172
    syn, weight = oracle.syndrome(oracle.Graph(toy_alist()), np.array([t["codeword"]], dtype=np.uint8))
    assert weight[0] == 0 and not syn.any()
