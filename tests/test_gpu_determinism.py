"""Repeated-run determinism of the row-record check-node kernel (cn_minsum_rec_kernel), every instantiated variant.

Round 4 met results that differed from run to run in two variants of that kernel (codeword 2 of lanes 12-15 of every 16).
Round 5 found the cause in the ISA (tools/mb/store_hazard_scan.py, tools/mb/store_hazard_repro.hip): a
128-bit buffer store whose data register the next vector instruction rewrites -- a hazard the compiler does not pad for the
SGPR-soffset form the library uses.  A hazard of that kind shows as run-to-run differences, so: the same batch decoded 20
times per variant must give 20 identical results (hard decisions, iteration counts, posterior LLRs), equal to the
per-edge-message kernels' (records = 0), which share none of the record code.
"""
import os

import numpy as np
import pytest

import ldpc_toolbox_amd as lt
from frames import alist, awgn_frames

pytestmark = pytest.mark.gpu

REPEATS = 20
# (LDPC_DET_BATCH: experiment runs at sizes where the stores queue up behind HBM -- tools/mb/store_hazard_gpu.sh)
BATCH = int(os.environ.get("LDPC_DET_BATCH", "768"))

# (code, Eb/N0, rule): rows of at most 8 edges (LONG = false) / longer rows (LONG = true) / rows beyond the 3-word record's
# sign field (RECW = 4: more than 26 edges in f32)
CASES = [("nr5g:2:24", 1.5, "Minsumf32"), ("nr5g:2:24", 1.5, "Minsumf64"), ("dvbs2:R1_2short", 1.3, "Minsumf32"),
         ("dvbs2:R1_2short", 1.3, "Minsumf64"), ("dvbs2:R8_9short", 4.0, "Minsumf32")]
# rec_run 1 and "vec" 4 / 2 / 1 are the run lengths and pack widths the kernel is instantiated for; rec_long forces the LONG
# variant on short rows; compact 0/1 covers both store paths of the L-free posterior
OPTIONS = [{"rec_run": 1}, {"rec_run": 8}, {"rec_run": 3, "vec": 2}, {"rec_run": 64, "vec": 1}, {"rec_run": 1, "rec_long": 1},
           {"rec_run": 8, "rec_quiet": 0, "compact": 0}, {"rec_run": 8, "vn_event": 0}]


def where(a, b, vec=4):
    bad = np.argwhere(np.atleast_2d(a != b).reshape(len(a), -1).any(axis=1)).ravel()
    return [(int(i), f"lane {(i // vec) % 64}, element {i % vec}") for i in bad[:12]]


@pytest.mark.parametrize("spec,ebn0,impl", CASES)
def test_row_record_variants_are_deterministic(spec, ebn0, impl):
    msgs, llrs, full = awgn_frames(spec, BATCH, ebn0, 99)
    gpu_in = llrs.astype(np.float64) if impl.endswith("f64") else llrs
    dec = lt.LdpcDecoder(alist(spec), impl)
    dec.set("group_size", min(BATCH, 4096))
    dec.set("records", 0)
    ref = dec.decode_batch(gpu_in, 30, want_posterior=True)
    assert 0 < (ref[1] >= 0).sum()
    for opts in OPTIONS:
        for key, v in {"records": 2, "rec_quiet": 1, "compact": 1, "vec": 0, "rec_long": 0, "vn_event": 1, **opts}.items():
            dec.set(key, v)
        assert dec.get("row_records") in (3, 4)
        for rep in range(REPEATS):
            got = dec.decode_batch(gpu_in, 30, want_posterior=True)
            for name, a, b in zip(("bits", "iterations", "posterior"), ref, got):
                assert np.array_equal(a, b), (opts, rep, name, where(a, b, opts.get("vec", 4) or 4))


@pytest.mark.parametrize("batch", [1, 2, 3, 4097])
def test_group_that_finishes_inside_the_variable_node_launch(oracle, batch):
    """The variable-node launch that also rebuilds the L-free posteriors of a slice's first convergences (vn_kernel<..., EVW>)
    subtracts the codewords it latches from the group's running count in that same launch.  When the LAST running codewords
    of a group converge there -- a group of one to three codewords, the one-codeword tail group of a batch of G + 1, or a full
    group whose slowest slices hold one frame each -- the count reaches zero while most of the launch's ~32 K waves have not
    started yet; a wave that then returned on "nothing running" skipped its share of the rebuild, and the staircase parity
    bits / their posteriors of exactly those codewords came back stale (round 5's advisor finding; timing dependent).
    DVB-S2 normal frames (far more waves than the chip holds at once), all n hard bits and the posterior, the batched
    kernels forced on small calls (latency = 0), many repeats, against the per-edge kernels and the oracle."""
    import torch
    spec, impl, max_it = "dvbs2:R1_2", "Minsumf32", 50
    distinct = min(batch, 8)
    msgs, llrs, full = awgn_frames(spec, distinct, 2.2, 4242)
    # the large batch: slices (256 codewords) of ONE frame each, so that a slice's first convergences are all of its codewords
    idx = (np.arange(batch) // 256) % distinct if batch > distinct else np.arange(batch)
    obits, oits, opost = oracle.decode_batch(oracle.Graph(alist(spec)), impl, full, max_it, threads=8)
    assert (oits > 0).all(), "the operating point must converge after at least one iteration"
    want = (obits[idx], oits[idx], opost[idx].astype(np.float32))
    dec = lt.LdpcDecoder(alist(spec), impl)
    dec.set("latency", 0)
    d_llrs = torch.from_numpy(llrs[idx]).cuda()
    d_bits = torch.zeros((batch, dec.n), dtype=torch.uint8, device="cuda")
    d_its = torch.zeros(batch, dtype=torch.int32, device="cuda")
    d_post = torch.zeros((batch, dec.n), dtype=torch.float32, device="cuda")

    def decode(host):
        if host:
            return dec.decode_batch(llrs[idx], max_it, want_posterior=True, output_len=dec.n)
        d_bits.zero_(), d_its.zero_(), d_post.zero_()
        torch.cuda.synchronize()
        dec.decode_batch_device(d_llrs.data_ptr(), False, batch, max_it, d_bits.data_ptr(), dec.n, d_its.data_ptr(), d_post.data_ptr(), 0)
        torch.cuda.synchronize()
        return d_bits.cpu().numpy(), d_its.cpu().numpy(), d_post.cpu().numpy()

    dec.set("records", 0)
    for host in (True, False):
        for name, a, b in zip(("bits", "iterations", "posterior"), decode(host), want):
            assert np.array_equal(a, b), ("per-edge kernels against the oracle", host, name)
    dec.set("records", 2)
    for key, v in {"rec_quiet": 1, "vn_event": 1, "compact": 1}.items():
        dec.set(key, v)
    assert dec.get("row_records") == 3
    for rep in range(10 if batch > 8 else 30):
        for host in (True, False):
            for name, a, b in zip(("bits", "iterations", "posterior"), decode(host), want):
                assert np.array_equal(a, b), (batch, rep, "host entry" if host else "device entry", name, where(a, b))


def test_streaming_record_variant_is_deterministic():
    """the STREAM instantiation (continuous batching through the simulator; -DLDPC_EXPERIMENTS builds only since round 5):
    same counters 20 times"""
    if lt.LdpcDecoder(alist("ar4ja:1/2:1024"), "Minsumf32").get("experiments") != 1:
        pytest.skip("continuous batching exists in -DLDPC_EXPERIMENTS builds only")
    s = lt.Simulator(alist("dvbs2:R1_2short"), "Minsumf32", "", device=0, pool_size=16, pool_seed=9)
    s.set("records", 2)
    s.set("streaming", 1)
    first = None
    for rep in range(REPEATS):
        c = list(s.run(1.5, seed=5, first_frame=3, frames=4096 + 1500, max_iterations=25))
        assert s.get("streamed_frames") == 4096 + 1500
        first = first or c
        assert c == first, rep


def test_one_wait_state_is_enough_behind_an_sgpr_soffset_store():
    """What store_data_pad relies on, checked on the hardware under the test: the stand-alone reproducer
    (tools/mb/store_hazard_repro.hip, built by __graft_entry__.build()) rewrites a 128-bit buffer store's data register N
    wait states behind the store.  With the soffset in an SGPR (the library's form) one wait state must leave every stored
    word intact, with a literal soffset two (the compiler pads those).  Whether N = 0 corrupts (it does on the MI355X boxes
    of the pool: 0.5 % of the stores, lanes 12-15) is reported, not asserted."""
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "mb", "store_hazard_repro")
    if not os.path.exists(exe):
        pytest.skip("tools/mb/store_hazard_repro not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    seen = 0
    for ln in r.stdout.splitlines():
        m = re.match(r"soffset=(SGPR|0)\s+nt=\d clobbered element \d, (\d) wait state\(s\):\s+(\d+) poisoned of \d+ words, (\d+) otherwise wrong", ln)
        if not m:
            continue
        seen += 1
        form, waits, poisoned, wrong = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
        assert wrong == 0, ln
        if waits >= (1 if form == "SGPR" else 2):
            assert poisoned == 0, ln
    assert seen >= 36
    print("\n".join(ln for ln in r.stdout.splitlines() if ", 0 wait state(s)" in ln)[:1500])
