"""Host-side frame pipeline and BER bookkeeping around the decode path.

Mirrors the reference's simulation driver over AWGN, BPSK and 8PSK (with the DVB-S2 bit
interleaver, simulation/interleaving.rs, modulation.rs:144-288):
  * sigma from Eb/N0:   /root/reference/src/simulation/ber.rs:299-302 (rate = k / n_tx)
  * one frame:          ber.rs:436-481  (random message -> encode -> puncture -> BPSK ->
                        AWGN -> LLR -> depuncture -> decode -> errors on the first k bits)
  * BPSK mapping / LLR: simulation/modulation.rs:87-95 (bit 0 -> -1... see note) and :127-140
  * statistics:         ber.rs:551-581
The reference draws from an OS-seeded ThreadRng (ber.rs:419), so its runs are not
reproducible; here every frame comes from a counter-based Philox stream keyed by
(seed, first frame index), which is what lets the GPU path and the CPU oracle be compared
on identical frames.

Note on the BPSK sign convention: the reference maps bit 1 -> +1.0 and bit 0 -> -1.0
(modulation.rs:87-95) and demodulates LLR = -2 y / sigma^2 (modulation.rs:127, 140), so a
received -1 (bit 0) gives a positive LLR: LLR > 0 <=> bit 0, as the decoder expects.
"""
from dataclasses import dataclass, field

import numpy as np


def noise_sigma(rate: float, ebn0_db: float, bits_per_symbol: float = 1.0) -> float:
    ebn0 = 10.0 ** (0.1 * float(ebn0_db))
    esn0 = rate * bits_per_symbol * ebn0
    return float(np.sqrt(0.5 / esn0))


def parse_puncturing_pattern(s: str):
    """src/cli/ber.rs:219-229"""
    out = []
    for tok in s.split(","):
        if tok == "0":
            out.append(False)
        elif tok == "1":
            out.append(True)
        else:
            raise ValueError("invalid puncturing pattern")
    return out


def puncture(codewords: np.ndarray, pattern) -> np.ndarray:
    """simulation/puncturing.rs:47-75 on a [B][n] array"""
    n = codewords.shape[-1]
    if n % len(pattern) != 0:
        raise ValueError("codeword size not divisible by puncturing pattern length")
    bs = n // len(pattern)
    keep = [codewords[..., k * bs:(k + 1) * bs] for k, p in enumerate(pattern) if p]
    return np.concatenate(keep, axis=-1)


def depuncture(llrs: np.ndarray, pattern) -> np.ndarray:
    """simulation/puncturing.rs:83-101 on a [B][n_tx] array: punctured blocks become 0.0"""
    trues = sum(bool(p) for p in pattern)
    if llrs.shape[-1] % trues != 0:
        raise ValueError("codeword size not divisible by puncturing pattern length")
    bs = llrs.shape[-1] // trues
    out = np.zeros(llrs.shape[:-1] + (bs * len(pattern),), dtype=llrs.dtype)
    j = 0
    for k, p in enumerate(pattern):
        if p:
            out[..., k * bs:(k + 1) * bs] = llrs[..., j * bs:(j + 1) * bs]
            j += 1
    return out


def bpsk_modulate(bits: np.ndarray) -> np.ndarray:
    return np.where(bits == 1, 1.0, -1.0)


def bpsk_demodulate(symbols: np.ndarray, sigma: float) -> np.ndarray:
    return -2.0 * symbols / (sigma * sigma)


class Interleaver:
    """simulation/interleaving.rs:20-87: matrix interleaver of the DVB-S2 standard.  `columns` is
    usually the number of bits per symbol; `read_rows_backwards` is DVB-S2's 8PSK rate-3/5 case.
    Works on the last axis of an array."""

    def __init__(self, columns: int, read_rows_backwards: bool = False):
        if columns <= 0:
            raise ValueError("interleaver columns must be positive")
        self.columns, self.read_rows_backwards = int(columns), bool(read_rows_backwards)

    @staticmethod
    def from_signed(columns: int):
        """ber.rs:250-252: Interleaver::new(|n|, n < 0); 0 / None = no interleaver"""
        return Interleaver(abs(int(columns)), columns < 0) if columns else None

    def interleave(self, codeword: np.ndarray) -> np.ndarray:
        n = codeword.shape[-1]
        if n % self.columns != 0:
            raise ValueError("codeword size not divisible by the interleaver columns")  # the reference asserts
        a = codeword.reshape(codeword.shape[:-1] + (self.columns, n // self.columns))
        t = np.swapaxes(a, -1, -2)                     # [rows][columns]
        if self.read_rows_backwards:
            t = t[..., ::-1]
        return np.ascontiguousarray(t).reshape(codeword.shape)

    def deinterleave(self, codeword: np.ndarray) -> np.ndarray:
        n = codeword.shape[-1]
        if n % self.columns != 0:
            raise ValueError("codeword size not divisible by the interleaver columns")
        a = codeword.reshape(codeword.shape[:-1] + (n // self.columns, self.columns))
        t = np.swapaxes(a, -1, -2)                     # [columns][rows]
        if self.read_rows_backwards:
            t = t[..., ::-1, :]
        return np.ascontiguousarray(t).reshape(codeword.shape)


_PSK8_A = float(np.sqrt(0.5))
# modulation.rs:166-179, indexed by b0 + 2 b1 + 4 b2
_PSK8_POINTS = np.array([complex(_PSK8_A, _PSK8_A), complex(0.0, 1.0), complex(-1.0, 0.0), complex(-_PSK8_A, _PSK8_A),
                         complex(1.0, 0.0), complex(_PSK8_A, -_PSK8_A), complex(-_PSK8_A, -_PSK8_A), complex(0.0, -1.0)])


def psk8_modulate(bits: np.ndarray) -> np.ndarray:
    """Psk8Modulator (modulation.rs:181-199): [..., 3 S] bits -> [..., S] complex symbols"""
    if bits.shape[-1] % 3 != 0:
        raise ValueError("8PSK needs a multiple of 3 bits")       # the reference asserts
    b = bits.reshape(bits.shape[:-1] + (-1, 3)).astype(np.int64)
    return _PSK8_POINTS[b[..., 0] + 2 * b[..., 1] + 4 * b[..., 2]]


def _maxstar(a, b):
    """modulation.rs:286-288"""
    return np.maximum(a, b) + np.log1p(np.exp(-np.abs(a - b)))


def psk8_demodulate(symbols: np.ndarray, sigma: float) -> np.ndarray:
    """Psk8Demodulator (modulation.rs:211-281): exact max* LLRs, [..., S] complex -> [..., 3 S]"""
    s = symbols * (1.0 / (sigma * sigma))
    d = {}
    for key, (b0, b1, b2) in {"000": (0, 0, 0), "100": (1, 0, 0), "110": (1, 1, 0), "010": (0, 1, 0), "011": (0, 1, 1),
                               "111": (1, 1, 1), "101": (1, 0, 1), "001": (0, 0, 1)}.items():
        p = _PSK8_POINTS[b0 + 2 * b1 + 4 * b2]
        d[key] = s.real * p.real + s.imag * p.imag

    def fold(keys):
        acc = d[keys[0]]
        for k in keys[1:]:
            acc = _maxstar(acc, d[k])
        return acc

    b0 = fold(["000", "001", "010", "011"]) - fold(["100", "101", "110", "111"])
    b1 = fold(["000", "001", "100", "101"]) - fold(["010", "011", "110", "111"])
    b2 = fold(["000", "010", "100", "110"]) - fold(["001", "011", "101", "111"])
    return np.stack([b0, b1, b2], axis=-1).reshape(symbols.shape[:-1] + (-1,))


MODULATIONS = {"BPSK": 1, "8PSK": 3}      # simulation/factory.rs:53-73, bits per symbol


def generate_frames(encode, k: int, batch: int, sigma: float, seed: int, first_frame: int = 0,
                    pattern=None, dtype=np.float32, modulation: str = "BPSK", interleaver=None):
    """`encode(message[k]) -> codeword[n]` (u8).  Returns (messages [B][k] u8, llrs [B][n_tx]).

    ber.rs:436-456: encode -> puncture -> interleave -> modulate -> AWGN -> demodulate ->
    deinterleave.  The LLRs are what the decoder boundary takes: still punctured (the decoder
    depunctures).
    """
    if modulation not in MODULATIONS:
        raise ValueError(f"invalid modulation {modulation}")      # factory.rs:70
    rng = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFFFFFFFFFF, first_frame]))
    messages = rng.integers(0, 2, size=(batch, k), dtype=np.uint8)
    codewords = np.stack([encode(m) for m in messages])
    tx = puncture(codewords, pattern) if pattern else codewords
    if interleaver is not None:
        tx = interleaver.interleave(tx)
    if sigma < 0.0:
        raise ValueError("noise sigma must be non-negative")  # channel.rs:100-113
    if modulation == "8PSK":
        symbols = psk8_modulate(tx)
        if sigma > 0.0:
            symbols = symbols + sigma * (rng.standard_normal(symbols.shape) + 1j * rng.standard_normal(symbols.shape))
        llrs = psk8_demodulate(symbols, sigma if sigma > 0.0 else 1.0)
    else:
        symbols = bpsk_modulate(tx)
        if sigma > 0.0:
            symbols = symbols + sigma * rng.standard_normal(symbols.shape)
        llrs = bpsk_demodulate(symbols, sigma if sigma > 0.0 else 1.0)
    if interleaver is not None:
        llrs = interleaver.deinterleave(llrs)
    return messages, llrs.astype(dtype)


@dataclass
class CodeStatistics:
    bit_errors: int = 0
    frame_errors: int = 0
    correct_iterations: int = 0
    ber: float = 0.0
    fer: float = 0.0
    average_iterations_correct: float = 0.0


@dataclass
class Statistics:
    """ber.rs:145-189 / :551-581"""
    ebn0_db: float = 0.0
    num_frames: int = 0
    total_iterations: int = 0
    false_decodes: int = 0
    average_iterations: float = 0.0
    elapsed: float = 0.0
    throughput_mbps: float = 0.0
    ldpc: CodeStatistics = field(default_factory=CodeStatistics)
    bch: "CodeStatistics | None" = None      # ber.rs:166: present when bch_max_errors > 0


def _finish_code_statistics(cs: CodeStatistics, k: int, n: int):
    """CodeStatistics::from_current (ber.rs:551-563)"""
    cs.ber = cs.bit_errors / (k * n) if n else 0.0
    cs.fer = cs.frame_errors / n if n else 0.0
    good = n - cs.frame_errors
    cs.average_iterations_correct = cs.correct_iterations / good if good else float("nan")


def fold_statistics(ebn0_db, k, messages, bits, iterations, max_iterations, elapsed, bch_max_errors: int = 0) -> Statistics:
    """Counters of ber.rs:313-338 over a decoded batch.  iterations: -1 = failed.  With
    bch_max_errors > 0 also the outer-BCH view (:328-337): a frame with at most that many bit errors
    counts as corrected."""
    errs = (bits[:, :k] != messages).sum(axis=1).astype(np.int64)  # first k bits only (:468-472)
    success = iterations >= 0
    its = np.where(success, iterations, max_iterations).astype(np.int64)
    frame_error = errs > 0
    st = Statistics(ebn0_db=ebn0_db, num_frames=int(len(errs)))
    st.total_iterations = int(its.sum())
    st.false_decodes = int((frame_error & success).sum())
    st.ldpc.bit_errors = int(errs.sum())
    st.ldpc.frame_errors = int(frame_error.sum())
    st.ldpc.correct_iterations = int(its[~frame_error].sum())
    st.elapsed = elapsed
    n = st.num_frames
    st.average_iterations = st.total_iterations / n if n else 0.0
    st.throughput_mbps = 1e-6 * k * n / elapsed if elapsed > 0 else 0.0
    st.ldpc.ber = st.ldpc.bit_errors / (k * n) if n else 0.0
    st.ldpc.fer = st.ldpc.frame_errors / n if n else 0.0
    good = n - st.ldpc.frame_errors
    st.ldpc.average_iterations_correct = st.ldpc.correct_iterations / good if good else float("nan")
    if bch_max_errors > 0:
        bad = errs > bch_max_errors
        st.bch = CodeStatistics(bit_errors=int(errs[bad].sum()), frame_errors=int(bad.sum()),
                                correct_iterations=int(its[~bad].sum()))
        _finish_code_statistics(st.bch, k, n)
    return st


def format_header() -> str:
    """src/cli/ber.rs:315-318"""
    return ("  Eb/N0 |   Frames | Bit errs | Frame er | False de |     BER |     FER | Avg iter | Avg corr | Throughp | Elapsed\n"
            "--------|----------|----------|----------|----------|---------|---------|----------|----------|----------|----------")


def format_duration(seconds: float) -> str:
    """humantime::format_duration on whole seconds (cli/ber.rs:339): 0s, 59s, 1m 5s, 2h 3s"""
    sec = int(seconds)
    parts = []
    for unit, size in (("h", 3600), ("m", 60), ("s", 1)):
        if sec >= size:
            parts.append(f"{sec // size}{unit}")
            sec %= size
    return " ".join(parts) if parts else "0s"


def rust_fixed(x: float, width: int, prec: int) -> str:
    """Rust's `{:W.P}` of a float: exact decimal rounding (as Python's), but `NaN` / `inf` spelled Rust's way
    (library/core/src/fmt/float.rs): an all-failing point prints `     NaN` in the Avg-corr column."""
    x = float(x)
    if x != x:
        body = "NaN"
    elif x in (float("inf"), float("-inf")):
        body = "inf" if x > 0 else "-inf"
    else:
        body = f"{x:.{prec}f}"
    return body.rjust(width)


def rust_lower_exp(x: float, width: int, prec: int) -> str:
    """Rust's `{:W.Pe}` (LowerExp) of a float: mantissa with P decimals, then `e` and the exponent with neither a
    plus sign nor padding -- `1.23e-2`, `0.00e0`, `1.00e0` -- where C/Python print `1.23e-02`, `0.00e+00`.  The
    mantissa's rounding is the exact-decimal round-half-even both languages use.  cli/ber.rs:327 (`{:7.2e}`)."""
    x = float(x)
    if x != x:
        body = "NaN"
    elif x in (float("inf"), float("-inf")):
        body = "inf" if x > 0 else "-inf"
    else:
        mant, exp = f"{x:.{prec}e}".split("e")
        body = f"{mant}e{int(exp)}"
    return body.rjust(width)


def format_progress(st: Statistics, force_ldpc: bool = False) -> str:
    """src/cli/ber.rs:320-340: the BCH columns when the run has BCH accounting, unless force_ldpc.  Every float goes
    through Rust's formatting rules so that a results file is byte-diffable with one written by `ldpc-toolbox ber`."""
    cs = st.ldpc if (force_ldpc or st.bch is None) else st.bch
    return (f"{rust_fixed(st.ebn0_db, 7, 2)} | {st.num_frames:8d} | {cs.bit_errors:8d} | {cs.frame_errors:8d} | "
            f"{st.false_decodes:8d} | {rust_lower_exp(cs.ber, 7, 2)} | {rust_lower_exp(cs.fer, 7, 2)} | "
            f"{rust_fixed(st.average_iterations, 8, 1)} | {rust_fixed(cs.average_iterations_correct, 8, 1)} | "
            f"{rust_fixed(st.throughput_mbps, 8, 3)} | {format_duration(st.elapsed)}")


def merge_statistics(a: Statistics, b: Statistics, k: int) -> Statistics:
    """Sum of the raw counters of two partial results at the same Eb/N0 (the fold of
    ber.rs:318-338); derived quantities recomputed (ber.rs:551-581)."""
    out = Statistics(ebn0_db=a.ebn0_db)
    out.num_frames = a.num_frames + b.num_frames
    out.total_iterations = a.total_iterations + b.total_iterations
    out.false_decodes = a.false_decodes + b.false_decodes
    out.elapsed = a.elapsed + b.elapsed
    out.ldpc.bit_errors = a.ldpc.bit_errors + b.ldpc.bit_errors
    out.ldpc.frame_errors = a.ldpc.frame_errors + b.ldpc.frame_errors
    out.ldpc.correct_iterations = a.ldpc.correct_iterations + b.ldpc.correct_iterations
    n = out.num_frames
    out.average_iterations = out.total_iterations / n if n else 0.0
    out.throughput_mbps = 1e-6 * k * n / out.elapsed if out.elapsed > 0 else 0.0
    out.ldpc.ber = out.ldpc.bit_errors / (k * n) if n else 0.0
    out.ldpc.fer = out.ldpc.frame_errors / n if n else 0.0
    good = n - out.ldpc.frame_errors
    out.ldpc.average_iterations_correct = out.ldpc.correct_iterations / good if good else float("nan")
    if a.bch is not None or b.bch is not None:
        za, zb = a.bch or CodeStatistics(), b.bch or CodeStatistics()
        out.bch = CodeStatistics(bit_errors=za.bit_errors + zb.bit_errors, frame_errors=za.frame_errors + zb.frame_errors,
                                 correct_iterations=za.correct_iterations + zb.correct_iterations)
        _finish_code_statistics(out.bch, k, n)
    return out


class BerTest:
    """Batched counterpart of the reference's BerTest (ber.rs:60-96 parameters, :297-368 run
    loop, :522-531 stop rule) for BPSK or 8PSK over AWGN.

    decode(llrs [B][n_tx] f32, max_iterations) -> (bits [B][>=k] u8, iterations [B] i32, -1 =
    failed) is the decode path under test (LdpcDecoder.decode_batch on the GPU).  Frames come in
    batches of `frames_per_batch` from the counter-based generator, so two runs with the same
    seed see the same frames whatever decodes them.  Stop rule per Eb/N0: the reference's
    (frame_errors >= max_frame_errors and elapsed >= min_run_time) or elapsed >= max_run_time,
    evaluated between batches, plus an optional fixed frame count (`max_frames`), which the
    reference lacks and reproducible comparisons need.
    """

    def __init__(self, alist: str, encode, decode, k: int, n: int, ebn0s_db, max_iterations: int = 100,
                 puncturing_pattern=None, max_frame_errors: int = 100, min_run_time: float = 0.0,
                 max_run_time: float = float("inf"), max_frames=None, frames_per_batch: int = 256,
                 seed: int = 0, reporter=None, modulation: str = "BPSK", interleaving_columns=None,
                 bch_max_errors: int = 0):
        if modulation not in MODULATIONS:
            raise ValueError(f"invalid modulation {modulation}")
        self.modulation = modulation
        self.bch_max_errors = int(bch_max_errors)      # ber.rs:93
        self.interleaver = Interleaver.from_signed(interleaving_columns) if interleaving_columns else None
        self.encode, self.decode = encode, decode
        self.k, self.n_cw = k, n
        self.pattern = puncturing_pattern
        rate_p = len(self.pattern) / sum(self.pattern) if self.pattern else 1.0
        self.n = int(round(n / rate_p))          # ber.rs:258
        self.rate = k / self.n                   # ber.rs:259
        self.ebn0s_db = [float(np.float32(e)) for e in ebn0s_db]   # stored as f32 (cli/ber.rs:106-109)
        self.max_iterations = max_iterations
        self.max_frame_errors = max_frame_errors
        self.min_run_time, self.max_run_time = min_run_time, max_run_time
        self.max_frames = max_frames
        self.frames_per_batch = frames_per_batch
        self.seed = seed
        self.reporter = reporter

    def run(self):
        import time
        results = []
        for ebn0_db in self.ebn0s_db:
            sigma = noise_sigma(self.rate, ebn0_db, MODULATIONS[self.modulation])   # ber.rs:301
            total = Statistics(ebn0_db=ebn0_db)
            if self.bch_max_errors > 0:
                total.bch = CodeStatistics()
            start = time.perf_counter()
            first_frame = 0
            while True:
                elapsed = time.perf_counter() - start
                errors = total.bch.frame_errors if total.bch is not None else total.ldpc.frame_errors  # ber.rs:514-520
                if (errors >= self.max_frame_errors and elapsed >= self.min_run_time) \
                        or elapsed >= self.max_run_time:
                    break
                if self.max_frames is not None and total.num_frames >= self.max_frames:
                    break
                nb = self.frames_per_batch
                if self.max_frames is not None:
                    nb = min(nb, self.max_frames - total.num_frames)
                msgs, llrs = generate_frames(self.encode, self.k, nb, sigma, self.seed, first_frame,
                                             pattern=self.pattern, modulation=self.modulation,
                                             interleaver=self.interleaver)
                t0 = time.perf_counter()
                bits, its = self.decode(llrs, self.max_iterations)
                dt = time.perf_counter() - t0
                part = fold_statistics(ebn0_db, self.k, msgs, bits, its, self.max_iterations, dt,
                                       bch_max_errors=self.bch_max_errors)
                total = merge_statistics(total, part, self.k)   # elapsed = decode time only
                first_frame += nb
                if self.reporter:
                    self.reporter(total)
            results.append(total)
        return results
