"""Host-side frame pipeline and BER bookkeeping around the decode path.

Mirrors the reference's simulation driver for the AWGN/BPSK case:
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


def generate_frames(encode, k: int, batch: int, sigma: float, seed: int, first_frame: int = 0,
                    pattern=None, dtype=np.float32):
    """`encode(message[k]) -> codeword[n]` (u8).  Returns (messages [B][k] u8, llrs [B][n_tx]).

    The LLRs are what the decoder boundary takes: still punctured (the decoder depunctures).
    """
    rng = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFFFFFFFFFF, first_frame]))
    messages = rng.integers(0, 2, size=(batch, k), dtype=np.uint8)
    codewords = np.stack([encode(m) for m in messages])
    tx = puncture(codewords, pattern) if pattern else codewords
    symbols = bpsk_modulate(tx)
    if sigma > 0.0:
        symbols = symbols + sigma * rng.standard_normal(symbols.shape)
    elif sigma < 0.0:
        raise ValueError("noise sigma must be non-negative")  # channel.rs:100-113
    llrs = bpsk_demodulate(symbols, sigma if sigma > 0.0 else 1.0)
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


def fold_statistics(ebn0_db, k, messages, bits, iterations, max_iterations, elapsed) -> Statistics:
    """Counters of ber.rs:313-338 over a decoded batch.  iterations: -1 = failed."""
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
    return st
