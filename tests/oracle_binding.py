"""ctypes binding of oracle/libldpc_oracle.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package (ldpc_toolbox_amd)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")
_LIB = None


def build():
    subprocess.run(["make", "-C", _DIR], check=True, capture_output=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_DIR, "libldpc_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_graph_from_alist.restype = C.c_void_p
        L.oracle_graph_from_alist.argtypes = [C.c_char_p]
        L.oracle_graph_free.argtypes = [C.c_void_p]
        for f in ("rows", "cols", "edges"):
            fn = getattr(L, "oracle_graph_" + f)
            fn.restype = C.c_size_t
            fn.argtypes = [C.c_void_p]
        L.oracle_decoder_new.restype = C.c_void_p
        L.oracle_decoder_new.argtypes = [C.c_void_p, C.c_char_p]
        L.oracle_decoder_free.argtypes = [C.c_void_p]
        L.oracle_decode.restype = C.c_int
        L.oracle_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.POINTER(C.c_uint32)]
        L.oracle_decode_batch_f32.restype = C.c_int
        L.oracle_decode_batch_f32.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t,
                                              C.c_uint32, C.c_uint, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
        L.oracle_decode_batch_timed_f32.restype = C.c_int
        L.oracle_decode_batch_timed_f32.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t,
                                                    C.c_uint32, C.c_uint, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.POINTER(C.c_double)]
        L.oracle_depuncture.restype = C.c_size_t
        L.oracle_depuncture.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_size_t]
        L.oracle_syndrome.restype = C.c_size_t
        L.oracle_syndrome.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_interleave_u8.restype = C.c_int
        L.oracle_interleave_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        L.oracle_deinterleave_f64.restype = C.c_int
        L.oracle_deinterleave_f64.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        L.oracle_psk8_modulate.restype = None
        L.oracle_psk8_modulate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_psk8_demodulate.restype = None
        L.oracle_psk8_demodulate.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_void_p]
        L.oracle_generate_llrs_psk8.restype = None
        L.oracle_generate_llrs_psk8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_int32,
                                                C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_philox4x32_10.restype = None
        L.oracle_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_generate_llrs.restype = None
        L.oracle_generate_llrs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_uint64,
                                           C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class Graph:
    def __init__(self, alist: str):
        self._h = lib().oracle_graph_from_alist(alist.encode())
        if not self._h:
            raise ValueError("oracle: malformed alist")
        self.rows = lib().oracle_graph_rows(self._h)
        self.cols = lib().oracle_graph_cols(self._h)
        self.edges = lib().oracle_graph_edges(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_graph_free(self._h)
            self._h = None


class Decoder:
    """One-codeword decoder with the reference's LdpcDecoder::decode contract."""

    def __init__(self, graph: Graph, implementation: str):
        self.graph = graph
        self._h = lib().oracle_decoder_new(graph._h, implementation.encode())
        if not self._h:
            raise ValueError("oracle: invalid decoder implementation")

    def decode(self, llrs, max_iterations):
        llrs = np.ascontiguousarray(llrs, dtype=np.float64)
        n = llrs.shape[0]
        bits = np.zeros(n, dtype=np.uint8)
        post = np.zeros(n, dtype=np.float64)
        it = C.c_uint32(0)
        ok = lib().oracle_decode(self._h, llrs.ctypes.data, n, max_iterations, bits.ctypes.data,
                                 post.ctypes.data, C.byref(it))
        if ok < 0:
            raise RuntimeError("oracle: reference would panic on this input")
        return bool(ok), bits, int(it.value), post

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_decoder_free(self._h)
            self._h = None


def decode_batch(graph: Graph, implementation: str, llrs, max_iterations, threads=1,
                 want_posterior=True):
    """llrs [B][n] f32 -> (bits [B][n] u8, iterations [B] i32 (-1 = failed), posterior [B][n] f64)"""
    llrs = np.ascontiguousarray(llrs, dtype=np.float32)
    B, n = llrs.shape
    assert n == graph.cols
    bits = np.zeros((B, n), dtype=np.uint8)
    its = np.zeros(B, dtype=np.int32)
    post = np.zeros((B, n), dtype=np.float64) if want_posterior else None
    rc = lib().oracle_decode_batch_f32(graph._h, implementation.encode(), llrs.ctypes.data, B,
                                       max_iterations, threads, bits.ctypes.data, its.ctypes.data,
                                       post.ctypes.data if want_posterior else None)
    if rc != 0:
        raise RuntimeError("oracle batch decode failed")
    return bits, its, post


def decode_batch_timed(graph: Graph, implementation: str, llrs, max_iterations, threads=1):
    """decode_batch without the posterior, timed inside the oracle: every worker thread builds its
    decoder first, all start together; returns (bits, iterations, decode-only wall seconds)"""
    llrs = np.ascontiguousarray(llrs, dtype=np.float32)
    B, n = llrs.shape
    assert n == graph.cols
    bits = np.zeros((B, n), dtype=np.uint8)
    its = np.zeros(B, dtype=np.int32)
    seconds = C.c_double(0.0)
    rc = lib().oracle_decode_batch_timed_f32(graph._h, implementation.encode(), llrs.ctypes.data, B,
                                             max_iterations, threads, bits.ctypes.data, its.ctypes.data,
                                             None, C.byref(seconds))
    if rc != 0:
        raise RuntimeError("oracle batch decode failed")
    return bits, its, seconds.value


def syndrome(graph: Graph, bits):
    """bits [B][n] u8 -> (syndrome [B][m] u8, weight [B])"""
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    B, n = bits.shape
    assert n == graph.cols
    syn = np.zeros((B, graph.rows), dtype=np.uint8)
    weight = np.zeros(B, dtype=np.uint32)
    for b in range(B):
        weight[b] = lib().oracle_syndrome(graph._h, bits[b].ctypes.data, syn[b].ctypes.data)
    return syn, weight


def depuncture(pattern, llrs):
    pattern = np.ascontiguousarray(pattern, dtype=np.uint8)
    llrs = np.ascontiguousarray(llrs, dtype=np.float64)
    trues = int(pattern.sum())
    cap = (len(llrs) // max(trues, 1) + 1) * len(pattern)
    out = np.zeros(cap, dtype=np.float64)
    n = lib().oracle_depuncture(pattern.ctypes.data, len(pattern), llrs.ctypes.data, len(llrs),
                                out.ctypes.data, cap)
    if n == 0:
        raise ValueError("codeword size not divisible by puncturing pattern length")
    return out[:n]


def philox4x32_10(counter, key):
    c = np.ascontiguousarray(counter, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().oracle_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def generate_llrs(tx_bits, rate, ebn0_db, seed, first_frame, frames):
    """tx_bits [pool][n_tx] u8 -> (llrs [frames][n_tx] f32, pool index [frames])"""
    tx_bits = np.ascontiguousarray(tx_bits, dtype=np.uint8)
    pool, n_tx = tx_bits.shape
    llrs = np.zeros((frames, n_tx), dtype=np.float32)
    idx = np.zeros(frames, dtype=np.uint32)
    lib().oracle_generate_llrs(tx_bits.ctypes.data, pool, n_tx, rate, ebn0_db, seed, first_frame, frames,
                               llrs.ctypes.data, idx.ctypes.data)
    return llrs, idx


def interleave(bits, columns, backwards=False):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros_like(bits)
    if lib().oracle_interleave_u8(bits.ctypes.data, len(bits), columns, int(backwards), out.ctypes.data) != 0:
        raise ValueError("codeword size not divisible by the interleaver columns")
    return out


def deinterleave(values, columns, backwards=False):
    values = np.ascontiguousarray(values, dtype=np.float64)
    out = np.zeros_like(values)
    if lib().oracle_deinterleave_f64(values.ctypes.data, len(values), columns, int(backwards), out.ctypes.data) != 0:
        raise ValueError("codeword size not divisible by the interleaver columns")
    return out


def psk8_modulate(bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(2 * (len(bits) // 3), dtype=np.float64)
    lib().oracle_psk8_modulate(bits.ctypes.data, len(bits) // 3, out.ctypes.data)
    return out[0::2] + 1j * out[1::2]


def psk8_demodulate(symbols, sigma):
    symbols = np.asarray(symbols, dtype=np.complex128)
    re_im = np.ascontiguousarray(np.stack([symbols.real, symbols.imag], axis=-1).reshape(-1))
    out = np.zeros(3 * len(symbols), dtype=np.float64)
    lib().oracle_psk8_demodulate(re_im.ctypes.data, len(symbols), float(sigma), out.ctypes.data)
    return out


def generate_llrs_psk8(tx_bits, rate, ebn0_db, interleaving, seed, first_frame, frames):
    """tx_bits [pool][n_tx] u8 -> (llrs [frames][n_tx] f32, pool index [frames])"""
    tx_bits = np.ascontiguousarray(tx_bits, dtype=np.uint8)
    pool, n_tx = tx_bits.shape
    llrs = np.zeros((frames, n_tx), dtype=np.float32)
    idx = np.zeros(frames, dtype=np.uint32)
    lib().oracle_generate_llrs_psk8(tx_bits.ctypes.data, pool, n_tx, rate, ebn0_db, int(interleaving), seed,
                                    first_frame, frames, llrs.ctypes.data, idx.ctypes.data)
    return llrs, idx
