"""Python mirror of the reference's decoder boundary, over the C ABI of libldpc_toolbox.so.

Names and argument meaning follow the reference so that tests read like its own:
  * `LdpcDecoder.decode(llrs, max_iterations)`  ~ trait LdpcDecoder::decode
    (/root/reference/src/decoder.rs:19-35): returns (ok, DecoderOutput) where ok=True is the
    trait's Ok(..) and ok=False its Err(..);
  * `DecoderOutput(codeword, iterations)`       ~ decoder.rs:38-48 (iterations ==
    max_iterations on failure);
  * `DecoderImplementation(name)`               ~ decoder/factory.rs:31-277, with
    `build_decoder(h)` as in trait DecoderFactory (factory.rs:19-25).
The batched calls (`decode_batch`, `decode_batch_device`) are the extension the GPU needs.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi
from .sparse import SparseMatrix

# implementation names accepted by the HIP path (factory.rs:240-277 float rows + Minsum)
_RULES = ("Phi", "Tanh", "Minstarapprox", "Aminstar", "Minsum")
IMPLEMENTATIONS = tuple(p + r + s for p in ("", "HL") for r in _RULES for s in ("f64", "f32"))
# the reference's 8-bit quantised names (factory.rs:246-263, 270-275)
I8_IMPLEMENTATIONS = tuple(b + j + h + d for b in ("Minstarapproxi8", "Aminstari8") for j in ("", "Jones")
                           for h in ("", "PartialHardLimit") for d in ("", "Deg1Clip")) + (
    "HLMinstarapproxi8", "HLMinstarapproxi8PartialHardLimit", "HLAminstari8", "HLAminstari8PartialHardLimit")
ALL_IMPLEMENTATIONS = IMPLEMENTATIONS + I8_IMPLEMENTATIONS
# opt-in approximate variants (native exp2 / log2 / rcp instead of the glibc-identical functions): NOT bit-identical to
# the reference, never chosen unless asked for by name
FAST_IMPLEMENTATIONS = ("Tanhf32@fast", "HLTanhf32@fast", "Phif32@fast", "HLPhif32@fast")


class DecoderUnavailable(RuntimeError):
    """ctor returned NULL (no GPU, bad alist, unknown implementation, bad pattern)."""


@dataclass
class DecoderOutput:
    codeword: np.ndarray  # u8, one byte per bit, all n bits
    iterations: int


class LdpcDecoder:
    """One Tanner graph + rule + schedule on one GPU."""

    def __init__(self, alist: str, implementation: str, puncturing: str = "", device=None):
        L = _capi.lib()
        if device is None:
            h = L.ldpc_toolbox_decoder_ctor_alist_string(alist.encode(), implementation.encode(),
                                                         puncturing.encode())
        else:
            h = L.ldpc_toolbox_decoder_ctor_alist_string_on_device(
                alist.encode(), implementation.encode(), puncturing.encode(), int(device))
        if not h:
            raise DecoderUnavailable(_capi.last_error() or "decoder constructor returned NULL")
        self._h = h
        self.implementation = implementation
        self.n = self.get("n")
        self.m = self.get("m")
        self.k = self.get("k")
        self.edges = self.get("edges")
        self.input_len = self.get("input_len")
        self.device = self.get("device")

    # -- properties / tunables ---------------------------------------------------------
    def get(self, key: str) -> int:
        v = C.c_int64(0)
        if _capi.lib().ldpc_toolbox_decoder_get(self._h, key.encode(), C.byref(v)) != 0:
            raise KeyError(key)
        return int(v.value)

    def set(self, key: str, value: int):
        if _capi.lib().ldpc_toolbox_decoder_set(self._h, key.encode(), int(value)) != 0:
            raise KeyError(key)

    def kernel_stats(self, kind: int, reset=False):
        n = C.c_uint64(0)
        ms = C.c_double(0.0)
        _capi.lib().ldpc_toolbox_decoder_kernel_stats(self._h, kind, C.byref(n), C.byref(ms), int(reset))
        return int(n.value), float(ms.value)

    # -- the reference's scalar contract ------------------------------------------------
    def decode(self, llrs, max_iterations: int):
        """trait LdpcDecoder::decode: llrs are f64 (f32 arrays go through the _f32 entry)."""
        llrs = np.ascontiguousarray(llrs)
        if llrs.dtype != np.float32:
            llrs = llrs.astype(np.float64, copy=False)
        if llrs.shape != (self.input_len,):
            raise ValueError("LLR length does not match the code")  # the reference asserts
        out = np.zeros(self.n, dtype=np.uint8)
        L = _capi.lib()
        fn = L.ldpc_toolbox_decoder_decode_f32 if llrs.dtype == np.float32 else L.ldpc_toolbox_decoder_decode_f64
        it = fn(self._h, out.ctypes.data, self.n, llrs.ctypes.data, llrs.shape[0], max_iterations)
        if it >= 0:
            return True, DecoderOutput(out, it)
        if it < -1:  # LDPC_TOOLBOX_ERR_*: the call failed (the reference would have panicked), nothing decoded
            raise RuntimeError(_capi.last_error() or f"decode failed with error {it}")
        return False, DecoderOutput(out, max_iterations)

    # -- batched extension ----------------------------------------------------------------
    def decode_batch(self, llrs, max_iterations: int, output_len=None, want_posterior=False, out=None):
        """llrs [B][input_len] f32/f64 host array -> (bits [B][output_len] u8,
        iterations [B] i32 with -1 = failed, posterior [B][n] or None).
        out: (bits, iterations[, posterior]) arrays of those shapes to fill instead of fresh ones (a
        caller that decodes batch after batch reuses its buffers, as a C caller would)."""
        llrs = np.ascontiguousarray(llrs)
        if llrs.dtype not in (np.float32, np.float64):
            llrs = llrs.astype(np.float64)
        B, ln = llrs.shape
        output_len = self.n if output_len is None else output_len
        if out is not None:
            bits, its = out[0], out[1]
            post = out[2] if want_posterior else None
            ok = (bits.shape == (B, output_len) and bits.dtype == np.uint8 and bits.flags.c_contiguous
                  and its.shape == (B,) and its.dtype == np.int32 and its.flags.c_contiguous
                  and (post is None or (post.shape == (B, self.n) and post.dtype == llrs.dtype and post.flags.c_contiguous)))
            if not ok:
                raise ValueError("out arrays do not match the batch")
        else:
            bits = np.zeros((B, output_len), dtype=np.uint8)
            its = np.zeros(B, dtype=np.int32)
            post = np.zeros((B, self.n), dtype=llrs.dtype) if want_posterior else None
        L = _capi.lib()
        fn = (L.ldpc_toolbox_decoder_decode_batch_f32 if llrs.dtype == np.float32
              else L.ldpc_toolbox_decoder_decode_batch_f64)
        rc = fn(self._h, bits.ctypes.data, output_len, llrs.ctypes.data, ln, B, max_iterations,
                its.ctypes.data, post.ctypes.data if want_posterior else None)
        if rc != 0:
            raise RuntimeError(f"decode_batch failed ({rc}): {_capi.last_error()}")
        return bits, its, post

    def decode_batch_device(self, llrs_ptr: int, f64: bool, batch: int, max_iterations: int,
                            bits_ptr: int, output_len: int, iterations_ptr: int = 0,
                            posterior_ptr: int = 0, stream: int = 0):
        """Raw device pointers (e.g. torch tensors' data_ptr()); stream = hipStream_t handle or 0."""
        L = _capi.lib()
        fn = (L.ldpc_toolbox_decoder_decode_batch_f64_device if f64
              else L.ldpc_toolbox_decoder_decode_batch_f32_device)
        rc = fn(self._h, bits_ptr, output_len, llrs_ptr, self.input_len, batch, max_iterations,
                iterations_ptr or None, posterior_ptr or None, stream or None)
        if rc != 0:
            raise RuntimeError(f"decode_batch_device failed ({rc}): {_capi.last_error()}")

    def syndrome(self, bits):
        """Syndrome of hard decisions (the reference's check_llrs, src/decoder.rs:157-164, with the
        parities returned): bits [B][n] u8 -> (syndrome [B][m] u8, weight [B] u32)."""
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        if bits.ndim != 2:
            raise ValueError("bits must be [batch][n]")
        B = bits.shape[0]
        syn = np.zeros((B, self.m), dtype=np.uint8)
        weight = np.zeros(B, dtype=np.uint32)
        rc = _capi.lib().ldpc_toolbox_decoder_syndrome(self._h, bits.ctypes.data, bits.shape[1], B,
                                                       syn.ctypes.data, weight.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"syndrome failed ({rc}): {_capi.last_error()}")
        return syn, weight

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib().ldpc_toolbox_decoder_dtor(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DecoderImplementation:
    """FromStr / Display / DecoderFactory of the reference's enum (factory.rs:211-236)."""

    def __init__(self, name: str):
        if name not in ALL_IMPLEMENTATIONS:
            raise ValueError("invalid decoder implementation")  # factory.rs:221
        self.name = name

    def __str__(self):
        return self.name

    def build_decoder(self, h, puncturing: str = "", device=None) -> LdpcDecoder:
        alist = h.alist() if isinstance(h, SparseMatrix) else str(h)
        return LdpcDecoder(alist, self.name, puncturing, device)


class Encoder:
    """C ABI encoder (/root/reference/src/c_api/encoder.rs:14-53)."""

    def __init__(self, alist: str, puncturing: str = ""):
        h = _capi.lib().ldpc_toolbox_encoder_ctor_alist_string(alist.encode(), puncturing.encode())
        if not h:
            raise ValueError(_capi.last_error() or "encoder constructor returned NULL")
        self._h = h

    def encode(self, message, output_len: int):
        message = np.ascontiguousarray(message, dtype=np.uint8)
        out = np.zeros(output_len, dtype=np.uint8)
        _capi.lib().ldpc_toolbox_encoder_encode(self._h, out.ctypes.data, output_len,
                                                message.ctypes.data, message.shape[0])
        if _capi.last_error():
            raise ValueError(_capi.last_error())
        return out

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib().ldpc_toolbox_encoder_dtor(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Simulator:
    """GPU-resident simulation step over the C ABI (include/ldpc_toolbox.h part 3): frames are
    generated, decoded and scored on the device; only six counters come back."""

    def __init__(self, alist: str, implementation: str, puncturing: str = "", device: int = 0, pool_size: int = 64,
                 pool_seed: int = 1, modulation: str = "BPSK", interleaving: int = 0):
        """modulation: "BPSK" or "8PSK" (factory.rs:53-73); interleaving: columns of the DVB-S2 bit
        interleaver, negative = rows read backwards, 0 = none (ber.rs:250-252)."""
        bits_per_symbol = {"BPSK": 1, "8PSK": 3}.get(modulation)
        if bits_per_symbol is None:
            raise ValueError(f"invalid modulation {modulation}")
        h = _capi.lib().ldpc_toolbox_sim_ctor(alist.encode(), implementation.encode(), puncturing.encode(),
                                              int(device), int(pool_size), int(pool_seed))
        if not h:
            raise DecoderUnavailable(_capi.last_error() or "simulator constructor returned NULL")
        self._h = h
        self.k, self.n, self.n_tx, self.pool = (self.get(x) for x in ("k", "n", "n_tx", "pool"))
        self.rate = self.k / self.n_tx
        self.modulation, self.interleaving = modulation, int(interleaving)
        for key, value in (("modulation", bits_per_symbol), ("interleaving", int(interleaving))):
            if _capi.lib().ldpc_toolbox_sim_set(self._h, key.encode(), value) != 0:
                msg = _capi.last_error() or f"cannot set {key}"
                self.close()
                raise ValueError(msg)

    def get(self, key):
        v = C.c_int64(0)
        if _capi.lib().ldpc_toolbox_sim_get(self._h, key.encode(), C.byref(v)) != 0:
            raise KeyError(key)
        return int(v.value)

    def set(self, key, value):
        if _capi.lib().ldpc_toolbox_sim_set(self._h, key.encode(), int(value)) != 0:
            raise KeyError(key)

    def run(self, ebn0_db, seed, first_frame, frames, max_iterations, bch_max_errors: int = 0):
        """-> int64[6]: frames, bit errors, frame errors, false decodes, total iterations, iterations of
        the correct frames (sharding.COUNTER_FIELDS); with bch_max_errors > 0 int64[9]: those, then
        the outer-BCH bit errors, frame errors and iterations of the corrected frames
        (sharding.BCH_COUNTER_FIELDS, ber.rs:328-337)"""
        out = np.zeros(9 if bch_max_errors > 0 else 6, dtype=np.uint64)
        if bch_max_errors > 0:
            rc = _capi.lib().ldpc_toolbox_sim_run_bch(self._h, float(ebn0_db), int(seed), int(first_frame), int(frames),
                                                      int(max_iterations), int(bch_max_errors), out.ctypes.data)
        else:
            rc = _capi.lib().ldpc_toolbox_sim_run(self._h, float(ebn0_db), int(seed), int(first_frame), int(frames),
                                                  int(max_iterations), out.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"sim_run failed ({rc}): {_capi.last_error()}")
        return out.astype(np.int64)

    def generate(self, ebn0_db, seed, first_frame, frames):
        llrs = np.zeros((frames, self.n_tx), dtype=np.float32)
        idx = np.zeros(frames, dtype=np.uint32)
        rc = _capi.lib().ldpc_toolbox_sim_generate(self._h, float(ebn0_db), int(seed), int(first_frame), int(frames),
                                                   llrs.ctypes.data, idx.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"sim_generate failed ({rc}): {_capi.last_error()}")
        return llrs, idx

    def generate_into(self, device_ptr: int, ebn0_db, seed, first_frame, frames):
        """the same frames written straight into a device buffer [frames][n_tx] f32 (e.g. a torch tensor's data_ptr() on
        the simulator's GPU) -> pool index per frame"""
        idx = np.zeros(frames, dtype=np.uint32)
        rc = _capi.lib().ldpc_toolbox_sim_generate(self._h, float(ebn0_db), int(seed), int(first_frame), int(frames),
                                                   C.c_void_p(device_ptr), idx.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"sim_generate failed ({rc}): {_capi.last_error()}")
        return idx

    def pool_data(self):
        msgs = np.zeros((self.pool, self.k), dtype=np.uint8)
        tx = np.zeros((self.pool, self.n_tx), dtype=np.uint8)
        _capi.lib().ldpc_toolbox_sim_pool(self._h, msgs.ctypes.data, tx.ctypes.data)
        return msgs, tx

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib().ldpc_toolbox_sim_dtor(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
