/*
 * C ABI of the MI355X-native LDPC decoder library (libldpc_toolbox.so).
 *
 * PART 1 re-exports, symbol for symbol, the C API of daniestevez/ldpc-toolbox v0.12.0
 * (reference header: /root/reference/include/ldpc_toolbox.h:12-29; Rust side:
 * /root/reference/src/c_api/decoder.rs:75-137 and src/c_api/encoder.rs:55-97), so that a
 * program linked against the reference's cdylib can link against this library unchanged.
 * Decoding runs on the GPU (hand-written HIP kernels for gfx950); there is NO CPU
 * fallback: constructors return NULL (with a message on stderr / ldpc_toolbox_last_error)
 * when no HIP device is usable.
 *
 * PART 2 is the batched extension the GPU path needs (the reference decodes one codeword
 * per call).  Per codeword the semantics are exactly those of the scalar call.
 *
 * Numerical contract.  Hard decisions, iteration counts and posterior LLRs are bit-identical to
 * the reference decoder for every implementation name -- for the rules that call transcendental
 * functions (Phi, Tanh, Minstarapprox, Aminstar; /root/reference/src/decoder/arithmetic.rs:184,
 * 357, 376, 510, 965-966) this means: identical to a reference whose Rust f32/f64::{exp, ln,
 * ln_1p, tanh} resolve to the glibc libm generation with the Szabolcs-Nagy expf/logf/exp/log and
 * the fdlibm log1pf/expm1f/tanhf (and double versions) -- glibc 2.28 through 2.40, checked
 * exhaustively against 2.35.  A libm that rounds these functions correctly (the CORE-MATH
 * routines of newer glibc releases) differs from it in rare last ulps, and so would a reference
 * built there; tests/test_libm_contract.py names that situation on a host where it applies.
 * Min-sum and the 8-bit rules use no libm function and carry no such condition.
 */
#ifndef _LDPC_TOOLBOX_H
#define _LDPC_TOOLBOX_H

#ifdef __cplusplus
extern "C" {
#endif

#include <stdint.h>
#include <stddef.h>

/* ===================================================================================
 * PART 1 -- the reference's nine symbols
 * =================================================================================== */

/* Replaces ldpc_toolbox_decoder_ctor (reference include/ldpc_toolbox.h:12-13,
 * src/c_api/decoder.rs:75-88).  alist_file_path: alist text file; implementation: a decoder
 * implementation name (src/decoder/factory.rs:240-277: all 36 are accepted, plus the added
 * Minsumf32/Minsumf64/HLMinsumf32/HLMinsumf64 -- all 40 bit-identical to the reference decoder;
 * "Tanhf32@fast", "HLTanhf32@fast", "Phif32@fast", "HLPhif32@fast" are opt-in approximate variants on the
 * GPU's native exp2 / log2 / rcp, NOT bit-identical and never chosen unless named;
 * an optional "@hip:N" suffix, last, selects GPU N); puncturing: "" or a pattern such as "1,1,1,0"
 * (src/cli/ber.rs:219-229).  Returns an opaque handle, or NULL on any error. */
void *ldpc_toolbox_decoder_ctor(const char *alist_file_path, const char *implementation,
                                const char *puncturing);

/* Replaces ldpc_toolbox_decoder_ctor_alist_string (reference header :14-15,
 * src/c_api/decoder.rs:90-102): same, with the alist text passed directly. */
void *ldpc_toolbox_decoder_ctor_alist_string(const char *alist, const char *implementation,
                                             const char *puncturing);

/* Replaces ldpc_toolbox_decoder_dtor (reference header :16, src/c_api/decoder.rs:104-107). */
void ldpc_toolbox_decoder_dtor(void *decoder);

/* Replaces ldpc_toolbox_decoder_decode_f64 (reference header :17-20,
 * src/c_api/decoder.rs:109-122 -> :50-67).  llrs: llrs_len channel LLRs (the punctured
 * length when a puncturing pattern was given, else exactly n); output: receives the first
 * output_len hard decisions, one byte per bit; returns the number of iterations (0 = the
 * input was already a codeword) or -1 when max_iterations were used without reaching a
 * codeword -- output is filled in both cases.  Where the reference would panic (length
 * mismatch) or when the GPU call itself fails, this returns a value BELOW -1 (one of the
 * LDPC_TOOLBOX_ERR_* codes) and writes nothing: a caller that counts -1 as a decoding failure
 * must treat < -1 as a fault (message: ldpc_toolbox_last_error). */
#define LDPC_TOOLBOX_ERR_DEVICE (-2)      /* HIP failure */
#define LDPC_TOOLBOX_ERR_UNSUPPORTED (-3) /* graph / rule combination the kernels do not take */
#define LDPC_TOOLBOX_ERR_ARGUMENT (-4)    /* null handle, LLR or output length not matching the code */
int32_t ldpc_toolbox_decoder_decode_f64(void *decoder,
                                        uint8_t *output, size_t output_len,
                                        const double *llrs, size_t llrs_len,
                                        uint32_t max_iterations);

/* Replaces ldpc_toolbox_decoder_decode_f32 (reference header :21-24,
 * src/c_api/decoder.rs:124-137 -> :69-73). */
int32_t ldpc_toolbox_decoder_decode_f32(void *decoder,
                                        uint8_t *output, size_t output_len,
                                        const float *llrs, size_t llrs_len,
                                        uint32_t max_iterations);

/* Replace ldpc_toolbox_encoder_{ctor,ctor_alist_string,dtor,encode} (reference header
 * :26-32, src/c_api/encoder.rs:55-97).  Host-side systematic encoder (+ puncturer).
 * encode: input = k message bytes (value 1 = one, anything else = zero), output =
 * the (punctured) codeword, one byte per bit; output_len must equal its length. */
void *ldpc_toolbox_encoder_ctor(const char *alist_file_path, const char *puncturing);
void *ldpc_toolbox_encoder_ctor_alist_string(const char *alist, const char *puncturing);
void ldpc_toolbox_encoder_dtor(void *encoder);
void ldpc_toolbox_encoder_encode(void *encoder,
                                 uint8_t *output, size_t output_len,
                                 const uint8_t *input, size_t input_len);

/* ===================================================================================
 * PART 2 -- batched extension (new symbols, same handles)
 * =================================================================================== */

/* Constructor with an explicit GPU index (the "@hip:N" suffix does the same). */
void *ldpc_toolbox_decoder_ctor_alist_string_on_device(const char *alist, const char *implementation,
                                                       const char *puncturing, int32_t device);

/* Batch decode, host buffers.  Semantically a loop of ldpc_toolbox_decoder_decode_f32 over
 * `batch` frames (replaces that loop: src/simulation/ber.rs:462-466 calls decode once per
 * frame):
 *   llrs        [batch][llrs_len]   one row per frame
 *   output      [batch][output_len] first output_len hard decisions of every frame
 *   iterations  [batch]             iterations used, -1 = failed (may be NULL)
 *   posterior   [batch][n]          final soft LLRs of the decoder (may be NULL)
 * returns 0, or one of the LDPC_TOOLBOX_ERR_* codes above -- the same vocabulary as the scalar entries (bad
 * handle / lengths: LDPC_TOOLBOX_ERR_ARGUMENT, HIP failure: _DEVICE, unsupported graph / rule: _UNSUPPORTED). */
int32_t ldpc_toolbox_decoder_decode_batch_f32(void *decoder, uint8_t *output, size_t output_len,
                                              const float *llrs, size_t llrs_len, size_t batch,
                                              uint32_t max_iterations, int32_t *iterations,
                                              float *posterior);
int32_t ldpc_toolbox_decoder_decode_batch_f64(void *decoder, uint8_t *output, size_t output_len,
                                              const double *llrs, size_t llrs_len, size_t batch,
                                              uint32_t max_iterations, int32_t *iterations,
                                              double *posterior);

/* Batch decode, buffers already resident in the decoder's GPU memory (all four pointers are
 * device pointers).  hip_stream: a hipStream_t to launch on (the call returns without
 * synchronising; the caller orders it after the producers of llrs), or NULL: the library uses the
 * handle's own stream, ordered after everything queued on the legacy default stream (handle 0) at
 * the time of the call, and synchronises before returning.  NULL is also what a framework's
 * "default stream" handle looks like (torch.cuda.default_stream().cuda_stream == 0): pass a real
 * stream to get the asynchronous form.  (Returning without synchronising is not returning at once: a call returns when
 * its last launch is enqueued, and a layered-schedule call large enough for two execution lanes enqueues from two
 * threads that follow their groups' progress one iteration behind, so it returns shortly before its work completes.) */
int32_t ldpc_toolbox_decoder_decode_batch_f32_device(void *decoder, uint8_t *output, size_t output_len,
                                                     const float *llrs, size_t llrs_len, size_t batch,
                                                     uint32_t max_iterations, int32_t *iterations,
                                                     float *posterior, void *hip_stream);
int32_t ldpc_toolbox_decoder_decode_batch_f64_device(void *decoder, uint8_t *output, size_t output_len,
                                                     const double *llrs, size_t llrs_len, size_t batch,
                                                     uint32_t max_iterations, int32_t *iterations,
                                                     double *posterior, void *hip_stream);

/* The syndrome test of the reference's decoders (src/decoder.rs:157-164, check_llrs: the parity
 * of the hard decisions over every row of H) as an operator that returns the parities instead of
 * only "all zero?".  bits: [batch][bits_len] hard decisions, one byte per bit (what the decode
 * entries write with output_len = n); bits_len must equal n.  syndrome: [batch][m], 1 = check
 * unsatisfied (may be NULL).  weight: [batch] number of unsatisfied checks (may be NULL).  A frame
 * the decoder reported as converged (iterations >= 0) has weight 0; a failed one does not.
 * Host pointers; the _device form takes device pointers and a hipStream_t (NULL = the handle's own
 * stream, synchronised on return), at most 65535 codewords per call.  returns 0 or an LDPC_TOOLBOX_ERR_* code. */
int32_t ldpc_toolbox_decoder_syndrome(void *decoder, const uint8_t *bits, size_t bits_len, size_t batch,
                                      uint8_t *syndrome, uint32_t *weight);
int32_t ldpc_toolbox_decoder_syndrome_device(void *decoder, const uint8_t *bits, size_t bits_len, size_t batch,
                                             uint8_t *syndrome, uint32_t *weight, void *hip_stream);

/* Integer properties: "n", "m", "k", "edges", "input_len", "device", "group_size",
 * "max_check_degree", "max_variable_degree", "layers" (dependency levels of the layered schedule),
 * "last_lanes" / "last_group" (execution lanes and codewords per group of the last decode call),
 * "preferred_group" (codewords per group of a large call: "group_size" if set, else 4096, more for small graphs),
 * "row_records" (words per check-row record when flooding min-sum keeps a row's messages as
 * {min1, min2, flip bits, argmin}; 0 = per-edge messages).  returns 0 or -1 (unknown key). */
int32_t ldpc_toolbox_decoder_get(void *decoder, const char *key, int64_t *value);
/* Tunables: "group_size" (codewords decoded together; 0 = automatic), "profiling" (0/1:
 * bracket the check/variable/layer launches with hipEvents), and 26 launch / execution choices -- "waves", "vec", "tile",
 * "lfree", "records", "rec_run", "rec_quiet", "rec_long", "vn_event", "staged_minsum", "cn_reg", "hl_reg", "hl_records",
 * "serial_levels", "latency", "latency_edge", "compact", "compact_first", "compact_every", "lanes", "lane_threads",
 * "lane_pace", "lead", "poll", "throttle", "pooling" (ldpc_toolbox_amd/csrc/device_decoder.h says what each selects; results never
 * depend on them: each chooses between forms the test suite compares bit for bit).  "throttle" (0/1, default 0):
 * a ..._device call on the CALLER's stream may pace its launches on the groups' progress words, i.e. return when the
 * work is within two iterations of its end instead of as soon as it is enqueued (fewer launches past convergence;
 * calls on the library's own stream always may).  The same switch governs the layered schedule's lane threads ("lane_pace",
 * one iteration ahead of their group): without it a call on the caller's stream returns as soon as everything is
 * enqueued.  A paced call that sees no progress for 200 ms (a stream gated behind something the caller releases later)
 * stops pacing and enqueues the rest at once.  Flooding schedule (round 5): a ..._device call with "throttle" (or on the library's
 * own stream) follows its groups two iterations ahead -- with two execution lanes through a host thread per lane -- and, once the first codewords have converged, ends every iteration with a re-packing
 * checkpoint instead of every second one (DVB-S2 1/2 at +2 dB: +2 %; a call in which nothing converges launches nothing
 * extra).
 * "pooling" (0/1, default 0): straggler pooling inside the batch entries.  A call of several chunks of frames learns from
 * its first chunk how many iterations its frames take; later chunks run a reduced iteration budget (2 x average + 8) and the
 * frames that have not converged by then are decoded again, together, with max_iterations -- per frame the outcome of one
 * full-budget decode, so outputs do not depend on the option.  It spares every chunk the nearly empty iterations its few slow
 * or failing frames would drag it through (the waterfall with the reference's default of 100 iterations).  The call
 * synchronises between chunks: host-buffer entries always may; a ..._device entry takes it on the library's own stream or
 * with "throttle".  ldpc_toolbox_decoder_get "last_pooled": frames of the last call that took the second pass.
 * returns 0 or -1. */
int32_t ldpc_toolbox_decoder_set(void *decoder, const char *key, int64_t value);
/* hipEvent statistics collected while "profiling" is 1.  kind: 0 = check-node kernel,
 * 1 = variable-node phase (vn_kernel and, with row records, the small vn_free_rec_kernel launch behind it: one bracket
 * per iteration), 2 = layered level kernel.  reset != 0 clears the counters
 * after reading. */
int32_t ldpc_toolbox_decoder_kernel_stats(void *decoder, int32_t kind, uint64_t *launches,
                                          double *total_ms, int32_t reset);

/* ===================================================================================
 * PART 3 -- GPU-resident simulation step (the frame pipeline either side of the decode path:
 * reference src/simulation/ber.rs:436-481 Worker::simulate, one frame per call there)
 * =================================================================================== */

/* Decoder + host encoder + a pool of `pool_size` pre-encoded random messages (seeded by
 * pool_seed) on GPU `device`.  BPSK over AWGN.  NULL on error. */
void *ldpc_toolbox_sim_ctor(const char *alist, const char *implementation, const char *puncturing,
                            int32_t device, uint32_t pool_size, uint64_t pool_seed);
void ldpc_toolbox_sim_dtor(void *sim);
/* Generates frames [first_frame, first_frame + frames) at ebn0_db on the device (noise is a pure
 * function of (seed, frame index, position): Philox4x32-10 + polar method), decodes them and counts
 * errors.  counters[6] = frames, bit errors (first k bits), frame errors, false decodes, total
 * iterations, iterations of the correct frames (the fields of ber.rs:113-138).  returns 0 or < 0. */
int32_t ldpc_toolbox_sim_run(void *sim, double ebn0_db, uint64_t seed, uint64_t first_frame,
                             size_t frames, uint32_t max_iterations, uint64_t *counters);
/* The same with the reference driver's outer-BCH accounting (src/simulation/ber.rs:328-337,
 * `ber --bch-max-errors`): a frame with at most bch_max_errors bit errors after LDPC decoding counts
 * as corrected.  counters[9] = the six above, then BCH bit errors, BCH frame errors, iterations of
 * the frames the BCH code corrects. */
int32_t ldpc_toolbox_sim_run_bch(void *sim, double ebn0_db, uint64_t seed, uint64_t first_frame,
                                 size_t frames, uint32_t max_iterations, uint64_t bch_max_errors,
                                 uint64_t *counters);
/* The LLRs of the same frames ([frames][n_tx]; a host buffer, or a buffer in the simulator's GPU memory, which is then
 * filled in place) and which pooled codeword each frame carries (host array, may be NULL): lets a CPU decoder be run on
 * identical frames, and a benchmark fill its device-resident batch from the library's own generator. */
int32_t ldpc_toolbox_sim_generate(void *sim, double ebn0_db, uint64_t seed, uint64_t first_frame,
                                  size_t frames, float *llrs, uint32_t *pool_index);
/* The pool: messages [pool][k] and transmitted (punctured) codewords [pool][n_tx]; either may be NULL. */
int32_t ldpc_toolbox_sim_pool(void *sim, uint8_t *messages, uint8_t *tx_bits);
/* "k", "n", "n_tx", "pool", "modulation", "interleaving"; "preferred_batch" (frames per run call that fill one
 * group of the decoder: 4096, more for small graphs); of the last run call: "pooled_frames" (frames that went
 * through the straggler pool, below), "streamed_frames", "stream_iterations".  returns 0 or -1. */
int32_t ldpc_toolbox_sim_get(void *sim, const char *key, int64_t *value);
/* "modulation": bits per symbol, 1 = BPSK (default), 3 = 8PSK with the DVB-S2 Gray mapping and the
 * exact max* demodulator (src/simulation/modulation.rs:144-288; n_tx must be a multiple of 3).
 * "interleaving": columns of the DVB-S2 bit interleaver applied before modulation and undone after
 * demodulation (src/simulation/interleaving.rs; negative = rows read backwards, 0 = none (default);
 * must divide n_tx), as the reference's `ber --modulation 8PSK --interleaving N` (src/cli/ber.rs:52-59).
 * "pooling" (0/1, default 1): once a run call has seen how many iterations its frames take, its later chunks run a
 * reduced iteration budget and the frames that have not converged by then are pooled and decoded together with the
 * full budget -- per frame the result of one full-budget decode, so the counters do not depend on it; it spares every
 * chunk the nearly empty iterations its few slow frames would otherwise drag it through.
 * "streaming" (0/1, default 0): continuous batching for flooding Minsumf32 (exact, slower in this layout).
 * Any other key is forwarded to the simulator's decoder (see ldpc_toolbox_decoder_set).
 * returns 0, or -1 (unknown key / unusable value, message via ldpc_toolbox_last_error). */
int32_t ldpc_toolbox_sim_set(void *sim, const char *key, int64_t value);

/* Standard-code generator (what the reference's `dvbs2` / `5g` / `ccsds` / `ccsds-c2` CLI
 * sub-commands print, src/cli/dvbs2.rs:91, src/cli/nr5g.rs:46, src/cli/ccsds.rs:70): writes the
 * padded alist text of `spec` ("dvbs2:R1_2", "nr5g:1:384", "ar4ja:1/2:1024", "c2") into
 * buffer (NUL-terminated when it fits) and returns the number of bytes needed excluding the
 * NUL; 0 for an unknown spec. */
size_t ldpc_toolbox_code_alist(const char *spec, char *buffer, size_t buffer_len);

/* alist parser + writer of the host library on their own (reference: SparseMatrix::from_alist
 * src/sparse.rs:352-389 then alist() / alist_no_padding() :301-341): parses `alist` and writes it
 * back (padding != 0: MacKay zero padding).  Same buffer convention as ldpc_toolbox_code_alist;
 * returns 0 when `alist` is malformed (message via ldpc_toolbox_last_error). */
size_t ldpc_toolbox_alist_normalize(const char *alist, int32_t padding, char *buffer, size_t buffer_len);

/* Number of usable HIP devices (0 when the HIP runtime finds none). */
int32_t ldpc_toolbox_device_count(void);

/* Message of the last failed call on this thread ("" if none). */
const char *ldpc_toolbox_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* _LDPC_TOOLBOX_H */
