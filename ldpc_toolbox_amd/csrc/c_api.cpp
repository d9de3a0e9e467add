// extern "C" boundary: the reference's nine symbols + the batched extension
// (include/ldpc_toolbox.h).  Behavioural model: /root/reference/src/c_api/decoder.rs and
// /root/reference/src/c_api/encoder.rs; every entry point below cites what it replaces.
#include "../../include/ldpc_toolbox.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "codes.h"
#include "device_decoder.h"
#include "encoder.h"
#include "implementation.h"
#include "simulator.h"
#include "sparse.h"

namespace {

thread_local std::string g_last_error;

void set_error(const std::string &m) { g_last_error = m; }

bool parse_puncturing(const char *s, std::vector<uint8_t> *out) {
  return ldpc::parse_puncturing_pattern(s ? s : "", out);
}

bool read_file(const char *path, std::string *out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}

struct DecoderHandle {
  std::unique_ptr<ldpc::DeviceDecoder> dec;
};

struct EncoderHandle {
  ldpc::Encoder enc;
  std::vector<uint8_t> pattern;
  size_t out_len = 0;
  std::vector<uint8_t> scratch;
};

// a device index: decimal digits only (no sign, no trailing characters)
bool parse_device_index(const char *s, int *out) {
  if (!s || !*s) return false;
  char *end = nullptr;
  errno = 0;
  const long v = std::strtol(s, &end, 10);
  if (errno != 0 || end == s || *end != '\0' || s[0] < '0' || s[0] > '9' || v < 0 || v > 1 << 20) return false;
  *out = static_cast<int>(v);
  return true;
}

// LDPC_TOOLBOX_DEVICE: GPU of the constructors that take no device argument; -1 = unusable value
int default_device() {
  const char *s = std::getenv("LDPC_TOOLBOX_DEVICE");
  if (!s || !*s) return 0;
  int d = 0;
  return parse_device_index(s, &d) ? d : -1;
}

void *make_decoder(const std::string &alist, const char *implementation, const char *puncturing,
                   int device) {
  g_last_error.clear();
  if (!implementation) {
    set_error("invalid decoder implementation");
    return nullptr;
  }
  std::string name = implementation;
  const size_t at = name.find("@hip");
  if (at != std::string::npos) {
    const std::string suffix = name.substr(at + 4);
    name = name.substr(0, at);
    if (!suffix.empty()) {
      if (suffix[0] != ':') {
        set_error("invalid device suffix (expected @hip or @hip:N)");
        return nullptr;
      }
      if (!parse_device_index(suffix.c_str() + 1, &device)) {
        set_error("invalid device suffix (expected @hip or @hip:N with N a decimal GPU index)");
        return nullptr;
      }
    }
  }
  if (device < 0) {
    set_error("LDPC_TOOLBOX_DEVICE is not a decimal GPU index");
    return nullptr;
  }
  ldpc::SparseMatrix h;
  std::string err;
  if (!ldpc::SparseMatrix::from_alist(alist, &h, &err)) {
    set_error(err);
    return nullptr;
  }
  ldpc::Implementation impl;
  if (!ldpc::parse_implementation(name, &impl, &err)) {
    set_error(err);
    return nullptr;
  }
  std::vector<uint8_t> pattern;
  if (!parse_puncturing(puncturing, &pattern)) {
    set_error("invalid puncturing pattern");
    return nullptr;
  }
  ldpc::DeviceDecoder *d = ldpc::DeviceDecoder::create(h, impl, pattern, device, &err);
  if (!d) {
    set_error(err);
    return nullptr;
  }
  auto *handle = new DecoderHandle();
  handle->dec.reset(d);
  return handle;
}

void *make_encoder(const std::string &alist, const char *puncturing) {
  g_last_error.clear();
  ldpc::SparseMatrix h;
  std::string err;
  if (!ldpc::SparseMatrix::from_alist(alist, &h, &err)) {
    set_error(err);
    return nullptr;
  }
  auto handle = std::make_unique<EncoderHandle>();
  if (!parse_puncturing(puncturing, &handle->pattern)) {
    set_error("invalid puncturing pattern");
    return nullptr;
  }
  if (!ldpc::Encoder::from_h(h, &handle->enc, &err)) {
    set_error(err);
    return nullptr;
  }
  const size_t n = handle->enc.n();
  handle->out_len = n;
  if (!handle->pattern.empty()) {
    size_t trues = 0;
    for (uint8_t p : handle->pattern) trues += p;
    // puncturing.rs:54-56: the codeword length must be divisible by the pattern length
    handle->out_len = (n % handle->pattern.size() == 0) ? n / handle->pattern.size() * trues : SIZE_MAX;
  }
  handle->scratch.resize(n);
  return handle.release();
}

template <typename F>
int32_t decode_scalar(void *decoder, uint8_t *output, size_t output_len, const F *llrs, size_t llrs_len,
                      uint32_t max_iterations) {
  g_last_error.clear();
  auto *h = static_cast<DecoderHandle *>(decoder);
  // -1 is the reference's "no codeword found" (output filled).  A call that could not run at all
  // (where the reference would panic) returns a value BELOW -1 and leaves `output` untouched, so a
  // caller counting -1 as a decoding failure never counts a GPU fault as one.
  if (!h || !h->dec) {
    set_error("null decoder handle");
    return LDPC_TOOLBOX_ERR_ARGUMENT;
  }
  if (llrs_len != h->dec->input_len() || output_len > h->dec->n()) {
    set_error("LLR or output length does not match the code");
    return LDPC_TOOLBOX_ERR_ARGUMENT;
  }
  int32_t iterations = -1;
  const int rc = h->dec->decode_host(llrs, sizeof(F) == 8, 1, max_iterations, output, output_len, &iterations,
                                     nullptr);
  if (rc != 0) {
    set_error(h->dec->last_error());
    return rc == -3 ? LDPC_TOOLBOX_ERR_UNSUPPORTED : LDPC_TOOLBOX_ERR_DEVICE;
  }
  return iterations;
}

template <typename F>
int32_t decode_batch(void *decoder, uint8_t *output, size_t output_len, const F *llrs, size_t llrs_len,
                     size_t batch, uint32_t max_iterations, int32_t *iterations, F *posterior, bool on_device,
                     void *stream) {
  g_last_error.clear();
  auto *h = static_cast<DecoderHandle *>(decoder);
  // one error vocabulary for the scalar and the batch entries (include/ldpc_toolbox.h, LDPC_TOOLBOX_ERR_*)
  if (!h || !h->dec) {
    set_error("null decoder handle");
    return LDPC_TOOLBOX_ERR_ARGUMENT;
  }
  if (llrs_len != h->dec->input_len() || output_len > h->dec->n()) {
    set_error("LLR or output length does not match the code");
    return LDPC_TOOLBOX_ERR_ARGUMENT;
  }
  const int rc = on_device ? h->dec->decode_device(llrs, sizeof(F) == 8, batch, max_iterations, output, output_len,
                                                   iterations, posterior, static_cast<hipStream_t>(stream))
                           : h->dec->decode_host(llrs, sizeof(F) == 8, batch, max_iterations, output, output_len,
                                                 iterations, posterior);
  if (rc == 0) return 0;
  set_error(h->dec->last_error());
  return rc == -3 ? LDPC_TOOLBOX_ERR_UNSUPPORTED : (rc == -1 ? LDPC_TOOLBOX_ERR_ARGUMENT : LDPC_TOOLBOX_ERR_DEVICE);
}

}  // namespace

extern "C" {

// ---- PART 1 ---------------------------------------------------------------------------------

void *ldpc_toolbox_decoder_ctor(const char *alist_file_path, const char *implementation,
                                const char *puncturing) {
  std::string text;
  if (!alist_file_path || !read_file(alist_file_path, &text)) {
    set_error("cannot read alist file");
    return nullptr;
  }
  return make_decoder(text, implementation, puncturing, default_device());
}

void *ldpc_toolbox_decoder_ctor_alist_string(const char *alist, const char *implementation,
                                             const char *puncturing) {
  if (!alist) {
    set_error("null alist");
    return nullptr;
  }
  return make_decoder(alist, implementation, puncturing, default_device());
}

void ldpc_toolbox_decoder_dtor(void *decoder) { delete static_cast<DecoderHandle *>(decoder); }

int32_t ldpc_toolbox_decoder_decode_f64(void *decoder, uint8_t *output, size_t output_len, const double *llrs,
                                        size_t llrs_len, uint32_t max_iterations) {
  return decode_scalar<double>(decoder, output, output_len, llrs, llrs_len, max_iterations);
}

int32_t ldpc_toolbox_decoder_decode_f32(void *decoder, uint8_t *output, size_t output_len, const float *llrs,
                                        size_t llrs_len, uint32_t max_iterations) {
  return decode_scalar<float>(decoder, output, output_len, llrs, llrs_len, max_iterations);
}

void *ldpc_toolbox_encoder_ctor(const char *alist_file_path, const char *puncturing) {
  std::string text;
  if (!alist_file_path || !read_file(alist_file_path, &text)) {
    set_error("cannot read alist file");
    return nullptr;
  }
  return make_encoder(text, puncturing);
}

void *ldpc_toolbox_encoder_ctor_alist_string(const char *alist, const char *puncturing) {
  if (!alist) {
    set_error("null alist");
    return nullptr;
  }
  return make_encoder(alist, puncturing);
}

void ldpc_toolbox_encoder_dtor(void *encoder) { delete static_cast<EncoderHandle *>(encoder); }

void ldpc_toolbox_encoder_encode(void *encoder, uint8_t *output, size_t output_len, const uint8_t *input,
                                 size_t input_len) {
  g_last_error.clear();
  auto *h = static_cast<EncoderHandle *>(encoder);
  if (!h) {
    set_error("null encoder handle");
    return;
  }
  // the reference panics on these (ndarray shape mismatch / assert_eq!, c_api/encoder.rs:50):
  // here nothing is written and the error is recorded
  if (input_len != h->enc.k() || output_len != h->out_len) {
    set_error("message or output length does not match the code");
    std::fprintf(stderr, "ldpc_toolbox: encoder: %s\n", g_last_error.c_str());
    return;
  }
  // c_api/encoder.rs:41-43: a byte equal to 1 is a one, anything else a zero
  std::vector<uint8_t> msg(input_len);
  for (size_t i = 0; i < input_len; i++) msg[i] = input[i] == 1;
  if (h->pattern.empty()) {
    h->enc.encode(msg.data(), output);
    return;
  }
  h->enc.encode(msg.data(), h->scratch.data());
  // puncturing.rs:47-75: keep the blocks whose pattern entry is true
  const size_t block = h->enc.n() / h->pattern.size();
  size_t j = 0;
  for (size_t k = 0; k < h->pattern.size(); k++) {
    if (!h->pattern[k]) continue;
    std::memcpy(output + j * block, h->scratch.data() + k * block, block);
    j++;
  }
}

// ---- PART 2 ---------------------------------------------------------------------------------

void *ldpc_toolbox_decoder_ctor_alist_string_on_device(const char *alist, const char *implementation,
                                                       const char *puncturing, int32_t device) {
  if (!alist) {
    set_error("null alist");
    return nullptr;
  }
  return make_decoder(alist, implementation, puncturing, device);
}

int32_t ldpc_toolbox_decoder_decode_batch_f32(void *decoder, uint8_t *output, size_t output_len,
                                              const float *llrs, size_t llrs_len, size_t batch,
                                              uint32_t max_iterations, int32_t *iterations, float *posterior) {
  return decode_batch<float>(decoder, output, output_len, llrs, llrs_len, batch, max_iterations, iterations,
                             posterior, false, nullptr);
}

int32_t ldpc_toolbox_decoder_decode_batch_f64(void *decoder, uint8_t *output, size_t output_len,
                                              const double *llrs, size_t llrs_len, size_t batch,
                                              uint32_t max_iterations, int32_t *iterations, double *posterior) {
  return decode_batch<double>(decoder, output, output_len, llrs, llrs_len, batch, max_iterations, iterations,
                              posterior, false, nullptr);
}

int32_t ldpc_toolbox_decoder_decode_batch_f32_device(void *decoder, uint8_t *output, size_t output_len,
                                                     const float *llrs, size_t llrs_len, size_t batch,
                                                     uint32_t max_iterations, int32_t *iterations,
                                                     float *posterior, void *hip_stream) {
  return decode_batch<float>(decoder, output, output_len, llrs, llrs_len, batch, max_iterations, iterations,
                             posterior, true, hip_stream);
}

int32_t ldpc_toolbox_decoder_decode_batch_f64_device(void *decoder, uint8_t *output, size_t output_len,
                                                     const double *llrs, size_t llrs_len, size_t batch,
                                                     uint32_t max_iterations, int32_t *iterations,
                                                     double *posterior, void *hip_stream) {
  return decode_batch<double>(decoder, output, output_len, llrs, llrs_len, batch, max_iterations, iterations,
                              posterior, true, hip_stream);
}

int32_t ldpc_toolbox_decoder_syndrome(void *decoder, const uint8_t *bits, size_t bits_len, size_t batch,
                                      uint8_t *syndrome, uint32_t *weight) {
  g_last_error.clear();
  auto *h = static_cast<DecoderHandle *>(decoder);
  if (!h || !h->dec) {
    set_error("null decoder handle");
    return LDPC_TOOLBOX_ERR_ARGUMENT;
  }
  if (bits_len != h->dec->n() || (!bits && batch)) {
    set_error("hard decisions must cover the whole codeword (bits_len == n)");
    return LDPC_TOOLBOX_ERR_ARGUMENT;
  }
  const int rc = h->dec->syndrome_host(bits, batch, syndrome, weight);
  if (rc == 0) return 0;
  set_error(h->dec->last_error());
  return rc == -1 ? LDPC_TOOLBOX_ERR_ARGUMENT : LDPC_TOOLBOX_ERR_DEVICE;
}

int32_t ldpc_toolbox_decoder_syndrome_device(void *decoder, const uint8_t *bits, size_t bits_len, size_t batch,
                                             uint8_t *syndrome, uint32_t *weight, void *hip_stream) {
  g_last_error.clear();
  auto *h = static_cast<DecoderHandle *>(decoder);
  if (!h || !h->dec) {
    set_error("null decoder handle");
    return LDPC_TOOLBOX_ERR_ARGUMENT;
  }
  if (bits_len != h->dec->n() || (!bits && batch)) {
    set_error("hard decisions must cover the whole codeword (bits_len == n)");
    return LDPC_TOOLBOX_ERR_ARGUMENT;
  }
  const int rc = h->dec->syndrome_device(bits, batch, syndrome, weight, static_cast<hipStream_t>(hip_stream));
  if (rc == 0) return 0;
  set_error(h->dec->last_error());
  return rc == -1 ? LDPC_TOOLBOX_ERR_ARGUMENT : LDPC_TOOLBOX_ERR_DEVICE;
}

int32_t ldpc_toolbox_decoder_get(void *decoder, const char *key, int64_t *value) {
  auto *h = static_cast<DecoderHandle *>(decoder);
  if (!h || !h->dec || !key || !value) return -1;
  const std::string k = key;
  const ldpc::DeviceDecoder &d = *h->dec;
  if (k == "n")
    *value = static_cast<int64_t>(d.n());
  else if (k == "m")
    *value = static_cast<int64_t>(d.m());
  else if (k == "k")
    *value = static_cast<int64_t>(d.n()) - static_cast<int64_t>(d.m());
  else if (k == "edges")
    *value = static_cast<int64_t>(d.edges());
  else if (k == "input_len")
    *value = static_cast<int64_t>(d.input_len());
  else if (k == "device")
    *value = d.device();
  else if (k == "group_size")
    *value = static_cast<int64_t>(d.group_size());
  else if (k == "max_check_degree")
    *value = d.max_check_degree();
  else if (k == "max_variable_degree")
    *value = d.max_variable_degree();
  else if (k == "layers")
    *value = static_cast<int64_t>(d.layers());
  else if (k == "last_lanes")
    *value = d.last_lanes();
  else if (k == "last_pooled")
    *value = static_cast<int64_t>(d.last_pooled());
  else if (k == "last_group")
    *value = static_cast<int64_t>(d.last_group());
  else if (k == "preferred_group")
    *value = static_cast<int64_t>(d.preferred_group(size_t(1) << 20));
  else if (k == "row_records")
    *value = d.row_records();
  else if (k == "last_persist")
    *value = d.last_persist();
  else if (k == "experiments")  // 1: a -DLDPC_EXPERIMENTS build (carries the opt-in forms the product left behind)
#ifdef LDPC_EXPERIMENTS
    *value = 1;
#else
    *value = 0;
#endif
  else
    return -1;
  return 0;
}

int32_t ldpc_toolbox_decoder_set(void *decoder, const char *key, int64_t value) {
  auto *h = static_cast<DecoderHandle *>(decoder);
  if (!h || !h->dec || !key) return -1;
  const std::string k = key;
  if (k == "group_size" && value >= 0)
    h->dec->set_group_size(static_cast<size_t>(value));
  else if (k == "profiling")
    h->dec->set_profiling(value != 0);
  else if (!h->dec->set_option(k, value))
    return -1;
  return 0;
}

int32_t ldpc_toolbox_decoder_kernel_stats(void *decoder, int32_t kind, uint64_t *launches, double *total_ms,
                                          int32_t reset) {
  auto *h = static_cast<DecoderHandle *>(decoder);
  if (!h || !h->dec || kind < 0 || kind >= ldpc::kKernelKinds) return -1;
  const ldpc::KernelStat s = h->dec->kernel_stat(kind);
  if (launches) *launches = s.launches;
  if (total_ms) *total_ms = s.total_ms;
  if (reset) h->dec->reset_kernel_stats();
  return 0;
}

// ---- PART 3: GPU-resident simulation step ----------------------------------------------------------

void *ldpc_toolbox_sim_ctor(const char *alist, const char *implementation, const char *puncturing, int32_t device,
                            uint32_t pool_size, uint64_t pool_seed) {
  g_last_error.clear();
  if (!alist || !implementation) {
    set_error("null argument");
    return nullptr;
  }
  std::string err;
  ldpc::Simulator *s = ldpc::Simulator::create(alist, implementation, puncturing ? puncturing : "", device,
                                               pool_size, pool_seed, &err);
  if (!s) set_error(err);
  return s;
}

void ldpc_toolbox_sim_dtor(void *sim) { delete static_cast<ldpc::Simulator *>(sim); }

int32_t ldpc_toolbox_sim_run(void *sim, double ebn0_db, uint64_t seed, uint64_t first_frame, size_t frames,
                             uint32_t max_iterations, uint64_t *counters) {
  g_last_error.clear();
  auto *s = static_cast<ldpc::Simulator *>(sim);
  if (!s || !counters) {
    set_error("null argument");
    return -1;
  }
  const int rc = s->run(ebn0_db, seed, first_frame, frames, max_iterations, counters);
  if (rc) set_error(s->last_error());
  return rc;
}

int32_t ldpc_toolbox_sim_run_bch(void *sim, double ebn0_db, uint64_t seed, uint64_t first_frame, size_t frames,
                                 uint32_t max_iterations, uint64_t bch_max_errors, uint64_t *counters) {
  g_last_error.clear();
  auto *s = static_cast<ldpc::Simulator *>(sim);
  if (!s || !counters) {
    set_error("null argument");
    return -1;
  }
  const int rc = s->run_bch(ebn0_db, seed, first_frame, frames, max_iterations, bch_max_errors, counters);
  if (rc) set_error(s->last_error());
  return rc;
}

int32_t ldpc_toolbox_sim_generate(void *sim, double ebn0_db, uint64_t seed, uint64_t first_frame, size_t frames,
                                  float *llrs, uint32_t *pool_index) {
  g_last_error.clear();
  auto *s = static_cast<ldpc::Simulator *>(sim);
  if (!s || !llrs) {
    set_error("null argument");
    return -1;
  }
  const int rc = s->generate(ebn0_db, seed, first_frame, frames, llrs, pool_index);
  if (rc) set_error(s->last_error());
  return rc;
}

int32_t ldpc_toolbox_sim_pool(void *sim, uint8_t *messages, uint8_t *tx_bits) {
  auto *s = static_cast<ldpc::Simulator *>(sim);
  if (!s) return -1;
  if (messages) std::memcpy(messages, s->messages().data(), s->messages().size());
  if (tx_bits) std::memcpy(tx_bits, s->tx_bits().data(), s->tx_bits().size());
  return 0;
}

int32_t ldpc_toolbox_sim_get(void *sim, const char *key, int64_t *value) {
  auto *s = static_cast<ldpc::Simulator *>(sim);
  if (!s || !key || !value) return -1;
  const std::string k = key;
  if (k == "k")
    *value = static_cast<int64_t>(s->k());
  else if (k == "n")
    *value = static_cast<int64_t>(s->n());
  else if (k == "n_tx")
    *value = static_cast<int64_t>(s->n_tx());
  else if (k == "pool")
    *value = s->pool();
  else if (k == "modulation")
    *value = s->modulation();
  else if (k == "interleaving")
    *value = s->interleaving();
  else if (k == "streamed_frames")
    *value = static_cast<int64_t>(s->streamed_frames());
  else if (k == "pooled_frames")  // frames of the last run() call that went through the straggler pool
    *value = static_cast<int64_t>(s->pooled_frames());
  else if (k == "preferred_batch")  // frames per run() call that fill one group of the decoder (4096; more for small graphs)
    *value = static_cast<int64_t>(s->decoder()->preferred_group(size_t(1) << 20));
  else if (k == "stream_iterations")
    *value = static_cast<int64_t>(s->decoder()->last_stream_iterations());
  else
    return -1;
  return 0;
}

int32_t ldpc_toolbox_sim_set(void *sim, const char *key, int64_t value) {
  auto *s = static_cast<ldpc::Simulator *>(sim);
  if (!s || !key) return -1;
  const std::string k = key;
  if (k == "group_size" && value >= 0) {
    s->decoder()->set_group_size(static_cast<size_t>(value));
    return 0;
  }
  if (k == "pooling") {  // 0: every chunk runs the full iteration budget (no straggler pool)
    s->set_pooling(value != 0);
    return 0;
  }
  if (k == "streaming") {  // 0: every chunk of 4096 frames is decoded to the end before the next starts
    s->set_streaming(value != 0);
    return 0;
  }
  if (k == "modulation" || k == "interleaving") {
    g_last_error.clear();
    const bool ok = k == "modulation" ? s->set_modulation(static_cast<int>(value)) : s->set_interleaving(value);
    if (!ok) set_error(s->last_error());
    return ok ? 0 : -1;
  }
  return s->decoder()->set_option(k, value) ? 0 : -1;
}

size_t ldpc_toolbox_code_alist(const char *spec, char *buffer, size_t buffer_len) {
  if (!spec) return 0;
  ldpc::SparseMatrix h;
  if (!ldpc::codes::by_spec(spec, &h)) return 0;
  const std::string text = h.alist(true);
  if (buffer && buffer_len > text.size()) std::memcpy(buffer, text.c_str(), text.size() + 1);
  return text.size();
}

size_t ldpc_toolbox_alist_normalize(const char *alist, int32_t padding, char *buffer, size_t buffer_len) {
  g_last_error.clear();
  if (!alist) return 0;
  ldpc::SparseMatrix h;
  std::string err;
  if (!ldpc::SparseMatrix::from_alist(alist, &h, &err)) {
    set_error(err);
    return 0;
  }
  const std::string text = h.alist(padding != 0);
  if (buffer && buffer_len > text.size()) std::memcpy(buffer, text.c_str(), text.size() + 1);
  return text.size();
}

int32_t ldpc_toolbox_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

const char *ldpc_toolbox_last_error(void) { return g_last_error.c_str(); }

}  // extern "C"
