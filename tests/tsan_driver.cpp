// ThreadSanitizer driver for the library's host threading (tests/test_host_sanitizers.py): the real c_api.cpp /
// device_decoder.hip / simulator.hip host code, compiled host-only with -fsanitize=thread and linked against
// tests/hip_stub (streams = worker threads, kernels = no-ops that publish the progress word).  Exercises what round 3
// added -- two execution lanes enqueued by two host threads, the staging thread of the host-buffer entry, the per-lane
// task queues and their event / counter hand-shakes, the progress-word polling with early exits -- and the error
// returns (HIP_STUB_FAIL).  Results are meaningless (no kernel runs); return codes and the absence of TSan reports are
// what is checked.  The reference's contract for a handle is `Send`, one call at a time (src/decoder.rs:19); two
// handles driven by two threads must not interfere either.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/ldpc_toolbox.h"

extern "C" unsigned long long hip_stub_launches(void);

static std::string alist_of(const char *spec) {
  const size_t need = ldpc_toolbox_code_alist(spec, nullptr, 0);
  std::string s(need + 1, '\0');
  ldpc_toolbox_code_alist(spec, &s[0], s.size());
  s.resize(need);
  return s;
}

static int64_t get(void *dec, const char *key) {
  int64_t v = -1;
  ldpc_toolbox_decoder_get(dec, key, &v);
  return v;
}

static std::atomic<int> g_errors{0};

// one handle, a sequence of calls; every return code is 0 or a fault code below -1 (never -1: no frame "fails to decode"
// at this level), and without an injected failure it is 0
static int drive(const char *spec, const char *impl, size_t batch, int lanes, int expect_error) {
  const std::string alist = alist_of(spec);
  void *dec = ldpc_toolbox_decoder_ctor_alist_string(alist.c_str(), impl, "");
  if (!dec) {
    if (expect_error) g_errors++;
    else std::fprintf(stderr, "ctor failed: %s\n", ldpc_toolbox_last_error());
    return expect_error ? 0 : 1;
  }
  const size_t n = size_t(get(dec, "n")), k = size_t(get(dec, "k"));
  ldpc_toolbox_decoder_set(dec, "latency", 0);      // the batched paths (the single-launch kernels wait for device flags)
  ldpc_toolbox_decoder_set(dec, "group_size", 256);
  ldpc_toolbox_decoder_set(dec, "lanes", lanes);
  std::vector<float> llrs(batch * n, 1.0f), post(batch * n);
  std::vector<uint8_t> out(batch * k);
  std::vector<int32_t> its(batch);
  int bad = 0;
  auto check = [&](int rc, const char *what) {
    if (rc < -1) g_errors++;
    if (rc == -1 || rc > 0 || (rc != 0 && !expect_error)) {
      std::fprintf(stderr, "%s %s %s: rc %d (%s)\n", spec, impl, what, rc, ldpc_toolbox_last_error());
      bad++;
    }
  };
  for (int threads : {1, 0}) {
    ldpc_toolbox_decoder_set(dec, "lane_threads", threads);
    // host buffers: staging thread + (layered, two lanes) one enqueuing thread per lane
    check(ldpc_toolbox_decoder_decode_batch_f32(dec, out.data(), k, llrs.data(), n, batch, 12, its.data(), post.data()), "host");
    check(ldpc_toolbox_decoder_decode_batch_f32(dec, out.data(), k, llrs.data(), n, batch / 3 + 1, 12, its.data(), nullptr), "host, ragged");
    // "device" buffers (host memory under the stub), the library's own stream, with and without pacing
    for (int throttle : {0, 1}) {
      ldpc_toolbox_decoder_set(dec, "throttle", throttle);
      check(ldpc_toolbox_decoder_decode_batch_f32_device(dec, out.data(), k, llrs.data(), n, batch, 12, its.data(), post.data(), nullptr),
            "device");
    }
  }
  ldpc_toolbox_decoder_set(dec, "profiling", 1);
  check(ldpc_toolbox_decoder_decode_batch_f32(dec, out.data(), k, llrs.data(), n, 300, 6, its.data(), nullptr), "host, profiling");
  uint64_t launches = 0;
  double ms = 0;
  ldpc_toolbox_decoder_kernel_stats(dec, 2, &launches, &ms, 1);
  ldpc_toolbox_decoder_dtor(dec);
  return bad;
}

int main(int argc, char **argv) {
  const int expect_error = argc > 1 ? std::atoi(argv[1]) : 0;
  int bad = 0;
  bad += drive("nr5g:1:8", "HLMinsumf32", 2048, 2, expect_error);   // layered: lane threads
  bad += drive("nr5g:2:8", "HLTanhf32", 1500, 2, expect_error);
  bad += drive("ar4ja:1/2:1024", "Minsumf32", 1024, 2, expect_error);  // flooding, two lanes
  bad += drive("nr5g:1:8", "Minstarapproxi8", 1024, 1, expect_error);
  // two handles on two threads at once
  int rc_a = 0, rc_b = 0;
  std::thread a([&] { rc_a = drive("nr5g:1:8", "HLTanhf32", 1024, 2, expect_error); });
  std::thread b([&] { rc_b = drive("nr5g:2:8", "HLMinsumf32", 1024, 2, expect_error); });
  a.join();
  b.join();
  bad += rc_a + rc_b;
  if (expect_error && g_errors.load() == 0) {
    std::fprintf(stderr, "the injected failure never surfaced\n");
    bad++;
  }
  std::printf("tsan driver: %s (%llu kernel launches through the stub)\n", bad ? "FAILED" : "ok", hip_stub_launches());
  return bad ? 1 : 0;
}
