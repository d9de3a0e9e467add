// Exhaustive host check of ldpc_toolbox_amd/csrc/exact_math.h against the host libm (glibc):
// every one of the 2^32 float arguments, bitwise (all NaNs count as equal).
//   g++ -O2 -std=c++17 -mfma -ffp-contract=off -pthread tools/check_exact_math.cpp -o /tmp/check_exact_math -lm
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../ldpc_toolbox_amd/csrc/exact_math.h"

using namespace ldpc::em;

struct Fn {
  const char *name;
  float (*mine)(float);
  float (*ref)(float);
};

static float r_expf(float x) { return ::expf(x); }
static float r_logf(float x) { return ::logf(x); }
static float r_log1pf(float x) { return ::log1pf(x); }
static float r_expm1f(float x) { return ::expm1f(x); }
static float r_tanhf(float x) { return ::tanhf(x); }
static float m_expf(float x) { return ldpc::em::expf(x); }
static float m_logf(float x) { return ldpc::em::logf(x); }
static float m_log1pf(float x) { return ldpc::em::log1pf(x); }
static float m_expm1f(float x) { return ldpc::em::expm1f(x); }
static float m_tanhf(float x) { return ldpc::em::tanhf(x); }

int main(int argc, char **argv) {
  const Fn fns[] = {{"expf", m_expf, r_expf},       {"logf", m_logf, r_logf},    {"log1pf", m_log1pf, r_log1pf},
                    {"expm1f", m_expm1f, r_expm1f}, {"tanhf", m_tanhf, r_tanhf}};
  const unsigned nthreads = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 4;
  int bad_total = 0;
  for (const Fn &f : fns) {
    if (argc > 1 && strcmp(argv[1], f.name) != 0) continue;
    std::atomic<unsigned long long> mism{0};
    std::atomic<unsigned> first_bad{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; t++)
      th.emplace_back([&, t] {
        unsigned long long local = 0;
        for (unsigned long long i = t; i < (1ull << 32); i += nthreads) {
          const float x = as_f32(static_cast<uint32_t>(i));
          const float a = f.mine(x), b = f.ref(x);
          if (as_u32(a) != as_u32(b) && !(a != a && b != b)) {
            if (local == 0 && first_bad.load() == 0) first_bad = static_cast<unsigned>(i);
            local++;
          }
        }
        mism += local;
      });
    for (auto &x : th) x.join();
    printf("%-8s mismatches: %llu", f.name, mism.load());
    if (mism.load()) {
      const float x = as_f32(first_bad.load());
      printf("   e.g. x=%a (0x%08x): mine %a ref %a", x, first_bad.load(), f.mine(x), f.ref(x));
      bad_total++;
    }
    printf("\n");
    fflush(stdout);
  }
  return bad_total ? 1 : 0;
}
