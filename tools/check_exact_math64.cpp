// Host check of the double-precision routines of ldpc_toolbox_amd/csrc/exact_math.h against the
// host libm (glibc): dense random sampling (uniform bit patterns + the ranges the decoder uses).
//   g++ -O2 -std=c++17 -mfma -ffp-contract=off -pthread tools/check_exact_math64.cpp -o /tmp/check64 -lm
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
#include "../ldpc_toolbox_amd/csrc/exact_math.h"
using namespace ldpc::em;

static inline uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct Fn { const char *name; double (*mine)(double); double (*ref)(double); };
static double r_exp(double x) { return ::exp(x); }
static double r_log(double x) { return ::log(x); }
static double r_log1p(double x) { return ::log1p(x); }
static double r_expm1(double x) { return ::expm1(x); }
static double r_tanh(double x) { return ::tanh(x); }
static double m_exp(double x) { return ldpc::em::exp(x); }
static double m_log(double x) { return ldpc::em::log(x); }
static double m_log1p(double x) { return ldpc::em::log1p(x); }
static double m_expm1(double x) { return ldpc::em::expm1(x); }
static double m_tanh(double x) { return ldpc::em::tanh(x); }
// the Phi rule's fused function against the composition it replaces (glibc's, and this header's own three functions)
static double m_phi(double x) { return ldpc::em::phi(x); }
static double r_phi(double x) { return -(::log(::tanh(0.5 * fmax(x, 1e-30)))); }
static double c_phi(double x) { return -(ldpc::em::log(ldpc::em::tanh(0.5 * fmax(x, 1e-30)))); }

int main(int argc, char **argv) {
  const Fn fns[] = {{"exp", m_exp, r_exp}, {"log", m_log, r_log}, {"log1p", m_log1p, r_log1p},
                    {"expm1", m_expm1, r_expm1}, {"tanh", m_tanh, r_tanh}, {"phi", m_phi, r_phi}, {"phi/em", m_phi, c_phi}};
  const unsigned long long per_thread = argc > 1 ? strtoull(argv[1], 0, 10) : 40000000ull;
  const unsigned nthreads = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 4;
  int bad_total = 0;
  for (const Fn &f : fns) {
    std::atomic<unsigned long long> mism{0}, total{0};
    std::atomic<uint64_t> first_bad{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; t++)
      th.emplace_back([&, t] {
        uint64_t s = 0x1234567ull * (t + 1) + (uint64_t)(uintptr_t)f.name;
        unsigned long long local = 0;
        for (unsigned long long i = 0; i < per_thread; i++) {
          const uint64_t r = splitmix(s);
          double x;
          switch (i & 7) {
            case 0: case 1: x = as_f64(r); break;                                              // any bit pattern
            case 2: x = (double)(int64_t)(r >> 11) * 0x1p-53 * 64.0 - 32.0; break;              // [-32, 32)
            case 3: x = -(double)(r >> 11) * 0x1p-53 * 800.0; break;                            // [-800, 0]
            case 4: x = (double)(r >> 11) * 0x1p-53; break;                                     // [0, 1)
            case 5: x = 1.0 + ((double)(int64_t)(r >> 11) * 0x1p-53 - 0.5) * 0.25; break;       // near 1
            case 6: x = as_f64((r & 0x800fffffffffffffull) | ((uint64_t)(0x3ff - 60 + (r >> 52) % 70) << 52)); break;  // 2^-60..2^9
            default: x = ((double)(int64_t)(r >> 11) * 0x1p-53 - 0.5) * 2.0; break;             // (-1, 1)
          }
          if (f.name[0] == 'p' && (i & 8)) {   // phi: half of the samples positive on a log scale, 2^-120 .. 2^7, and around 2 * {1, 19.4, 22}
            const uint64_t e = 0x3ff - 120 + (r >> 52) % 128;
            x = as_f64((r & 0x000fffffffffffffull) | (e << 52));
            if ((i & 0x30) == 0x30) x = 2.0 * ((i & 0x40) ? 19.4 : ((i & 0x80) ? 22.0 : 1.0)) + ((double)(int64_t)(r >> 11) * 0x1p-53 - 0.5) * 0.01;
          }
          const double a = f.mine(x), b = f.ref(x);
          if (as_u64(a) != as_u64(b) && !(a != a && b != b)) {
            if (local == 0 && first_bad.load() == 0) first_bad = as_u64(x);
            local++;
          }
        }
        mism += local;
        total += per_thread;
      });
    for (auto &x : th) x.join();
    printf("%-6s %llu arguments, mismatches: %llu", f.name, total.load(), mism.load());
    if (mism.load()) {
      const double x = as_f64(first_bad.load());
      printf("   e.g. x=%a: mine %a ref %a", x, f.mine(x), f.ref(x));
      bad_total++;
    }
    printf("\n");
    fflush(stdout);
  }
  // the class boundaries of the routines (range-reduction thresholds, every k * ln2 up to overflow,
  // the sqrt(2) normalisation threshold of log1p, tiny/huge cut-offs): every double within 2^18
  // ulps on both sides of each boundary, for every function
  std::vector<double> edges = {0.0, 0x1p-54, 0x1p-55, 0x1p-29, 0x1p-28, 0x1p-20, 0.41421356237309503, -0.29289321881345248, 1.0, -1.0,
                               2.0, 22.0, 0x1p53, 709.782712893384, -745.13321910194111, 0x1p-1022, 0.5, 0.25, -0.25,
                               1.4142135623730951, 0.70710678118654757, 2.8284271247461903, 0.41421356237309515 * 2 + 1};
  // phi: h = x / 2 crosses tanh's classes at 2^-55, 1, 22 and expm1's at (k +- 0.5) ln2 / 2; tanh(h) crosses log's "close to 1" at atanh(0.9375)
  for (double hb : {0x1p-55, 1.0, 22.0, 19.407, 1.7166400194, 0.5 * 0.34657359027997264, 0.5 * 1.0397207708399179})
    edges.push_back(2.0 * hb);
  edges.push_back(1e-30);
  edges.push_back(2e-30);
  for (int k = 1; k <= 1100; k++) {
    edges.push_back((k - 0.5) * 0.69314718055994529);   // rounding boundary of k = round(x / ln2)
    edges.push_back(k * 0.69314718055994529);
  }
  for (int e = -60; e <= 60; e++) {                     // 1 + x crossing sqrt(2) * 2^e, and powers of two
    edges.push_back(ldexp(1.4142135623730951, e) - 1.0);
    edges.push_back(ldexp(1.0, e) - 1.0);
    edges.push_back(ldexp(1.0, e));
  }
  for (const Fn &f : fns) {
    std::atomic<unsigned long long> mism{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; t++)
      th.emplace_back([&, t] {
        unsigned long long local = 0;
        for (size_t i = t; i < edges.size(); i += nthreads)
          for (int sign = 0; sign < 2; sign++) {
            const uint64_t centre = as_u64(sign ? -edges[i] : edges[i]);
            for (int64_t d = -(1 << 18); d <= (1 << 18); d++) {
              const double x = as_f64(centre + (uint64_t)d);
              const double a = f.mine(x), b = f.ref(x);
              if (as_u64(a) != as_u64(b) && !(a != a && b != b)) local++;
            }
          }
        mism += local;
      });
    for (auto &x : th) x.join();
    printf("%-6s boundaries (%zu edges x 2 signs x 2^19 neighbours), mismatches: %llu\n", f.name, edges.size(), mism.load());
    if (mism.load()) bad_total++;
  }
  return bad_total ? 1 : 0;
}
