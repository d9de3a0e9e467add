/* Elementwise calls into the platform libm for tests/independent_restatement.py (numpy's own
 * tanh/log/exp are SIMD re-implementations, not libm).  Built with -fno-builtin so that every
 * element really is one libm call.  Test infrastructure only. */
#include <math.h>
#include <stddef.h>

#define MAP(name, fn32, fn64)                                                        \
  void name##_f32(const float *x, float *y, size_t n) {                             \
    for (size_t i = 0; i < n; i++) y[i] = fn32(x[i]);                               \
  }                                                                                  \
  void name##_f64(const double *x, double *y, size_t n) {                           \
    for (size_t i = 0; i < n; i++) y[i] = fn64(x[i]);                               \
  }

MAP(map_tanh, tanhf, tanh)
MAP(map_log, logf, log)
MAP(map_exp, expf, exp)
MAP(map_log1p, log1pf, log1p)
