/* What replaces the reference's one-frame loop (Worker::simulate, src/simulation/ber.rs:462-466: `decoder.decode(&llrs, max_iter)` once
 * per frame) in a C caller: ONE call of the batched extension over B frames (include/ldpc_toolbox.h, part 2), host buffers, with the
 * decoder's own straggler pooling switched on.  The program decodes the same frames three ways -- frame by frame through the
 * reference's scalar symbol, as one batch, and as one batch with "pooling" = 1 -- and checks that all three agree bit for bit.
 *
 *   gcc -O2 -Iinclude examples/batched_worker.c -Lldpc_toolbox_amd/lib -lldpc_toolbox -Wl,-rpath,$PWD/ldpc_toolbox_amd/lib -lm \
 *       -o /tmp/batched_worker && /tmp/batched_worker code.alist [implementation] [frames] [sigma]
 * Exit code 0: the three agree (and at least one frame decoded).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ldpc_toolbox.h"

static uint32_t lcg(uint32_t *s) { return *s = *s * 1664525u + 1013904223u; }
static double uniform01(uint32_t *s) { return (lcg(s) >> 8) * (1.0 / 16777216.0) + 1e-9; }
static double gauss(uint32_t *s) { return sqrt(-2.0 * log(uniform01(s))) * cos(6.283185307179586 * uniform01(s)); }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s code.alist [implementation] [frames] [sigma]\n", argv[0]);
    return 2;
  }
  const char *impl = argc > 2 ? argv[2] : "Minsumf32";
  const size_t frames = argc > 3 ? (size_t)atol(argv[3]) : 6000;
  const double sigma = argc > 4 ? atof(argv[4]) : 0.78;
  const uint32_t max_iterations = 100; /* the reference CLI's default, src/cli/ber.rs:64-66 */
  FILE *f = fopen(argv[1], "r");
  unsigned n = 0, m = 0;
  if (!f || fscanf(f, "%u %u", &n, &m) != 2) return 2;
  fclose(f);
  const unsigned k = n - m;
  void *enc = ldpc_toolbox_encoder_ctor(argv[1], "");
  void *dec = ldpc_toolbox_decoder_ctor(argv[1], impl, "");
  if (!enc || !dec) {
    fprintf(stderr, "constructor returned NULL: %s\n", ldpc_toolbox_last_error());
    return 3;
  }
  uint8_t *msg = malloc(k), *cw = malloc(n);
  float *llrs = malloc(frames * n * sizeof(float));
  uint8_t *bits[3];
  int32_t *its[3];
  for (int v = 0; v < 3; v++) {
    bits[v] = calloc(frames, k);
    its[v] = calloc(frames, sizeof(int32_t));
  }
  uint32_t seed = 11;
  for (size_t fr = 0; fr < frames; fr++) {
    for (unsigned i = 0; i < k; i++) msg[i] = (lcg(&seed) >> 16) & 1;
    ldpc_toolbox_encoder_encode(enc, cw, n, msg, k);
    for (unsigned i = 0; i < n; i++) llrs[fr * n + i] = (float)(-2.0 * ((cw[i] ? 1.0 : -1.0) + sigma * gauss(&seed)) / (sigma * sigma));
  }
  /* 1: the reference's way, one frame per call (first 64 frames: each call is a launch sequence of its own) */
  const size_t scalar = frames < 64 ? frames : 64;
  double t0 = now();
  for (size_t fr = 0; fr < scalar; fr++) its[0][fr] = ldpc_toolbox_decoder_decode_f32(dec, bits[0] + fr * k, k, llrs + fr * n, n, max_iterations);
  const double t_scalar = now() - t0;
  /* 2 and 3: one batched call, without and with straggler pooling.  (Groups of 1024 frames: a call of this size is then several
   * chunks of the decoder, which is what pooling works on -- a production caller with 10^5 frames per call keeps the default.
   * The first batched call is not timed: it allocates the pinned staging and the workspace.) */
  if (ldpc_toolbox_decoder_set(dec, "group_size", 1024) != 0) return 4;
  if (ldpc_toolbox_decoder_decode_batch_f32(dec, bits[1], k, llrs, n, frames, max_iterations, its[1], NULL) != 0) return 5;
  double t_batch[2];
  int64_t pooled = 0;
  for (int pooling = 0; pooling < 2; pooling++) {
    if (ldpc_toolbox_decoder_set(dec, "pooling", pooling) != 0) return 4;
    t0 = now();
    const int32_t rc = ldpc_toolbox_decoder_decode_batch_f32(dec, bits[1 + pooling], k, llrs, n, frames, max_iterations, its[1 + pooling], NULL);
    t_batch[pooling] = now() - t0;
    if (rc != 0) {
      fprintf(stderr, "decode_batch failed (%d): %s\n", rc, ldpc_toolbox_last_error());
      return 5;
    }
    if (pooling) ldpc_toolbox_decoder_get(dec, "last_pooled", &pooled);
  }
  size_t decoded = 0, differ = 0;
  for (size_t fr = 0; fr < frames; fr++) {
    decoded += its[1][fr] >= 0;
    differ += its[1][fr] != its[2][fr] || memcmp(bits[1] + fr * k, bits[2] + fr * k, k) != 0;
    if (fr < scalar) differ += its[0][fr] != its[1][fr] || memcmp(bits[0] + fr * k, bits[1] + fr * k, k) != 0;
  }
  printf("%s %s: %zu frames, %zu decoded; scalar calls %.0f frames/s, one batch %.0f frames/s, one batch with pooling %.0f frames/s "
         "(%lld frames took the second pass); results %s\n",
         argv[1], impl, frames, decoded, scalar / t_scalar, frames / t_batch[0], frames / t_batch[1], (long long)pooled,
         differ ? "DIFFER" : "identical");
  ldpc_toolbox_decoder_dtor(dec);
  ldpc_toolbox_encoder_dtor(enc);
  return (differ || decoded == 0) ? 1 : 0;
}
