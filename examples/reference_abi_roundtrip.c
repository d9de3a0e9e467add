/* A C program written against the REFERENCE's header only -- the nine symbols of
 * daniestevez/ldpc-toolbox's include/ldpc_toolbox.h (part 1 of this repo's header) -- linked against
 * this library instead of the reference's cdylib: encode, add noise, decode on the GPU, compare.
 *
 *   gcc -O2 -Iinclude examples/reference_abi_roundtrip.c -Lldpc_toolbox_amd/lib -lldpc_toolbox \
 *       -Wl,-rpath,$PWD/ldpc_toolbox_amd/lib -lm -o /tmp/roundtrip && /tmp/roundtrip code.alist
 *
 * Exit code 0: every frame decoded to the transmitted message.  argv[1]: alist file; argv[2]
 * (optional): decoder implementation (default Phif64, the reference CLI's default).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ldpc_toolbox.h"

static uint32_t lcg(uint32_t *s) { return *s = *s * 1664525u + 1013904223u; }
static double uniform01(uint32_t *s) { return (lcg(s) >> 8) * (1.0 / 16777216.0) + 1e-9; }
static double gauss(uint32_t *s) { return sqrt(-2.0 * log(uniform01(s))) * cos(6.283185307179586 * uniform01(s)); }

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s code.alist [implementation]\n", argv[0]);
    return 2;
  }
  const char *impl = argc > 2 ? argv[2] : "Phif64";
  FILE *f = fopen(argv[1], "r");
  if (!f) return 2;
  unsigned n = 0, m = 0;
  if (fscanf(f, "%u %u", &n, &m) != 2) return 2;
  fclose(f);
  const unsigned k = n - m;

  void *enc = ldpc_toolbox_encoder_ctor(argv[1], "");
  void *dec = ldpc_toolbox_decoder_ctor(argv[1], impl, "");
  if (!enc || !dec) {
    fprintf(stderr, "constructor returned NULL (no GPU, bad alist or unknown implementation)\n");
    return 3;
  }
  uint8_t *msg = malloc(k), *cw = malloc(n), *out = malloc(k);
  double *llr64 = malloc(n * sizeof(double));
  float *llr32 = malloc(n * sizeof(float));
  const double sigma = 0.55; /* comfortably above threshold for a rate-1/2 code */
  uint32_t seed = 7;
  int bad = 0;
  for (int frame = 0; frame < 8; frame++) {
    for (unsigned i = 0; i < k; i++) msg[i] = (lcg(&seed) >> 16) & 1;
    ldpc_toolbox_encoder_encode(enc, cw, n, msg, k);
    for (unsigned i = 0; i < n; i++) { /* BPSK: bit 1 -> +1, LLR = -2 y / sigma^2 */
      const double y = (cw[i] ? 1.0 : -1.0) + sigma * gauss(&seed);
      llr64[i] = -2.0 * y / (sigma * sigma);
      llr32[i] = (float)llr64[i];
    }
    const int32_t it = (frame & 1) ? ldpc_toolbox_decoder_decode_f32(dec, out, k, llr32, n, 100)
                                   : ldpc_toolbox_decoder_decode_f64(dec, out, k, llr64, n, 100);
    const int same = memcmp(out, msg, k) == 0;
    printf("frame %d: %s, %d iterations, message %s\n", frame, it >= 0 ? "decoded" : "FAILED", it, same ? "recovered" : "WRONG");
    if (it < 0 || !same) bad++;
  }
  /* a wrong LLR length (a panic in the reference) is an error code below -1 here, not a crash and
   * never the "-1 = no codeword found" of a real decode */
  if (ldpc_toolbox_decoder_decode_f32(dec, out, k, llr32, n - 1, 10) >= -1) bad++;
  ldpc_toolbox_decoder_dtor(dec);
  ldpc_toolbox_encoder_dtor(enc);
  free(msg); free(cw); free(out); free(llr64); free(llr32);
  return bad ? 1 : 0;
}
