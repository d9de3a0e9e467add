/* The CPU oracle (oracle/ldpc_oracle.c) under AddressSanitizer + UBSan: decodes a few noisy
 * frames with every implementation name given on the command line.  argv[1] = alist file. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../oracle/ldpc_oracle.h"

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  long len = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *text = (char *)malloc((size_t)len + 1);
  if (fread(text, 1, (size_t)len, f) != (size_t)len) return 2;
  text[len] = 0;
  fclose(f);
  oracle_graph *g = oracle_graph_from_alist(text);
  free(text);
  if (!g) return 3;
  const size_t n = oracle_graph_cols(g), m = oracle_graph_rows(g), batch = 6;
  float *llrs = (float *)malloc(batch * n * sizeof(float));
  unsigned state = 12345u;
  for (size_t i = 0; i < batch * n; i++) {
    state = state * 1664525u + 1013904223u;
    const float u = (float)(state >> 8) / 16777216.0f;           /* all-zero codeword: LLR mostly > 0 */
    llrs[i] = 6.0f * u - 1.2f;
  }
  for (size_t i = 0; i < n; i++) llrs[i] = 3.0f;                  /* frame 0: clean, 0 iterations */
  uint8_t *bits = (uint8_t *)malloc(batch * n), *syn = (uint8_t *)malloc(m ? m : 1);
  int32_t *its = (int32_t *)malloc(batch * sizeof(int32_t));
  double *post = (double *)malloc(batch * n * sizeof(double));
  int bad = 0;
  for (int a = 2; a < argc; a++) {
    if (oracle_decode_batch_f32(g, argv[a], llrs, batch, 12, 2, bits, its, post) != 0) {
      fprintf(stderr, "%s: batch decode failed\n", argv[a]);
      bad++;
      continue;
    }
    if (its[0] != 0) bad++;
    for (size_t b = 0; b < batch; b++) {
      const size_t w = oracle_syndrome(g, bits + b * n, syn);
      if ((its[b] >= 0) != (w == 0)) {
        fprintf(stderr, "%s frame %zu: iterations %d but syndrome weight %zu\n", argv[a], b, its[b], w);
        bad++;
      }
    }
  }
  if (oracle_decoder_new(g, "NoSuchRule") != NULL) bad++;
  free(llrs);
  free(bits);
  free(syn);
  free(its);
  free(post);
  oracle_graph_free(g);
  if (bad) return 1;
  puts("oracle sanitizer driver: ok");
  return 0;
}
