/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see ldpc_oracle.h for scope, citations and the
 * parity status).  Plain C11; build: make -C oracle
 */
#include "ldpc_oracle.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- graph: adjacency lists in the reference's insertion order ------------------------ */

struct oracle_graph {
  size_t nrows, ncols, nedges;
  size_t **rows; /* rows[r][i]: i-th variable of check r (iter_row order) */
  size_t *row_len, *row_cap;
  size_t **cols; /* cols[c][j]: j-th check of variable c (iter_col order) */
  size_t *col_len, *col_cap;
  size_t max_row_len;
};

static void list_push(size_t **l, size_t *len, size_t *cap, size_t v) {
  if (*len == *cap) {
    *cap = *cap ? *cap * 2 : 4;
    *l = (size_t *)realloc(*l, *cap * sizeof(size_t));
  }
  (*l)[(*len)++] = v;
}

/* sparse.rs:114-119 -- insert de-duplicates (search in the column) and appends */
static void graph_insert(oracle_graph *g, size_t r, size_t c) {
  for (size_t j = 0; j < g->col_len[c]; j++)
    if (g->cols[c][j] == r) return;
  list_push(&g->rows[r], &g->row_len[r], &g->row_cap[r], c);
  list_push(&g->cols[c], &g->col_len[c], &g->col_cap[c], r);
  g->nedges++;
}

void oracle_graph_free(oracle_graph *g) {
  if (!g) return;
  for (size_t r = 0; r < g->nrows; r++) free(g->rows[r]);
  for (size_t c = 0; c < g->ncols; c++) free(g->cols[c]);
  free(g->rows);
  free(g->row_len);
  free(g->row_cap);
  free(g->cols);
  free(g->col_len);
  free(g->col_cap);
  free(g);
}

/* next '\n'-separated line of [*p, end); returns 0 when the input is exhausted */
static int next_line(const char **p, const char *end, const char **ls, const char **le, int *done) {
  if (*done) return 0;
  const char *s = *p;
  const char *nl = (const char *)memchr(s, '\n', (size_t)(end - s));
  *ls = s;
  if (nl) {
    *le = nl;
    *p = nl + 1;
  } else {
    *le = end;
    *done = 1;
  }
  return 1;
}

/* whitespace-separated unsigned token; 1 = got one, 0 = end of line, -1 = not a number */
static int next_uint(const char **s, const char *le, size_t *v) {
  const char *p = *s;
  while (p < le && isspace((unsigned char)*p)) p++;
  if (p >= le) {
    *s = p;
    return 0;
  }
  size_t acc = 0;
  int digits = 0, bad = 0;
  if (*p == '+') p++;
  while (p < le && !isspace((unsigned char)*p)) {
    if (*p < '0' || *p > '9')
      bad = 1;
    else {
      acc = acc * 10 + (size_t)(*p - '0');
      digits++;
    }
    p++;
  }
  *s = p;
  *v = acc;
  return (bad || !digits) ? -1 : 1;
}

/* sparse.rs:352-389 */
oracle_graph *oracle_graph_from_alist(const char *alist) {
  const char *p = alist, *end = alist + strlen(alist), *ls, *le;
  int done = 0;
  if (!next_line(&p, end, &ls, &le, &done)) return NULL;
  size_t ncols, nrows;
  if (next_uint(&ls, le, &ncols) != 1) return NULL;
  if (next_uint(&ls, le, &nrows) != 1) return NULL;
  oracle_graph *g = (oracle_graph *)calloc(1, sizeof(*g));
  g->nrows = nrows;
  g->ncols = ncols;
  g->rows = (size_t **)calloc(nrows ? nrows : 1, sizeof(size_t *));
  g->row_len = (size_t *)calloc(nrows ? nrows : 1, sizeof(size_t));
  g->row_cap = (size_t *)calloc(nrows ? nrows : 1, sizeof(size_t));
  g->cols = (size_t **)calloc(ncols ? ncols : 1, sizeof(size_t *));
  g->col_len = (size_t *)calloc(ncols ? ncols : 1, sizeof(size_t));
  g->col_cap = (size_t *)calloc(ncols ? ncols : 1, sizeof(size_t));
  /* skip max weights, column weights, row weights */
  next_line(&p, end, &ls, &le, &done);
  next_line(&p, end, &ls, &le, &done);
  next_line(&p, end, &ls, &le, &done);
  for (size_t c = 0; c < ncols; c++) {
    if (!next_line(&p, end, &ls, &le, &done)) {
      oracle_graph_free(g);
      return NULL;
    }
    size_t r;
    int t;
    while ((t = next_uint(&ls, le, &r)) != 0) {
      if (t < 0 || r > nrows) {
        oracle_graph_free(g);
        return NULL;
      }
      if (r != 0) graph_insert(g, r - 1, c); /* 0 = padding */
    }
  }
  for (size_t r = 0; r < nrows; r++)
    if (g->row_len[r] > g->max_row_len) g->max_row_len = g->row_len[r];
  return g;
}

size_t oracle_graph_rows(const oracle_graph *g) { return g->nrows; }
size_t oracle_graph_cols(const oracle_graph *g) { return g->ncols; }
size_t oracle_graph_edges(const oracle_graph *g) { return g->nedges; }

/* flooding.rs:57 / horizontal_layered.rs:55: the pre-check runs on the RAW f64 input */
static int check_llrs_f64in(const oracle_graph *g, const double *llrs) {
  for (size_t r = 0; r < g->nrows; r++) {
    size_t ones = 0;
    for (size_t i = 0; i < g->row_len[r]; i++)
      if (llrs[g->rows[r][i]] <= 0.0) ones++;
    if (ones % 2 == 1) return 0;
  }
  return 1;
}

enum { RULE_PHI, RULE_TANH, RULE_MINSTARAPPROX, RULE_AMINSTAR, RULE_MINSUM };

/* ---- f32 instantiation ------------------------------------------------------------ */
#define REAL float
#define FN(name) name##_f32
#define R_TANH tanhf
#define R_LOG logf
#define R_EXP expf
#define R_LOG1P log1pf
#define R_FABS fabsf
#define R_FMIN fminf
#define R_FMAX fmaxf
#define TANH_CLAMP 9.0
#define PHI_MIN_X ((float)1e-30)
#include "rules.inc"
#undef REAL
#undef FN
#undef R_TANH
#undef R_LOG
#undef R_EXP
#undef R_LOG1P
#undef R_FABS
#undef R_FMIN
#undef R_FMAX
#undef TANH_CLAMP
#undef PHI_MIN_X

/* ---- f64 instantiation ------------------------------------------------------------ */
#define REAL double
#define FN(name) name##_f64
#define R_TANH tanh
#define R_LOG log
#define R_EXP exp
#define R_LOG1P log1p
#define R_FABS fabs
#define R_FMIN fmin
#define R_FMAX fmax
#define TANH_CLAMP 18.0
#define PHI_MIN_X (1e-30)
#include "rules.inc"
#undef REAL
#undef FN

/* ---- names (factory.rs:240-277 + the added Minsum family) ----------------------------- */

struct oracle_decoder {
  int rule, is_f64, layered;
  oracle_graph *g; /* private copy */
  void *impl;
};

static int parse_name(const char *name, int *rule, int *is_f64, int *layered) {
  const char *p = name;
  *layered = 0;
  if (strncmp(p, "HL", 2) == 0) {
    *layered = 1;
    p += 2;
  }
  static const struct {
    const char *stem;
    int rule;
  } stems[] = {{"Phi", RULE_PHI},
               {"Tanh", RULE_TANH},
               {"Minstarapprox", RULE_MINSTARAPPROX},
               {"Aminstar", RULE_AMINSTAR},
               {"Minsum", RULE_MINSUM}};
  for (size_t i = 0; i < sizeof(stems) / sizeof(stems[0]); i++) {
    size_t l = strlen(stems[i].stem);
    if (strncmp(p, stems[i].stem, l) == 0) {
      if (strcmp(p + l, "f32") == 0) {
        *rule = stems[i].rule;
        *is_f64 = 0;
        return 1;
      }
      if (strcmp(p + l, "f64") == 0) {
        *rule = stems[i].rule;
        *is_f64 = 1;
        return 1;
      }
    }
  }
  return 0;
}

static oracle_graph *graph_clone(const oracle_graph *g) {
  oracle_graph *c = (oracle_graph *)calloc(1, sizeof(*c));
  *c = *g;
  c->rows = (size_t **)calloc(g->nrows ? g->nrows : 1, sizeof(size_t *));
  c->row_len = (size_t *)calloc(g->nrows ? g->nrows : 1, sizeof(size_t));
  c->row_cap = (size_t *)calloc(g->nrows ? g->nrows : 1, sizeof(size_t));
  c->cols = (size_t **)calloc(g->ncols ? g->ncols : 1, sizeof(size_t *));
  c->col_len = (size_t *)calloc(g->ncols ? g->ncols : 1, sizeof(size_t));
  c->col_cap = (size_t *)calloc(g->ncols ? g->ncols : 1, sizeof(size_t));
  for (size_t r = 0; r < g->nrows; r++) {
    c->row_len[r] = c->row_cap[r] = g->row_len[r];
    c->rows[r] = (size_t *)malloc((g->row_len[r] ? g->row_len[r] : 1) * sizeof(size_t));
    memcpy(c->rows[r], g->rows[r], g->row_len[r] * sizeof(size_t));
  }
  for (size_t v = 0; v < g->ncols; v++) {
    c->col_len[v] = c->col_cap[v] = g->col_len[v];
    c->cols[v] = (size_t *)malloc((g->col_len[v] ? g->col_len[v] : 1) * sizeof(size_t));
    memcpy(c->cols[v], g->cols[v], g->col_len[v] * sizeof(size_t));
  }
  return c;
}

oracle_decoder *oracle_decoder_new(const oracle_graph *g, const char *implementation) {
  int rule, is_f64, layered;
  if (!g || !implementation || !parse_name(implementation, &rule, &is_f64, &layered)) return NULL;
  oracle_decoder *d = (oracle_decoder *)calloc(1, sizeof(*d));
  d->rule = rule;
  d->is_f64 = is_f64;
  d->layered = layered;
  d->g = graph_clone(g);
  if (layered)
    d->impl = is_f64 ? (void *)layered_new_f64(d->g, rule) : (void *)layered_new_f32(d->g, rule);
  else
    d->impl = is_f64 ? (void *)flooding_new_f64(d->g, rule) : (void *)flooding_new_f32(d->g, rule);
  return d;
}

void oracle_decoder_free(oracle_decoder *d) {
  if (!d) return;
  if (d->layered) {
    if (d->is_f64)
      layered_free_f64((layered_f64 *)d->impl);
    else
      layered_free_f32((layered_f32 *)d->impl);
  } else {
    if (d->is_f64)
      flooding_free_f64((flooding_f64 *)d->impl);
    else
      flooding_free_f32((flooding_f32 *)d->impl);
  }
  oracle_graph_free(d->g);
  free(d);
}

int oracle_decode(oracle_decoder *d, const double *llrs, size_t n, uint32_t max_iterations,
                  uint8_t *bits, double *posterior, uint32_t *iterations) {
  if (!d || n != d->g->ncols) return -1; /* assert_eq!(llrs.len(), n) */
  if (d->layered)
    return d->is_f64 ? layered_decode_f64((layered_f64 *)d->impl, llrs, max_iterations, bits,
                                          posterior, iterations)
                     : layered_decode_f32((layered_f32 *)d->impl, llrs, max_iterations, bits,
                                          posterior, iterations);
  return d->is_f64 ? flooding_decode_f64((flooding_f64 *)d->impl, llrs, max_iterations, bits,
                                         posterior, iterations)
                   : flooding_decode_f32((flooding_f32 *)d->impl, llrs, max_iterations, bits,
                                         posterior, iterations);
}

/* ---- batch driver: N worker threads, one private decoder each ----------------------- */

typedef struct {
  const oracle_graph *g;
  const char *impl;
  const float *llrs;
  size_t n, begin, end;
  uint32_t max_iterations;
  uint8_t *bits;
  int32_t *iterations;
  double *posterior;
  int status;
} batch_job;

static void *batch_worker(void *arg) {
  batch_job *j = (batch_job *)arg;
  oracle_decoder *d = oracle_decoder_new(j->g, j->impl);
  if (!d) {
    j->status = -1;
    return NULL;
  }
  size_t n = j->n;
  double *wide = (double *)malloc((n ? n : 1) * sizeof(double));
  uint8_t *tmp_bits = (uint8_t *)malloc(n ? n : 1);
  for (size_t b = j->begin; b < j->end; b++) {
    /* c_api/decoder.rs:69-72: the f32 entry widens to f64 */
    for (size_t v = 0; v < n; v++) wide[v] = (double)j->llrs[b * n + v];
    uint32_t it = 0;
    int ok = oracle_decode(d, wide, n, j->max_iterations, j->bits ? j->bits + b * n : tmp_bits,
                           j->posterior ? j->posterior + b * n : NULL, &it);
    if (ok < 0) {
      j->status = -1;
      break;
    }
    if (j->iterations) j->iterations[b] = ok ? (int32_t)it : -1;
  }
  free(wide);
  free(tmp_bits);
  oracle_decoder_free(d);
  return NULL;
}

int oracle_decode_batch_f32(const oracle_graph *g, const char *implementation, const float *llrs,
                            size_t batch, uint32_t max_iterations, unsigned threads, uint8_t *bits,
                            int32_t *iterations, double *posterior) {
  if (!g || !implementation) return -1;
  if (threads == 0) threads = 1;
  if (threads > batch) threads = batch ? (unsigned)batch : 1;
  batch_job *jobs = (batch_job *)calloc(threads, sizeof(*jobs));
  pthread_t *tids = (pthread_t *)calloc(threads, sizeof(*tids));
  size_t per = batch / threads, extra = batch % threads, at = 0;
  for (unsigned t = 0; t < threads; t++) {
    size_t cnt = per + (t < extra ? 1 : 0);
    jobs[t] = (batch_job){g, implementation, llrs, g->ncols, at, at + cnt, max_iterations,
                          bits, iterations, posterior, 0};
    at += cnt;
  }
  for (unsigned t = 1; t < threads; t++) pthread_create(&tids[t], NULL, batch_worker, &jobs[t]);
  batch_worker(&jobs[0]);
  int status = jobs[0].status;
  for (unsigned t = 1; t < threads; t++) {
    pthread_join(tids[t], NULL);
    if (jobs[t].status) status = jobs[t].status;
  }
  free(jobs);
  free(tids);
  return status;
}

/* simulation/puncturing.rs:83-101 */
size_t oracle_depuncture(const uint8_t *pattern, size_t pattern_len, const double *llrs,
                         size_t llrs_len, double *out, size_t out_cap) {
  size_t trues = 0;
  for (size_t i = 0; i < pattern_len; i++) trues += pattern[i] ? 1 : 0;
  if (trues == 0 || llrs_len % trues != 0) return 0;
  size_t block = llrs_len / trues, total = pattern_len * block;
  if (total > out_cap) return 0;
  for (size_t i = 0; i < total; i++) out[i] = 0.0;
  size_t j = 0;
  for (size_t k = 0; k < pattern_len; k++) {
    if (!pattern[k]) continue;
    memcpy(out + k * block, llrs + j * block, block * sizeof(double));
    j++;
  }
  return total;
}
