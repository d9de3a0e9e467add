/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see ldpc_oracle.h for scope, citations and the
 * parity status).  Plain C11; build: make -C oracle
 */
#ifndef _POSIX_C_SOURCE
#define _POSIX_C_SOURCE 200809L /* pthread_barrier_t, clock_gettime under -std=c11 */
#endif
#include "ldpc_oracle.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

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

/* ---- 8-bit quantised arithmetics (arithmetic.rs:582-897 Minstarapproxi8*, :1074-1304 Aminstari8*) --
 * Llr = CheckMessage = VarMessage = i8, VarLlr = i16.  Options per implementation name:
 * Jones clipping of the variable-node sum (:806-810), partial hard limiting of check messages
 * (:812-824), clipping of the channel LLR of degree-1 variables (:826-842). */

typedef struct {
  int aminstar, jones, hardlimit, deg1clip;
  int8_t table[128];
  size_t table_len;
} i8_opts;

/* arithmetic.rs:588-601: round(8 ln(1 + e^(-t/8))) for t = 0.. while > 0 */
static void i8_table(i8_opts *o) {
  o->table_len = 0;
  for (int t = 0; t <= 127; t++) {
    double v = round(8.0 * log1p(exp(-((double)t / 8.0))));
    int8_t x = (int8_t)v;
    if (x > 0)
      o->table[o->table_len++] = x;
    else
      break;
  }
}
static int i8_lookup(const i8_opts *o, int x) { return (x >= 0 && (size_t)x < o->table_len) ? o->table[x] : 0; }
static int i8_clip(int x) { return x >= 127 ? 127 : (x <= -127 ? -127 : x); } /* :609-617 */
static int i8_hardlimit(const i8_opts *o, int x) {
  if (!o->hardlimit) return x;
  return x <= -100 ? -127 : (x >= 100 ? 127 : x);
}
static int i8_sat_add(int a, int b) {
  int s = a + b;
  return s > 127 ? 127 : (s < -128 ? -128 : s);
}
/* arithmetic.rs:690-699 */
static int i8_quantize(double llr) {
  double x = 8.0 * llr;
  if (x >= 127.0) return 127;
  if (x <= -127.0) return -127;
  if (x != x) return 0; /* Rust's float -> int `as` cast turns NaN into 0 */
  return (int)round(x);
}

/* check node on x[0..d) (i8 values) -> out[0..d); -1 where the reference panics */
static int i8_check_node(const i8_opts *o, const int *x, size_t d, int *out) {
  if (!o->aminstar) { /* :722-753 */
    for (size_t i = 0; i < d; i++) {
      unsigned sign = 0;
      int have = 0, acc = 0;
      for (size_t j = 0; j < d; j++) {
        if (j == i) continue;
        int v = x[j];
        if (v < 0) sign ^= 1u;
        v = abs(v);
        if (!have) {
          acc = v;
          have = 1;
        } else {
          int m = (v < acc ? v : acc) - i8_lookup(o, abs(v - acc));
          acc = m > 0 ? m : 0;
        }
      }
      if (!have) return -1;
      out[i] = i8_hardlimit(o, sign == 0 ? acc : -acc);
    }
    return 0;
  }
  /* A-Min*: :1134-1191; min_by_key keeps the first minimum */
  if (d == 0) return -1;
  size_t argmin = 0;
  for (size_t i = 1; i < d; i++)
    if (abs(x[i]) < abs(x[argmin])) argmin = i;
  unsigned sign = 0;
  int have = 0, delta = 0;
  for (size_t j = 0; j < d; j++) {
    int v = x[j];
    if (v < 0) sign ^= 1u;
    if (j != argmin) {
      v = abs(v);
      if (!have) {
        delta = v;
        have = 1;
      } else {
        int m = (v < delta ? v : delta) - i8_lookup(o, abs(v - delta)) + i8_lookup(o, i8_sat_add(v, delta));
        delta = m > 0 ? m : 0;
      }
    }
  }
  if (!have) return -1;
  int dhl = i8_hardlimit(o, delta);
  out[argmin] = ((sign != 0) ^ (x[argmin] < 0)) ? -dhl : dhl;
  int vmin = abs(x[argmin]);
  int m = (delta < vmin ? delta : vmin) - i8_lookup(o, abs(delta - vmin)) + i8_lookup(o, i8_sat_add(delta, vmin));
  delta = m > 0 ? m : 0;
  dhl = i8_hardlimit(o, delta);
  for (size_t j = 0; j < d; j++)
    if (j != argmin) out[j] = ((sign != 0) ^ (x[j] < 0)) ? -dhl : dhl;
  return 0;
}

typedef struct {
  size_t source;
  int8_t value;
} msg_i8;
typedef struct {
  msg_i8 *m;
  size_t len;
} msglist_i8;

typedef struct {
  i8_opts o;
  const oracle_graph *g;
  int layered;
  /* flooding */
  int8_t *input_llrs, *output_llrs;
  msglist_i8 *check_messages, *variable_messages;
  /* layered */
  int16_t *llrs;
  int8_t **rcv;
  int *x, *out;
} decoder_i8;

static void send_i8(msglist_i8 *per_destination, size_t source, size_t destination, int8_t value) {
  msglist_i8 *l = &per_destination[destination];
  for (size_t i = 0; i < l->len; i++)
    if (l->m[i].source == source) {
      l->m[i].value = value;
      return;
    }
  abort();
}

static msglist_i8 *msglists_i8_new(size_t count, size_t *const *lists, const size_t *lens) {
  msglist_i8 *l = (msglist_i8 *)calloc(count ? count : 1, sizeof(*l));
  for (size_t i = 0; i < count; i++) {
    l[i].len = lens[i];
    l[i].m = (msg_i8 *)calloc(lens[i] ? lens[i] : 1, sizeof(msg_i8));
    for (size_t j = 0; j < lens[i]; j++) l[i].m[j].source = lists[i][j];
  }
  return l;
}

static decoder_i8 *decoder_i8_new(const oracle_graph *g, const i8_opts *o, int layered) {
  decoder_i8 *d = (decoder_i8 *)calloc(1, sizeof(*d));
  d->o = *o;
  i8_table(&d->o);
  d->g = g;
  d->layered = layered;
  size_t n = g->ncols ? g->ncols : 1, w = g->max_row_len ? g->max_row_len : 1;
  d->x = (int *)calloc(w, sizeof(int));
  d->out = (int *)calloc(w, sizeof(int));
  if (layered) {
    d->llrs = (int16_t *)calloc(n, sizeof(int16_t));
    d->rcv = (int8_t **)calloc(g->nrows ? g->nrows : 1, sizeof(int8_t *));
    for (size_t r = 0; r < g->nrows; r++) d->rcv[r] = (int8_t *)calloc(g->row_len[r] ? g->row_len[r] : 1, 1);
  } else {
    d->input_llrs = (int8_t *)calloc(n, 1);
    d->output_llrs = (int8_t *)calloc(n, 1);
    d->check_messages = msglists_i8_new(g->ncols, g->cols, g->col_len);
    d->variable_messages = msglists_i8_new(g->nrows, g->rows, g->row_len);
  }
  return d;
}

static void decoder_i8_free(decoder_i8 *d) {
  if (!d) return;
  free(d->x);
  free(d->out);
  if (d->layered) {
    free(d->llrs);
    for (size_t r = 0; r < d->g->nrows; r++) free(d->rcv[r]);
    free(d->rcv);
  } else {
    free(d->input_llrs);
    free(d->output_llrs);
    for (size_t i = 0; i < d->g->ncols; i++) free(d->check_messages[i].m);
    free(d->check_messages);
    for (size_t i = 0; i < d->g->nrows; i++) free(d->variable_messages[i].m);
    free(d->variable_messages);
  }
  free(d);
}

static int decode_i8(decoder_i8 *d, const double *llrs, uint32_t max_iterations, uint8_t *bits,
                     double *posterior, uint32_t *iterations) {
  const oracle_graph *g = d->g;
  const size_t n = g->ncols;
  if (check_llrs_f64in(g, llrs)) {
    for (size_t v = 0; v < n; v++) {
      bits[v] = (uint8_t)(llrs[v] <= 0.0);
      /* the soft output of an i8 decoder is in its 8-bit units: the quantised input here */
      if (posterior) posterior[v] = (double)i8_quantize(llrs[v]);
    }
    *iterations = 0;
    return 1;
  }
  int ok = 0;
  uint32_t it_used = max_iterations;
  if (!d->layered) {
    /* flooding.rs:88-125 with the i8 variable rule (arithmetic.rs:622-654) */
    for (size_t v = 0; v < n; v++) d->input_llrs[v] = (int8_t)i8_quantize(llrs[v]);
    for (size_t v = 0; v < n; v++)
      for (size_t j = 0; j < g->col_len[v]; j++) send_i8(d->variable_messages, v, g->cols[v][j], d->input_llrs[v]);
    for (uint32_t it = 1; it <= max_iterations; it++) {
      for (size_t c = 0; c < g->nrows; c++) {
        msglist_i8 *l = &d->variable_messages[c];
        for (size_t i = 0; i < l->len; i++) d->x[i] = l->m[i].value;
        if (i8_check_node(&d->o, d->x, l->len, d->out) != 0) return -1;
        for (size_t i = 0; i < l->len; i++) send_i8(d->check_messages, c, l->m[i].source, (int8_t)d->out[i]);
      }
      for (size_t v = 0; v < n; v++) {
        msglist_i8 *l = &d->check_messages[v];
        int in = d->input_llrs[v];
        if (d->o.deg1clip && l->len == 1) in = in <= -116 ? -116 : (in >= 116 ? 116 : in);
        int llr = in;
        for (size_t i = 0; i < l->len; i++) llr += l->m[i].value;
        if (d->o.jones) llr = i8_clip(llr);
        for (size_t i = 0; i < l->len; i++)
          send_i8(d->variable_messages, v, l->m[i].source, (int8_t)i8_clip(llr - l->m[i].value));
        d->output_llrs[v] = (int8_t)i8_clip(llr);
      }
      int good = 1;
      for (size_t r = 0; r < g->nrows && good; r++) {
        size_t ones = 0;
        for (size_t i = 0; i < g->row_len[r]; i++) ones += d->output_llrs[g->rows[r][i]] <= 0;
        if (ones % 2) good = 0;
      }
      if (good) {
        ok = 1;
        it_used = it;
        break;
      }
    }
    for (size_t v = 0; v < n; v++) {
      bits[v] = (uint8_t)(d->output_llrs[v] <= 0);
      if (posterior) posterior[v] = (double)d->output_llrs[v];
    }
  } else {
    /* horizontal_layered.rs:90-110 with update_check_messages_and_vars (:759-801, :1197-1257) */
    for (size_t v = 0; v < n; v++) d->llrs[v] = (int16_t)i8_quantize(llrs[v]);
    for (size_t r = 0; r < g->nrows; r++) memset(d->rcv[r], 0, g->row_len[r]);
    for (uint32_t it = 1; it <= max_iterations; it++) {
      for (size_t r = 0; r < g->nrows; r++) {
        const size_t dd = g->row_len[r];
        const size_t *cols = g->rows[r];
        int8_t *R = d->rcv[r];
        for (size_t i = 0; i < dd; i++) d->x[i] = i8_clip(d->llrs[cols[i]] - R[i]);
        if (i8_check_node(&d->o, d->x, dd, d->out) != 0) return -1;
        if (!d->o.aminstar) {
          for (size_t i = 0; i < dd; i++) {
            d->llrs[cols[i]] = (int16_t)(d->llrs[cols[i]] + d->out[i] - R[i]);
            R[i] = (int8_t)d->out[i];
          }
        } else {
          /* :1243-1256: the unclipped x = Qv - R decides the sign of the non-minimum edges and is
           * the base of the new Qv; i8_check_node used the clipped x for every sign, so redo
           * the non-minimum signs from the unclipped value */
          size_t argmin = 0;
          for (size_t i = 1; i < dd; i++)
            if (abs(d->x[i]) < abs(d->x[argmin])) argmin = i;
          unsigned sign = 0;
          for (size_t i = 0; i < dd; i++) sign ^= (unsigned)(d->x[i] < 0);
          int mag = 0;
          for (size_t i = 0; i < dd; i++)
            if (i != argmin) {
              mag = abs(d->out[i]);
              break;
            }
          for (size_t i = 0; i < dd; i++) {
            int xu = d->llrs[cols[i]] - R[i];
            int rcv = d->out[i];
            if (i != argmin) rcv = ((sign != 0) ^ (xu < 0)) ? -mag : mag;
            d->llrs[cols[i]] = (int16_t)(xu + rcv);
            R[i] = (int8_t)rcv;
          }
        }
      }
      int good = 1;
      for (size_t r = 0; r < g->nrows && good; r++) {
        size_t ones = 0;
        for (size_t i = 0; i < g->row_len[r]; i++) ones += d->llrs[g->rows[r][i]] <= 0;
        if (ones % 2) good = 0;
      }
      if (good) {
        ok = 1;
        it_used = it;
        break;
      }
    }
    for (size_t v = 0; v < n; v++) {
      bits[v] = (uint8_t)(d->llrs[v] <= 0);
      if (posterior) posterior[v] = (double)i8_clip(d->llrs[v]);
    }
  }
  *iterations = it_used;
  return ok;
}

/* ---- names (factory.rs:240-277 + the added Minsum family) ----------------------------- */

struct oracle_decoder {
  int rule, is_f64, layered, is_i8;
  i8_opts i8o;
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

/* factory.rs:246-263, 270-275: the i8 names; only four exist with the HL prefix */
static int parse_name_i8(const char *name, i8_opts *o, int *layered) {
  const char *p = name;
  memset(o, 0, sizeof(*o));
  *layered = 0;
  if (strncmp(p, "HL", 2) == 0) {
    *layered = 1;
    p += 2;
  }
  if (strncmp(p, "Minstarapproxi8", 15) == 0)
    p += 15;
  else if (strncmp(p, "Aminstari8", 10) == 0) {
    o->aminstar = 1;
    p += 10;
  } else
    return 0;
  if (strncmp(p, "Jones", 5) == 0) {
    o->jones = 1;
    p += 5;
  }
  if (strncmp(p, "PartialHardLimit", 16) == 0) {
    o->hardlimit = 1;
    p += 16;
  }
  if (strncmp(p, "Deg1Clip", 8) == 0) {
    o->deg1clip = 1;
    p += 8;
  }
  if (*p) return 0;
  if (*layered && (o->jones || o->deg1clip)) return 0;
  return 1;
}

oracle_decoder *oracle_decoder_new(const oracle_graph *g, const char *implementation) {
  int rule, is_f64, layered;
  i8_opts i8o;
  if (g && implementation && parse_name_i8(implementation, &i8o, &layered)) {
    oracle_decoder *d = (oracle_decoder *)calloc(1, sizeof(*d));
    d->is_i8 = 1;
    d->layered = layered;
    d->g = graph_clone(g);
    d->impl = decoder_i8_new(d->g, &i8o, layered);
    return d;
  }
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
  if (d->is_i8) {
    decoder_i8_free((decoder_i8 *)d->impl);
    oracle_graph_free(d->g);
    free(d);
    return;
  }
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
  if (d->is_i8) return decode_i8((decoder_i8 *)d->impl, llrs, max_iterations, bits, posterior, iterations);
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
  pthread_barrier_t *start; /* NULL, or the start line of a timed run */
} batch_job;

static void *batch_worker(void *arg) {
  batch_job *j = (batch_job *)arg;
  oracle_decoder *d = oracle_decoder_new(j->g, j->impl);
  /* timed runs: every worker has built its decoder before any starts decoding (the reference's
   * workers build theirs in make_worker, simulation/ber.rs:387, before the frames are timed) */
  if (j->start) pthread_barrier_wait(j->start);
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

static double now_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int oracle_decode_batch_f32(const oracle_graph *g, const char *implementation, const float *llrs,
                            size_t batch, uint32_t max_iterations, unsigned threads, uint8_t *bits,
                            int32_t *iterations, double *posterior) {
  return oracle_decode_batch_timed_f32(g, implementation, llrs, batch, max_iterations, threads, bits,
                                       iterations, posterior, NULL);
}

int oracle_decode_batch_timed_f32(const oracle_graph *g, const char *implementation,
                                  const float *llrs, size_t batch, uint32_t max_iterations,
                                  unsigned threads, uint8_t *bits, int32_t *iterations,
                                  double *posterior, double *decode_seconds) {
  if (!g || !implementation) return -1;
  if (threads == 0) threads = 1;
  if (threads > batch) threads = batch ? (unsigned)batch : 1;
  batch_job *jobs = (batch_job *)calloc(threads, sizeof(*jobs));
  pthread_t *tids = (pthread_t *)calloc(threads, sizeof(*tids));
  size_t per = batch / threads, extra = batch % threads, at = 0;
  for (unsigned t = 0; t < threads; t++) {
    size_t cnt = per + (t < extra ? 1 : 0);
    jobs[t] = (batch_job){g, implementation, llrs, g->ncols, at, at + cnt, max_iterations,
                          bits, iterations, posterior, 0, NULL};
    at += cnt;
  }
  int status = 0;
  if (decode_seconds) {
    /* all workers are threads here; the caller waits at the same start line and holds the clock */
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, threads + 1);
    for (unsigned t = 0; t < threads; t++) jobs[t].start = &start;
    for (unsigned t = 0; t < threads; t++) pthread_create(&tids[t], NULL, batch_worker, &jobs[t]);
    pthread_barrier_wait(&start);
    const double t0 = now_seconds();
    for (unsigned t = 0; t < threads; t++) {
      pthread_join(tids[t], NULL);
      if (jobs[t].status) status = jobs[t].status;
    }
    *decode_seconds = now_seconds() - t0;
    pthread_barrier_destroy(&start);
  } else {
    for (unsigned t = 1; t < threads; t++) pthread_create(&tids[t], NULL, batch_worker, &jobs[t]);
    batch_worker(&jobs[0]);
    status = jobs[0].status;
    for (unsigned t = 1; t < threads; t++) {
      pthread_join(tids[t], NULL);
      if (jobs[t].status) status = jobs[t].status;
    }
  }
  free(jobs);
  free(tids);
  return status;
}

/* simulation/puncturing.rs:83-101 */
/* decoder.rs:157-164: for every check, the parity of the hard decisions of its variables */
size_t oracle_syndrome(const oracle_graph *g, const uint8_t *bits, uint8_t *syndrome) {
  size_t weight = 0;
  for (size_t r = 0; r < g->nrows; r++) {
    unsigned parity = 0;
    for (size_t i = 0; i < g->row_len[r]; i++) parity ^= bits[g->rows[r][i]] & 1u;
    if (syndrome) syndrome[r] = (uint8_t)parity;
    weight += parity;
  }
  return weight;
}

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

/* ---- frame generator ----------------------------------------------------------------------------- */

void oracle_philox4x32_10(const uint32_t c_in[4], const uint32_t k_in[2], uint32_t out[4]) {
  uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3], k0 = k_in[0], k1 = k_in[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

static float unit_f(uint32_t w) { return (float)(w >> 8) * 0x1p-23f - 1.0f; }

void oracle_generate_llrs(const uint8_t *tx_bits, uint32_t pool, uint32_t n_tx, double rate, double ebn0_db,
                          uint64_t seed, uint64_t first_frame, uint32_t frames, float *llrs, uint32_t *pool_idx) {
  const double ebn0 = pow(10.0, 0.1 * ebn0_db);
  const double s = sqrt(0.5 / (rate * ebn0));
  const float sigma = (float)s, scale = (float)(-2.0 / (s * s));
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (uint32_t f = 0; f < frames; f++) {
    const uint64_t frame = first_frame + f;
    uint32_t c[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, (uint32_t)frame, (uint32_t)(frame >> 32)}, o[4];
    oracle_philox4x32_10(c, key, o);
    const uint32_t pi = o[0] % pool;
    if (pool_idx) pool_idx[f] = pi;
    const uint8_t *cw = tx_bits + (size_t)pi * n_tx;
    float *row = llrs + (size_t)f * n_tx;
    for (uint32_t pair = 0; pair < (n_tx + 1) / 2; pair++) {
      float z0 = 0.0f, z1 = 0.0f;
      int got = 0;
      for (uint32_t attempt = 0; !got; attempt++) {
        uint32_t cc[4] = {attempt, pair, (uint32_t)frame, (uint32_t)(frame >> 32)};
        oracle_philox4x32_10(cc, key, o);
        for (int h = 0; h < 2 && !got; h++) {
          const float v1 = unit_f(o[2 * h]), v2 = unit_f(o[2 * h + 1]);
          const float ss = v1 * v1 + v2 * v2;
          if (ss > 0.0f && ss < 1.0f) {
            const float fac = sqrtf(-2.0f * logf(ss) / ss);
            z0 = v1 * fac;
            z1 = v2 * fac;
            got = 1;
          }
        }
      }
      const uint32_t j = 2 * pair;
      {
        const float sym = cw[j] ? 1.0f : -1.0f;
        const float y = sym + sigma * z0;
        row[j] = scale * y;
      }
      if (j + 1 < n_tx) {
        const float sym = cw[j + 1] ? 1.0f : -1.0f;
        const float y = sym + sigma * z1;
        row[j + 1] = scale * y;
      }
    }
  }
}

/* ---- 8PSK and the bit interleaver ------------------------------------------------------------- */

/* interleaving.rs:40-58: view the codeword as [columns][len / columns], transpose, optionally
 * reverse each row of the transpose, read out row by row */
int oracle_interleave_u8(const uint8_t *in, size_t len, size_t columns, int backwards, uint8_t *out) {
  if (columns == 0 || len % columns != 0) return -1;
  const size_t rows = len / columns;
  for (size_t r = 0; r < rows; r++)
    for (size_t cp = 0; cp < columns; cp++) {
      const size_t c = backwards ? columns - 1 - cp : cp;
      out[r * columns + cp] = in[c * rows + r];
    }
  return 0;
}

/* interleaving.rs:65-86: view as [len / columns][columns], transpose, optionally reverse the row
 * order of the transpose, read out row by row */
int oracle_deinterleave_f64(const double *in, size_t len, size_t columns, int backwards, double *out) {
  if (columns == 0 || len % columns != 0) return -1;
  const size_t rows = len / columns;
  for (size_t c = 0; c < columns; c++)
    for (size_t r = 0; r < rows; r++) {
      const size_t cp = backwards ? columns - 1 - c : c;
      out[c * rows + r] = in[r * columns + cp];
    }
  return 0;
}

/* modulation.rs:166-179 */
static void psk8_point(unsigned b0, unsigned b1, unsigned b2, double *re, double *im) {
  const double a = sqrt(0.5);
  if (!b0 && !b1 && !b2) { *re = a; *im = a; }
  else if (b0 && !b1 && !b2) { *re = 0.0; *im = 1.0; }
  else if (b0 && b1 && !b2) { *re = -a; *im = a; }
  else if (!b0 && b1 && !b2) { *re = -1.0; *im = 0.0; }
  else if (!b0 && b1 && b2) { *re = -a; *im = -a; }
  else if (b0 && b1 && b2) { *re = 0.0; *im = -1.0; }
  else if (b0 && !b1 && b2) { *re = a; *im = -a; }
  else { *re = 1.0; *im = 0.0; }
}

void oracle_psk8_modulate(const uint8_t *bits, size_t symbols, double *re_im) {
  for (size_t s = 0; s < symbols; s++)
    psk8_point(bits[3 * s] & 1u, bits[3 * s + 1] & 1u, bits[3 * s + 2] & 1u, &re_im[2 * s], &re_im[2 * s + 1]);
}

/* modulation.rs:286-288 */
static double maxstar(double a, double b) { return fmax(a, b) + log1p(exp(-fabs(a - b))); }

/* modulation.rs:282-284 */
static double dot2(double are, double aim, double bre, double bim) { return are * bre + aim * bim; }

/* modulation.rs:225-267 */
static void psk8_demodulate_symbol(double re, double im, double scale, double out[3]) {
  const double a = sqrt(0.5);
  re *= scale;
  im *= scale;
  const double d000 = dot2(re, im, a, a), d100 = dot2(re, im, 0.0, 1.0), d110 = dot2(re, im, -a, a),
               d010 = dot2(re, im, -1.0, 0.0), d011 = dot2(re, im, -a, -a), d111 = dot2(re, im, 0.0, -1.0),
               d101 = dot2(re, im, a, -a), d001 = dot2(re, im, 1.0, 0.0);
  out[0] = maxstar(maxstar(maxstar(d000, d001), d010), d011) - maxstar(maxstar(maxstar(d100, d101), d110), d111);
  out[1] = maxstar(maxstar(maxstar(d000, d001), d100), d101) - maxstar(maxstar(maxstar(d010, d011), d110), d111);
  out[2] = maxstar(maxstar(maxstar(d000, d010), d100), d110) - maxstar(maxstar(maxstar(d001, d011), d101), d111);
}

void oracle_psk8_demodulate(const double *re_im, size_t symbols, double noise_sigma, double *llrs) {
  const double scale = 1.0 / (noise_sigma * noise_sigma); /* modulation.rs:220-222 */
  for (size_t s = 0; s < symbols; s++) psk8_demodulate_symbol(re_im[2 * s], re_im[2 * s + 1], scale, llrs + 3 * s);
}

void oracle_generate_llrs_psk8(const uint8_t *tx_bits, uint32_t pool, uint32_t n_tx, double rate, double ebn0_db,
                               int32_t interleaving, uint64_t seed, uint64_t first_frame, uint32_t frames,
                               float *llrs, uint32_t *pool_idx) {
  const double ebn0 = pow(10.0, 0.1 * ebn0_db);
  const double sigma = sqrt(0.5 / (rate * 3.0 * ebn0));
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const uint32_t symbols = n_tx / 3;
  const size_t columns = interleaving < 0 ? (size_t)(-(int64_t)interleaving) : (size_t)interleaving;
  uint8_t *inter = (uint8_t *)malloc(n_tx ? n_tx : 1);
  double *sym = (double *)malloc(sizeof(double) * 2 * (symbols ? symbols : 1));
  double *dem = (double *)malloc(sizeof(double) * (n_tx ? n_tx : 1));
  double *dei = (double *)malloc(sizeof(double) * (n_tx ? n_tx : 1));
  for (uint32_t f = 0; f < frames; f++) {
    const uint64_t frame = first_frame + f;
    uint32_t c[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, (uint32_t)frame, (uint32_t)(frame >> 32)}, o[4];
    oracle_philox4x32_10(c, key, o);
    const uint32_t pi = o[0] % pool;
    if (pool_idx) pool_idx[f] = pi;
    const uint8_t *cw = tx_bits + (size_t)pi * n_tx;
    /* ber.rs:446-452: interleave, modulate, add noise, demodulate, deinterleave */
    if (columns) oracle_interleave_u8(cw, n_tx, columns, interleaving < 0, inter);
    else memcpy(inter, cw, n_tx);
    oracle_psk8_modulate(inter, symbols, sym);
    for (uint32_t s = 0; s < symbols; s++) {
      float z0 = 0.0f, z1 = 0.0f;
      int got = 0;
      for (uint32_t attempt = 0; !got; attempt++) {
        uint32_t cc[4] = {attempt, s, (uint32_t)frame, (uint32_t)(frame >> 32)};
        oracle_philox4x32_10(cc, key, o);
        for (int h = 0; h < 2 && !got; h++) {
          const float v1 = unit_f(o[2 * h]), v2 = unit_f(o[2 * h + 1]);
          const float ss = v1 * v1 + v2 * v2;
          if (ss > 0.0f && ss < 1.0f) {
            const float fac = sqrtf(-2.0f * logf(ss) / ss);
            z0 = v1 * fac;
            z1 = v2 * fac;
            got = 1;
          }
        }
      }
      sym[2 * s] = sym[2 * s] + sigma * (double)z0;
      sym[2 * s + 1] = sym[2 * s + 1] + sigma * (double)z1;
    }
    oracle_psk8_demodulate(sym, symbols, sigma, dem);
    if (columns) oracle_deinterleave_f64(dem, n_tx, columns, interleaving < 0, dei);
    else memcpy(dei, dem, sizeof(double) * n_tx);
    float *row = llrs + (size_t)f * n_tx;
    for (uint32_t j = 0; j < n_tx; j++) row[j] = (float)dei[j];
  }
  free(inter);
  free(sym);
  free(dem);
  free(dei);
}
