#include "codes.h"

#include <cstdint>
#include <cstdlib>

namespace ldpc {
namespace codes {

namespace {
#include "code_tables.inc"

const Dvbs2Entry *find_dvbs2(const std::string &name) {
  for (const auto &e : kDvbs2Codes)
    if (name == e.name) return &e;
  return nullptr;
}

// set index iLS of a lifting size (TS 38.212 Table 5.3.2-1): Zc = a * 2^j with
// a in {2,3,5,7,9,11,13,15}.
int nr5g_set_index(unsigned zc) {
  static const unsigned a[8] = {2, 3, 5, 7, 9, 11, 13, 15};
  static const unsigned jmax[8] = {7, 7, 6, 5, 5, 5, 4, 4};
  for (int s = 0; s < 8; s++)
    for (unsigned j = 0; j <= jmax[s]; j++)
      if (zc == (a[s] << j)) return s;
  return -1;
}

}  // namespace

std::vector<std::string> dvbs2_names() {
  std::vector<std::string> v;
  for (const auto &e : kDvbs2Codes) v.emplace_back(e.name);
  return v;
}

bool dvbs2(const std::string &name, SparseMatrix *h) {
  const Dvbs2Entry *e = find_dvbs2(name);
  if (!e) return false;
  const size_t n = e->n, m = e->m, k = n - m, q = e->q;
  SparseMatrix mat(m, n);
  // systematic part: column j of group t = j / 360 hits rows (a + (j % 360) q) mod m
  const uint32_t *p = kDvbs2Addr + e->offset;
  for (size_t t = 0; t * 360 < k && t < e->groups; t++) {
    const uint32_t len = *p++;
    for (size_t w = 0; w < 360 && t * 360 + w < k; w++) {
      const size_t j = t * 360 + w;
      for (uint32_t i = 0; i < len; i++) mat.insert((p[i] + w * q) % m, j);
    }
    p += len;
  }
  // parity part: the staircase
  mat.insert(0, k);
  for (size_t j = 1; j < m; j++) {
    mat.insert(j, j + k);
    mat.insert(j, j + k - 1);
  }
  *h = std::move(mat);
  return true;
}

bool nr5g(int base_graph, unsigned zc, SparseMatrix *h) {
  const int ils = nr5g_set_index(zc);
  if (ils < 0 || (base_graph != 1 && base_graph != 2)) return false;
  const uint16_t *tab = base_graph == 1 ? kNr5gBg1 : kNr5gBg2;
  const uint32_t *off = base_graph == 1 ? kNr5gBg1RowOffset : kNr5gBg2RowOffset;
  const size_t base_rows = base_graph == 1 ? 46 : 42;
  const size_t base_cols = base_graph == 1 ? 68 : 52;
  SparseMatrix mat(base_rows * zc, base_cols * zc);
  for (size_t j = 0; j < base_rows; j++) {
    const uint16_t *p = tab + off[j];
    const unsigned count = *p++;
    for (unsigned t = 0; t < count; t++, p += 9) {
      const size_t col = p[0];
      const size_t v = p[1 + ils];
      for (size_t r = 0; r < zc; r++) mat.insert(zc * j + r, zc * col + ((r + v) % zc));
    }
  }
  *h = std::move(mat);
  return true;
}

bool ar4ja(const std::string &rate, unsigned k, SparseMatrix *h) {
  int rate_idx;  // 0: 1/2, 1: 2/3, 2: 4/5
  if (rate == "1/2")
    rate_idx = 0;
  else if (rate == "2/3")
    rate_idx = 1;
  else if (rate == "4/5")
    rate_idx = 2;
  else
    return false;
  int k_idx;
  if (k == 1024)
    k_idx = 0;
  else if (k == 4096)
    k_idx = 1;
  else if (k == 16384)
    k_idx = 2;
  else
    return false;
  // CCSDS 131.0-B-5 Table 7-2: log2(M)
  static const unsigned log2m_tab[3][3] = {{9, 8, 7}, {11, 10, 9}, {13, 12, 11}};
  const unsigned lg = log2m_tab[k_idx][rate_idx];
  const size_t m = size_t{1} << lg;
  // pi_k(i) = M/4 ((theta_k + floor(4i/M)) mod 4) + (phi_k(floor(4i/M), M) + i) mod M/4
  auto pi = [&](unsigned kk, size_t i) -> size_t {
    const size_t j = 4 * i / m;
    const size_t a = (kAr4jaTheta[kk - 1] + j) & 3;
    const size_t quarter = m >> 2;
    const size_t phi = kAr4jaPhi[(j * 26 + (kk - 1)) * 7 + (lg - 7)];
    return a * quarter + ((phi + i) & (quarter - 1));
  };
  const size_t extra_blocks = rate_idx == 0 ? 0 : (rate_idx == 1 ? 2 : 6);
  const size_t x = m * extra_blocks;
  SparseMatrix mat(3 * m, x + 5 * m);
  // H_1/2 part (blocks addressed as (block row, block col) of the 3x5 protograph)
  for (size_t i = 0; i < m; i++) {
    mat.insert(i, x + 2 * m + i);
    mat.insert(i, x + 4 * m + i);
    mat.toggle(i, x + 4 * m + pi(1, i));
    mat.insert(m + i, x + i);
    mat.insert(m + i, x + m + i);
    mat.insert(m + i, x + 3 * m + i);
    mat.insert(m + i, x + 4 * m + pi(2, i));
    mat.toggle(m + i, x + 4 * m + pi(3, i));
    mat.toggle(m + i, x + 4 * m + pi(4, i));
    mat.insert(2 * m + i, x + i);
    mat.insert(2 * m + i, x + m + pi(5, i));
    mat.toggle(2 * m + i, x + m + pi(6, i));
    mat.insert(2 * m + i, x + 3 * m + pi(7, i));
    mat.toggle(2 * m + i, x + 3 * m + pi(8, i));
    mat.insert(2 * m + i, x + 4 * m + i);
  }
  if (rate_idx != 0) {
    // the two extra column blocks of H_2/3
    const size_t y = rate_idx == 1 ? 0 : 4 * m;
    for (size_t i = 0; i < m; i++) {
      mat.insert(m + i, y + pi(9, i));
      mat.toggle(m + i, y + pi(10, i));
      mat.toggle(m + i, y + pi(11, i));
      mat.insert(m + i, y + m + i);
      mat.insert(2 * m + i, y + i);
      mat.insert(2 * m + i, y + m + pi(12, i));
      mat.toggle(2 * m + i, y + m + pi(13, i));
      mat.toggle(2 * m + i, y + m + pi(14, i));
    }
  }
  if (rate_idx == 2) {
    // the four extra column blocks of H_4/5
    for (size_t i = 0; i < m; i++) {
      mat.insert(m + i, pi(21, i));
      mat.toggle(m + i, pi(22, i));
      mat.toggle(m + i, pi(23, i));
      mat.insert(m + i, m + i);
      mat.insert(m + i, 2 * m + pi(15, i));
      mat.toggle(m + i, 2 * m + pi(16, i));
      mat.toggle(m + i, 2 * m + pi(17, i));
      mat.insert(m + i, 3 * m + i);
      mat.insert(2 * m + i, i);
      mat.insert(2 * m + i, m + pi(24, i));
      mat.toggle(2 * m + i, m + pi(25, i));
      mat.toggle(2 * m + i, m + pi(26, i));
      mat.insert(2 * m + i, 2 * m + i);
      mat.insert(2 * m + i, 3 * m + pi(18, i));
      mat.toggle(2 * m + i, 3 * m + pi(19, i));
      mat.toggle(2 * m + i, 3 * m + pi(20, i));
    }
  }
  *h = std::move(mat);
  return true;
}

SparseMatrix c2() {
  const size_t n = 511;
  SparseMatrix mat(2 * n, 16 * n);
  for (size_t rb = 0; rb < 2; rb++)
    for (size_t cb = 0; cb < 16; cb++)
      for (size_t w = 0; w < 2; w++) {
        const size_t circ = kC2Circulants[(rb * 16 + cb) * 2 + w];
        for (size_t j = 0; j < n; j++) mat.insert(rb * n + j, cb * n + (j + circ) % n);
      }
  return mat;
}

bool by_spec(const std::string &spec, SparseMatrix *h) {
  std::vector<std::string> parts;
  size_t start = 0;
  while (true) {
    size_t c = spec.find(':', start);
    if (c == std::string::npos) {
      parts.push_back(spec.substr(start));
      break;
    }
    parts.push_back(spec.substr(start, c - start));
    start = c + 1;
  }
  if (parts[0] == "dvbs2" && parts.size() == 2) return dvbs2(parts[1], h);
  if (parts[0] == "nr5g" && parts.size() == 3)
    return nr5g(std::atoi(parts[1].c_str()), static_cast<unsigned>(std::atoi(parts[2].c_str())), h);
  if (parts[0] == "ar4ja" && parts.size() == 3)
    return ar4ja(parts[1], static_cast<unsigned>(std::atoi(parts[2].c_str())), h);
  if (parts[0] == "c2" && parts.size() == 1) {
    *h = c2();
    return true;
  }
  return false;
}

}  // namespace codes
}  // namespace ldpc
