#include "sparse.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>

namespace ldpc {

size_t SparseMatrix::num_edges() const {
  size_t e = 0;
  for (const auto &r : rows_) e += r.size();
  return e;
}

bool SparseMatrix::contains(size_t r, size_t c) const {
  // columns are normally the shorter lists
  const auto &v = cols_[c];
  return std::find(v.begin(), v.end(), static_cast<uint32_t>(r)) != v.end();
}

void SparseMatrix::insert(size_t r, size_t c) {
  if (contains(r, c)) return;
  rows_[r].push_back(static_cast<uint32_t>(c));
  cols_[c].push_back(static_cast<uint32_t>(r));
}

void SparseMatrix::remove(size_t r, size_t c) {
  auto &rv = rows_[r];
  rv.erase(std::remove(rv.begin(), rv.end(), static_cast<uint32_t>(c)), rv.end());
  auto &cv = cols_[c];
  cv.erase(std::remove(cv.begin(), cv.end(), static_cast<uint32_t>(r)), cv.end());
}

void SparseMatrix::toggle(size_t r, size_t c) {
  if (contains(r, c))
    remove(r, c);
  else
    insert(r, c);
}

namespace {

void append_uint(std::string &s, size_t v) {
  char buf[24];
  int n = 0;
  do {
    buf[n++] = static_cast<char>('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) s.push_back(buf[--n]);
}

// One whitespace-separated unsigned token; returns false at end of line.
// *bad is set when the token is not a plain decimal number.
bool next_token(const std::string &line, size_t *pos, uint64_t *val, bool *bad) {
  size_t i = *pos;
  while (i < line.size() && std::isspace(static_cast<unsigned char>(line[i]))) i++;
  if (i >= line.size()) {
    *pos = i;
    return false;
  }
  size_t j = i;
  uint64_t v = 0;
  bool ok = true;
  if (line[j] == '+') j++;
  size_t digits = 0;
  while (j < line.size() && !std::isspace(static_cast<unsigned char>(line[j]))) {
    if (line[j] < '0' || line[j] > '9')
      ok = false;
    else {
      v = v * 10 + static_cast<uint64_t>(line[j] - '0');
      digits++;
    }
    j++;
  }
  if (!digits) ok = false;
  *bad = !ok;
  *val = v;
  *pos = j;
  return true;
}

}  // namespace

std::string SparseMatrix::alist(bool padding) const {
  std::string s;
  s.reserve(16 * (num_edges() + num_rows() + num_cols()) + 64);
  append_uint(s, num_cols());
  s.push_back(' ');
  append_uint(s, num_rows());
  s.push_back('\n');
  const std::vector<std::vector<uint32_t>> *dirs[2] = {&cols_, &rows_};
  size_t maxlen[2] = {0, 0};
  for (int d = 0; d < 2; d++)
    for (const auto &l : *dirs[d]) maxlen[d] = std::max(maxlen[d], l.size());
  append_uint(s, maxlen[0]);
  s.push_back(' ');
  append_uint(s, maxlen[1]);
  s.push_back('\n');
  for (int d = 0; d < 2; d++) {
    bool first = true;
    for (const auto &l : *dirs[d]) {
      if (!first) s.push_back(' ');
      first = false;
      append_uint(s, l.size());
    }
    s.push_back('\n');
  }
  std::vector<uint32_t> v;
  for (int d = 0; d < 2; d++) {
    for (const auto &l : *dirs[d]) {
      v = l;
      std::sort(v.begin(), v.end());
      for (size_t i = 0; i < v.size(); i++) {
        if (i) s.push_back(' ');
        append_uint(s, static_cast<size_t>(v[i]) + 1);
      }
      if (padding) {
        if (v.empty()) s.push_back('0');
        size_t pad = maxlen[d] - std::max<size_t>(v.size(), 1);
        for (size_t i = 0; i < pad; i++) {
          s.push_back(' ');
          s.push_back('0');
        }
      }
      s.push_back('\n');
    }
  }
  return s;
}

bool SparseMatrix::from_alist(const std::string &text, SparseMatrix *out, std::string *err) {
  auto fail = [&](const char *m) {
    if (err) *err = m;
    return false;
  };
  size_t cursor = 0;
  bool exhausted = false;
  auto next_line = [&](std::string *line) {
    if (exhausted) return false;
    size_t nl = text.find('\n', cursor);
    if (nl == std::string::npos) {
      *line = text.substr(cursor);
      exhausted = true;
    } else {
      *line = text.substr(cursor, nl - cursor);
      cursor = nl + 1;
    }
    return true;
  };
  std::string line;
  if (!next_line(&line)) return fail("alist first line not found");
  size_t pos = 0;
  uint64_t ncols = 0, nrows = 0;
  bool bad = false;
  if (!next_token(line, &pos, &ncols, &bad))
    return fail("alist first line does not contain enough elements");
  if (bad) return fail("ncols is not a number");
  if (!next_token(line, &pos, &nrows, &bad))
    return fail("alist first line does not contain enough elements");
  if (bad) return fail("nrows is not a number");
  if (ncols > 0xFFFFFFFFull || nrows > 0xFFFFFFFFull) return fail("alist dimensions too large");
  SparseMatrix h(static_cast<size_t>(nrows), static_cast<size_t>(ncols));
  // max weights, column weights, row weights: skipped unread
  next_line(&line);
  next_line(&line);
  next_line(&line);
  for (size_t c = 0; c < ncols; c++) {
    if (!next_line(&line)) return fail("alist does not contain expected number of lines");
    pos = 0;
    uint64_t r = 0;
    while (next_token(line, &pos, &r, &bad)) {
      if (bad) return fail("row value is not a number");
      if (r == 0) continue;  // padding
      if (r > nrows) return fail("row value out of range");
      h.insert(static_cast<size_t>(r - 1), c);
    }
  }
  // the row section is not read
  *out = std::move(h);
  return true;
}

SparseMatrix::Csr SparseMatrix::csr() const {
  Csr g;
  g.n_rows = static_cast<uint32_t>(num_rows());
  g.n_cols = static_cast<uint32_t>(num_cols());
  g.row_ptr.resize(g.n_rows + 1);
  uint32_t e = 0;
  for (uint32_t r = 0; r < g.n_rows; r++) {
    g.row_ptr[r] = e;
    e += static_cast<uint32_t>(rows_[r].size());
    g.max_row_weight = std::max<uint32_t>(g.max_row_weight, static_cast<uint32_t>(rows_[r].size()));
  }
  g.row_ptr[g.n_rows] = e;
  g.n_edges = e;
  g.edge_col.resize(e);
  for (uint32_t r = 0; r < g.n_rows; r++)
    std::copy(rows_[r].begin(), rows_[r].end(), g.edge_col.begin() + g.row_ptr[r]);
  g.col_ptr.resize(g.n_cols + 1);
  g.col_edge.resize(e);
  uint32_t s = 0;
  for (uint32_t c = 0; c < g.n_cols; c++) {
    g.col_ptr[c] = s;
    g.max_col_weight = std::max<uint32_t>(g.max_col_weight, static_cast<uint32_t>(cols_[c].size()));
    for (uint32_t r : cols_[c]) {
      const auto &rv = rows_[r];
      uint32_t slot = static_cast<uint32_t>(std::find(rv.begin(), rv.end(), c) - rv.begin());
      g.col_edge[s++] = g.row_ptr[r] + slot;
    }
  }
  g.col_ptr[g.n_cols] = s;
  return g;
}

}  // namespace ldpc
