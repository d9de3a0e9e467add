// Tanner-graph owner for the decode path.
//
// Host-side counterpart of the reference's sparse::SparseMatrix
// (/root/reference/src/sparse.rs:23-26, 114-119, 240-248, 250-389).  What the
// decode path needs from it, and what is therefore kept *bit for bit*:
//   * adjacency lists whose ORDER is semantic: the f32 variable-node sum runs in
//     cols[v] order and the check-node folds run in rows[c] order
//     (SURVEY.md Appendix C);
//   * insert() de-duplicates and appends (sparse.rs:114-119);
//   * from_alist() reads only the column section, in file order, ignoring zero
//     padding (sparse.rs:352-389); alist() writes sorted, 1-based, padded lists
//     (sparse.rs:250-299).
// New here (not in the reference): flat CSR/CSC exports in the layout the HIP
// kernels consume (edge ids are row-major: e = row_ptr[r] + slot).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace ldpc {

class SparseMatrix {
 public:
  SparseMatrix() = default;
  SparseMatrix(size_t nrows, size_t ncols) : rows_(nrows), cols_(ncols) {}

  size_t num_rows() const { return rows_.size(); }
  size_t num_cols() const { return cols_.size(); }
  size_t row_weight(size_t r) const { return rows_[r].size(); }
  size_t col_weight(size_t c) const { return cols_[c].size(); }
  size_t num_edges() const;

  bool contains(size_t r, size_t c) const;
  void insert(size_t r, size_t c);
  void remove(size_t r, size_t c);
  void toggle(size_t r, size_t c);

  // slot-ordered neighbour lists (the reference's iter_row / iter_col)
  const std::vector<uint32_t> &row(size_t r) const { return rows_[r]; }
  const std::vector<uint32_t> &col(size_t c) const { return cols_[c]; }

  // alist text; `padding` = MacKay zero padding (the reference's alist()).
  std::string alist(bool padding = true) const;
  // Parses alist text.  On malformed input returns false and sets *err.
  static bool from_alist(const std::string &text, SparseMatrix *out, std::string *err);

  // Flat graph tables for the device.
  struct Csr {
    uint32_t n_rows = 0, n_cols = 0, n_edges = 0;
    uint32_t max_row_weight = 0, max_col_weight = 0;
    std::vector<uint32_t> row_ptr;   // [n_rows+1]   edge id range of check r
    std::vector<uint32_t> edge_col;  // [n_edges]    variable of edge e (rows[r] order)
    std::vector<uint32_t> col_ptr;   // [n_cols+1]   slot range of variable v in col_edge
    std::vector<uint32_t> col_edge;  // [n_edges]    row-major edge id of v's j-th slot (cols[v] order)
  };
  Csr csr() const;

 private:
  std::vector<std::vector<uint32_t>> rows_;
  std::vector<std::vector<uint32_t>> cols_;
};

}  // namespace ldpc
