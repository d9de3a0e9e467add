// Host-side tables of the layered schedule: dependency levels of the check rows and the task lists of the
// slice-persistent kernel (kernels.hip.h, hl_slice_kernel).  Shared by DeviceDecoder::create and tools/mb/slice_bench.hip.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace ldpc {

struct LevelTables {
  std::vector<uint32_t> level_ptr;   // [n_levels + 1] into rows
  std::vector<uint32_t> rows;        // rows grouped by level, in row order inside a level
  std::vector<uint32_t> maxdeg;      // [n_levels] longest row of the level
};

// level(r) = 1 + max level of the earlier rows that share a variable with r: the rows of one level are
// variable-disjoint, so processing level after level equals the serial row order of horizontal_layered.rs:105-110
inline LevelTables build_levels(const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &edge_col, uint32_t n_rows,
                                uint32_t n_cols) {
  LevelTables t;
  std::vector<uint32_t> last(n_cols, 0), level(n_rows, 0);
  uint32_t n_levels = 0;
  for (uint32_t r = 0; r < n_rows; r++) {
    uint32_t lv = 0;
    for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) lv = std::max(lv, last[edge_col[e]]);
    lv += 1;
    level[r] = lv;
    n_levels = std::max(n_levels, lv);
    for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) last[edge_col[e]] = lv;
  }
  t.level_ptr.assign(n_levels + 1, 0);
  for (uint32_t r = 0; r < n_rows; r++) t.level_ptr[level[r]]++;
  for (uint32_t l = 1; l <= n_levels; l++) t.level_ptr[l] += t.level_ptr[l - 1];
  std::vector<uint32_t> cursor(t.level_ptr.begin(), t.level_ptr.end() - 1);
  t.rows.assign(n_rows, 0);
  for (uint32_t r = 0; r < n_rows; r++) t.rows[cursor[level[r] - 1]++] = r;
  t.maxdeg.assign(n_levels, 0);
  for (uint32_t r = 0; r < n_rows; r++)
    t.maxdeg[level[r] - 1] = std::max(t.maxdeg[level[r] - 1], row_ptr[r + 1] - row_ptr[r]);
  return t;
}

// Row records of the register-resident level kernels (kernels.hip.h, hl_level_reg_kernel): per row of a level whose
// longest row has at most 24 edges, in level order, [first edge, degree, variable of edge 0, 1, ...] padded with the
// last variable to 16 words (levels of at most 12 edges) or 32 words -- one scalar load per row instead of the chain
// level_rows -> row_ptr -> edge_col.  rec_ptr[l] = first word of level l's records (kNoLevelRecs: the level has none).
constexpr uint32_t kNoLevelRecs = 0xFFFFFFFFu;
constexpr uint32_t kLevelRecShort = 12, kLevelRecLong = 24;  // edges of a 16-word / 32-word record
struct LevelRecs {
  std::vector<uint32_t> words;
  std::vector<uint32_t> rec_ptr;  // [n_levels]
};
inline LevelRecs build_level_recs(const LevelTables &lv, const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &edge_col) {
  LevelRecs t;
  const size_t n_levels = lv.maxdeg.size();
  t.rec_ptr.assign(n_levels, kNoLevelRecs);
  for (size_t l = 0; l < n_levels; l++) {
    if (lv.maxdeg[l] > kLevelRecLong) continue;
    const uint32_t stride = lv.maxdeg[l] <= kLevelRecShort ? 16u : 32u;
    t.rec_ptr[l] = static_cast<uint32_t>(t.words.size());
    for (uint32_t idx = lv.level_ptr[l]; idx < lv.level_ptr[l + 1]; idx++) {
      const uint32_t r = lv.rows[idx], e0 = row_ptr[r], d = row_ptr[r + 1] - e0;
      const size_t base = t.words.size();
      t.words.resize(base + stride, 0);
      t.words[base] = e0;
      t.words[base + 1] = d;
      for (uint32_t i = 0; i + 2 < stride; i++) t.words[base + 2 + i] = d ? edge_col[e0 + std::min(i, d - 1)] : 0u;
    }
  }
  if (t.words.empty()) t.words.assign(16, 0);  // (an empty upload is an error)
  return t;
}

struct SliceTasks {
  std::vector<uint32_t> tasks;     // [n_tasks + 1][4 + rpt * kSliceWords]: see kernels.hip.h, hl_slice_kernel
  std::vector<uint32_t> task_ptr;  // [n_levels + 1]
  bool fits = true;                // false: some row fits no task (the kernel cannot take this graph)
};

constexpr uint32_t kSliceEdges = 10;  // = dev::kSliceD: edges per lane of the kernel's register sets
constexpr uint32_t kSliceWords = 12;  // = dev::kSliceW: index words per row in a record
constexpr uint32_t kSliceNoRow = 0xFFFFFFFFu, kSliceSplit = 0x80000000u;
constexpr uint32_t kSlicePadIndex = 0x003FFFFFu;  // = dev::kSlicePad: an index whose row offset is out of range

// Per level the rows by falling degree (long rows first: the short ones even out the waves at the barrier), rpt rows of
// equal degree to a task; empty rows carry no message and are left out.  A row of more than kSliceEdges edges is, with
// can_split (two rows per task and a rule whose row can be shared by two lanes), a task of its own whose two
// half-waves take half of the edges each.
inline SliceTasks build_slice_tasks(const LevelTables &lv, const std::vector<uint32_t> &row_ptr,
                                    const std::vector<uint32_t> &edge_col, uint32_t rpt, bool can_split) {
  SliceTasks s;
  s.task_ptr.assign(1, 0);
  const size_t n_levels = lv.level_ptr.size() - 1;
  const uint32_t tw = 4 + rpt * kSliceWords;
  auto deg = [&](uint32_t r) { return row_ptr[r + 1] - row_ptr[r]; };
  auto record = [&]() {
    s.tasks.resize(s.tasks.size() + tw, 0);
    const size_t at = s.tasks.size() - tw;
    s.tasks[at] = s.tasks[at + 1] = kSliceNoRow;
    for (uint32_t i = 4; i < tw; i++) s.tasks[at + i] = kSlicePadIndex;
    return at;
  };
  auto put_indices = [&](size_t at, uint32_t k, uint32_t e0, uint32_t count) {
    for (uint32_t i = 0; i < count && i < kSliceWords; i++) s.tasks[at + 4 + k * kSliceWords + i] = edge_col[e0 + i];
  };
  for (size_t l = 0; l < n_levels; l++) {
    std::vector<uint32_t> lr(lv.rows.begin() + lv.level_ptr[l], lv.rows.begin() + lv.level_ptr[l + 1]);
    std::stable_sort(lr.begin(), lr.end(), [&](uint32_t a, uint32_t b) { return deg(a) > deg(b); });
    size_t i = 0;
    while (i < lr.size()) {
      const uint32_t dr = deg(lr[i]);
      if (dr == 0) break;
      const size_t at = record();
      if (dr > kSliceEdges && !(can_split && rpt == 2 && dr <= 2 * kSliceEdges)) {
        s.fits = false;  // (the table is not used then; keep it well-formed)
        s.tasks[at + 2] = dr;
        i++;
        continue;
      }
      if (dr > kSliceEdges) {
        // (both lanes take ceil(dr / 2) edges, the second one the LAST ones: an odd row's middle edge is done twice)
        const uint32_t e0 = row_ptr[lr[i]], half = (dr + 1) / 2;
        s.tasks[at] = e0;
        s.tasks[at + 1] = e0 + (dr - half);
        s.tasks[at + 2] = dr | kSliceSplit;
        put_indices(at, 0, e0, half);
        put_indices(at, 1, e0 + (dr - half), half);
        i++;
        continue;
      }
      for (uint32_t j = 0; j < rpt; j++) {
        const bool same = i < lr.size() && deg(lr[i]) == dr;
        s.tasks[at + j] = same ? row_ptr[lr[i]] : kSliceNoRow;
        if (same) put_indices(at, j, row_ptr[lr[i]], dr);
        if (same) i++;
      }
      s.tasks[at + 2] = dr;
    }
    s.task_ptr.push_back(static_cast<uint32_t>(s.tasks.size() / tw));
  }
  record();  // an all-padding record behind the last task (tickets past the end fetch it)
  return s;
}

}  // namespace ldpc
