// CPU check of the layered schedule's host tables (csrc/slice_tasks.h): dependency levels, the row records of the
// register-resident level kernels and the task records of the slice-persistent kernel, built for the alist files given on the command line under ASan/UBSan.  Invariants:
//  * rows of one level share no variable, and a row's level is one more than the highest level among the earlier rows
//    it shares a variable with (horizontal_layered.rs:105-110: level order == row order, as far as any result can tell);
//  * every non-empty row is in exactly one task of its level, every edge in exactly one lane slot -- except the middle
//    edge of a shared row with an odd number of edges, which both lanes take;
//  * a record's unused slots carry the out-of-range padding index;
//  * a level's row records list its rows in level order: first edge, degree, variables, padded with the last variable.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <string>

#include "../ldpc_toolbox_amd/csrc/slice_tasks.h"
#include "../ldpc_toolbox_amd/csrc/sparse.h"

using namespace ldpc;

#define REQUIRE(c)                                                        \
  do {                                                                    \
    if (!(c)) {                                                           \
      std::fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #c, path); \
      return 1;                                                           \
    }                                                                     \
  } while (0)

static int check(const char *path) {
  std::string err, text;
  if (FILE *f = std::fopen(path, "rb")) {
    char buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
    std::fclose(f);
  }
  SparseMatrix h;
  if (!SparseMatrix::from_alist(text, &h, &err)) {
    std::fprintf(stderr, "%s: %s\n", path, err.c_str());
    return 1;
  }
  const SparseMatrix::Csr g = h.csr();
  const LevelTables lv = build_levels(g.row_ptr, g.edge_col, g.n_rows, g.n_cols);
  const size_t n_levels = lv.maxdeg.size();
  REQUIRE(lv.level_ptr.size() == n_levels + 1 && lv.level_ptr.back() == g.n_rows);
  std::vector<uint32_t> level_of(g.n_rows, 0);
  for (size_t l = 0; l < n_levels; l++) {
    std::set<uint32_t> vars;
    for (uint32_t k = lv.level_ptr[l]; k < lv.level_ptr[l + 1]; k++) {
      const uint32_t r = lv.rows[k];
      level_of[r] = uint32_t(l);
      REQUIRE(k == lv.level_ptr[l] || lv.rows[k - 1] < r);  // row order inside a level
      for (uint32_t e = g.row_ptr[r]; e < g.row_ptr[r + 1]; e++) REQUIRE(vars.insert(g.edge_col[e]).second);
    }
  }
  std::vector<int64_t> last(g.n_cols, -1);
  for (uint32_t r = 0; r < g.n_rows; r++) {
    int64_t want = 0;
    for (uint32_t e = g.row_ptr[r]; e < g.row_ptr[r + 1]; e++) want = std::max(want, last[g.edge_col[e]] + 1);
    REQUIRE(int64_t(level_of[r]) == want);
    for (uint32_t e = g.row_ptr[r]; e < g.row_ptr[r + 1]; e++) last[g.edge_col[e]] = level_of[r];
  }
  // row records of the register-resident level kernels: every row of a level with records, in level order, with its
  // first edge, degree and variables, padded with the last variable to the level's record size
  {
    const LevelRecs lr = build_level_recs(lv, g.row_ptr, g.edge_col);
    REQUIRE(lr.rec_ptr.size() == n_levels);
    size_t expect = 0;
    for (size_t l = 0; l < n_levels; l++) {
      if (lv.maxdeg[l] > kLevelRecLong) {
        REQUIRE(lr.rec_ptr[l] == kNoLevelRecs);
        continue;
      }
      const uint32_t stride = lv.maxdeg[l] <= kLevelRecShort ? 16u : 32u;
      REQUIRE(lr.rec_ptr[l] == expect && lr.rec_ptr[l] % 16 == 0);
      for (uint32_t k = lv.level_ptr[l]; k < lv.level_ptr[l + 1]; k++) {
        const uint32_t r = lv.rows[k], e0 = g.row_ptr[r], d = g.row_ptr[r + 1] - e0;
        REQUIRE(expect + stride <= lr.words.size());
        const uint32_t *rec = &lr.words[expect];
        REQUIRE(rec[0] == e0 && rec[1] == d && d + 2 <= stride);
        for (uint32_t i = 0; i + 2 < stride; i++) REQUIRE(rec[2 + i] == (d ? g.edge_col[e0 + std::min(i, d - 1)] : 0u));
        expect += stride;
      }
    }
    REQUIRE(lr.words.size() == std::max<size_t>(expect, 16));
  }
  for (uint32_t rpt : {2u, 1u}) {
    for (bool can_split : {true, false}) {
      const SliceTasks st = build_slice_tasks(lv, g.row_ptr, g.edge_col, rpt, can_split);
      const uint32_t tw = 4 + rpt * kSliceWords;
      REQUIRE(st.task_ptr.size() == n_levels + 1);
      REQUIRE(st.tasks.size() == size_t(st.task_ptr.back() + 1) * tw);
      bool fits = true;
      std::vector<uint32_t> edge_seen(g.n_edges, 0);
      std::map<uint32_t, uint32_t> row_of_e0;
      for (uint32_t r = 0; r < g.n_rows; r++)
        if (g.row_ptr[r + 1] > g.row_ptr[r]) row_of_e0[g.row_ptr[r]] = r;
      for (size_t l = 0; l < n_levels; l++) {
        uint32_t prev_deg = 0xFFFFFFFFu;
        for (uint32_t t = st.task_ptr[l]; t < st.task_ptr[l + 1]; t++) {
          const uint32_t *rec = &st.tasks[size_t(t) * tw];
          const uint32_t d = rec[2] & 0xFFFFu;
          REQUIRE(d >= 1 && d <= prev_deg);  // falling degree inside a level
          prev_deg = d;
          if (d > kSliceEdges && !(rec[2] & kSliceSplit)) {
            fits = false;
            continue;
          }
          if (rec[2] & kSliceSplit) {
            REQUIRE(rpt == 2 && can_split && d > kSliceEdges && d <= 2 * kSliceEdges);
            const uint32_t half = (d + 1) / 2, e0 = rec[0];
            REQUIRE(row_of_e0.count(e0) && level_of[row_of_e0[e0]] == l && g.row_ptr[row_of_e0[e0] + 1] - e0 == d);
            REQUIRE(rec[1] == e0 + d - half);
            for (uint32_t k = 0; k < 2; k++)
              for (uint32_t i = 0; i < kSliceWords; i++) {
                const uint32_t v = rec[4 + k * kSliceWords + i];
                if (i < half) {
                  REQUIRE(v == g.edge_col[rec[k] + i]);
                  edge_seen[rec[k] + i]++;
                } else {
                  REQUIRE(v == kSlicePadIndex);
                }
              }
            if (d & 1) edge_seen[e0 + half - 1]--;  // the middle edge, taken by both lanes
            continue;
          }
          for (uint32_t k = 0; k < rpt; k++) {
            const uint32_t e0 = rec[k];
            for (uint32_t i = 0; i < kSliceWords; i++) {
              const uint32_t v = rec[4 + k * kSliceWords + i];
              if (e0 != kSliceNoRow && i < d) {
                REQUIRE(v == g.edge_col[e0 + i]);
                edge_seen[e0 + i]++;
              } else {
                REQUIRE(v == kSlicePadIndex);
              }
            }
            if (e0 == kSliceNoRow) continue;
            REQUIRE(row_of_e0.count(e0) && level_of[row_of_e0[e0]] == l && g.row_ptr[row_of_e0[e0] + 1] - e0 == d);
          }
          if (rpt == 1) REQUIRE(rec[0] != kSliceNoRow);
        }
      }
      REQUIRE(fits == st.fits);
      if (st.fits)
        for (uint32_t e = 0; e < g.n_edges; e++) REQUIRE(edge_seen[e] == 1);
      // the record behind the last task: all padding, no rows
      const uint32_t *tail = &st.tasks[size_t(st.task_ptr.back()) * tw];
      REQUIRE(tail[0] == kSliceNoRow && tail[1] == kSliceNoRow && tail[2] == 0);
      for (uint32_t i = 4; i < tw; i++) REQUIRE(tail[i] == kSlicePadIndex);
    }
  }
  std::printf("%s: %u rows, %zu levels: ok\n", path, g.n_rows, n_levels);
  return 0;
}

int main(int argc, char **argv) {
  for (int i = 1; i < argc; i++)
    if (check(argv[i])) return 1;
  std::printf("slice tasks driver: ok\n");
  return 0;
}
