#!/usr/bin/env python3
"""Writes a code's check-row adjacency (rows[c] order of the alist, as the decoder uses it) as a flat binary for the
standalone kernel benches: u32 n_rows, n_cols, n_edges, row_ptr[n_rows + 1], edge_col[n_edges].
  python tools/mb/make_graph.py nr5g:1:384 tools/mb/graph.bin"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ldpc_toolbox_amd as lt

spec, out = sys.argv[1], sys.argv[2]
lines = lt.code_alist(spec).split("\n")
n_cols, n_rows = (int(x) for x in lines[0].split())
# alist: line 0 sizes, 1 max weights, 2 column weights, 3 row weights, then n_cols column lists, then n_rows row lists
row_lists = lines[4 + n_cols: 4 + n_cols + n_rows]
row_ptr, edge_col = [0], []
for ln in row_lists:
    vs = [int(x) - 1 for x in ln.split() if int(x) > 0]
    edge_col.extend(vs)
    row_ptr.append(len(edge_col))
hdr = np.array([n_rows, n_cols, len(edge_col)], dtype=np.uint32)
with open(out, "wb") as f:
    f.write(hdr.tobytes())
    f.write(np.array(row_ptr, dtype=np.uint32).tobytes())
    f.write(np.array(edge_col, dtype=np.uint32).tobytes())
print(spec, n_rows, n_cols, len(edge_col))
