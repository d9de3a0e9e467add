// Standard code definitions -> parity check matrices.
//
// Same constructions (and therefore the same insertion/edge order) as the
// reference's code modules:
//   DVB-S2   /root/reference/src/codes/dvbs2.rs:79-98   (ETSI EN 302 307-1 5.3.2.1)
//   5G NR    /root/reference/src/codes/nr5g.rs:40-53    (3GPP TS 38.212 5.3.2)
//   AR4JA    /root/reference/src/codes/ccsds.rs:51-187  (CCSDS 131.0-B-5 7.4)
//   C2       /root/reference/src/codes/ccsds.rs:353-368 (CCSDS 131.0-B-5 7.3)
// The numeric tables live in code_tables.inc (generated, see tools/extract_tables.py).
#pragma once
#include <string>
#include <vector>

#include "sparse.h"

namespace ldpc {
namespace codes {

// DVB-S2: `name` is the reference's Code variant name ("R1_2", "R3_4short", ...)
// or the CLI spelling ("1/2", "3/4short" is not accepted: use variant names).
bool dvbs2(const std::string &name, SparseMatrix *h);
std::vector<std::string> dvbs2_names();

// 5G NR: base_graph in {1,2}; lifting size one of the 51 values of TS 38.212 Table 5.3.2-1.
bool nr5g(int base_graph, unsigned lifting_size, SparseMatrix *h);

// CCSDS AR4JA: rate in {"1/2","2/3","4/5"}, k in {1024, 4096, 16384}.
bool ar4ja(const std::string &rate, unsigned k, SparseMatrix *h);

// CCSDS C2 basic (8176, 7156) code.
SparseMatrix c2();

// Generic front door used by the C ABI helper and the Python layer:
//   "dvbs2:R1_2", "nr5g:1:384", "ar4ja:1/2:1024", "c2".
bool by_spec(const std::string &spec, SparseMatrix *h);

}  // namespace codes
}  // namespace ldpc
