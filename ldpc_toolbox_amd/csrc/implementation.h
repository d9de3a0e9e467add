// Decoder implementation names <-> (check-node rule, precision, schedule).
//
// Mirrors the name table of the reference's decoder::factory::DecoderImplementation
// (/root/reference/src/decoder/factory.rs:240-277; FromStr error text :221) and adds the
// Minsum family this build defines.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ldpc {

enum class Rule { Phi, Tanh, Minstarapprox, Aminstar, Minsum };
enum class Schedule { Flooding, Layered };

struct Implementation {
  Rule rule = Rule::Minsum;
  bool f64 = false;
  // 8-bit quantised arithmetics (rule is Minstarapprox or Aminstar) and their options
  // (arithmetic.rs:806-848): Jones clipping, partial hard limiting, degree-one clipping
  bool i8 = false, jones = false, hardlimit = false, deg1clip = false;
  Schedule schedule = Schedule::Flooding;
  // "@fast" (Tanhf32 / Phif32, both schedules; this build's addition, never the default): the rule's formulas with the
  // GPU's native exp2 / log2 / rcp instead of the glibc-identical functions -- NOT bit-identical to the reference
  bool fast = false;
  std::string name;
};

// Returns false and sets *err ("invalid decoder implementation" for unknown names,
// factory.rs:221).  All 36 names of the reference are accepted, plus the Minsum family.
bool parse_implementation(const std::string &name, Implementation *out, std::string *err);

// "1,1,1,0" -> {1,1,1,0}; "" -> empty (no puncturing).  Only "0"/"1" tokens are legal
// (src/cli/ber.rs:219-229); returns false otherwise.
bool parse_puncturing_pattern(const std::string &text, std::vector<uint8_t> *out);

// Every name the HIP path accepts with results identical to the reference decoder's.
std::vector<std::string> implementation_names();
// The opt-in approximate variants ("Tanhf32@fast", "HLTanhf32@fast", "Phif32@fast", "HLPhif32@fast").
std::vector<std::string> fast_implementation_names();

}  // namespace ldpc
