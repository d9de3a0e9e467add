// Decoder implementation names <-> (check-node rule, precision, schedule).
//
// Mirrors the name table of the reference's decoder::factory::DecoderImplementation
// (/root/reference/src/decoder/factory.rs:240-277; FromStr error text :221) and adds the
// Minsum family this build defines.
#pragma once
#include <string>
#include <vector>

namespace ldpc {

enum class Rule { Phi, Tanh, Minstarapprox, Aminstar, Minsum };
enum class Schedule { Flooding, Layered };

struct Implementation {
  Rule rule = Rule::Minsum;
  bool f64 = false;
  Schedule schedule = Schedule::Flooding;
  std::string name;
};

// Returns false and sets *err ("invalid decoder implementation" for unknown names, a
// specific message for the reference's i8 names, which have no HIP kernels yet).
bool parse_implementation(const std::string &name, Implementation *out, std::string *err);

// Every name the HIP path accepts.
std::vector<std::string> implementation_names();

}  // namespace ldpc
