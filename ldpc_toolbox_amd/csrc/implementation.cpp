#include "implementation.h"

#include <cstring>

namespace ldpc {

namespace {
struct Stem {
  const char *text;
  Rule rule;
};
const Stem kStems[] = {{"Phi", Rule::Phi},
                       {"Tanh", Rule::Tanh},
                       {"Minstarapprox", Rule::Minstarapprox},
                       {"Aminstar", Rule::Aminstar},
                       {"Minsum", Rule::Minsum}};
}  // namespace

bool parse_implementation(const std::string &name, Implementation *out, std::string *err) {
  Implementation impl;
  impl.name = name;
  const char *p = name.c_str();
  if (std::strncmp(p, "HL", 2) == 0) {
    impl.schedule = Schedule::Layered;
    p += 2;
  }
  for (const Stem &s : kStems) {
    const size_t l = std::strlen(s.text);
    if (std::strncmp(p, s.text, l) != 0) continue;
    const char *suffix = p + l;
    if (std::strcmp(suffix, "f32") == 0 || std::strcmp(suffix, "f64") == 0) {
      impl.rule = s.rule;
      impl.f64 = suffix[1] == '6';
      *out = impl;
      return true;
    }
    if (std::strncmp(suffix, "i8", 2) == 0 && s.rule != Rule::Minsum && s.rule != Rule::Phi &&
        s.rule != Rule::Tanh) {
      if (err) *err = "decoder implementation '" + name + "' (8-bit quantised) has no HIP kernels yet";
      return false;
    }
  }
  if (err) *err = "invalid decoder implementation";
  return false;
}

std::vector<std::string> implementation_names() {
  std::vector<std::string> v;
  for (const char *prefix : {"", "HL"})
    for (const Stem &s : kStems)
      for (const char *suffix : {"f64", "f32"}) v.push_back(std::string(prefix) + s.text + suffix);
  return v;
}

}  // namespace ldpc
