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
    if (std::strcmp(suffix, "f32@fast") == 0 && (s.rule == Rule::Tanh || s.rule == Rule::Phi)) {
      impl.rule = s.rule;
      impl.fast = true;
      *out = impl;
      return true;
    }
    if (std::strncmp(suffix, "i8", 2) == 0 && (s.rule == Rule::Minstarapprox || s.rule == Rule::Aminstar)) {
      // factory.rs:246-263, 270-275: optional Jones / PartialHardLimit / Deg1Clip, in this order;
      // the layered schedule exists only with and without PartialHardLimit
      const char *q = suffix + 2;
      impl.rule = s.rule;
      impl.i8 = true;
      if (std::strncmp(q, "Jones", 5) == 0) {
        impl.jones = true;
        q += 5;
      }
      if (std::strncmp(q, "PartialHardLimit", 16) == 0) {
        impl.hardlimit = true;
        q += 16;
      }
      if (std::strncmp(q, "Deg1Clip", 8) == 0) {
        impl.deg1clip = true;
        q += 8;
      }
      if (*q == 0 && !(impl.schedule == Schedule::Layered && (impl.jones || impl.deg1clip))) {
        *out = impl;
        return true;
      }
    }
  }
  if (err) *err = "invalid decoder implementation";
  return false;
}

bool parse_puncturing_pattern(const std::string &text, std::vector<uint8_t> *out) {
  out->clear();
  if (text.empty()) return true;
  size_t start = 0;
  while (true) {
    const size_t comma = text.find(',', start);
    const std::string tok = text.substr(start, comma == std::string::npos ? std::string::npos : comma - start);
    if (tok != "0" && tok != "1") return false;
    out->push_back(tok == "1");
    if (comma == std::string::npos) break;
    start = comma + 1;
  }
  return true;
}

std::vector<std::string> fast_implementation_names() {
  return {"Tanhf32@fast", "HLTanhf32@fast", "Phif32@fast", "HLPhif32@fast"};
}

std::vector<std::string> implementation_names() {
  std::vector<std::string> v;
  for (const char *prefix : {"", "HL"})
    for (const Stem &s : kStems)
      for (const char *suffix : {"f64", "f32"}) v.push_back(std::string(prefix) + s.text + suffix);
  for (const char *base : {"Minstarapproxi8", "Aminstari8"}) {
    for (const char *j : {"", "Jones"})
      for (const char *h : {"", "PartialHardLimit"})
        for (const char *d : {"", "Deg1Clip"}) v.push_back(std::string(base) + j + h + d);
    v.push_back(std::string("HL") + base);
    v.push_back(std::string("HL") + base + "PartialHardLimit");
  }
  return v;
}

}  // namespace ldpc
