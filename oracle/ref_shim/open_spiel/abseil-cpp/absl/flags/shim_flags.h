// TEST INFRASTRUCTURE ONLY: stand-in for absl/flags (flag.h, parse.h) so that the reference's own
// command-line tools (examples/benchmark_game.cc, examples/mcts_example.cc) build against oracle/_ref.
// ABSL_FLAG(type, name, default, help), absl::GetFlag / SetFlag, absl::ParseCommandLine(argc, argv) with
// --name=value, --name value, --boolflag / --noboolflag; returns the positional arguments (argv[0] first).
#ifndef ORACLE_REF_SHIM_ABSL_FLAGS_H_
#define ORACLE_REF_SHIM_ABSL_FLAGS_H_

#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace absl {
namespace shim_flags {
struct Entry {
  std::function<bool(const std::string&)> set;
  bool is_bool = false;
  std::string help;
};
inline std::map<std::string, Entry>& Registry() {
  static std::map<std::string, Entry> r;
  return r;
}
template <class T>
bool ParseValue(const std::string& text, T* out) {
  if constexpr (std::is_same_v<T, std::string>) {
    *out = text;
    return true;
  } else if constexpr (std::is_same_v<T, bool>) {
    if (text == "true" || text == "1" || text == "yes" || text.empty()) { *out = true; return true; }
    if (text == "false" || text == "0" || text == "no") { *out = false; return true; }
    return false;
  } else {
    std::istringstream is(text);
    is >> *out;
    return !is.fail();
  }
}
}  // namespace shim_flags

template <class T>
class Flag {
 public:
  Flag(const char* name, T dflt, const char* help) : value_(std::move(dflt)) {
    shim_flags::Entry e;
    e.set = [this](const std::string& text) { return shim_flags::ParseValue<T>(text, &value_); };
    e.is_bool = std::is_same_v<T, bool>;
    e.help = help;
    shim_flags::Registry()[name] = std::move(e);
  }
  const T& Get() const { return value_; }
  void Set(T v) { value_ = std::move(v); }

 private:
  T value_;
};
template <class T>
T GetFlag(const Flag<T>& f) { return f.Get(); }
template <class T, class V>
void SetFlag(Flag<T>* f, const V& v) { f->Set(T(v)); }

inline std::vector<char*> ParseCommandLine(int argc, char** argv) {
  std::vector<char*> positional;
  if (argc > 0) positional.push_back(argv[0]);
  auto& reg = shim_flags::Registry();
  for (int i = 1; i < argc; ++i) {
    std::string arg = argv[i];
    if (arg == "--") {
      for (int j = i + 1; j < argc; ++j) positional.push_back(argv[j]);
      break;
    }
    if (arg.rfind("--", 0) != 0 && arg.rfind("-", 0) != 0) { positional.push_back(argv[i]); continue; }
    std::string body = arg.substr(arg.rfind("--", 0) == 0 ? 2 : 1), value;
    bool has_value = false;
    size_t eq = body.find('=');
    if (eq != std::string::npos) { value = body.substr(eq + 1); body = body.substr(0, eq); has_value = true; }
    auto it = reg.find(body);
    if (it == reg.end() && body.rfind("no", 0) == 0) {
      auto neg = reg.find(body.substr(2));
      if (neg != reg.end() && neg->second.is_bool) { neg->second.set("false"); continue; }
    }
    if (it == reg.end()) { std::cerr << "unknown flag --" << body << std::endl; std::exit(1); }
    if (!has_value && !it->second.is_bool) {
      if (i + 1 >= argc) { std::cerr << "flag --" << body << " needs a value" << std::endl; std::exit(1); }
      value = argv[++i];
    }
    if (!it->second.set(value)) { std::cerr << "bad value for --" << body << ": " << value << std::endl; std::exit(1); }
  }
  return positional;
}
inline void SetProgramUsageMessage(const std::string&) {}
}  // namespace absl

#define ABSL_FLAG(Type, name, default_value, help) absl::Flag<Type> FLAGS_##name(#name, default_value, help)
#define ABSL_DECLARE_FLAG(Type, name) extern absl::Flag<Type> FLAGS_##name

#endif  // ORACLE_REF_SHIM_ABSL_FLAGS_H_
