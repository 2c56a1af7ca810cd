// TEST INFRASTRUCTURE ONLY — part of the CPU checker, never of the product path.
//
// A small JSON value type with the slice of the nlohmann/json interface that the reference's
// hot-path headers need to COMPILE (spiel.h:244-299,952-971; tic_tac_toe.h:58-75;
// connect_four.h:72-113; spiel_utils.h:456-545).  nlohmann/json is not vendored in
// /root/reference.  The JSON "struct" API is not on the hot path; this exists so
// that the genuine sources build unmodified into oracle/_ref (recipe: oracle/Makefile.ref).
// Written from the library's documented interface, not from its sources.
#ifndef ORACLE_REF_SHIM_NLOHMANN_JSON_HPP_
#define ORACLE_REF_SHIM_NLOHMANN_JSON_HPP_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <map>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace nlohmann {

class json;

namespace shim_detail {
// Free functions, so that the unqualified calls below are resolved by ADL alone (inside
// adl_serializer the member names would hide the friends the DEFINE_TYPE macros declare).
template <class J, class T>
auto call_to_json(J& j, const T& v) -> decltype(to_json(j, v), void()) {
  to_json(j, v);
}
template <class J, class T>
auto call_from_json(const J& j, T& v) -> decltype(from_json(j, v), void()) {
  from_json(j, v);
}
}  // namespace shim_detail

template <class T, class SFINAE = void>
struct adl_serializer {
  template <class J, class U = T>
  static auto to_json(J& j, const U& v) -> decltype(shim_detail::call_to_json(j, v)) {
    shim_detail::call_to_json(j, v);
  }
  template <class J, class U = T>
  static auto from_json(const J& j, U& v) -> decltype(shim_detail::call_from_json(j, v)) {
    shim_detail::call_from_json(j, v);
  }
};

class json {
 public:
  enum class value_t { null, boolean, number_integer, number_float, string, array, object };
  using object_t = std::map<std::string, json>;
  using array_t = std::vector<json>;

  class exception : public std::runtime_error {
   public:
    explicit exception(const std::string& m) : std::runtime_error(m) {}
  };
  class parse_error : public exception { public: using exception::exception; };
  class type_error : public exception { public: using exception::exception; };
  class out_of_range : public exception { public: using exception::exception; };

  json() = default;
  json(std::nullptr_t) {}
  json(const json&) = default;
  json(json&&) = default;
  json& operator=(const json&) = default;
  json& operator=(json&&) = default;

  // {{"k", v}, ...} is an object when every element is a [string, value] pair, else an array.
  json(std::initializer_list<json> init) {
    bool is_obj = true;
    for (const json& e : init)
      if (!(e.type_ == value_t::array && e.arr_.size() == 2 && e.arr_[0].type_ == value_t::string)) is_obj = false;
    if (is_obj) {
      type_ = value_t::object;
      for (const json& e : init) obj_[e.arr_[0].str_] = e.arr_[1];
    } else {
      type_ = value_t::array;
      arr_.assign(init.begin(), init.end());
    }
  }

  template <class T, class D = std::decay_t<T>,
            class = std::enable_if_t<!std::is_same_v<D, json> && !std::is_same_v<D, std::nullptr_t> &&
                                     !std::is_same_v<D, std::initializer_list<json>>>>
  json(T&& v) {  // NOLINT: implicit by design, as in the library
    Assign(std::forward<T>(v));
  }

  static json object() { json j; j.type_ = value_t::object; return j; }
  static json array() { json j; j.type_ = value_t::array; return j; }

  value_t type() const { return type_; }
  bool is_null() const { return type_ == value_t::null; }
  bool is_boolean() const { return type_ == value_t::boolean; }
  bool is_number_integer() const { return type_ == value_t::number_integer; }
  bool is_number_float() const { return type_ == value_t::number_float; }
  bool is_number() const { return is_number_integer() || is_number_float(); }
  bool is_string() const { return type_ == value_t::string; }
  bool is_array() const { return type_ == value_t::array; }
  bool is_object() const { return type_ == value_t::object; }
  const char* type_name() const {
    switch (type_) {
      case value_t::null: return "null";
      case value_t::boolean: return "boolean";
      case value_t::string: return "string";
      case value_t::array: return "array";
      case value_t::object: return "object";
      default: return "number";
    }
  }

  bool contains(const std::string& k) const { return type_ == value_t::object && obj_.count(k) != 0; }
  size_t size() const {
    return type_ == value_t::array ? arr_.size() : type_ == value_t::object ? obj_.size() : type_ == value_t::null ? 0 : 1;
  }
  bool empty() const { return size() == 0; }

  json& operator[](const std::string& k) {
    if (type_ == value_t::null) type_ = value_t::object;
    if (type_ != value_t::object) throw type_error(std::string("cannot use operator[] with a string argument with ") + type_name());
    return obj_[k];
  }
  json& operator[](const char* k) { return (*this)[std::string(k)]; }
  const json& operator[](const std::string& k) const { return at(k); }
  const json& operator[](const char* k) const { return at(std::string(k)); }
  json& operator[](size_t i) {
    if (type_ == value_t::null) type_ = value_t::array;
    if (type_ != value_t::array) throw type_error("cannot use operator[] with a numeric argument");
    if (i >= arr_.size()) arr_.resize(i + 1);
    return arr_[i];
  }
  const json& operator[](size_t i) const { return at(i); }
  json& operator[](int i) { return (*this)[static_cast<size_t>(i)]; }
  const json& operator[](int i) const { return at(static_cast<size_t>(i)); }
  json& at(const std::string& k) {
    if (type_ != value_t::object) throw type_error(std::string("cannot use at() with ") + type_name());
    auto it = obj_.find(k);
    if (it == obj_.end()) throw out_of_range("key '" + k + "' not found");
    return it->second;
  }
  const json& at(const std::string& k) const { return const_cast<json*>(this)->at(k); }
  json& at(size_t i) {
    if (type_ != value_t::array) throw type_error(std::string("cannot use at() with ") + type_name());
    if (i >= arr_.size()) throw out_of_range("array index out of range");
    return arr_[i];
  }
  const json& at(size_t i) const { return const_cast<json*>(this)->at(i); }

  template <class T>
  T value(const std::string& k, const T& dflt) const {
    if (contains(k)) return obj_.at(k).template get<T>();
    return dflt;
  }
  std::string value(const std::string& k, const char* dflt) const {
    if (contains(k)) return obj_.at(k).get<std::string>();
    return dflt;
  }

  void push_back(json v) {
    if (type_ == value_t::null) type_ = value_t::array;
    if (type_ != value_t::array) throw type_error("cannot use push_back()");
    arr_.push_back(std::move(v));
  }
  template <class... A>
  json& emplace_back(A&&... a) {
    push_back(json(std::forward<A>(a)...));
    return arr_.back();
  }
  void merge_patch(const json& patch) {
    if (patch.type_ != value_t::object) { *this = patch; return; }
    if (type_ != value_t::object) { *this = object(); }
    for (const auto& kv : patch.obj_) {
      if (kv.second.is_null()) obj_.erase(kv.first);
      else obj_[kv.first].merge_patch(kv.second);
    }
  }
  void update(const json& other) {
    if (type_ == value_t::null) type_ = value_t::object;
    for (const auto& kv : other.obj_) obj_[kv.first] = kv.second;
  }

  // Array iteration.
  array_t::iterator begin() { return arr_.begin(); }
  array_t::iterator end() { return arr_.end(); }
  array_t::const_iterator begin() const { return arr_.begin(); }
  array_t::const_iterator end() const { return arr_.end(); }

  // `for (auto& [key, value] : j.items())`
  template <class J>
  struct item_t {
    const std::string& key;
    J& value;
  };
  std::vector<item_t<json>> items() {
    std::vector<item_t<json>> v;
    for (auto& kv : obj_) v.push_back({kv.first, kv.second});
    return v;
  }
  std::vector<item_t<const json>> items() const {
    std::vector<item_t<const json>> v;
    for (const auto& kv : obj_) v.push_back({kv.first, kv.second});
    return v;
  }

  template <class T>
  T get() const {
    T v{};
    get_to(v);
    return v;
  }
  template <class T>
  T& get_to(T& v) const {
    Extract(v);
    return v;
  }
  template <class T, class = std::enable_if_t<!std::is_same_v<T, json> && !std::is_pointer_v<T> &&
                                              !std::is_same_v<T, std::nullptr_t> && !std::is_same_v<T, char> &&
                                              !std::is_same_v<T, std::initializer_list<char>>>>
  operator T() const {  // NOLINT
    return get<T>();
  }

  friend bool operator==(const json& a, const json& b) {
    if (a.is_number() && b.is_number() && a.type_ != b.type_) return a.AsDouble() == b.AsDouble();
    if (a.type_ != b.type_) return false;
    switch (a.type_) {
      case value_t::null: return true;
      case value_t::boolean: return a.bool_ == b.bool_;
      case value_t::number_integer: return a.int_ == b.int_;
      case value_t::number_float: return a.dbl_ == b.dbl_;
      case value_t::string: return a.str_ == b.str_;
      case value_t::array: return a.arr_ == b.arr_;
      case value_t::object: return a.obj_ == b.obj_;
    }
    return false;
  }
  friend bool operator!=(const json& a, const json& b) { return !(a == b); }

  std::string dump(int indent = -1) const {
    std::string out;
    Dump(out, indent, 0);
    return out;
  }
  static json parse(const std::string& text) {
    size_t pos = 0;
    json j = ParseValue(text, pos);
    SkipWs(text, pos);
    if (pos != text.size()) throw parse_error("unexpected trailing characters at offset " + std::to_string(pos));
    return j;
  }
  friend std::ostream& operator<<(std::ostream& os, const json& j) { return os << j.dump(); }

 private:
  double AsDouble() const { return type_ == value_t::number_integer ? static_cast<double>(int_) : dbl_; }

  template <class T>
  void Assign(T&& v) {
    using D = std::decay_t<T>;
    if constexpr (std::is_same_v<D, bool>) {
      type_ = value_t::boolean; bool_ = v;
    } else if constexpr (std::is_integral_v<D> || std::is_enum_v<D>) {
      type_ = value_t::number_integer; int_ = static_cast<int64_t>(v);
    } else if constexpr (std::is_floating_point_v<D>) {
      type_ = value_t::number_float; dbl_ = static_cast<double>(v);
    } else if constexpr (std::is_convertible_v<T, std::string>) {
      type_ = value_t::string; str_ = std::string(std::forward<T>(v));
    } else if constexpr (IsMap<D>::value) {
      type_ = value_t::object;
      for (const auto& kv : v) obj_[kv.first] = json(kv.second);
    } else if constexpr (IsSequence<D>::value) {
      type_ = value_t::array;
      for (const auto& e : v) arr_.push_back(json(e));
    } else {
      adl_serializer<D>::to_json(*this, static_cast<const D&>(v));
    }
  }
  template <class T>
  void Extract(T& v) const {
    if constexpr (std::is_same_v<T, json>) {
      v = *this;
    } else if constexpr (std::is_same_v<T, bool>) {
      if (type_ != value_t::boolean) throw type_error(std::string("type must be boolean, but is ") + type_name());
      v = bool_;
    } else if constexpr (std::is_integral_v<T> || std::is_enum_v<T>) {
      if (type_ == value_t::number_integer) v = static_cast<T>(int_);
      else if (type_ == value_t::number_float) v = static_cast<T>(static_cast<int64_t>(dbl_));
      else if (type_ == value_t::boolean) v = static_cast<T>(bool_);
      else throw type_error(std::string("type must be number, but is ") + type_name());
    } else if constexpr (std::is_floating_point_v<T>) {
      if (!is_number()) throw type_error(std::string("type must be number, but is ") + type_name());
      v = static_cast<T>(AsDouble());
    } else if constexpr (std::is_same_v<T, std::string>) {
      if (type_ != value_t::string) throw type_error(std::string("type must be string, but is ") + type_name());
      v = str_;
    } else if constexpr (IsMap<T>::value) {
      if (type_ != value_t::object) throw type_error(std::string("type must be object, but is ") + type_name());
      v.clear();
      for (const auto& kv : obj_) v[kv.first] = kv.second.template get<typename T::mapped_type>();
    } else if constexpr (IsSequence<T>::value) {
      if (type_ != value_t::array) throw type_error(std::string("type must be array, but is ") + type_name());
      v.clear();
      for (const json& e : arr_) v.push_back(e.template get<typename T::value_type>());
    } else {
      adl_serializer<T>::from_json(*this, v);
    }
  }

  template <class T, class = void>
  struct IsMap : std::false_type {};
  template <class T>
  struct IsMap<T, std::void_t<typename T::mapped_type, typename T::key_type>> : std::true_type {};
  template <class T, class = void>
  struct IsSequence : std::false_type {};
  template <class T>
  struct IsSequence<T, std::void_t<typename T::value_type, decltype(std::declval<T&>().push_back(std::declval<typename T::value_type>()))>>
      : std::bool_constant<!std::is_same_v<T, std::string>> {};

  static void DumpString(std::string& out, const std::string& s) {
    out.push_back('"');
    for (unsigned char c : s) {
      switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        case '\b': out += "\\b"; break;
        case '\f': out += "\\f"; break;
        default:
          if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); out += b; }
          else out.push_back(static_cast<char>(c));
      }
    }
    out.push_back('"');
  }
  void Dump(std::string& out, int indent, int depth) const {
    auto newline = [&](int d) {
      if (indent >= 0) { out.push_back('\n'); out.append(static_cast<size_t>(indent) * d, ' '); }
    };
    switch (type_) {
      case value_t::null: out += "null"; break;
      case value_t::boolean: out += bool_ ? "true" : "false"; break;
      case value_t::number_integer: out += std::to_string(int_); break;
      case value_t::number_float: {
        if (!std::isfinite(dbl_)) { out += "null"; break; }
        char b[40];
        std::snprintf(b, sizeof b, "%.17g", dbl_);
        // shortest representation that round-trips
        for (int p = 1; p < 17; ++p) {
          char t[40];
          std::snprintf(t, sizeof t, "%.*g", p, dbl_);
          if (std::strtod(t, nullptr) == dbl_) { std::snprintf(b, sizeof b, "%s", t); break; }
        }
        std::string s(b);
        if (s.find_first_of(".eE") == std::string::npos) s += ".0";
        out += s;
        break;
      }
      case value_t::string: DumpString(out, str_); break;
      case value_t::array: {
        if (arr_.empty()) { out += "[]"; break; }
        out.push_back('[');
        for (size_t i = 0; i < arr_.size(); ++i) {
          if (i) out.push_back(',');
          newline(depth + 1);
          arr_[i].Dump(out, indent, depth + 1);
        }
        newline(depth);
        out.push_back(']');
        break;
      }
      case value_t::object: {
        if (obj_.empty()) { out += "{}"; break; }
        out.push_back('{');
        bool first = true;
        for (const auto& kv : obj_) {
          if (!first) out.push_back(',');
          first = false;
          newline(depth + 1);
          DumpString(out, kv.first);
          out.push_back(':');
          if (indent >= 0) out.push_back(' ');
          kv.second.Dump(out, indent, depth + 1);
        }
        newline(depth);
        out.push_back('}');
        break;
      }
    }
  }

  static void SkipWs(const std::string& t, size_t& p) {
    while (p < t.size() && (t[p] == ' ' || t[p] == '\n' || t[p] == '\r' || t[p] == '\t')) ++p;
  }
  static std::string ParseString(const std::string& t, size_t& p) {
    std::string s;
    ++p;  // opening quote
    while (p < t.size() && t[p] != '"') {
      char c = t[p++];
      if (c != '\\') { s.push_back(c); continue; }
      if (p >= t.size()) break;
      char e = t[p++];
      switch (e) {
        case 'n': s.push_back('\n'); break;
        case 't': s.push_back('\t'); break;
        case 'r': s.push_back('\r'); break;
        case 'b': s.push_back('\b'); break;
        case 'f': s.push_back('\f'); break;
        case 'u': {
          if (p + 4 > t.size()) throw parse_error("bad \\u escape");
          unsigned cp = static_cast<unsigned>(std::strtoul(t.substr(p, 4).c_str(), nullptr, 16));
          p += 4;
          if (cp < 0x80) s.push_back(static_cast<char>(cp));
          else if (cp < 0x800) { s.push_back(static_cast<char>(0xC0 | (cp >> 6))); s.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
          else { s.push_back(static_cast<char>(0xE0 | (cp >> 12))); s.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F))); s.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
          break;
        }
        default: s.push_back(e);
      }
    }
    if (p >= t.size()) throw parse_error("unterminated string");
    ++p;  // closing quote
    return s;
  }
  static json ParseValue(const std::string& t, size_t& p) {
    SkipWs(t, p);
    if (p >= t.size()) throw parse_error("unexpected end of input");
    char c = t[p];
    if (c == '{') {
      json j = object();
      ++p;
      SkipWs(t, p);
      if (p < t.size() && t[p] == '}') { ++p; return j; }
      for (;;) {
        SkipWs(t, p);
        if (p >= t.size() || t[p] != '"') throw parse_error("expected a string key at offset " + std::to_string(p));
        std::string k = ParseString(t, p);
        SkipWs(t, p);
        if (p >= t.size() || t[p] != ':') throw parse_error("expected ':' at offset " + std::to_string(p));
        ++p;
        j.obj_[k] = ParseValue(t, p);
        SkipWs(t, p);
        if (p < t.size() && t[p] == ',') { ++p; continue; }
        if (p < t.size() && t[p] == '}') { ++p; return j; }
        throw parse_error("expected ',' or '}' at offset " + std::to_string(p));
      }
    }
    if (c == '[') {
      json j = array();
      ++p;
      SkipWs(t, p);
      if (p < t.size() && t[p] == ']') { ++p; return j; }
      for (;;) {
        j.arr_.push_back(ParseValue(t, p));
        SkipWs(t, p);
        if (p < t.size() && t[p] == ',') { ++p; continue; }
        if (p < t.size() && t[p] == ']') { ++p; return j; }
        throw parse_error("expected ',' or ']' at offset " + std::to_string(p));
      }
    }
    if (c == '"') return json(ParseString(t, p));
    if (t.compare(p, 4, "true") == 0) { p += 4; return json(true); }
    if (t.compare(p, 5, "false") == 0) { p += 5; return json(false); }
    if (t.compare(p, 4, "null") == 0) { p += 4; return json(); }
    size_t q = p;
    bool is_float = false;
    if (q < t.size() && (t[q] == '-' || t[q] == '+')) ++q;
    while (q < t.size() && (std::isdigit(static_cast<unsigned char>(t[q])) || t[q] == '.' || t[q] == 'e' || t[q] == 'E' || t[q] == '-' || t[q] == '+')) {
      if (t[q] == '.' || t[q] == 'e' || t[q] == 'E') is_float = true;
      ++q;
    }
    if (q == p) throw parse_error("unexpected character at offset " + std::to_string(p));
    std::string num = t.substr(p, q - p);
    p = q;
    if (is_float) return json(std::strtod(num.c_str(), nullptr));
    return json(static_cast<int64_t>(std::strtoll(num.c_str(), nullptr, 10)));
  }

  value_t type_ = value_t::null;
  bool bool_ = false;
  int64_t int_ = 0;
  double dbl_ = 0;
  std::string str_;
  array_t arr_;
  object_t obj_;
};

using ordered_json = json;

}  // namespace nlohmann

// NLOHMANN_DEFINE_TYPE_INTRUSIVE(Type, members...): friend to_json / from_json over the members.
#define ORACLE_SHIM_JSON_EXPAND(x) x
#define ORACLE_SHIM_JSON_GET(_1, _2, _3, _4, _5, _6, _7, _8, _9, _10, _11, _12, _13, _14, _15, _16, NAME, ...) NAME
#define ORACLE_SHIM_JSON_FE1(F, a) F(a)
#define ORACLE_SHIM_JSON_FE2(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE1(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE3(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE2(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE4(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE3(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE5(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE4(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE6(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE5(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE7(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE6(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE8(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE7(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE9(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE8(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE10(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE9(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE11(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE10(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE12(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE11(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE13(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE12(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE14(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE13(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE15(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE14(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FE16(F, a, ...) F(a) ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_FE15(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_FOR_EACH(F, ...)                                                              \
  ORACLE_SHIM_JSON_EXPAND(ORACLE_SHIM_JSON_GET(                                                        \
      __VA_ARGS__, ORACLE_SHIM_JSON_FE16, ORACLE_SHIM_JSON_FE15, ORACLE_SHIM_JSON_FE14,                \
      ORACLE_SHIM_JSON_FE13, ORACLE_SHIM_JSON_FE12, ORACLE_SHIM_JSON_FE11, ORACLE_SHIM_JSON_FE10,      \
      ORACLE_SHIM_JSON_FE9, ORACLE_SHIM_JSON_FE8, ORACLE_SHIM_JSON_FE7, ORACLE_SHIM_JSON_FE6,          \
      ORACLE_SHIM_JSON_FE5, ORACLE_SHIM_JSON_FE4, ORACLE_SHIM_JSON_FE3, ORACLE_SHIM_JSON_FE2,          \
      ORACLE_SHIM_JSON_FE1)(F, __VA_ARGS__))
#define ORACLE_SHIM_JSON_TO(m) nlohmann_json_j[#m] = nlohmann_json_t.m;
#define ORACLE_SHIM_JSON_FROM(m) nlohmann_json_j.at(#m).get_to(nlohmann_json_t.m);
#define ORACLE_SHIM_JSON_FROM_DEFAULT(m) \
  if (nlohmann_json_j.contains(#m)) nlohmann_json_j.at(#m).get_to(nlohmann_json_t.m);

#define NLOHMANN_DEFINE_TYPE_INTRUSIVE(Type, ...)                                          \
  friend void to_json(nlohmann::json& nlohmann_json_j, const Type& nlohmann_json_t) {      \
    if (!nlohmann_json_j.is_object()) nlohmann_json_j = nlohmann::json::object();          \
    ORACLE_SHIM_JSON_FOR_EACH(ORACLE_SHIM_JSON_TO, __VA_ARGS__)                            \
  }                                                                                        \
  friend void from_json(const nlohmann::json& nlohmann_json_j, Type& nlohmann_json_t) {    \
    ORACLE_SHIM_JSON_FOR_EACH(ORACLE_SHIM_JSON_FROM, __VA_ARGS__)                          \
  }
#define NLOHMANN_DEFINE_TYPE_INTRUSIVE_WITH_DEFAULT(Type, ...)                             \
  friend void to_json(nlohmann::json& nlohmann_json_j, const Type& nlohmann_json_t) {      \
    if (!nlohmann_json_j.is_object()) nlohmann_json_j = nlohmann::json::object();          \
    ORACLE_SHIM_JSON_FOR_EACH(ORACLE_SHIM_JSON_TO, __VA_ARGS__)                            \
  }                                                                                        \
  friend void from_json(const nlohmann::json& nlohmann_json_j, Type& nlohmann_json_t) {    \
    ORACLE_SHIM_JSON_FOR_EACH(ORACLE_SHIM_JSON_FROM_DEFAULT, __VA_ARGS__)                  \
  }
#define NLOHMANN_DEFINE_TYPE_NON_INTRUSIVE(Type, ...)                                      \
  inline void to_json(nlohmann::json& nlohmann_json_j, const Type& nlohmann_json_t) {      \
    if (!nlohmann_json_j.is_object()) nlohmann_json_j = nlohmann::json::object();          \
    ORACLE_SHIM_JSON_FOR_EACH(ORACLE_SHIM_JSON_TO, __VA_ARGS__)                            \
  }                                                                                        \
  inline void from_json(const nlohmann::json& nlohmann_json_j, Type& nlohmann_json_t) {    \
    ORACLE_SHIM_JSON_FOR_EACH(ORACLE_SHIM_JSON_FROM, __VA_ARGS__)                          \
  }

#endif  // ORACLE_REF_SHIM_NLOHMANN_JSON_HPP_
