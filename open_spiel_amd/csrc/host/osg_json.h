// A small JSON value for the host mirror's struct API (State::ToJson, Game::NewInitialState(json), LoadGameFromJson:
// open_spiel/spiel.h:235-299, 464-473, 967-971, 1332-1340).  The reference uses nlohmann::json, which is not in this
// image; this class covers what that API needs — parse, dump, typed access — and prints as nlohmann's dump() does
// with default arguments: no whitespace, object keys in sorted order (nlohmann::json keeps objects in a std::map),
// strings escaped per RFC 8259, integers as integers.  include/open_spiel/json/include/nlohmann/json.hpp opens it as
// nlohmann::json for sources written against the reference.
#ifndef OSG_HOST_JSON_H_
#define OSG_HOST_JSON_H_

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace open_spiel {
namespace hip {

class Json {
 public:
  enum class Type { kNull, kBool, kInt, kDouble, kString, kArray, kObject };
  using array_t = std::vector<Json>;
  using object_t = std::map<std::string, Json>;
  struct exception : public std::runtime_error {  // nlohmann::json::exception
    explicit exception(const std::string& what) : std::runtime_error(what) {}
  };
  using parse_error = exception;
  using type_error = exception;
  using out_of_range = exception;

  Json() = default;
  Json(std::nullptr_t) {}
  Json(bool v) : type_(Type::kBool), int_(v ? 1 : 0) {}
  template <class T, std::enable_if_t<std::is_integral_v<T> && !std::is_same_v<T, bool>, int> = 0>
  Json(T v) : type_(Type::kInt), int_(static_cast<int64_t>(v)) {}
  Json(double v) : type_(Type::kDouble), double_(v) {}
  Json(const char* v) : type_(Type::kString), string_(v) {}
  Json(const std::string& v) : type_(Type::kString), string_(v) {}
  template <class T>
  Json(const std::vector<T>& v) : type_(Type::kArray) {
    for (const T& e : v) array_.emplace_back(e);
  }
  static Json array() { Json j; j.type_ = Type::kArray; return j; }
  static Json object() { Json j; j.type_ = Type::kObject; return j; }

  Type type() const { return type_; }
  bool is_null() const { return type_ == Type::kNull; }
  bool is_boolean() const { return type_ == Type::kBool; }
  bool is_number_integer() const { return type_ == Type::kInt; }
  bool is_number() const { return type_ == Type::kInt || type_ == Type::kDouble; }
  bool is_string() const { return type_ == Type::kString; }
  bool is_array() const { return type_ == Type::kArray; }
  bool is_object() const { return type_ == Type::kObject; }
  size_t size() const { return is_array() ? array_.size() : (is_object() ? object_.size() : (is_null() ? 0 : 1)); }
  bool contains(const std::string& key) const { return is_object() && object_.count(key) != 0; }

  Json& operator[](const std::string& key) {  // a null value becomes an object, as in nlohmann::json
    if (is_null()) type_ = Type::kObject;
    if (!is_object()) throw exception("cannot use operator[] with a string argument on a non-object");
    return object_[key];
  }
  const Json& at(const std::string& key) const {
    if (!is_object()) throw exception("cannot use at() with a string argument on a non-object");
    auto it = object_.find(key);
    if (it == object_.end()) throw exception("key '" + key + "' not found");
    return it->second;
  }
  const Json& at(size_t i) const {
    if (!is_array() || i >= array_.size()) throw exception("array index out of range");
    return array_[i];
  }
  const Json& operator[](size_t i) const { return at(i); }
  void push_back(Json v) {
    if (is_null()) type_ = Type::kArray;
    if (!is_array()) throw exception("cannot use push_back() on a non-array");
    array_.push_back(std::move(v));
  }
  const array_t& items_array() const { return array_; }
  const object_t& items_object() const { return object_; }

  // typed access (nlohmann's get<T>() / get_to(T&))
  template <class T>
  T get() const {
    T v{};
    get_to(v);
    return v;
  }
  void get_to(bool& v) const {
    if (!is_boolean()) throw exception("type must be boolean");
    v = int_ != 0;
  }
  template <class T, std::enable_if_t<std::is_integral_v<T> && !std::is_same_v<T, bool>, int> = 0>
  void get_to(T& v) const {
    // (a value the target type cannot hold is an error, not a wrapped or undefined conversion)
    if (type_ == Type::kInt) {
      if (std::is_signed_v<T> ? (int_ < static_cast<int64_t>(std::numeric_limits<T>::min()) ||
                                 int_ > static_cast<int64_t>(std::numeric_limits<T>::max()))
                              : (int_ < 0 || static_cast<uint64_t>(int_) > static_cast<uint64_t>(std::numeric_limits<T>::max())))
        throw exception("number out of range of the requested integer type");
      v = static_cast<T>(int_);
    } else if (type_ == Type::kDouble) {
      // (2^63 and 2^64 are exact doubles; the comparison keeps NaN out as well)
      const double lo = std::is_signed_v<T> ? static_cast<double>(std::numeric_limits<T>::min()) : 0.0;
      const double hi = static_cast<double>(std::numeric_limits<T>::max());
      if (!(double_ >= lo && double_ <= hi) || (sizeof(T) == 8 && double_ >= hi))
        throw exception("number out of range of the requested integer type");
      v = static_cast<T>(double_);
    } else {
      throw exception("type must be number");
    }
  }
  void get_to(double& v) const {
    if (type_ == Type::kInt) v = static_cast<double>(int_);
    else if (type_ == Type::kDouble) v = double_;
    else throw exception("type must be number");
  }
  void get_to(std::string& v) const {
    if (!is_string()) throw exception("type must be string");
    v = string_;
  }
  template <class T>
  void get_to(std::vector<T>& v) const {
    if (!is_array()) throw exception("type must be array");
    v.clear();
    for (const Json& e : array_) {
      T x{};
      e.get_to(x);
      v.push_back(std::move(x));
    }
  }
  // any type with a `void from_json(const Json&)` member (the struct types of the mirror)
  template <class T, class = decltype(std::declval<T&>().from_json(std::declval<const Json&>()))>
  void get_to(T& v) const { v.from_json(*this); }

  bool operator==(const Json& o) const { return dump() == o.dump(); }
  bool operator!=(const Json& o) const { return !(*this == o); }

  std::string dump() const {
    std::string out;
    Dump(&out);
    return out;
  }

  static Json parse(const std::string& text) {
    size_t pos = 0;
    Json v = ParseValue(text, &pos);
    SkipSpace(text, &pos);
    if (pos != text.size()) throw exception("parse error at byte " + std::to_string(pos) + ": unexpected trailing characters");
    return v;
  }

 private:
  static void SkipSpace(const std::string& t, size_t* p) {
    while (*p < t.size() && (t[*p] == ' ' || t[*p] == '\t' || t[*p] == '\n' || t[*p] == '\r')) ++*p;
  }
  [[noreturn]] static void Fail(size_t pos, const std::string& what) {
    throw exception("parse error at byte " + std::to_string(pos) + ": " + what);
  }
  static void AppendUtf8(std::string* out, uint32_t cp) {
    if (cp < 0x80) out->push_back(static_cast<char>(cp));
    else if (cp < 0x800) { out->push_back(static_cast<char>(0xC0 | (cp >> 6))); out->push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
      out->push_back(static_cast<char>(0xE0 | (cp >> 12))); out->push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
      out->push_back(static_cast<char>(0x80 | (cp & 0x3F)));
    } else {
      out->push_back(static_cast<char>(0xF0 | (cp >> 18))); out->push_back(static_cast<char>(0x80 | ((cp >> 12) & 0x3F)));
      out->push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F))); out->push_back(static_cast<char>(0x80 | (cp & 0x3F)));
    }
  }
  static uint32_t ParseHex4(const std::string& t, size_t* p) {
    if (*p + 4 > t.size()) Fail(*p, "truncated \\u escape");
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = t[(*p)++];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= static_cast<uint32_t>(c - '0');
      else if (c >= 'a' && c <= 'f') v |= static_cast<uint32_t>(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= static_cast<uint32_t>(c - 'A' + 10);
      else Fail(*p - 1, "bad hex digit in \\u escape");
    }
    return v;
  }
  static std::string ParseString(const std::string& t, size_t* p) {
    std::string out;
    ++*p;  // the opening quote
    for (;;) {
      if (*p >= t.size()) Fail(*p, "unterminated string");
      const char c = t[(*p)++];
      if (c == '"') return out;
      if (static_cast<unsigned char>(c) < 0x20) Fail(*p - 1, "control character in string");
      if (c != '\\') { out.push_back(c); continue; }
      if (*p >= t.size()) Fail(*p, "unterminated escape");
      const char e = t[(*p)++];
      switch (e) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          uint32_t cp = ParseHex4(t, p);
          if (cp >= 0xD800 && cp <= 0xDBFF) {  // a surrogate pair
            if (*p + 2 > t.size() || t[*p] != '\\' || t[*p + 1] != 'u') Fail(*p, "missing low surrogate");
            *p += 2;
            const uint32_t lo = ParseHex4(t, p);
            if (lo < 0xDC00 || lo > 0xDFFF) Fail(*p, "bad low surrogate");
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          AppendUtf8(&out, cp);
          break;
        }
        default: Fail(*p - 1, "bad escape");
      }
    }
  }
  static Json ParseNumber(const std::string& t, size_t* p) {
    const size_t start = *p;
    bool integral = true;
    if (*p < t.size() && t[*p] == '-') ++*p;
    if (*p >= t.size() || t[*p] < '0' || t[*p] > '9') Fail(*p, "bad number");
    while (*p < t.size() && t[*p] >= '0' && t[*p] <= '9') ++*p;
    if (*p < t.size() && t[*p] == '.') {
      integral = false;
      ++*p;
      if (*p >= t.size() || t[*p] < '0' || t[*p] > '9') Fail(*p, "bad fraction");
      while (*p < t.size() && t[*p] >= '0' && t[*p] <= '9') ++*p;
    }
    if (*p < t.size() && (t[*p] == 'e' || t[*p] == 'E')) {
      integral = false;
      ++*p;
      if (*p < t.size() && (t[*p] == '+' || t[*p] == '-')) ++*p;
      if (*p >= t.size() || t[*p] < '0' || t[*p] > '9') Fail(*p, "bad exponent");
      while (*p < t.size() && t[*p] >= '0' && t[*p] <= '9') ++*p;
    }
    const std::string token = t.substr(start, *p - start);
    {  // RFC 8259: no leading zeros ("01", "-007"); "0", "-0", "0.5" are fine
      const size_t d = token[0] == '-' ? 1 : 0;
      if (token.size() > d + 1 && token[d] == '0' && token[d + 1] >= '0' && token[d + 1] <= '9') Fail(start, "leading zero in a number");
    }
    if (integral) {   // every integer an int64 holds stays an integer; beyond that the value is a double, as in nlohmann
      errno = 0;
      char* end = nullptr;
      const long long v = std::strtoll(token.c_str(), &end, 10);
      if (errno != ERANGE && end && *end == '\0') return Json(static_cast<int64_t>(v));
    }
    return Json(std::strtod(token.c_str(), nullptr));
  }
  static constexpr int kMaxDepth = 256;   // (a bound on nesting: the parser recurses)
  static Json ParseValue(const std::string& t, size_t* p, int depth = 0) {
    if (depth > kMaxDepth) Fail(*p, "nesting deeper than 256 levels");
    SkipSpace(t, p);
    if (*p >= t.size()) Fail(*p, "unexpected end of input");
    const char c = t[*p];
    if (c == '{') {
      Json obj = object();
      ++*p;
      SkipSpace(t, p);
      if (*p < t.size() && t[*p] == '}') { ++*p; return obj; }
      for (;;) {
        SkipSpace(t, p);
        if (*p >= t.size() || t[*p] != '"') Fail(*p, "expected a string key");
        std::string key = ParseString(t, p);
        SkipSpace(t, p);
        if (*p >= t.size() || t[*p] != ':') Fail(*p, "expected ':'");
        ++*p;
        obj.object_[key] = ParseValue(t, p, depth + 1);
        SkipSpace(t, p);
        if (*p < t.size() && t[*p] == ',') { ++*p; continue; }
        if (*p < t.size() && t[*p] == '}') { ++*p; return obj; }
        Fail(*p, "expected ',' or '}'");
      }
    }
    if (c == '[') {
      Json arr = array();
      ++*p;
      SkipSpace(t, p);
      if (*p < t.size() && t[*p] == ']') { ++*p; return arr; }
      for (;;) {
        arr.array_.push_back(ParseValue(t, p, depth + 1));
        SkipSpace(t, p);
        if (*p < t.size() && t[*p] == ',') { ++*p; continue; }
        if (*p < t.size() && t[*p] == ']') { ++*p; return arr; }
        Fail(*p, "expected ',' or ']'");
      }
    }
    if (c == '"') return Json(ParseString(t, p));
    if (t.compare(*p, 4, "true") == 0) { *p += 4; return Json(true); }
    if (t.compare(*p, 5, "false") == 0) { *p += 5; return Json(false); }
    if (t.compare(*p, 4, "null") == 0) { *p += 4; return Json(); }
    if (c == '-' || (c >= '0' && c <= '9')) return ParseNumber(t, p);
    Fail(*p, "unexpected character");
  }
  static void DumpString(const std::string& s, std::string* out) {
    out->push_back('"');
    for (const char ch : s) {
      const unsigned char c = static_cast<unsigned char>(ch);
      switch (c) {
        case '"': *out += "\\\""; break;
        case '\\': *out += "\\\\"; break;
        case '\b': *out += "\\b"; break;
        case '\f': *out += "\\f"; break;
        case '\n': *out += "\\n"; break;
        case '\r': *out += "\\r"; break;
        case '\t': *out += "\\t"; break;
        default:
          if (c < 0x20) {
            char buf[8];
            std::snprintf(buf, sizeof buf, "\\u%04x", c);
            *out += buf;
          } else {
            out->push_back(ch);
          }
      }
    }
    out->push_back('"');
  }
  void Dump(std::string* out) const {
    switch (type_) {
      case Type::kNull: *out += "null"; return;
      case Type::kBool: *out += int_ ? "true" : "false"; return;
      case Type::kInt: *out += std::to_string(int_); return;
      case Type::kDouble: {
        if (!std::isfinite(double_)) { *out += "null"; return; }  // as nlohmann::json prints non-finite numbers
        // the shortest digits that read back to the same double, laid out as nlohmann's dump() lays them out
        // (to_chars.hpp format_buffer: plain digits for decimal exponents in (-4, 15], else d[.ddd]e+XX; a float
        // keeps a ".0")
        char buf[40];
        int prec = 1;
        for (; prec <= 17; ++prec) {
          std::snprintf(buf, sizeof buf, "%.*e", prec - 1, double_);
          if (std::strtod(buf, nullptr) == double_) break;
        }
        std::string sci = buf;                       // [-]d[.ddd]e[+-]XX
        const bool negative = sci[0] == '-';
        if (negative) sci.erase(0, 1);
        const size_t epos = sci.find('e');
        std::string digits = sci.substr(0, epos);
        digits.erase(std::remove(digits.begin(), digits.end(), '.'), digits.end());
        const int exp10 = std::atoi(sci.c_str() + epos + 1);
        const int k = static_cast<int>(digits.size()), n = exp10 + 1;   // value = 0.digits x 10^n
        std::string text;
        if (k <= n && n <= 15) {
          text = digits + std::string(static_cast<size_t>(n - k), '0') + ".0";
        } else if (0 < n && n <= 15) {
          text = digits.substr(0, static_cast<size_t>(n)) + "." + digits.substr(static_cast<size_t>(n));
        } else if (-4 < n && n <= 0) {
          text = "0." + std::string(static_cast<size_t>(-n), '0') + digits;
        } else {
          text = digits.substr(0, 1);
          if (k > 1) text += "." + digits.substr(1);
          const int e = n - 1;
          char ebuf[16];
          std::snprintf(ebuf, sizeof ebuf, "e%c%02d", e < 0 ? '-' : '+', e < 0 ? -e : e);
          text += ebuf;
        }
        if (negative) text.insert(text.begin(), '-');
        *out += text;
        return;
      }
      case Type::kString: DumpString(string_, out); return;
      case Type::kArray: {
        out->push_back('[');
        bool first = true;
        for (const Json& e : array_) {
          if (!first) out->push_back(',');
          first = false;
          e.Dump(out);
        }
        out->push_back(']');
        return;
      }
      case Type::kObject: {
        out->push_back('{');
        bool first = true;
        for (const auto& kv : object_) {
          if (!first) out->push_back(',');
          first = false;
          DumpString(kv.first, out);
          out->push_back(':');
          kv.second.Dump(out);
        }
        out->push_back('}');
        return;
      }
    }
  }

  Type type_ = Type::kNull;
  int64_t int_ = 0;
  double double_ = 0.0;
  std::string string_;
  array_t array_;
  object_t object_;
};

}  // namespace hip
}  // namespace open_spiel

#endif  // OSG_HOST_JSON_H_
