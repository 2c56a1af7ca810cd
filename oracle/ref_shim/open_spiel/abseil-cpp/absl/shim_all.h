// TEST INFRASTRUCTURE ONLY — part of the CPU checker, never of the product path.
//
// A private stand-in for the subset of abseil-cpp (pinned by the reference at 20250814.1,
// open_spiel/scripts/install.sh:20) that the hot-path files of /root/reference use.  abseil is
// not vendored in /root/reference and there is no network, so the genuine reference sources
// cannot be compiled against the real library; with this header on the include path
// (`-I oracle/ref_shim -I /root/reference`) they compile UNMODIFIED, from where they lie, into
// oracle/_ref/libspiel_ref.so (recipe: oracle/Makefile.ref).  Written from abseil's documented
// behaviour, not from its sources.  What it does NOT reproduce: abseil's random-number streams
// (absl::Uniform / BitGen draw from the std engines here) and its hash-table iteration order;
// neither is pinned by any reference test.
//
// Every `open_spiel/abseil-cpp/absl/**.h` path the reference includes is a one-line stub that
// includes this file.
#ifndef ORACLE_REF_SHIM_ABSL_ALL_H_
#define ORACLE_REF_SHIM_ABSL_ALL_H_

#include <algorithm>
#include <array>
#include <charconv>
#include <chrono>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <iterator>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <optional>
#include <ostream>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <string_view>
#include <system_error>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

// ---- base/attributes.h, base/thread_annotations.h, base/macros.h -----------------------------
#define ABSL_GUARDED_BY(x)
#define ABSL_PT_GUARDED_BY(x)
#define ABSL_EXCLUSIVE_LOCKS_REQUIRED(...)
#define ABSL_SHARED_LOCKS_REQUIRED(...)
#define ABSL_LOCKS_EXCLUDED(...)
#define ABSL_NO_THREAD_SAFETY_ANALYSIS
#define ABSL_MUST_USE_RESULT [[nodiscard]]
#define ABSL_ATTRIBUTE_UNUSED __attribute__((unused))
#define ABSL_ATTRIBUTE_NORETURN __attribute__((noreturn))
#define ABSL_ATTRIBUTE_ALWAYS_INLINE __attribute__((always_inline))
#define ABSL_ATTRIBUTE_NOINLINE __attribute__((noinline))
#define ABSL_DEPRECATED(msg) [[deprecated(msg)]]
#define ABSL_FALLTHROUGH_INTENDED [[fallthrough]]
#define ABSL_PREDICT_TRUE(x) (__builtin_expect(false || (x), true))
#define ABSL_PREDICT_FALSE(x) (__builtin_expect(false || (x), false))

namespace absl {

// ---- types/optional.h, strings/string_view.h, memory/memory.h -------------------------------
using string_view = std::string_view;
template <class T>
using optional = std::optional<T>;
using nullopt_t = std::nullopt_t;
inline constexpr nullopt_t nullopt = std::nullopt;
using std::make_optional;
using std::make_unique;
template <class T>
std::unique_ptr<T> WrapUnique(T* p) {
  return std::unique_ptr<T>(p);
}

// ---- types/span.h ----------------------------------------------------------------------------
template <class T>
class Span {
  using Mutable = std::remove_const_t<T>;
  template <class C>
  using DataOf = decltype(std::declval<C&>().data());
  template <class C>
  using EnableIfContainer =
      std::enable_if_t<std::is_convertible_v<std::remove_pointer_t<DataOf<C>> (*)[], T (*)[]> &&
                       std::is_integral_v<decltype(std::declval<C&>().size())>>;

 public:
  using element_type = T;
  using value_type = Mutable;
  using pointer = T*;
  using const_pointer = const T*;
  using reference = T&;
  using const_reference = const T&;
  using iterator = T*;
  using const_iterator = const T*;
  using reverse_iterator = std::reverse_iterator<iterator>;
  using size_type = size_t;
  using difference_type = ptrdiff_t;
  static constexpr size_type npos = ~size_type(0);

  constexpr Span() noexcept : p_(nullptr), n_(0) {}
  constexpr Span(T* p, size_type n) noexcept : p_(p), n_(n) {}
  template <size_t N>
  constexpr Span(T (&a)[N]) noexcept : p_(a), n_(N) {}
  // Mutable containers bind to Span<T>; const containers only to Span<const T>.
  template <class C, class = EnableIfContainer<C>,
            class = std::enable_if_t<!std::is_const_v<T>, C>>
  explicit Span(C& c) noexcept : p_(c.data()), n_(c.size()) {}
  template <class C, class = EnableIfContainer<const C>,
            class = std::enable_if_t<std::is_const_v<T>, C>>
  constexpr Span(const C& c) noexcept : p_(c.data()), n_(c.size()) {}
  template <class U = T, class = std::enable_if_t<std::is_const_v<U>>>
  Span(std::initializer_list<value_type> l) noexcept : p_(l.begin()), n_(l.size()) {}

  constexpr pointer data() const noexcept { return p_; }
  constexpr size_type size() const noexcept { return n_; }
  constexpr size_type length() const noexcept { return n_; }
  constexpr bool empty() const noexcept { return n_ == 0; }
  constexpr reference operator[](size_type i) const noexcept { return p_[i]; }
  constexpr reference at(size_type i) const {
    if (i >= n_) throw std::out_of_range("Span::at");
    return p_[i];
  }
  constexpr reference front() const noexcept { return p_[0]; }
  constexpr reference back() const noexcept { return p_[n_ - 1]; }
  constexpr iterator begin() const noexcept { return p_; }
  constexpr iterator end() const noexcept { return p_ + n_; }
  constexpr const_iterator cbegin() const noexcept { return p_; }
  constexpr const_iterator cend() const noexcept { return p_ + n_; }
  reverse_iterator rbegin() const noexcept { return reverse_iterator(end()); }
  reverse_iterator rend() const noexcept { return reverse_iterator(begin()); }
  void remove_prefix(size_type n) noexcept { p_ += n; n_ -= n; }
  void remove_suffix(size_type n) noexcept { n_ -= n; }
  constexpr Span subspan(size_type pos = 0, size_type len = npos) const {
    if (pos > n_) throw std::out_of_range("Span::subspan");
    return Span(p_ + pos, std::min(n_ - pos, len));
  }
  constexpr Span first(size_type n) const { return subspan(0, n); }
  constexpr Span last(size_type n) const { return subspan(n_ - n, n); }

 private:
  pointer p_;
  size_type n_;
};
template <class T>
bool operator==(Span<T> a, Span<T> b) {
  return std::equal(a.begin(), a.end(), b.begin(), b.end());
}
template <class T>
bool operator!=(Span<T> a, Span<T> b) {
  return !(a == b);
}
template <class T>
constexpr Span<T> MakeSpan(T* p, size_t n) noexcept {
  return Span<T>(p, n);
}
template <class T>
Span<T> MakeSpan(T* b, T* e) noexcept {
  return Span<T>(b, e - b);
}
template <class C>
constexpr auto MakeSpan(C& c) noexcept -> Span<std::remove_pointer_t<decltype(c.data())>> {
  return Span<std::remove_pointer_t<decltype(c.data())>>(c.data(), c.size());
}
template <class T, size_t N>
constexpr Span<T> MakeSpan(T (&a)[N]) noexcept {
  return Span<T>(a, N);
}
template <class T>
constexpr Span<const T> MakeConstSpan(T* p, size_t n) noexcept {
  return Span<const T>(p, n);
}
template <class C>
constexpr auto MakeConstSpan(const C& c) noexcept
    -> Span<const std::remove_pointer_t<decltype(c.data())>> {
  return Span<const std::remove_pointer_t<decltype(c.data())>>(c.data(), c.size());
}

// ---- container/*.h ---------------------------------------------------------------------------
template <class T, size_t N, class A = std::allocator<T>>
using InlinedVector = std::vector<T, A>;

namespace shim_internal {
inline void HashMix(size_t& seed, size_t v) {
  seed ^= v + 0x9e3779b97f4a7c15ULL + (seed << 6) + (seed >> 2);
}
template <class T, class = void>
struct Hash : std::hash<T> {};
template <class A, class B>
struct Hash<std::pair<A, B>> {
  size_t operator()(const std::pair<A, B>& p) const {
    size_t s = Hash<A>()(p.first);
    HashMix(s, Hash<B>()(p.second));
    return s;
  }
};
template <class T, class A>
struct Hash<std::vector<T, A>> {
  size_t operator()(const std::vector<T, A>& v) const {
    size_t s = v.size();
    for (const auto& x : v) HashMix(s, Hash<T>()(x));
    return s;
  }
};
template <class T>
struct Hash<T, std::enable_if_t<std::is_enum_v<T>>> {
  size_t operator()(T v) const { return std::hash<std::underlying_type_t<T>>()(static_cast<std::underlying_type_t<T>>(v)); }
};
}  // namespace shim_internal
template <class T>
using Hash = shim_internal::Hash<T>;

template <class K, class V, class H = Hash<K>, class E = std::equal_to<K>>
struct flat_hash_map : std::unordered_map<K, V, H, E> {
  using Base = std::unordered_map<K, V, H, E>;
  using Base::Base;
  flat_hash_map() = default;
  template <class K2>
  bool contains(const K2& k) const { return this->find(k) != this->end(); }
};
template <class K, class V, class H = Hash<K>, class E = std::equal_to<K>>
struct node_hash_map : std::unordered_map<K, V, H, E> {
  using Base = std::unordered_map<K, V, H, E>;
  using Base::Base;
  node_hash_map() = default;
  template <class K2>
  bool contains(const K2& k) const { return this->find(k) != this->end(); }
};
template <class K, class H = Hash<K>, class E = std::equal_to<K>>
struct flat_hash_set : std::unordered_set<K, H, E> {
  using Base = std::unordered_set<K, H, E>;
  using Base::Base;
  flat_hash_set() = default;
  template <class K2>
  bool contains(const K2& k) const { return this->find(k) != this->end(); }
};
template <class K, class H = Hash<K>, class E = std::equal_to<K>>
struct node_hash_set : std::unordered_set<K, H, E> {
  using Base = std::unordered_set<K, H, E>;
  using Base::Base;
  node_hash_set() = default;
  template <class K2>
  bool contains(const K2& k) const { return this->find(k) != this->end(); }
};
template <class K, class V, class C = std::less<K>>
struct btree_map : std::map<K, V, C> {
  using Base = std::map<K, V, C>;
  using Base::Base;
  btree_map() = default;
  template <class K2>
  bool contains(const K2& k) const { return this->find(k) != this->end(); }
};
template <class K, class C = std::less<K>>
struct btree_set : std::set<K, C> {
  using Base = std::set<K, C>;
  using Base::Base;
  btree_set() = default;
  template <class K2>
  bool contains(const K2& k) const { return this->find(k) != this->end(); }
};

// ---- algorithm/container.h -------------------------------------------------------------------
template <class C, class T>
auto c_find(C& c, const T& v) { return std::find(std::begin(c), std::end(c), v); }
template <class C, class P>
auto c_find_if(C& c, P&& p) { return std::find_if(std::begin(c), std::end(c), std::forward<P>(p)); }
template <class C, class T>
bool c_linear_search(const C& c, const T& v) { return std::find(std::begin(c), std::end(c), v) != std::end(c); }
template <class C, class T>
auto c_count(const C& c, const T& v) { return std::count(std::begin(c), std::end(c), v); }
template <class C, class P>
auto c_count_if(const C& c, P&& p) { return std::count_if(std::begin(c), std::end(c), std::forward<P>(p)); }
template <class C, class P>
bool c_all_of(const C& c, P&& p) { return std::all_of(std::begin(c), std::end(c), std::forward<P>(p)); }
template <class C, class P>
bool c_any_of(const C& c, P&& p) { return std::any_of(std::begin(c), std::end(c), std::forward<P>(p)); }
template <class C, class P>
bool c_none_of(const C& c, P&& p) { return std::none_of(std::begin(c), std::end(c), std::forward<P>(p)); }
template <class C, class F>
F c_for_each(C&& c, F&& f) { return std::for_each(std::begin(c), std::end(c), std::forward<F>(f)); }
template <class C, class T>
void c_fill(C& c, const T& v) { std::fill(std::begin(c), std::end(c), v); }
template <class C, class T>
void c_iota(C& c, const T& v) { std::iota(std::begin(c), std::end(c), v); }
template <class C, class T>
std::decay_t<T> c_accumulate(const C& c, T&& init) { return std::accumulate(std::begin(c), std::end(c), std::forward<T>(init)); }
template <class C, class T, class Op>
std::decay_t<T> c_accumulate(const C& c, T&& init, Op&& op) { return std::accumulate(std::begin(c), std::end(c), std::forward<T>(init), std::forward<Op>(op)); }
template <class C>
void c_sort(C& c) { std::sort(std::begin(c), std::end(c)); }
template <class C, class L>
void c_sort(C& c, L&& l) { std::sort(std::begin(c), std::end(c), std::forward<L>(l)); }
template <class C>
void c_stable_sort(C& c) { std::stable_sort(std::begin(c), std::end(c)); }
template <class C, class L>
void c_stable_sort(C& c, L&& l) { std::stable_sort(std::begin(c), std::end(c), std::forward<L>(l)); }
template <class C>
void c_reverse(C& c) { std::reverse(std::begin(c), std::end(c)); }
template <class C, class G>
void c_shuffle(C& c, G&& g) { std::shuffle(std::begin(c), std::end(c), std::forward<G>(g)); }
template <class C>
auto c_max_element(C& c) { return std::max_element(std::begin(c), std::end(c)); }
template <class C, class L>
auto c_max_element(C& c, L&& l) { return std::max_element(std::begin(c), std::end(c), std::forward<L>(l)); }
template <class C>
auto c_min_element(C& c) { return std::min_element(std::begin(c), std::end(c)); }
template <class C, class L>
auto c_min_element(C& c, L&& l) { return std::min_element(std::begin(c), std::end(c), std::forward<L>(l)); }
template <class C1, class C2>
bool c_equal(const C1& a, const C2& b) { return std::equal(std::begin(a), std::end(a), std::begin(b), std::end(b)); }
template <class C, class O>
O c_copy(const C& c, O o) { return std::copy(std::begin(c), std::end(c), o); }
template <class C, class O, class F>
O c_transform(const C& c, O o, F&& f) { return std::transform(std::begin(c), std::end(c), o, std::forward<F>(f)); }
template <class C, class T>
bool c_binary_search(const C& c, const T& v) { return std::binary_search(std::begin(c), std::end(c), v); }
template <class C, class T>
auto c_lower_bound(C& c, const T& v) { return std::lower_bound(std::begin(c), std::end(c), v); }
template <class C, class T>
auto c_upper_bound(C& c, const T& v) { return std::upper_bound(std::begin(c), std::end(c), v); }
template <class C>
bool c_is_sorted(const C& c) { return std::is_sorted(std::begin(c), std::end(c)); }

// ---- strings/str_cat.h -----------------------------------------------------------------------
// AlphaNum: integers in decimal, floating point like printf("%g") (six significant digits).
class AlphaNum {
 public:
  AlphaNum(int v) { Int(v); }
  AlphaNum(unsigned v) { Int(v); }
  AlphaNum(long v) { Int(v); }
  AlphaNum(unsigned long v) { Int(v); }
  AlphaNum(long long v) { Int(v); }
  AlphaNum(unsigned long long v) { Int(v); }
  AlphaNum(float v) { Flt(v); }
  AlphaNum(double v) { Flt(v); }
  AlphaNum(const char* s) : piece_(s ? s : "") {}
  AlphaNum(string_view s) : piece_(s) {}
  AlphaNum(const std::string& s) : piece_(s) {}
  template <class E, class = std::enable_if_t<std::is_enum_v<E>>>
  AlphaNum(E v) { Int(static_cast<std::underlying_type_t<E>>(v)); }  // enums print as their integer value
  AlphaNum(char) = delete;
  AlphaNum(const AlphaNum&) = delete;
  AlphaNum& operator=(const AlphaNum&) = delete;
  string_view Piece() const { return piece_; }
  size_t size() const { return piece_.size(); }
  const char* data() const { return piece_.data(); }

 private:
  template <class I>
  void Int(I v) {
    auto r = std::to_chars(buf_, buf_ + sizeof(buf_), v);
    piece_ = string_view(buf_, r.ptr - buf_);
  }
  void Flt(double v) {
    int n = std::snprintf(buf_, sizeof(buf_), "%g", v);
    piece_ = string_view(buf_, n);
  }
  string_view piece_;
  char buf_[32];
};

namespace shim_internal {
inline std::string CatPieces(std::initializer_list<string_view> pieces) {
  size_t total = 0;
  for (string_view p : pieces) total += p.size();
  std::string out;
  out.reserve(total);
  for (string_view p : pieces) out.append(p.data(), p.size());
  return out;
}
inline void AppendPieces(std::string* dest, std::initializer_list<string_view> pieces) {
  size_t total = dest->size();
  for (string_view p : pieces) total += p.size();
  if (total > dest->capacity()) dest->reserve(std::max(total, 2 * dest->capacity()));
  for (string_view p : pieces) dest->append(p.data(), p.size());
}
}  // namespace shim_internal
// The AlphaNum temporaries live until the end of the full expression, i.e. through the call.
template <class... A>
std::string StrCat(const A&... a) {
  return shim_internal::CatPieces({AlphaNum(a).Piece()...});
}
template <class... A>
void StrAppend(std::string* dest, const A&... a) {
  shim_internal::AppendPieces(dest, {AlphaNum(a).Piece()...});
}

// ---- strings/ascii.h, match.h, str_replace.h, numbers.h, charconv.h --------------------------
inline string_view StripAsciiWhitespace(string_view s) {
  size_t b = 0, e = s.size();
  while (b < e && std::isspace(static_cast<unsigned char>(s[b]))) ++b;
  while (e > b && std::isspace(static_cast<unsigned char>(s[e - 1]))) --e;
  return s.substr(b, e - b);
}
inline void StripAsciiWhitespace(std::string* s) { *s = std::string(StripAsciiWhitespace(string_view(*s))); }
inline string_view StripLeadingAsciiWhitespace(string_view s) {
  size_t b = 0;
  while (b < s.size() && std::isspace(static_cast<unsigned char>(s[b]))) ++b;
  return s.substr(b);
}
inline string_view StripTrailingAsciiWhitespace(string_view s) {
  size_t e = s.size();
  while (e > 0 && std::isspace(static_cast<unsigned char>(s[e - 1]))) --e;
  return s.substr(0, e);
}
inline std::string AsciiStrToLower(string_view s) {
  std::string r(s);
  for (char& c : r) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  return r;
}
inline std::string AsciiStrToUpper(string_view s) {
  std::string r(s);
  for (char& c : r) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  return r;
}
inline bool ascii_isdigit(unsigned char c) { return c >= '0' && c <= '9'; }
inline bool ascii_isspace(unsigned char c) { return std::isspace(c) != 0; }
inline bool ascii_isalpha(unsigned char c) { return std::isalpha(c) != 0; }
inline bool ascii_isalnum(unsigned char c) { return std::isalnum(c) != 0; }
inline char ascii_tolower(unsigned char c) { return static_cast<char>(std::tolower(c)); }
inline char ascii_toupper(unsigned char c) { return static_cast<char>(std::toupper(c)); }

inline bool StartsWith(string_view s, string_view p) { return s.size() >= p.size() && s.compare(0, p.size(), p) == 0; }
inline bool EndsWith(string_view s, string_view p) { return s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0; }
inline bool StrContains(string_view s, string_view p) { return s.find(p) != string_view::npos; }
inline bool StrContains(string_view s, char c) { return s.find(c) != string_view::npos; }

inline std::string StrReplaceAll(string_view s, std::initializer_list<std::pair<string_view, string_view>> reps) {
  std::string out;
  size_t i = 0;
  while (i < s.size()) {
    // leftmost match; among matches at the same position the longest pattern wins
    const std::pair<string_view, string_view>* best = nullptr;
    for (const auto& r : reps)
      if (!r.first.empty() && s.compare(i, r.first.size(), r.first) == 0 &&
          (!best || r.first.size() > best->first.size()))
        best = &r;
    if (best) {
      out.append(best->second);
      i += best->first.size();
    } else {
      out.push_back(s[i++]);
    }
  }
  return out;
}

template <class I>
bool SimpleAtoi(string_view s, I* out) {
  static_assert(std::is_integral_v<I>, "SimpleAtoi needs an integer type");
  s = StripAsciiWhitespace(s);
  if (!s.empty() && s[0] == '+') {
    s.remove_prefix(1);
    if (!s.empty() && (s[0] == '-' || s[0] == '+')) return false;
  }
  if (s.empty()) return false;
  I v{};
  auto r = std::from_chars(s.data(), s.data() + s.size(), v, 10);
  if (r.ec != std::errc() || r.ptr != s.data() + s.size()) return false;
  *out = v;
  return true;
}
inline bool SimpleAtod(string_view s, double* out) {
  s = StripAsciiWhitespace(s);
  if (s.empty()) return false;
  std::string z(s);
  char* end = nullptr;
  double v = std::strtod(z.c_str(), &end);
  if (end != z.c_str() + z.size()) return false;
  *out = v;
  return true;
}
inline bool SimpleAtof(string_view s, float* out) {
  double d;
  if (!SimpleAtod(s, &d)) return false;
  *out = static_cast<float>(d);
  return true;
}
inline bool SimpleAtob(string_view s, bool* out) {
  std::string l = AsciiStrToLower(s);
  if (l == "true" || l == "t" || l == "yes" || l == "y" || l == "1") { *out = true; return true; }
  if (l == "false" || l == "f" || l == "no" || l == "n" || l == "0") { *out = false; return true; }
  return false;
}

enum class chars_format { scientific = 1, fixed = 2, hex = 4, general = fixed | scientific };
struct from_chars_result {
  const char* ptr;
  std::errc ec;
};
// Decimal and (with or without a 0x prefix) hexadecimal floating point, like strtod.
inline from_chars_result from_chars(const char* first, const char* last, double& value,
                                    chars_format = chars_format::general) {
  std::string z(first, last);
  char* end = nullptr;
  double v = std::strtod(z.c_str(), &end);
  if (end == z.c_str()) return {first, std::errc::invalid_argument};
  value = v;
  return {first + (end - z.c_str()), std::errc()};
}
inline from_chars_result from_chars(const char* first, const char* last, float& value,
                                    chars_format f = chars_format::general) {
  double d = 0;
  from_chars_result r = from_chars(first, last, d, f);
  if (r.ec == std::errc()) value = static_cast<float>(d);
  return r;
}

// ---- strings/str_join.h ----------------------------------------------------------------------
struct AlphaNumFormatterImpl {
  template <class T>
  void operator()(std::string* out, const T& v) const { StrAppend(out, v); }
  void operator()(std::string* out, const std::string& v) const { out->append(v); }
  void operator()(std::string* out, string_view v) const { out->append(v.data(), v.size()); }
  void operator()(std::string* out, const char* v) const { out->append(v); }
  void operator()(std::string* out, bool v) const { out->push_back(v ? '1' : '0'); }
};
inline AlphaNumFormatterImpl AlphaNumFormatter() { return {}; }
struct StreamFormatterImpl {
  template <class T>
  void operator()(std::string* out, const T& v) const {
    std::ostringstream os;
    os << v;
    out->append(os.str());
  }
};
inline StreamFormatterImpl StreamFormatter() { return {}; }
template <class F1, class F2>
struct PairFormatterImpl {
  F1 f1;
  std::string sep;
  F2 f2;
  template <class P>
  void operator()(std::string* out, const P& p) const {
    f1(out, p.first);
    out->append(sep);
    f2(out, p.second);
  }
};
template <class F1, class F2>
PairFormatterImpl<F1, F2> PairFormatter(F1 f1, string_view sep, F2 f2) {
  return {std::move(f1), std::string(sep), std::move(f2)};
}
inline PairFormatterImpl<AlphaNumFormatterImpl, AlphaNumFormatterImpl> PairFormatter(string_view sep) {
  return {{}, std::string(sep), {}};
}
template <class F>
struct DereferenceFormatterImpl {
  F f;
  template <class P>
  void operator()(std::string* out, const P& p) const { f(out, *p); }
};
template <class F>
DereferenceFormatterImpl<F> DereferenceFormatter(F f) { return {std::move(f)}; }
inline DereferenceFormatterImpl<AlphaNumFormatterImpl> DereferenceFormatter() { return {}; }

template <class It, class F>
std::string StrJoin(It b, It e, string_view sep, F&& f) {
  std::string out;
  bool first = true;
  for (; b != e; ++b) {
    if (!first) out.append(sep.data(), sep.size());
    first = false;
    f(&out, *b);
  }
  return out;
}
template <class It, class = typename std::iterator_traits<It>::iterator_category>
std::string StrJoin(It b, It e, string_view sep) { return StrJoin(b, e, sep, AlphaNumFormatterImpl()); }
template <class R, class F, class = decltype(std::begin(std::declval<const R&>())),
          class = std::enable_if_t<!std::is_convertible_v<F, string_view>>>
std::string StrJoin(const R& r, string_view sep, F&& f) {
  return StrJoin(std::begin(r), std::end(r), sep, std::forward<F>(f));
}
template <class R, class = decltype(std::begin(std::declval<const R&>()))>
std::string StrJoin(const R& r, string_view sep) {
  return StrJoin(std::begin(r), std::end(r), sep, AlphaNumFormatterImpl());
}
template <class T>
std::string StrJoin(std::initializer_list<T> l, string_view sep) {
  return StrJoin(l.begin(), l.end(), sep, AlphaNumFormatterImpl());
}
template <class T, class F>
std::string StrJoin(std::initializer_list<T> l, string_view sep, F&& f) {
  return StrJoin(l.begin(), l.end(), sep, std::forward<F>(f));
}
template <class... T>
std::string StrJoin(const std::tuple<T...>& t, string_view sep) {
  std::string out;
  bool first = true;
  std::apply([&](const auto&... v) {
    ((out.append(first ? "" : std::string(sep)), first = false, AlphaNumFormatterImpl()(&out, v)), ...);
  }, t);
  return out;
}

// ---- strings/str_split.h ---------------------------------------------------------------------
struct ByString {
  std::string d;
  explicit ByString(string_view s) : d(s) {}
  // position and length of the next delimiter at or after pos (npos: none)
  std::pair<size_t, size_t> Find(string_view t, size_t pos) const {
    if (d.empty()) return pos + 1 < t.size() ? std::make_pair(pos + 1, size_t(0)) : std::make_pair(string_view::npos, size_t(0));
    size_t p = t.find(d, pos);
    return {p, d.size()};
  }
};
struct ByChar {
  char c;
  explicit ByChar(char ch) : c(ch) {}
  std::pair<size_t, size_t> Find(string_view t, size_t pos) const { return {t.find(c, pos), 1}; }
};
struct ByAnyChar {
  std::string set;
  explicit ByAnyChar(string_view s) : set(s) {}
  std::pair<size_t, size_t> Find(string_view t, size_t pos) const { return {t.find_first_of(set, pos), 1}; }
};
namespace shim_internal {
inline ByChar ToDelimiter(char c) { return ByChar(c); }
inline ByString ToDelimiter(const char* s) { return ByString(s); }
inline ByString ToDelimiter(const std::string& s) { return ByString(s); }
inline ByString ToDelimiter(string_view s) { return ByString(s); }
inline ByString ToDelimiter(ByString d) { return d; }
inline ByChar ToDelimiter(ByChar d) { return d; }
inline ByAnyChar ToDelimiter(ByAnyChar d) { return d; }
template <class D>
struct MaxSplitsImpl {
  D d;
  int limit;
  mutable int used = 0;
  std::pair<size_t, size_t> Find(string_view t, size_t pos) const {
    if (used >= limit) return {string_view::npos, 0};
    ++used;
    return d.Find(t, pos);
  }
};
template <class D>
MaxSplitsImpl<D> ToDelimiter(MaxSplitsImpl<D> d) { return d; }
template <class T>
struct IsInitList : std::false_type {};
template <class T>
struct IsInitList<std::initializer_list<T>> : std::true_type {};
}  // namespace shim_internal
template <class D>
auto MaxSplits(D d, int limit) {
  auto inner = shim_internal::ToDelimiter(d);
  return shim_internal::MaxSplitsImpl<decltype(inner)>{inner, limit};
}
struct AllowEmpty {
  bool operator()(string_view) const { return true; }
};
struct SkipEmpty {
  bool operator()(string_view s) const { return !s.empty(); }
};
struct SkipWhitespace {
  bool operator()(string_view s) const { return !StripAsciiWhitespace(s).empty(); }
};

class Splitter {
 public:
  template <class D, class P>
  Splitter(string_view text, std::string owned, bool own, const D& delim, P pred) : owned_(std::move(owned)) {
    if (own) text = owned_;
    size_t pos = 0;
    for (;;) {
      auto [at, len] = delim.Find(text, pos);
      if (at == string_view::npos) {
        if (pred(text.substr(pos))) parts_.push_back(text.substr(pos));
        break;
      }
      if (pred(text.substr(pos, at - pos))) parts_.push_back(text.substr(pos, at - pos));
      pos = at + len;
    }
  }
  // The pieces point into owned_ when the input was a temporary string: fix them up on copy / move.
  Splitter(const Splitter& o) : owned_(o.owned_), parts_(o.parts_) { Rebase(o); }
  Splitter(Splitter&& o) noexcept : parts_(o.parts_) {
    const char* old = o.owned_.data();
    size_t n = o.owned_.size();
    owned_ = std::move(o.owned_);
    RebaseFrom(old, n);
  }
  using const_iterator = std::vector<string_view>::const_iterator;
  const_iterator begin() const { return parts_.begin(); }
  const_iterator end() const { return parts_.end(); }

  template <class C, class V = typename C::value_type,
            class = std::enable_if_t<!shim_internal::IsInitList<C>::value &&
                                     std::is_constructible_v<V, string_view>>>
  operator C() const {  // NOLINT: implicit by design, as in the library
    C c;
    for (string_view p : parts_) c.insert(c.end(), V(p));
    return c;
  }
  template <class A, class B>
  operator std::pair<A, B>() const {  // NOLINT
    return std::pair<A, B>(parts_.size() > 0 ? A(parts_[0]) : A(), parts_.size() > 1 ? B(parts_[1]) : B());
  }

 private:
  void Rebase(const Splitter& o) { RebaseFrom(o.owned_.data(), o.owned_.size()); }
  void RebaseFrom(const char* old, size_t n) {
    if (n == 0) return;
    for (string_view& p : parts_)
      if (p.data() >= old && p.data() <= old + n) p = string_view(owned_.data() + (p.data() - old), p.size());
  }
  std::string owned_;
  std::vector<string_view> parts_;
};
namespace shim_internal {
// Temporaries of std::string are copied (the pieces must outlive the full expression in
// range-for statements); everything else is viewed in place.
template <class T>
constexpr bool kOwnText = std::is_same_v<std::decay_t<T>, std::string> && !std::is_lvalue_reference_v<T>;
}  // namespace shim_internal
template <class T, class D, class P>
Splitter StrSplit(T&& text, D delim, P pred) {
  if constexpr (shim_internal::kOwnText<T&&>)
    return Splitter(string_view(), std::string(std::forward<T>(text)), true, shim_internal::ToDelimiter(delim), pred);
  else
    return Splitter(string_view(text), std::string(), false, shim_internal::ToDelimiter(delim), pred);
}
template <class T, class D>
Splitter StrSplit(T&& text, D delim) {
  return StrSplit(std::forward<T>(text), delim, AllowEmpty());
}

// ---- strings/str_format.h --------------------------------------------------------------------
namespace shim_internal {
struct FormatArg {
  enum Kind { kInt, kUInt, kDouble, kStr, kPtr, kChar } kind;
  long long i = 0;
  unsigned long long u = 0;
  double d = 0;
  string_view s;
  const void* p = nullptr;
  FormatArg(bool v) : kind(kInt), i(v) {}
  FormatArg(char v) : kind(kChar), i(v) {}
  FormatArg(signed char v) : kind(kInt), i(v) {}
  FormatArg(unsigned char v) : kind(kUInt), u(v) {}
  FormatArg(short v) : kind(kInt), i(v) {}
  FormatArg(unsigned short v) : kind(kUInt), u(v) {}
  FormatArg(int v) : kind(kInt), i(v) {}
  FormatArg(unsigned v) : kind(kUInt), u(v) {}
  FormatArg(long v) : kind(kInt), i(v) {}
  FormatArg(unsigned long v) : kind(kUInt), u(v) {}
  FormatArg(long long v) : kind(kInt), i(v) {}
  FormatArg(unsigned long long v) : kind(kUInt), u(v) {}
  FormatArg(float v) : kind(kDouble), d(v) {}
  FormatArg(double v) : kind(kDouble), d(v) {}
  FormatArg(long double v) : kind(kDouble), d(static_cast<double>(v)) {}
  FormatArg(const char* v) : kind(kStr), s(v ? v : "(null)") {}
  FormatArg(const std::string& v) : kind(kStr), s(v) {}
  FormatArg(string_view v) : kind(kStr), s(v) {}
  template <class T, class = std::enable_if_t<!std::is_same_v<std::remove_cv_t<T>, char>>>
  FormatArg(T* v) : kind(kPtr), p(v) {}
  template <class E, class = std::enable_if_t<std::is_enum_v<E>>, class = void>
  FormatArg(E v) : kind(kInt), i(static_cast<long long>(v)) {}
};
inline std::string FormatImpl(string_view fmt, const FormatArg* args, size_t nargs) {
  std::string out;
  size_t next = 0;
  char buf[512];
  for (size_t k = 0; k < fmt.size(); ++k) {
    if (fmt[k] != '%') { out.push_back(fmt[k]); continue; }
    if (k + 1 < fmt.size() && fmt[k + 1] == '%') { out.push_back('%'); ++k; continue; }
    std::string spec = "%";
    size_t j = k + 1;
    auto take_star = [&]() {
      long long v = next < nargs ? (args[next].kind == FormatArg::kUInt ? static_cast<long long>(args[next].u) : args[next].i) : 0;
      ++next;
      spec += std::to_string(v);
    };
    while (j < fmt.size() && std::strchr("-+ #0", fmt[j])) spec.push_back(fmt[j++]);
    if (j < fmt.size() && fmt[j] == '*') { take_star(); ++j; }
    while (j < fmt.size() && std::isdigit(static_cast<unsigned char>(fmt[j]))) spec.push_back(fmt[j++]);
    if (j < fmt.size() && fmt[j] == '.') {
      spec.push_back(fmt[j++]);
      if (j < fmt.size() && fmt[j] == '*') { take_star(); ++j; }
      while (j < fmt.size() && std::isdigit(static_cast<unsigned char>(fmt[j]))) spec.push_back(fmt[j++]);
    }
    while (j < fmt.size() && std::strchr("hlqjztL", fmt[j])) ++j;  // length modifiers carry no information here
    if (j >= fmt.size()) break;
    char conv = fmt[j];
    k = j;
    if (next >= nargs) { out.append("<missing arg>"); continue; }
    const FormatArg& a = args[next++];
    if (conv == 'v') conv = a.kind == FormatArg::kDouble ? 'g' : a.kind == FormatArg::kStr ? 's' : a.kind == FormatArg::kUInt ? 'u' : a.kind == FormatArg::kChar ? 'c' : a.kind == FormatArg::kPtr ? 'p' : 'd';
    int n = 0;
    switch (conv) {
      case 'd': case 'i':
        if (a.kind == FormatArg::kUInt) { spec += "llu"; n = std::snprintf(buf, sizeof buf, spec.c_str(), a.u); }
        else if (a.kind == FormatArg::kDouble) { spec += "lld"; n = std::snprintf(buf, sizeof buf, spec.c_str(), static_cast<long long>(a.d)); }
        else { spec += "lld"; n = std::snprintf(buf, sizeof buf, spec.c_str(), a.i); }
        break;
      case 'u': case 'o': case 'x': case 'X':
        spec += "ll"; spec.push_back(conv);
        n = std::snprintf(buf, sizeof buf, spec.c_str(), a.kind == FormatArg::kUInt ? a.u : static_cast<unsigned long long>(a.i));
        break;
      case 'c':
        spec.push_back('c');
        n = std::snprintf(buf, sizeof buf, spec.c_str(), static_cast<int>(a.kind == FormatArg::kUInt ? static_cast<long long>(a.u) : a.i));
        break;
      case 'f': case 'F': case 'e': case 'E': case 'g': case 'G': case 'a': case 'A': {
        spec.push_back(conv);
        double v = a.kind == FormatArg::kDouble ? a.d : a.kind == FormatArg::kUInt ? static_cast<double>(a.u) : static_cast<double>(a.i);
        n = std::snprintf(buf, sizeof buf, spec.c_str(), v);
        if (n >= static_cast<int>(sizeof buf)) {
          std::string big(n + 1, '\0');
          std::snprintf(big.data(), big.size(), spec.c_str(), v);
          out.append(big.data(), n);
          continue;
        }
        break;
      }
      case 's': {
        // width / precision applied by hand: the piece is not NUL-terminated
        std::string piece = a.kind == FormatArg::kStr ? std::string(a.s)
                            : a.kind == FormatArg::kDouble ? std::to_string(a.d)
                            : a.kind == FormatArg::kUInt ? std::to_string(a.u) : std::to_string(a.i);
        bool left = spec.find('-') != std::string::npos;
        size_t dot = spec.find('.');
        if (dot != std::string::npos) piece = piece.substr(0, std::strtoul(spec.c_str() + dot + 1, nullptr, 10));
        size_t wb = 1;
        while (wb < spec.size() && std::strchr("-+ #0", spec[wb])) ++wb;
        size_t width = std::strtoul(spec.c_str() + wb, nullptr, 10);
        if (piece.size() < width) piece.insert(left ? piece.size() : 0, width - piece.size(), ' ');
        out.append(piece);
        continue;
      }
      case 'p':
        n = std::snprintf(buf, sizeof buf, "%p", a.p);
        break;
      default:
        out.append("<bad conversion>");
        continue;
    }
    if (n > 0) out.append(buf, std::min<size_t>(n, sizeof buf - 1));
  }
  return out;
}
}  // namespace shim_internal
template <class... A>
std::string StrFormat(string_view fmt, const A&... a) {
  const shim_internal::FormatArg args[] = {shim_internal::FormatArg(0), shim_internal::FormatArg(a)...};
  return shim_internal::FormatImpl(fmt, args + 1, sizeof...(A));
}
template <class... A>
std::string StreamFormat(string_view fmt, const A&... a) { return StrFormat(fmt, a...); }
template <class... A>
void StrAppendFormat(std::string* dst, string_view fmt, const A&... a) { dst->append(StrFormat(fmt, a...)); }
template <class... A>
int PrintF(string_view fmt, const A&... a) {
  std::string s = StrFormat(fmt, a...);
  return static_cast<int>(std::fwrite(s.data(), 1, s.size(), stdout));
}
template <class... A>
int FPrintF(std::FILE* f, string_view fmt, const A&... a) {
  std::string s = StrFormat(fmt, a...);
  return static_cast<int>(std::fwrite(s.data(), 1, s.size(), f));
}

// ---- time/time.h, time/clock.h ---------------------------------------------------------------
class Duration {
 public:
  constexpr Duration() : ns_(0) {}
  constexpr explicit Duration(int64_t ns) : ns_(ns) {}
  constexpr int64_t ns() const { return ns_; }
  Duration& operator+=(Duration o) { ns_ += o.ns_; return *this; }
  Duration& operator-=(Duration o) { ns_ -= o.ns_; return *this; }
 private:
  int64_t ns_;
};
constexpr Duration operator+(Duration a, Duration b) { return Duration(a.ns() + b.ns()); }
constexpr Duration operator-(Duration a, Duration b) { return Duration(a.ns() - b.ns()); }
constexpr bool operator<(Duration a, Duration b) { return a.ns() < b.ns(); }
constexpr bool operator>(Duration a, Duration b) { return a.ns() > b.ns(); }
constexpr bool operator<=(Duration a, Duration b) { return a.ns() <= b.ns(); }
constexpr bool operator>=(Duration a, Duration b) { return a.ns() >= b.ns(); }
constexpr bool operator==(Duration a, Duration b) { return a.ns() == b.ns(); }
constexpr bool operator!=(Duration a, Duration b) { return a.ns() != b.ns(); }
template <class T> constexpr Duration Nanoseconds(T n) { return Duration(static_cast<int64_t>(n)); }
template <class T> constexpr Duration Microseconds(T n) { return Duration(static_cast<int64_t>(n * 1000)); }
template <class T> constexpr Duration Milliseconds(T n) { return Duration(static_cast<int64_t>(n * 1000000)); }
template <class T> constexpr Duration Seconds(T n) { return Duration(static_cast<int64_t>(n * 1000000000LL)); }
constexpr Duration ZeroDuration() { return Duration(); }
inline double ToDoubleSeconds(Duration d) { return d.ns() * 1e-9; }
inline double ToDoubleMilliseconds(Duration d) { return d.ns() * 1e-6; }
inline int64_t ToInt64Nanoseconds(Duration d) { return d.ns(); }
inline int64_t ToInt64Microseconds(Duration d) { return d.ns() / 1000; }
inline int64_t ToInt64Milliseconds(Duration d) { return d.ns() / 1000000; }
inline int64_t ToInt64Seconds(Duration d) { return d.ns() / 1000000000LL; }
class Time {
 public:
  constexpr Time() : ns_(0) {}
  constexpr explicit Time(int64_t ns_since_epoch) : ns_(ns_since_epoch) {}
  constexpr int64_t ns() const { return ns_; }
 private:
  int64_t ns_;
};
constexpr Duration operator-(Time a, Time b) { return Duration(a.ns() - b.ns()); }
constexpr Time operator+(Time a, Duration d) { return Time(a.ns() + d.ns()); }
constexpr Time operator-(Time a, Duration d) { return Time(a.ns() - d.ns()); }
constexpr bool operator<(Time a, Time b) { return a.ns() < b.ns(); }
constexpr bool operator>(Time a, Time b) { return a.ns() > b.ns(); }
constexpr bool operator<=(Time a, Time b) { return a.ns() <= b.ns(); }
constexpr bool operator>=(Time a, Time b) { return a.ns() >= b.ns(); }
constexpr bool operator==(Time a, Time b) { return a.ns() == b.ns(); }
constexpr Time UnixEpoch() { return Time(); }
inline Time Now() {
  return Time(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count());
}
inline int64_t ToUnixNanos(Time t) { return t.ns(); }
inline int64_t ToUnixMicros(Time t) { return t.ns() / 1000; }
inline int64_t ToUnixMillis(Time t) { return t.ns() / 1000000; }
inline int64_t ToUnixSeconds(Time t) { return t.ns() / 1000000000LL; }

// ---- synchronization/mutex.h -----------------------------------------------------------------
class Mutex {
 public:
  void Lock() { m_.lock(); }
  void Unlock() { m_.unlock(); }
  void lock() { m_.lock(); }
  void unlock() { m_.unlock(); }
  bool TryLock() { return m_.try_lock(); }
  void ReaderLock() { m_.lock(); }
  void ReaderUnlock() { m_.unlock(); }
 private:
  std::mutex m_;
};
class MutexLock {
 public:
  explicit MutexLock(Mutex* m) : m_(m) { m_->Lock(); }
  explicit MutexLock(Mutex& m) : m_(&m) { m_->Lock(); }
  ~MutexLock() { m_->Unlock(); }
  MutexLock(const MutexLock&) = delete;
  MutexLock& operator=(const MutexLock&) = delete;
 private:
  Mutex* m_;
};
using ReaderMutexLock = MutexLock;
using WriterMutexLock = MutexLock;

// ---- random/*.h ------------------------------------------------------------------------------
// NOT abseil's streams: a std engine behind the same interface.
class BitGen {
 public:
  using result_type = uint64_t;
  BitGen() {
    std::random_device rd;
    e_.seed((static_cast<uint64_t>(rd()) << 32) ^ rd());
  }
  template <class SeedSeq, class = std::enable_if_t<!std::is_same_v<std::decay_t<SeedSeq>, BitGen>>>
  explicit BitGen(SeedSeq&& s) : e_(s) {}
  static constexpr result_type min() { return 0; }
  static constexpr result_type max() { return ~result_type(0); }
  result_type operator()() { return e_(); }
 private:
  std::mt19937_64 e_;
};
using InsecureBitGen = BitGen;
namespace shim_internal {
// One engine output -> a double in [0, 1): 53 bits from a 64-bit engine, 27 bits from a 32-bit
// one (std::mt19937).  The 27-bit convention is the restatement's (spiel_oracle_algos.cpp), chosen
// so that MCTSBot searches of the two builds can be compared node for node; abseil's own
// conversion is different and, like its streams, pinned by no reference test.
template <class URBG>
double UnitFrom(URBG& g) {
  constexpr auto kRange = static_cast<uint64_t>(URBG::max()) - static_cast<uint64_t>(URBG::min());
  if constexpr (kRange == ~uint64_t(0)) {
    return static_cast<double>(static_cast<uint64_t>(g() - URBG::min()) >> 11) * (1.0 / 9007199254740992.0);
  } else if constexpr (kRange == 0xFFFFFFFFULL) {
    return static_cast<double>(static_cast<uint32_t>(g() - URBG::min()) >> 5) * (1.0 / 134217728.0);
  } else {
    return std::generate_canonical<double, 53>(g);
  }
}
}  // namespace shim_internal
class BitGenRef {
 public:
  using result_type = uint64_t;
  template <class URBG, class = std::enable_if_t<!std::is_same_v<std::decay_t<URBG>, BitGenRef>>>
  BitGenRef(URBG& g) : p_(&g), call_(&Call<URBG>), unit_(&Unit<URBG>) {}  // NOLINT: implicit, as in the library
  static constexpr result_type min() { return 0; }
  static constexpr result_type max() { return ~result_type(0); }
  result_type operator()() { return call_(p_); }
  double UnitDraw() { return unit_(p_); }
 private:
  template <class URBG>
  static uint64_t Call(void* p) {
    URBG& g = *static_cast<URBG*>(p);
    if constexpr (static_cast<uint64_t>(URBG::max()) - static_cast<uint64_t>(URBG::min()) == ~uint64_t(0)) {
      return static_cast<uint64_t>(g() - URBG::min());
    } else {
      uint64_t hi = static_cast<uint64_t>(g() - URBG::min());
      uint64_t lo = static_cast<uint64_t>(g() - URBG::min());
      return (hi << 32) ^ lo;
    }
  }
  template <class URBG>
  static double Unit(void* p) { return shim_internal::UnitFrom(*static_cast<URBG*>(p)); }
  void* p_;
  uint64_t (*call_)(void*);
  double (*unit_)(void*);
};
namespace shim_internal {
inline double UnitFromRef(BitGenRef& g) { return g.UnitDraw(); }
}  // namespace shim_internal
template <class T = int>
using uniform_int_distribution = std::uniform_int_distribution<T>;
template <class T = double>
using uniform_real_distribution = std::uniform_real_distribution<T>;
template <class T = int>
using discrete_distribution = std::discrete_distribution<T>;
using bernoulli_distribution = std::bernoulli_distribution;
template <class T = double>
using gaussian_distribution = std::normal_distribution<T>;
template <class T = double>
using exponential_distribution = std::exponential_distribution<T>;

namespace shim_internal {
template <class R, class A, class B>
using UniformResult = std::conditional_t<std::is_void_v<R>, std::common_type_t<A, B>, R>;
}
// Uniform(gen, lo, hi): [lo, hi) for both integers and reals.
template <class R = void, class URBG, class A, class B>
shim_internal::UniformResult<R, A, B> Uniform(URBG&& gen, A lo, B hi) {
  using T = shim_internal::UniformResult<R, A, B>;
  if constexpr (std::is_floating_point_v<T>) {
    double u;
    if constexpr (std::is_same_v<std::decay_t<URBG>, BitGenRef>) u = gen.UnitDraw();
    else u = shim_internal::UnitFrom(gen);
    return static_cast<T>(static_cast<T>(lo) + (static_cast<T>(hi) - static_cast<T>(lo)) * static_cast<T>(u));
  } else {
    T l = static_cast<T>(lo), h = static_cast<T>(hi);
    if (!(l < h)) return l;
    return std::uniform_int_distribution<T>(l, static_cast<T>(h - 1))(gen);
  }
}
template <class URBG>
bool Bernoulli(URBG&& gen, double p) { return std::bernoulli_distribution(p)(gen); }
template <class R = double, class URBG>
R Gaussian(URBG&& gen, R mean = 0, R stddev = 1) { return std::normal_distribution<R>(mean, stddev)(gen); }

}  // namespace absl

#endif  // ORACLE_REF_SHIM_ABSL_ALL_H_
