// TEST INFRASTRUCTURE ONLY: self-test of the abseil / nlohmann stand-ins in oracle/ref_shim, against the
// documented behaviour of the real libraries for exactly the forms the reference's hot-path files use
// (call sites cited).  Built and run by tests/test_ref_shim.py:  g++ -std=c++17 -Ioracle/ref_shim ...
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "open_spiel/abseil-cpp/absl/shim_all.h"
#include "open_spiel/json/include/nlohmann/json.hpp"

static int g_failures = 0;
#define EXPECT_EQ(a, b)                                                                          \
  do {                                                                                           \
    auto va = (a);                                                                               \
    auto vb = (b);                                                                               \
    if (!(va == vb)) {                                                                           \
      ++g_failures;                                                                              \
      std::fprintf(stderr, "%s:%d: %s != %s\n", __FILE__, __LINE__, #a, #b);                     \
    }                                                                                            \
  } while (0)
#define EXPECT_TRUE(c) EXPECT_EQ(static_cast<bool>(c), true)

struct Contents {  // the shape of connect_four.h:72-79
  std::vector<std::vector<std::string>> board;
  std::string current_player;
  bool is_terminal = false;
  std::string winner;
  NLOHMANN_DEFINE_TYPE_INTRUSIVE(Contents, board, current_player, is_terminal, winner);
};
struct Derived : public Contents {  // SPIEL_DEFINE_STRUCT: conversion through the base's friends
  nlohmann::json to_json_base() const { return *this; }
};
enum class Dyn { kA = 0, kB = 7 };

int main() {
  using std::string;
  // ---- StrCat / StrAppend (spiel.cc, leduc_poker.cc:198-239: ints, int64, doubles as %g, strings) ----
  EXPECT_EQ(absl::StrCat("a", 1, "b", int64_t{-5}, size_t{7}, 2.5, 100.0, 1e-7, string("z")), string("a1b-572.51001e-07z"));
  EXPECT_EQ(absl::StrCat(true, false), string("10"));
  EXPECT_EQ(absl::StrCat(Dyn::kB), string("7"));           // spiel.cc:658
  EXPECT_EQ(absl::StrCat(), string(""));
  EXPECT_EQ(absl::StrCat(0.1 + 0.2, " ", 1.0 / 3), string("0.3 0.333333"));  // six significant digits
  string s = "x";
  absl::StrAppend(&s, 12, "-", 3.0, absl::string_view("sv"));
  EXPECT_EQ(s, string("x12-3sv"));
  absl::StrAppend(&s);
  EXPECT_EQ(s, string("x12-3sv"));
  // ---- StrJoin (leduc "[Money: 100 100]" joins doubles; policy.h:221 PairFormatter) ----
  EXPECT_EQ(absl::StrJoin(std::vector<double>{100, 99.5}, " "), string("100 99.5"));
  EXPECT_EQ(absl::StrJoin(std::vector<int>{}, ","), string(""));
  EXPECT_EQ(absl::StrJoin(std::vector<string>{"a", "b", "c"}, ", "), string("a, b, c"));
  std::map<int, double> m{{1, 0.5}, {2, 0.25}};
  EXPECT_EQ(absl::StrJoin(m, " ", absl::PairFormatter(absl::AlphaNumFormatter(), "=", absl::AlphaNumFormatter())),
            string("1=0.5 2=0.25"));
  EXPECT_EQ(absl::StrJoin(std::vector<int>{1, 2}, "+", [](string* out, int v) { absl::StrAppend(out, v * 10); }),
            string("10+20"));
  // ---- StrSplit (spiel.cc:551,869; policy.cc:145-186; cfr.cc:540-541,770) ----
  std::vector<string> v = absl::StrSplit("a,b,,c", ',');
  EXPECT_EQ(v, (std::vector<string>{"a", "b", "", "c"}));
  v = absl::StrSplit("", '\n');
  EXPECT_EQ(v, (std::vector<string>{""}));                  // one empty piece, like the library
  v = absl::StrSplit("k: v: w", absl::MaxSplits(": ", 1));
  EXPECT_EQ(v, (std::vector<string>{"k", "v: w"}));
  std::pair<string, string> pr = absl::StrSplit("cls:content:more", absl::MaxSplits(':', 1));
  EXPECT_EQ(pr.first, string("cls"));
  EXPECT_EQ(pr.second, string("content:more"));
  std::pair<absl::string_view, absl::string_view> halves = absl::StrSplit("head[T]\ntail", absl::StrCat("[T]", "\n"));
  EXPECT_EQ(halves.second, absl::string_view("tail"));      // temporary std::string delimiter (cfr.cc:770)
  std::vector<std::vector<absl::string_view>> nested;
  for (absl::string_view piece : absl::StrSplit("0,1;x,y", ';')) nested.push_back(absl::StrSplit(piece, ','));
  EXPECT_EQ(nested.size(), size_t{2});
  EXPECT_EQ(nested[1][1], absl::string_view("y"));
  int count = 0;
  for (absl::string_view tok : absl::StrSplit(string("1 2 3"), ' ')) count += tok.size();  // owned temporary
  EXPECT_EQ(count, 3);
  v = absl::StrSplit("a<~>b<~>c", "<~>");
  EXPECT_EQ(v.size(), size_t{3});
  // ---- StrFormat (spiel_utils.cc:105 "%.15f"; mcts.cc:165-174; serialization.h:44 "%a"; policy.cc:585 "%i") ----
  EXPECT_EQ(absl::StrFormat("%.15f", 0.1), string("0.100000000000000"));
  EXPECT_EQ(absl::StrFormat("%d/%i/%5d/%-3d|", int64_t{1} << 40, size_t{3}, 42, 7), string("1099511627776/3/   42/7  |"));
  EXPECT_EQ(absl::StrFormat("%4.1f%%", 12.345), string("12.3%"));
  EXPECT_EQ(absl::StrFormat("%s=%s", string("k"), "v"), string("k=v"));
  EXPECT_EQ(absl::StrFormat("%6s|%-6s|", "ab", "cd"), string("    ab|cd    |"));
  EXPECT_EQ(absl::StrFormat("%a", 1.5), string("0x1.8p+0"));
  EXPECT_EQ(absl::StrFormat("(%i, %f), ", 3, 0.25), string("(3, 0.250000), "));
  EXPECT_EQ(absl::StrFormat("%v %v", 5, "x"), string("5 x"));
  // ---- numbers / charconv / ascii / replace / match ----
  int i = 0;
  int64_t l = 0;
  double d = 0;
  EXPECT_TRUE(absl::SimpleAtoi(" 42 ", &i) && i == 42);
  EXPECT_TRUE(absl::SimpleAtoi("-9000000000", &l) && l == -9000000000LL);
  EXPECT_TRUE(!absl::SimpleAtoi("4x", &i) && !absl::SimpleAtoi("", &i) && !absl::SimpleAtoi("99999999999", &i));
  EXPECT_TRUE(absl::SimpleAtod("1e-3", &d) && d == 1e-3);
  EXPECT_TRUE(!absl::SimpleAtod("1.5abc", &d));
  const string hex = "0x1.3f42448f051cfp-3";
  absl::from_chars(hex.data(), hex.data() + hex.size(), d);   // cfr.cc:555 on a HexDoubleFormatter value
  EXPECT_EQ(absl::StrFormat("%a", d), hex);
  EXPECT_EQ(string(absl::StripAsciiWhitespace("  a b \n")), string("a b"));
  EXPECT_EQ(absl::StrReplaceAll("a\nb\nc", {{"\n", "\\n"}}), string("a\\nb\\nc"));  // game_parameters.cc:81
  EXPECT_TRUE(absl::StartsWith("kuhn_poker", "kuhn") && absl::EndsWith("kuhn_poker", "poker") && absl::StrContains("abc", "bc"));
  // ---- Span (observer.h, spiel.h:713-714) ----
  std::vector<float> buf(6, 0.f);
  absl::Span<float> mut = absl::MakeSpan(buf);
  mut[2] = 1.f;
  absl::Span<const float> ro = buf;
  EXPECT_EQ(ro.size(), size_t{6});
  EXPECT_EQ(ro.subspan(2, 2)[0], 1.f);
  absl::Span<const float> from_mut = mut;                  // Span<T> -> Span<const T>
  EXPECT_EQ(from_mut.data(), buf.data());
  // ---- containers / algorithms / optional ----
  absl::flat_hash_map<std::string, int> fm{{"a", 1}};
  EXPECT_TRUE(fm.contains("a") && !fm.contains("b"));
  absl::btree_map<int, int> bm{{2, 1}, {1, 2}};
  EXPECT_EQ(bm.begin()->first, 1);
  absl::flat_hash_map<std::pair<int, int>, int> pm;        // pair keys need the stand-in's Hash
  pm[{1, 2}] = 3;
  EXPECT_EQ(pm.at({1, 2}), 3);
  std::vector<int> iv{3, 1, 2};
  EXPECT_TRUE(absl::c_linear_search(iv, 2) && absl::c_find(iv, 9) == iv.end());
  EXPECT_EQ(absl::c_accumulate(iv, 0), 6);
  absl::optional<int> none = absl::nullopt;
  EXPECT_TRUE(!none.has_value());
  // ---- random: Uniform over a 32-bit engine (mcts.cc:54, spiel.cc:368-371) ----
  std::mt19937 rng(7), rng2(7);
  for (int k = 0; k < 1000; ++k) {
    size_t pick = absl::Uniform(rng, 0u, size_t{5});
    double u = absl::Uniform(rng, 0.0, 1.0);
    EXPECT_TRUE(pick < 5 && u >= 0.0 && u < 1.0);
  }
  absl::BitGenRef ref(rng2);
  std::mt19937 rng3(7);
  EXPECT_EQ(absl::Uniform(ref, 0.0, 1.0), (rng3() >> 5) * (1.0 / 134217728.0));  // one draw, 27 bits: the restatement's convention
  // ---- time / mutex ----
  absl::Time t0 = absl::Now();
  EXPECT_TRUE(absl::ToDoubleSeconds(absl::Now() - t0) >= 0.0 && absl::ToInt64Nanoseconds(t0 - absl::UnixEpoch()) > 0);
  absl::Mutex mu;
  { absl::MutexLock lock(&mu); }
  { absl::MutexLock lock(mu); }
  // ---- nlohmann stand-in (spiel.cc:301-352, connect_four.h:72-113, spiel_utils.h:471-504) ----
  nlohmann::json j = nlohmann::json::parse(R"({"game_name":"connect_four","rows":5,"x":1.5,"ego":true,"b":[["x","."],["o","."]],"n":null})");
  EXPECT_TRUE(j.contains("rows") && j["rows"].is_number_integer() && j["x"].is_number_float() && j["ego"].is_boolean());
  EXPECT_EQ(j.at("game_name").get<string>(), string("connect_four"));
  EXPECT_EQ(j["rows"].get<int>(), 5);
  int seen = 0;
  for (auto& [key, value] : j.items()) seen += key.empty() ? 0 : 1;
  EXPECT_EQ(seen, 6);
  EXPECT_EQ(j.dump(), string(R"({"b":[["x","."],["o","."]],"ego":true,"game_name":"connect_four","n":null,"rows":5,"x":1.5})"));
  EXPECT_EQ((nlohmann::json{{"game_name", string("g")}}).dump(), string(R"({"game_name":"g"})"));
  Derived c;
  c.board = {{"x", "o"}, {".", "."}};
  c.current_player = "x";
  c.winner = "";
  nlohmann::json cj = c.to_json_base();
  EXPECT_EQ(cj.dump(), string(R"({"board":[["x","o"],[".","."]],"current_player":"x","is_terminal":false,"winner":""})"));
  Derived back;
  nlohmann::json::parse(cj.dump()).get_to(back);
  EXPECT_TRUE(back.board == c.board && back.current_player == "x" && !back.is_terminal);
  bool threw = false;
  try { nlohmann::json::parse("{\"a\":}"); } catch (const nlohmann::json::exception&) { threw = true; }
  EXPECT_TRUE(threw);
  threw = false;
  try { Derived bad; nlohmann::json::parse("{\"board\":[]}").get_to(bad); } catch (const nlohmann::json::exception&) { threw = true; }
  EXPECT_TRUE(threw);                                       // missing keys are an error, as in the library

  if (g_failures) {
    std::fprintf(stderr, "%d check(s) failed\n", g_failures);
    return 1;
  }
  std::printf("ref_shim selftest: all checks passed\n");
  return 0;
}
