// Host-side check of osg_ttt_step.h (the body of k_step_vec<Ttt, ...>) against a plain array model of
// tic_tac_toe written from the rules as the reference states them (open_spiel/games/tic_tac_toe/
// tic_tac_toe.cc:109-148,215-227: x = player 0 starts; the game is over when a player holds one of the
// eight lines or the board is full; LegalActions = empty cells, none once the game is over).
// Random games to the end, illegal and out-of-range actions, steps on finished games, "no action" (0xFF),
// and EVERY one of the 2^18 packed words as a status query (line tests of both players at once).
//   hipcc --cuda-host-only -x hip -O2 -I open_spiel_amd/csrc tests/native/ttt_step_host_test.cpp
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "osg_ttt_step.h"

static bool model_line(const int* g, int who) {
  static const int L[8][3] = {{0, 1, 2}, {3, 4, 5}, {6, 7, 8}, {0, 3, 6}, {1, 4, 7}, {2, 5, 8}, {0, 4, 8}, {2, 4, 6}};
  for (auto& l : L) if (g[l[0]] == who && g[l[1]] == who && g[l[2]] == who) return true;
  return false;
}
struct Model {
  int g[9];  // 0 empty, 1 x, 2 o
  Model() { memset(g, 0, sizeof g); }
  int plies() const { int n = 0; for (int c : g) n += c != 0; return n; }
  bool over() const { return model_line(g, 1) || model_line(g, 2) || plies() == 9; }
  int outcome() const { return model_line(g, 1) ? 0 : (model_line(g, 2) ? 1 : 2); }
  bool apply(int a) {
    if (over() || a < 0 || a >= 9 || g[a]) return false;
    g[a] = 1 + (plies() & 1);
    return true;
  }
  unsigned mask() const { unsigned m = 0; if (!over()) for (int c = 0; c < 9; ++c) if (!g[c]) m |= 1u << c; return m; }
  unsigned word() const { unsigned w = 0; for (int c = 0; c < 9; ++c) { if (g[c] == 1) w |= 1u << c; if (g[c] == 2) w |= 1u << (16 + c); } return w; }
  unsigned status(bool illegal) const {
    return (over() ? (0x80u | outcome()) : ((plies() & 1) + 1u)) | (illegal ? 0x40u : 0u);
  }
};

int main() {
  unsigned long long z = 0x5EED;
  auto rnd = [&]() { z += 0x9E3779B97F4A7C15ull; unsigned long long x = z; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); };
  long steps = 0, rejected = 0;
  for (int game = 0; game < 200000; ++game) {
    Model m;
    uint32_t w = 0;
    for (int t = 0; t < 14; ++t) {
      unsigned long long r = rnd();
      int a = static_cast<int>(r % 9);
      const int kind = static_cast<int>((r >> 32) % 16);
      if (kind == 0) a = 9 + static_cast<int>((r >> 40) % 240);      // out of range (never 0xFF)
      if (kind == 1) a = 0xFF;                                        // no action
      if (a == 0xFF - 0 && kind == 0) a = 200;
      bool illegal = false;
      if (a != 0xFF) illegal = !m.apply(a);
      rejected += illegal;
      uint32_t word = w | ((r >> 48) & 0xFE00u) | (((r >> 20) & 0x7Fu) << 25);  // junk in the unused bits is dropped
      const uint32_t got = osg::ttt_fused_step(word, static_cast<uint32_t>(a));
      const uint32_t want = m.mask() | (m.status(illegal) << 16);
      if (word != m.word() || got != want) {
        printf("MISMATCH game %d step %d action %d: word %08x vs %08x, result %08x vs %08x\n", game, t, a, word, m.word(), got, want);
        return 1;
      }
      w = word;
      ++steps;
    }
  }
  // every pair of 9-bit boards (also unreachable ones: both players with lines, overlapping stones excluded)
  long words = 0;
  for (unsigned x = 0; x < 512; ++x) for (unsigned o = 0; o < 512; ++o) {
    if (x & o) continue;
    Model m;
    for (int c = 0; c < 9; ++c) m.g[c] = ((x >> c) & 1) ? 1 : (((o >> c) & 1) ? 2 : 0);
    uint32_t word = x | (o << 16);
    const uint32_t got = osg::ttt_fused_step(word, 0xFFu);
    // the model's plies / mover come from the stone count, as in the packed state
    const uint32_t want = m.mask() | (m.status(false) << 16);
    if (word != (x | (o << 16)) || got != want) { printf("MISMATCH word x=%03x o=%03x: %08x vs %08x\n", x, o, got, want); return 1; }
    ++words;
  }
  printf("ok: 200000 games, %ld steps (%ld rejected), %ld packed words\n", steps, rejected, words);
  return 0;
}
