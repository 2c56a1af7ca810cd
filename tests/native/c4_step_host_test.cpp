// Host-side check of osg_c4_step.h (the body of the headline kernel k_step_c4std) against a plain
// array model of connect_four written from the rules as the reference states them
// (open_spiel/games/connect_four/connect_four.cc:130-209: a stone drops to the lowest empty row;
// the mover wins with four in a row in any of the four directions; a full board is a draw;
// LegalActions = columns whose top cell is empty, none once the game is over).
// Random games to the end, illegal and out-of-range actions, steps on finished games, "no action" (0xFF).
//   hipcc --cuda-host-only -x hip -O2 -I open_spiel_amd/csrc tests/native/c4_step_host_test.cpp
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "osg_c4_step.h"

struct Model {
  int g[6][7];  // 0 empty, 1 x, 2 o
  int plies = 0, outcome = -1;  // outcome: -1 running, 0 x, 1 o, 2 draw
  Model() { memset(g, 0, sizeof g); }
  bool room(int c) const { return g[5][c] == 0; }
  bool line(int who) const {
    const int dr[4] = {0, 1, 1, 1}, dc[4] = {1, 0, 1, -1};
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 7; ++c) for (int d = 0; d < 4; ++d) {
      int k = 0;
      for (; k < 4; ++k) {
        int rr = r + k * dr[d], cc = c + k * dc[d];
        if (rr < 0 || rr >= 6 || cc < 0 || cc >= 7 || g[rr][cc] != who) break;
      }
      if (k == 4) return true;
    }
    return false;
  }
  bool apply(int a) {  // false = rejected
    if (outcome >= 0 || a < 0 || a >= 7 || !room(a)) return false;
    int r = 0;
    while (g[r][a]) ++r;
    const int who = 1 + (plies & 1);
    g[r][a] = who;
    ++plies;
    if (line(who)) outcome = who - 1; else if (plies == 42) outcome = 2;
    return true;
  }
  unsigned mask() const {
    unsigned m = 0;
    if (outcome < 0) for (int c = 0; c < 7; ++c) if (room(c)) m |= 1u << c;
    return m;
  }
  void planes(uint64_t* x, uint64_t* o) const {
    *x = *o = 0;
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 7; ++c) {
      if (g[r][c] == 1) *x |= 1ull << (c * 7 + r);
      if (g[r][c] == 2) *o |= 1ull << (c * 7 + r);
    }
    if (outcome >= 0) *x |= (uint64_t)(1u | (outcome << 1)) << 56;
  }
};

int main() {
  uint64_t z = 0x5EED;
  long steps = 0, games = 0, rejected = 0;
  for (int game = 0; game < 200000; ++game) {
    Model m;
    uint64_t x = 0, o = 0;
    for (int t = 0; t < 80; ++t) {
      z = osg::mix64(z + 0x9E3779B97F4A7C15ULL);
      uint32_t a;
      const unsigned kind = (z >> 40) % 16;
      if (kind == 0) a = 0xFF;                       // no action
      else if (kind == 1) a = 7 + (z >> 8) % 248;    // out of range (may be 0xFF: no action)
      else a = z % 7;                                // any column, full ones included
      const bool ok = a == 0xFF ? true : m.apply((int)a);
      const uint32_t r = osg::c4_fused_step(x, o, a);
      uint64_t wx, wo;
      m.planes(&wx, &wo);
      const uint32_t want_mask = m.mask();
      const uint32_t want_st = (ok ? 0u : 0x40u) | (m.outcome >= 0 ? (0x80u | (unsigned)m.outcome) : (unsigned)((m.plies & 1) + 1));
      if (x != wx || o != wo || (r & 0xFF) != want_mask || (r >> 8) != want_st) {
        printf("MISMATCH game %d ply %d action %u: planes %016llx %016llx want %016llx %016llx, mask %02x want %02x, status %02x want %02x\n",
               game, t, a, (unsigned long long)x, (unsigned long long)o, (unsigned long long)wx, (unsigned long long)wo,
               r & 0xFF, want_mask, r >> 8, want_st);
        return 1;
      }
      ++steps;
      rejected += !ok;
      if (m.outcome >= 0 && t > 50) break;
    }
    ++games;
  }
  printf("ok: %ld games, %ld fused steps (%ld rejected actions) equal the array model\n", games, steps, rejected);
  return 0;
}
