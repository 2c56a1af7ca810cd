// Device-side rules of the three perfect-information board games, on bitboards.
// Each struct restates one reference State implementation for ONE state held in
// registers; the kernels in osg_kernels.hip map them over SoA batches.
//
//   Ttt : open_spiel/games/tic_tac_toe/tic_tac_toe.{h,cc}
//   C4  : open_spiel/games/connect_four/connect_four.{h,cc}
//   Hex : open_spiel/games/hex/hex.{h,cc}
#ifndef OSG_GAME_BOARDS_H_
#define OSG_GAME_BOARDS_H_

#include "osg_common.h"

namespace osg {

// Default tensor walker: entry idx, idx + 1, ... through the game's obs_at.
template <class G>
struct GenericObsCursor {
  int idx;
  OSG_HD void init(const typename G::Params&, const typename G::State&, int, int, int idx0) { idx = idx0; }
  OSG_HD float next(const typename G::Params& p, const typename G::State& s, int player, int which) {
    return G::obs_at(p, s, player, which, idx++);
  }
};

// ===========================================================================
// tic_tac_toe.  HBM layout: ONE u32 per state, bits 0-8 = x stones (player 0,
// kCross), bits 16-24 = o stones (player 1, kNought); cell = 3*row + col.
// Player to move = parity of the stone count (player 0 starts,
// tic_tac_toe.h:127); outcome is recomputed from the lines.
// ===========================================================================
struct Ttt {
  using word_t = uint32_t;
  static constexpr int kMaskW = kMaskWords;
  struct Params {
    int words;  // = 1
  };
  struct State {
    uint32_t x, o;
  };
  OSG_D static State initial(const Params&) { return {0u, 0u}; }
  OSG_D static State load(const Params&, const word_t* base, int64_t, int64_t i) {
    uint32_t v = base[i];
    return {v & 0x1FFu, (v >> 16) & 0x1FFu};
  }
  OSG_D static void store(const Params&, word_t* base, int64_t, int64_t i, const State& s) {
    base[i] = s.x | (s.o << 16);
  }
  // BoardHasLine, tic_tac_toe.cc:109-120: the 8 lines as 9-bit masks.
  OSG_D static bool line(uint32_t b) {
    return ((b & 0x007) == 0x007) | ((b & 0x038) == 0x038) | ((b & 0x1C0) == 0x1C0) |
           ((b & 0x049) == 0x049) | ((b & 0x092) == 0x092) | ((b & 0x124) == 0x124) |
           ((b & 0x111) == 0x111) | ((b & 0x054) == 0x054);
  }
  // A position from its cells ('.', 'x', 'o'; cell a = action a): TicTacToeState(game, TicTacToeStateStruct),
  // tic_tac_toe.cc:273-336.  0, or 1 wrong cell count, 2 bad character, 4 both players have a line (the piece-count
  // rules are the caller's: this layout derives the mover from the stone count).
  OSG_D static int from_cells(const Params&, const unsigned char* cells, int n_cells, State& s) {
    if (n_cells != 9) return 1;
    s = {0u, 0u};
    for (int a = 0; a < 9; ++a) {
      if (cells[a] == 'x') s.x |= 1u << a;
      else if (cells[a] == 'o') s.o |= 1u << a;
      else if (cells[a] != '.') return 2;
    }
    return line(s.x) && line(s.o) ? 4 : 0;
  }
  OSG_D static int plies(const State& s) { return __builtin_popcount(s.x | s.o); }
  OSG_D static bool terminal(const Params&, const State& s) {  // tic_tac_toe.cc:215-217
    return line(s.x) | line(s.o) | (plies(s) == 9);
  }
  OSG_D static int current_player(const Params& p, const State& s) {  // tic_tac_toe.h:87-89
    return terminal(p, s) ? kTerminalPlayer : (plies(s) & 1);
  }
  OSG_D static Mask legal(const Params& p, const State& s) {  // tic_tac_toe.cc:138-148
    Mask m;
    if (!terminal(p, s)) m.w[0] = ~(s.x | s.o) & 0x1FFu;
    return m;
  }
  OSG_D static void apply(const Params&, State& s, int a) {  // tic_tac_toe.cc:128-136
    if (plies(s) & 1) s.o |= 1u << a; else s.x |= 1u << a;
  }
  OSG_D static int outcome_code(const Params&, const State& s) {  // tic_tac_toe.cc:219-227
    return line(s.x) ? 0 : (line(s.o) ? 1 : 2);
  }
  OSG_D static void returns(const Params& p, const State& s, double* out) {
    int c = outcome_code(p, s);
    out[0] = c == 0 ? 1.0 : (c == 1 ? -1.0 : 0.0);
    out[1] = -out[0] + 0.0;  // +0.0: never produce -0.0 for draws / running games
  }
  OSG_D static double chance_prob(const Params&, const State&, int) { return 0.0; }
  // ObservationTensor, tic_tac_toe.cc:241-251: plane = raw CellState enum value
  // (0 empty, 1 o, 2 x), flat index plane*9 + cell.  Returns the value at `idx`.
  OSG_D static float obs_at(const Params&, const State& s, int /*player*/, int /*which*/, int idx) {
    int plane = idx / 9, cell = idx - plane * 9;
    uint32_t bits = plane == 0 ? ~(s.x | s.o) : (plane == 1 ? s.o : s.x);
    return static_cast<float>((bits >> cell) & 1u);
  }
  // The 27 entries as one bit set: plane 0 empty cells, plane 1 o, plane 2 x.
  struct ObsCursor {
    uint32_t bits;
    int idx;
    OSG_HD void init(const Params&, const State& s, int, int, int idx0) {
      idx = idx0;
      bits = (~(s.x | s.o) & 0x1FFu) | (s.o << 9) | (s.x << 18);
    }
    OSG_HD float next(const Params&, const State&, int, int) { return static_cast<float>((bits >> idx++) & 1u); }
  };
};

// ===========================================================================
// connect_four (rows R, columns C, x_in_row K; (R+1)*C <= 64).  HBM layout: TWO
// u64 planes per state: plane 0 = x stones (player 0), plane 1 = o stones.
// Bit index = col*(R+1) + row, row 0 = bottom; the extra row per column is an
// always-empty sentinel so shifted line tests never wrap between columns.
// Player to move = parity of the stone count.
//
// The default 6x7 board uses 49 bits per plane, so plane 0 carries the result of
// the game in its spare top byte (bit 56 = terminal, bits 57-58 = outcome: 0 x
// won, 1 o won, 2 draw) — written by apply(), which already has to test the
// mover's lines (connect_four.cc:138-142).  IsTerminal / Returns / LegalActions then
// cost no line test at all.  Other geometries have no spare bits guaranteed and
// recompute the outcome (a reachable position has at most one player with a line).
// ===========================================================================
struct C4Params {
  int words;  // = 2 (boards of up to 64 bits), 4 (up to 128 bits: two words per colour)
  int rows, cols, k, ego;
  uint64_t board;  // all playable cells (low 64 bits)
  uint64_t top;    // top playable cell of every column (low 64 bits)
  uint64_t board_hi, top_hi;   // bits 64-127 of the same (wide boards)
};
typedef unsigned __int128 osg_u128;
template <class BB> struct BitboardOps;
template <> struct BitboardOps<uint64_t> {
  static constexpr int kWords = 1;
  OSG_HD static int popcount(uint64_t v) { return __builtin_popcountll(v); }
  OSG_HD static uint64_t make(uint64_t lo, uint64_t) { return lo; }
  OSG_HD static uint64_t shr_small(uint64_t v, int s) { return v >> s; }   // 0 < s < 64
};
template <> struct BitboardOps<osg_u128> {
  static constexpr int kWords = 2;
  OSG_HD static int popcount(osg_u128 v) {
    return __builtin_popcountll(static_cast<uint64_t>(v)) + __builtin_popcountll(static_cast<uint64_t>(v >> 64));
  }
  OSG_HD static osg_u128 make(uint64_t lo, uint64_t hi) { return (static_cast<osg_u128>(hi) << 64) | lo; }
  // v >> s for 0 < s < 64, on the two halves: three 64-bit shifts and an or — the generic 128-bit shift by a run-time
  // amount also handles s >= 64 with a compare and four selects (round 6: the wide boards' step is instruction-bound)
  OSG_HD static osg_u128 shr_small(osg_u128 v, int s) {
    const uint64_t lo = static_cast<uint64_t>(v), hi = static_cast<uint64_t>(v >> 64);
    return make((lo >> s) | (hi << (64 - s)), hi >> s);
  }
};
// R_/C_/K_ = 0: geometry read from Params at run time.  The default 6x7x4 game
// is instantiated with compile-time constants (shifts by immediates, the
// column loop fully unrolled): this is the headline kernel's game.
// BB: the bitboard type — uint64_t while (rows + 1) * cols <= 64, unsigned __int128 (two plane words per colour:
// x.lo, x.hi, o.lo, o.hi) up to 128 bits, e.g. 8 x 8, 9 x 9, 10 x 10, 7 x 15.
template <int R_, int C_, int K_, class BB = uint64_t>
struct C4T {
  using Ops = BitboardOps<BB>;
  using word_t = uint64_t;
  static constexpr int kMaskW = kMaskWords;
  using Params = C4Params;
  static constexpr bool kStored = R_ != 0 && (R_ + 1) * C_ <= 56;  // result kept in plane 0's top byte
  struct State {
    BB x, o;
    uint32_t flags;  // kStored only: bit 0 terminal, bits 1-2 outcome
  };
  OSG_D static int R(const Params& p) { return R_ ? R_ : p.rows; }
  OSG_D static int C(const Params& p) { return C_ ? C_ : p.cols; }
  OSG_D static int K(const Params& p) { return K_ ? K_ : p.k; }
  OSG_D static BB top(const Params& p) {
    if (R_ == 0) return Ops::make(p.top, p.top_hi);
    BB t = 0;
#pragma unroll
    for (int c = 0; c < C_; ++c) t |= BB(1) << (c * (R_ + 1) + R_ - 1);
    return t;
  }
  OSG_D static State initial(const Params&) { return {BB(0), BB(0), 0u}; }
  OSG_D static State unpack(uint64_t w0, uint64_t w1) {
    if (kStored) return {w0 & ((1ull << 56) - 1ull), w1, static_cast<uint32_t>(w0 >> 56)};
    return {w0, w1, 0u};
  }
  OSG_D static uint64_t pack0(const State& s) {
    return kStored ? (s.x | (static_cast<uint64_t>(s.flags) << 56)) : s.x;
  }
  OSG_D static State load(const Params&, const word_t* base, int64_t n, int64_t i) {
    if constexpr (Ops::kWords == 2) return {Ops::make(base[i], base[n + i]), Ops::make(base[2 * n + i], base[3 * n + i]), 0u};
    else return unpack(base[i], base[n + i]);
  }
  OSG_D static void store(const Params&, word_t* base, int64_t n, int64_t i, const State& s) {
    if constexpr (Ops::kWords == 2) {
      base[i] = static_cast<uint64_t>(s.x); base[n + i] = static_cast<uint64_t>(s.x >> 64);
      base[2 * n + i] = static_cast<uint64_t>(s.o); base[3 * n + i] = static_cast<uint64_t>(s.o >> 64);
    } else {
      base[i] = pack0(s);
      base[n + i] = s.o;
    }
  }
  // HasLine, connect_four.cc:163-201, as the classic shifted-AND test along the
  // four directions: vertical (1), horizontal (H), the two diagonals (H-1, H+1).
  OSG_D static bool line(const Params& p, BB b) {
    const int H = R(p) + 1;
    const int dirs[4] = {1, H, H - 1, H + 1};
    BB hit = 0;
    if (R_ == 0 && K(p) == 4 && 2 * (H + 1) < 64) {   // run-time geometry (wave-uniform): every shift amount is in [1, 63]
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        BB m = b & Ops::shr_small(b, dirs[d]);
        hit |= m & Ops::shr_small(m, 2 * dirs[d]);
      }
      return hit != 0;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      BB m = b;
      if (K(p) == 4) {
        m = m & (m >> dirs[d]);
        m = m & (m >> (2 * dirs[d]));
      } else {
        for (int i = 1; i < K(p); ++i) m &= b >> (i * dirs[d]);
      }
      hit |= m;
    }
    return hit != 0;
  }
  OSG_D static int plies(const State& s) { return Ops::popcount(s.x | s.o); }
  OSG_D static bool full(const Params& p, const State& s) {  // connect_four.cc:203-209
    return ((s.x | s.o) & top(p)) == top(p);
  }
  OSG_D static bool terminal(const Params& p, const State& s) {
    if (kStored) return s.flags & 1u;
    return line(p, s.x) | line(p, s.o) | full(p, s);
  }
  OSG_D static int current_player(const Params& p, const State& s) {  // connect_four.cc:122-128
    return terminal(p, s) ? kTerminalPlayer : (plies(s) & 1);
  }
  OSG_D static bool column_has_room(const Params& p, const State& s, int col) {  // connect_four.cc:153
    return col < C(p) && !static_cast<uint32_t>(((s.x | s.o) >> (col * (R(p) + 1) + R(p) - 1)) & BB(1));
  }
  // Bit c set = column c still has room, whatever the state of the game.
  OSG_D static uint32_t open_columns(const Params& p, const State& s) {
    const int H = R(p) + 1;
    BB free_top = ~(s.x | s.o) & top(p);
    if constexpr (R_ == 6 && C_ == 7) {
      // One multiply instead of seven extracts: the top cells sit at bits 5 + 7c; after >> 5 they
      // are at 7c, and multiplying by sum_k 2^(36 - 6k) moves bit 7k to 36 + k.  No two partial
      // products share a bit (7(i - i') = 6(k - k') has no solution with |k - k'| <= 6): no carries.
      const uint64_t M = (1ull << 36) | (1ull << 30) | (1ull << 24) | (1ull << 18) | (1ull << 12) | (1ull << 6) | 1ull;
      return static_cast<uint32_t>(((free_top >> 5) * M) >> 36) & 0x7Fu;
    }
    uint32_t m = 0;
    if (C_ != 0) {
#pragma unroll
      for (int c = 0; c < C_; ++c) m |= static_cast<uint32_t>((free_top >> (c * H + R(p) - 1)) & BB(1)) << c;
    } else {
      // a running shift by H (< 64) per column instead of a 128-bit shift by c * H + R - 1 per column
      BB t = free_top >> (R(p) - 1);
      for (int c = 0; c < p.cols; ++c) {
        m |= (static_cast<uint32_t>(t) & 1u) << c;
        t = H < 64 ? Ops::shr_small(t, H) : (t >> H);
      }
    }
    return m;
  }
  OSG_D static Mask legal(const Params& p, const State& s) {  // connect_four.cc:147-156
    Mask m;
    if (!terminal(p, s)) m.w[0] = open_columns(p, s);
    return m;
  }
  OSG_D static void apply(const Params& p, State& s, int col) {  // connect_four.cc:130-145
    const int H = R(p) + 1;
    BB all = s.x | s.o;
    BB colmask = ((BB(1) << R(p)) - BB(1)) << (col * H);
    BB cell = (all + (BB(1) << (col * H))) & colmask;  // lowest empty cell
    const int mover = Ops::popcount(all) & 1;
    if (mover) s.o |= cell; else s.x |= cell;
    if (kStored) {  // outcome_ = mover if HasLine(mover) else draw if IsFull (connect_four.cc:138-142)
      const bool win = line(p, mover ? s.o : s.x);
      const bool done = win | full(p, s);
      s.flags = done ? (1u | (static_cast<uint32_t>(win ? mover : 2) << 1)) : 0u;
    }
  }
  // The whole step of k_step in one pass (round 6; the geometries without a stored result: run-time sizes, boards above
  // 56 bits): the generic sequence legal / apply / terminal / legal / outcome runs the line test on BOTH colours three
  // times over; here the non-mover's plane is tested once (it does not change), the mover's once AFTER the tentative
  // placement — a line it has not got then, it had not got before either (a subset) — and only where that test fires
  // (a winning move, or an uploaded position that was over already) the plane as it was is tested too.  Returns the
  // successor's open columns (0 when the game is over); same results as the generic sequence, case by case
  // (tests/test_gpu_parity.py steps these boards against the oracle at every ply).
  OSG_D static uint32_t fused_step(const Params& p, State& s, int a, bool& illegal, bool& term, int& outcome) {
    const int H = R(p) + 1;
    const BB all = s.x | s.o, tp = top(p);
    const int mover = Ops::popcount(all) & 1;
    const BB mine = mover ? s.o : s.x, theirs = mover ? s.x : s.o;
    BB cell = 0;
    if (a != 0xFF && a < C(p)) {
      const BB colmask = ((BB(1) << R(p)) - BB(1)) << (a * H);
      cell = (all + (BB(1) << (a * H))) & colmask;   // the lowest empty cell of the column; 0 when it is full
    }
    const bool their_line = line(p, theirs);
    const bool full_before = (all & tp) == tp;
    bool my_line_after = line(p, mine | cell), my_line_before = false;
    if (my_line_after && cell != 0) my_line_before = line(p, mine);
    else my_line_before = my_line_after;              // (no cell placed: the same plane)
    const bool term_before = their_line | my_line_before | full_before;
    const bool placed = cell != 0 && !term_before;
    illegal = a != 0xFF && !placed;
    const BB mine2 = placed ? (mine | cell) : mine;
    const bool my_line = placed ? my_line_after : my_line_before;
    if (mover) s.o = mine2; else s.x = mine2;
    const BB all2 = placed ? (all | cell) : all;
    term = their_line | my_line | ((all2 & tp) == tp);
    const bool x_line = mover ? their_line : my_line, o_line = mover ? my_line : their_line;
    outcome = x_line ? 0 : (o_line ? 1 : 2);
    if (term) return 0u;
    State t{mover ? theirs : mine2, mover ? mine2 : theirs, 0u};
    return open_columns(p, t);
  }
  // A position from its cells ('.', 'x', 'o'; cell r * cols + c, row 0 = the bottom row: ConnectFourStateStruct::board,
  // connect_four.h:72-79): ConnectFourState(game, struct, strict) / ConnectFourState(game, string),
  // connect_four.cc:352-470.  0, or 1 wrong cell count, 2 bad character, 3 a gap in a column, 4 both players have a
  // line (the piece-count rules are the caller's: this layout derives the mover from the stone count).
  OSG_D static int from_cells(const Params& p, const unsigned char* cells, int n_cells, State& s) {
    const int rows = R(p), cols = C(p), H = rows + 1;
    if (n_cells != rows * cols) return 1;
    s = {BB(0), BB(0), 0u};
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        const unsigned char ch = cells[r * cols + c];
        if (ch == 'x') s.x |= BB(1) << (c * H + r);
        else if (ch == 'o') s.o |= BB(1) << (c * H + r);
        else if (ch != '.') return 2;
      }
    for (int c = 0; c < cols; ++c) {   // stacked from the bottom: the column's stones are a run of low bits
      const BB col = ((s.x | s.o) >> (c * H)) & ((BB(1) << rows) - BB(1));
      if ((col & (col + BB(1))) != BB(0)) return 3;
    }
    const bool xw = line(p, s.x), ow = line(p, s.o);
    if (xw && ow) return 4;
    if (kStored) {
      const bool done = xw | ow | full(p, s);
      s.flags = done ? (1u | (static_cast<uint32_t>(xw ? 0 : (ow ? 1 : 2)) << 1)) : 0u;
    }
    return 0;
  }
  OSG_D static int outcome_code(const Params& p, const State& s) {  // connect_four.cc:281-285
    if (kStored) return static_cast<int>((s.flags >> 1) & 3u);
    return line(p, s.x) ? 0 : (line(p, s.o) ? 1 : 2);
  }
  OSG_D static void returns(const Params& p, const State& s, double* out) {
    int c = terminal(p, s) ? outcome_code(p, s) : 2;
    out[0] = c == 0 ? 1.0 : (c == 1 ? -1.0 : 0.0);
    out[1] = -out[0] + 0.0;
  }
  OSG_D static double chance_prob(const Params&, const State&, int) { return 0.0; }
  // ObservationTensor, connect_four.cc:312-328.  Shape [3, R, C]; default planes
  // via StateToPlayer (:75-86): 0 = x, 1 = o, 2 = empty.  Egocentric planes via
  // PlayerRelative (:299-310): nought -> (player==0 ? 0 : 1), cross -> (player==1 ? 0 : 1).
  OSG_D static float obs_at(const Params& p, const State& s, int player, int /*which*/, int idx) {
    const int RC = R(p) * C(p);
    int plane = idx / RC, rem = idx - plane * RC;
    int r = rem / C(p), c = rem - r * C(p);
    int bit = c * (R(p) + 1) + r;
    BB first = s.x, second = s.o;
    if (p.ego) {  // plane 0 holds kNought iff player == 0, kCross iff player == 1
      first = player == 0 ? s.o : s.x;
      second = player == 0 ? s.x : s.o;
    }
    BB bits = plane == 0 ? first : (plane == 1 ? second : ~(s.x | s.o));
    return static_cast<float>(static_cast<uint32_t>((bits >> bit) & BB(1)));
  }
  // Walks the tensor of one state entry by entry (same values as obs_at; plane / row / column are
  // advanced incrementally instead of being re-derived by division for every float).
  struct ObsCursor {
    BB first, second, empty;
    int plane, r, c;
    OSG_D void init(const Params& p, const State& s, int player, int /*which*/, int idx) {
      const int RC = R(p) * C(p);
      plane = idx / RC;
      const int rem = idx - plane * RC;
      r = rem / C(p);
      c = rem - r * C(p);
      first = s.x;
      second = s.o;
      if (p.ego) {
        first = player == 0 ? s.o : s.x;
        second = player == 0 ? s.x : s.o;
      }
      empty = ~(s.x | s.o);
    }
    OSG_D float next(const Params& p, const State&, int, int) {
      const BB bits = plane == 0 ? first : (plane == 1 ? second : empty);
      const float v = static_cast<float>(static_cast<uint32_t>((bits >> (c * (R(p) + 1) + r)) & BB(1)));
      if (++c == C(p)) {
        c = 0;
        if (++r == R(p)) { r = 0; ++plane; }
      }
      return v;
    }
  };
};
using C4 = C4T<0, 0, 0>;     // any geometry with (rows+1)*cols <= 64
using C4Wide = C4T<0, 0, 0, osg_u128>;  // (rows+1)*cols <= 128: two plane words per colour
using C4Std = C4T<6, 7, 4>;  // the default game, constants folded

// ===========================================================================
// hex (num_cols C, num_rows R, C*R <= 32*NW).  HBM layout: 4*NW + 1 u32 planes:
//   [0,NW)    black stones      [NW,2NW)   white stones
//   [2NW,3NW) edge-A connected  [3NW,4NW)  edge-B connected
//   4NW       meta: bit0 player to move, bits1-2 result (1 black won, 2 white
//             won), bits 8-15 plies (saturating), bits 16-31 first move
// Cell = row*C + col.  The reference's 9 labels (hex.h:68-78) are
//   black: plain 1, South(B) 2, North(A) 3, Win(A&B) 4   (A = first row)
//   white: plain -1, East(B) -2, West(A) -3, Win -4       (A = first column)
// i.e. label = +-(1 + 2*A + B), which is what the 9-plane tensor needs.
// ===========================================================================
template <int NW>
struct HexBitsT {
  uint32_t w[NW];
};
template <int NW>
struct HexParamsT {
  int words;  // = 4*NW + 1, or 4*NW in the folded form (HexT::folded)
  int cols, rows, cells, swap, plain_obs;
  HexBitsT<NW> board, col_first, col_last, row_first, row_last;
};
// The folded record (round 5): where the board leaves five spare bits at the top of each plane's last word
// (cells <= 32*NW - 5: hex(9) 81 of 96, the default 11 x 11 121 of 128, 13 x 13, 15 x 15, 19 x 19 ...), the meta word
// rides in them (4 x 5 = 20 bits: mover 1, result 2, plies 8, first move 9) and a state is 4*NW words instead of
// 4*NW + 1 — 12 instead of 13 for hex(9), 8 % fewer bytes per step both ways.  Which form a batch has is a field of
// its Params (words == 4*NW), a wave-uniform branch in load / store; everything above them sees the same State.
template <int NW>
struct HexT {
  using word_t = uint32_t;
  // boards of up to 128 actions share the engine's 4-word mask (and with it the search kernels); the big boards
  // (13 x 13 ... 19 x 19: NW = 6, 8, 12) carry one mask word per plane word
  static constexpr int kMaskW = NW > kMaskWords ? NW : kMaskWords;
  static constexpr int kMaxWords = 4 * NW + 1;
  static constexpr uint32_t kFoldedLastWordMask = 0x07FFFFFFu;
  using MaskType = MaskT<kMaskW>;
  using Bits = HexBitsT<NW>;
  using Params = HexParamsT<NW>;
  struct State {
    Bits black, white, ea, eb;
    uint32_t meta;
  };
  // ---- multiword bit helpers (cell index grows with word index) ----
  OSG_D static Bits zero() {
    Bits b;
#pragma unroll
    for (int i = 0; i < NW; ++i) b.w[i] = 0;
    return b;
  }
  OSG_D static Bits band(const Bits& a, const Bits& b) {
    Bits r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = a.w[i] & b.w[i];
    return r;
  }
  OSG_D static Bits bor(const Bits& a, const Bits& b) {
    Bits r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = a.w[i] | b.w[i];
    return r;
  }
  OSG_D static Bits bandn(const Bits& a, const Bits& b) {  // a & ~b
    Bits r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = a.w[i] & ~b.w[i];
    return r;
  }
  OSG_D static Bits sel(bool c, const Bits& a, const Bits& b) {  // word-wise select: never a pointer select
    Bits r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = c ? a.w[i] : b.w[i];
    return r;
  }
  OSG_D static bool any(const Bits& a) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) v |= a.w[i];
    return v != 0;
  }
  OSG_D static bool test(const Bits& a, int c) {  // select, not a.w[c >> 5]: keeps Bits in VGPRs
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) v |= (i == (c >> 5)) ? a.w[i] : 0u;
    return (v >> (c & 31)) & 1u;
  }
  OSG_D static Bits single(int c) {
    Bits b = zero();
#pragma unroll
    for (int i = 0; i < NW; ++i) b.w[i] = (i == (c >> 5)) ? (1u << (c & 31)) : 0u;
    return b;
  }
  // cells moved to HIGHER indices by s (0 < s < 32)
  OSG_D static Bits shl(const Bits& a, int s) {
    Bits r;
#pragma unroll
    for (int i = NW - 1; i >= 0; --i) {
      uint32_t lo = (i > 0) ? (a.w[i - 1] >> (32 - s)) : 0u;
      r.w[i] = (a.w[i] << s) | lo;
    }
    return r;
  }
  OSG_D static Bits shr(const Bits& a, int s) {
    Bits r;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      uint32_t hi = (i + 1 < NW) ? (a.w[i + 1] << (32 - s)) : 0u;
      r.w[i] = (a.w[i] >> s) | hi;
    }
    return r;
  }
  OSG_D static int popcount(const Bits& a) {
    int c = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) c += __builtin_popcount(a.w[i]);
    return c;
  }
  // Union of the six neighbours of every cell in `s` (AdjacentCells,
  // hex.cc:316-329: -C, -C+1, +1, +C, +C-1, -1 with the edge guards).
  OSG_D static Bits neighbours(const Params& p, const Bits& s) {
    const int C = p.cols;
    Bits not_e = bandn(s, p.col_last), not_w = bandn(s, p.col_first);
    const Bits east = shl(not_e, 1), west = shr(not_w, 1);   // +1, -1 (both stay inside their row)
#ifndef OSG_HEX_NB6
#define OSG_HEX_NB6 0   // 1: the six-shift form (A/B builds)
#endif
    if (!OSG_HEX_NB6 && C > 1 && C < 32) {
      // -C and -C+1 are ONE shift of (s | east) by C, +C and +C-1 one shift of (s | west): four multiword shifts
      // instead of six (round 6; the flood of an apply runs this per step)
      Bits r = bor(east, west);
      r = bor(r, shr(bor(s, east), C));
      r = bor(r, shl(bor(s, west), C));
      return band(r, p.board);
    }
    Bits r = shr(s, C);                 // -C   (cells in the first row shift out)
    r = bor(r, shl(s, C));              // +C
    r = bor(r, east);                   // +1
    r = bor(r, west);                   // -1
    if (C > 1) {
      r = bor(r, shr(not_e, C - 1));    // -C+1
      r = bor(r, shl(not_w, C - 1));    // +C-1
    }
    return band(r, p.board);
  }

  OSG_D static State initial(const Params&) {
    State s;
    s.black = zero(); s.white = zero(); s.ea = zero(); s.eb = zero();
    s.meta = 0;
    return s;
  }
  OSG_D static bool folded(const Params& p) { return p.words == 4 * NW; }
  // (the record's form as a template argument: the byte-bound step kernel is compiled per form; everything else asks
  // the Params — load / store below)
  template <bool kFold>
  OSG_D static State load_as(const word_t* base, int64_t n, int64_t i) {
    State s;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      s.black.w[k] = base[(0 * NW + k) * n + i];
      s.white.w[k] = base[(1 * NW + k) * n + i];
      s.ea.w[k] = base[(2 * NW + k) * n + i];
      s.eb.w[k] = base[(3 * NW + k) * n + i];
    }
    if constexpr (kFold) {
      const uint32_t pk = (s.black.w[NW - 1] >> 27) | ((s.white.w[NW - 1] >> 27) << 5) | ((s.ea.w[NW - 1] >> 27) << 10) |
                          ((s.eb.w[NW - 1] >> 27) << 15);
      s.black.w[NW - 1] &= kFoldedLastWordMask; s.white.w[NW - 1] &= kFoldedLastWordMask;
      s.ea.w[NW - 1] &= kFoldedLastWordMask; s.eb.w[NW - 1] &= kFoldedLastWordMask;
      s.meta = (pk & 7u) | (((pk >> 3) & 0xFFu) << 8) | ((pk >> 11) << 16);   // mover | result, plies, first move
    } else {
      s.meta = base[(4 * NW) * n + i];
    }
    return s;
  }
  template <bool kFold>
  OSG_D static void store_as(word_t* base, int64_t n, int64_t i, const State& s) {
    uint32_t pk = 0u;
    if constexpr (kFold) pk = (s.meta & 7u) | (((s.meta >> 8) & 0xFFu) << 3) | (((s.meta >> 16) & 0x1FFu) << 11);
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const bool last = kFold && k == NW - 1;
      base[(0 * NW + k) * n + i] = s.black.w[k] | (last ? (pk & 31u) << 27 : 0u);
      base[(1 * NW + k) * n + i] = s.white.w[k] | (last ? ((pk >> 5) & 31u) << 27 : 0u);
      base[(2 * NW + k) * n + i] = s.ea.w[k] | (last ? ((pk >> 10) & 31u) << 27 : 0u);
      base[(3 * NW + k) * n + i] = s.eb.w[k] | (last ? ((pk >> 15) & 31u) << 27 : 0u);
    }
    if constexpr (!kFold) base[(4 * NW) * n + i] = s.meta;
  }
  OSG_D static State load(const Params& p, const word_t* base, int64_t n, int64_t i) {
    return folded(p) ? load_as<true>(base, n, i) : load_as<false>(base, n, i);   // (wave-uniform)
  }
  OSG_D static void store(const Params& p, word_t* base, int64_t n, int64_t i, const State& s) {
    if (folded(p)) store_as<true>(base, n, i, s); else store_as<false>(base, n, i, s);
  }
  OSG_D static int to_move(const State& s) { return s.meta & 1u; }
  OSG_D static int result(const State& s) { return (s.meta >> 1) & 3u; }
  OSG_D static int plies(const State& s) { return (s.meta >> 8) & 0xFFu; }
  OSG_D static bool terminal(const Params&, const State& s) { return result(s) != 0; }  // hex.cc:361
  OSG_D static int current_player(const Params& p, const State& s) {                     // hex.h:96-98
    return terminal(p, s) ? kTerminalPlayer : to_move(s);
  }
  OSG_D static MaskType legal(const Params& p, const State& s) {  // hex.cc:280-293
    MaskType m;
    if (terminal(p, s)) return m;
    Bits empty = bandn(p.board, bor(s.black, s.white));
#pragma unroll
    for (int i = 0; i < NW; ++i) m.w[i] = empty.w[i];
    if (p.swap && plies(s) == 1 && to_move(s) == 1) m.set(p.cells);
    return m;
  }
  // PlayerAndActionToState (hex.cc:108-171) + DoApplyAction (hex.cc:229-278).
  OSG_D static void place(const Params& p, State& s, int player, int move, bool& a, bool& b) {
    Bits cell = single(move);
    // black: first row -> North(A), ELSE IF last row -> South(B); white: first column -> West(A),
    // ELSE IF last column -> East(B) (hex.cc:122-126,146-150).  Written as plain boolean algebra: the
    // if / else-if form made the compiler index {a, b} as a two-byte array in scratch memory.
    const bool on_first = player == 0 ? test(p.row_first, move) : test(p.col_first, move);
    const bool on_last = player == 0 ? test(p.row_last, move) : test(p.col_last, move);
    a = on_first;
    b = !on_first && on_last;
    const Bits own = sel(player == 0, s.black, s.white);
    Bits nb = band(neighbours(p, cell), own);
    // a neighbour labelled exactly A (not Win) / exactly B
    a |= any(bandn(band(nb, s.ea), s.eb));
    b |= any(bandn(band(nb, s.eb), s.ea));
    if (player == 0) s.black = bor(s.black, cell); else s.white = bor(s.white, cell);
    if (a) s.ea = bor(s.ea, cell);
    if (b) s.eb = bor(s.eb, cell);
  }
  OSG_D static void apply(const Params& p, State& s, int move) {
    uint32_t ply = plies(s);
    uint32_t ply_next = ply < 255u ? ply + 1u : 255u;
    if (p.swap && move == p.cells) {  // hex.cc:230-244
      int first = (s.meta >> 16) & 0xFFFFu;   // (16 bits: cells up to 360 on the big boards)
      s.black = zero(); s.ea = zero(); s.eb = zero();  // only the first stone was on the board
      int r = first / p.cols, c = first - r * p.cols;
      int mirrored = c * p.cols + r;
      bool a, b;
      place(p, s, 1, mirrored, a, b);
      s.meta = (s.meta & 0xFFFF0000u) | (ply_next << 8) | 0u;  // black to move
      return;
    }
    int player = to_move(s);
    bool a, b;
    place(p, s, player, move, a, b);
    uint32_t res = 0;
    if (a && b) {
      res = player == 0 ? 1u : 2u;  // Win label; no flood fill (hex.cc:248-252)
    } else if (a || b) {
      // flood the plain same-colour group reachable from the new stone
      const Bits own = sel(player == 0, s.black, s.white);
      Bits plain = bandn(bandn(own, s.ea), s.eb);
      Bits region = zero();
      Bits frontier = single(move);
      for (int it = 0; it < 128; ++it) {
        Bits grow = bandn(band(neighbours(p, frontier), plain), region);
        if (!any(grow)) break;
        region = bor(region, grow);
        frontier = grow;
      }
      if (a) s.ea = bor(s.ea, region); else s.eb = bor(s.eb, region);
    }
    uint32_t first = ply == 0 ? static_cast<uint32_t>(move) : ((s.meta >> 16) & 0xFFFFu);
    s.meta = static_cast<uint32_t>(1 - player) | (res << 1) | (ply_next << 8) | (first << 16);
  }
  // A uniformly random playout from `start` whose WINNER is all that is asked for (RandomRolloutEvaluator inside a search,
  // mcts.cc:43-72; round 6): the same draws and the same moves as the move-by-move playout — legal[rng.below(count)] on
  // the empty cells in ascending order — but no edge labels and no test for the end on the way: the stones are placed
  // until the board is full and the winner is read off the filled board by ONE flood of black's stones from its first
  // row.  A hex game cannot be un-won (the connected side stays connected, the other can no longer connect), so the
  // filled board's winner is the winner at the ply where the reference's loop would have stopped; the draws beyond
  // that ply belong to this playout's own counter stream and are seen by nobody else.  The move-by-move form relabels
  // a group at every stone (neighbour sets, often a flood): ~4 x the instructions.  The swap rule's plies (the first
  // two) run through the generic rules.  Returns 0 = black won, 1 = white won.  Boards with at least two rows and
  // two columns (the entry points refuse the others for playouts).
  template <class RngT>
  OSG_D static int fill_playout_winner(const Params& p, const State& start, RngT& rng) {
    State w = start;
    if (p.swap) {
      for (int g = 0; g < 2 && plies(w) < 2 && !terminal(p, w); ++g) {
        const MaskType m = legal(p, w);
        apply(p, w, select_action(m, static_cast<int>(rng.below(static_cast<uint32_t>(m.count())))));
      }
    }
    if (terminal(p, w)) return result(w) == 1 ? 0 : 1;
    Bits black = w.black;
    Bits empty = bandn(p.board, bor(w.black, w.white));
    int left = popcount(empty);
    bool black_turn = to_move(w) == 0;
    for (; left > 0; --left) {
      int rem = static_cast<int>(rng.below(static_cast<uint32_t>(left)));
      // the rem-th empty cell: its word by a running count (selects, no indexing: the planes stay in registers)
      bool found = false;
      uint32_t word = 0;
      int wi = 0, kin = 0;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int c = __builtin_popcount(empty.w[i]);
        const bool here = !found && rem < c;
        word = here ? empty.w[i] : word;
        wi = here ? i : wi;
        kin = here ? rem : kin;
        found |= here;
        rem -= c;
      }
      const uint32_t bit = 1u << select32(word, kin);
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const uint32_t m = i == wi ? bit : 0u;
        empty.w[i] &= ~m;
        black.w[i] |= black_turn ? m : 0u;
      }
      black_turn = !black_turn;
    }
    Bits reach = band(black, p.row_first), frontier = reach;
    for (int it = 0; it < 32 * NW; ++it) {
      if (any(band(reach, p.row_last))) return 0;
      const Bits grow = bandn(band(neighbours(p, frontier), black), reach);
      if (!any(grow)) return 1;
      reach = bor(reach, grow);
      frontier = grow;
    }
    return any(band(reach, p.row_last)) ? 0 : 1;
  }
  OSG_D static int outcome_code(const Params&, const State& s) { return result(s) == 1 ? 0 : (result(s) == 2 ? 1 : 2); }
  OSG_D static void returns(const Params&, const State& s, double* out) {  // hex.cc:363-365
    double r = result(s) == 1 ? 1.0 : (result(s) == 2 ? -1.0 : 0.0);
    out[0] = r;
    out[1] = -r + 0.0;  // the reference yields -0.0 for a running game; compared as integers
  }
  OSG_D static double chance_prob(const Params&, const State&, int) { return 0.0; }
  OSG_D static int label(const State& s, int cell) {
    int mag = 1 + 2 * static_cast<int>(test(s.ea, cell)) + static_cast<int>(test(s.eb, cell));
    return test(s.black, cell) ? mag : (test(s.white, cell) ? -mag : 0);
  }
  // ObservationTensor, hex.cc:379-398.
  OSG_D static float obs_at(const Params& p, const State& s, int /*player*/, int /*which*/, int idx) {
    if (p.plain_obs) {  // view {3, num_cols, num_rows}[plane, cell / C, cell % C]
      int per = p.cols * p.rows;
      int plane = idx / per, rem = idx - plane * per;
      int i0 = rem / p.rows, i1 = rem - i0 * p.rows;
      // inverse of (cell / C, cell % C) -> i0 * rows + i1
      if (i1 >= p.cols || i0 * p.cols + i1 >= p.cells) return 0.0f;
      int cell = i0 * p.cols + i1;
      if (cell / p.cols != i0) return 0.0f;
      int l = label(s, cell);
      int pl = l == 0 ? 2 : (l < 0 ? 1 : 0);  // CellStateToPlainPlane, hex.cc:76-93
      return pl == plane ? 1.0f : 0.0f;
    }
    int plane = idx / p.cells, cell = idx - plane * p.cells;
    return (label(s, cell) + 4) == plane ? 1.0f : 0.0f;
  }
  // Cells whose label puts them on tensor plane `plane` (hex.cc:392-396: plane = label + 4).
  OSG_D static Bits plane_mask(const Params& p, const State& s, int plane) {
    const int l = plane - 4;
    if (l == 0) return bandn(p.board, bor(s.black, s.white));
    const Bits own = sel(l > 0, s.black, s.white);
    const int mag = l > 0 ? l : -l;  // 1 plain, 2 edge-B only, 3 edge-A only, 4 both
    const Bits a = (mag == 3 || mag == 4) ? band(own, s.ea) : bandn(own, s.ea);
    return (mag == 2 || mag == 4) ? band(a, s.eb) : bandn(a, s.eb);
  }
  // Walks the 9-plane tensor cell by cell: the plane's membership mask sits in a one-word shift
  // register (bit 0 = the current cell), refilled every 32 cells and recomputed at a plane boundary.
  struct ObsCursor {
    Bits m;
    uint32_t cur;
    int plane, cell, idx;
    OSG_D static uint32_t word(const Bits& b, int i) {  // static selects keep Bits in registers
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < NW; ++k) v |= (i == k) ? b.w[k] : 0u;
      return v;
    }
    OSG_D void init(const Params& p, const State& s, int /*player*/, int /*which*/, int idx0) {
      idx = idx0;
      plane = idx0 / p.cells;
      cell = idx0 - plane * p.cells;
      m = p.plain_obs ? zero() : plane_mask(p, s, plane);
      cur = word(m, cell >> 5) >> (cell & 31);
    }
    OSG_D float next(const Params& p, const State& s, int player, int which) {
      if (p.plain_obs) return obs_at(p, s, player, which, idx++);
      // The kernel cuts chunks at plane boundaries (segment = plane), so a cursor stays in one plane.
      const float v = static_cast<float>(cur & 1u);
      cur >>= 1;
      if ((++cell & 31) == 0) cur = word(m, cell >> 5);
      return v;
    }
  };
};

template <class G> struct is_hex : std::false_type {};
template <int NW> struct is_hex<HexT<NW>> : std::true_type {};

}  // namespace osg
#endif  // OSG_GAME_BOARDS_H_
