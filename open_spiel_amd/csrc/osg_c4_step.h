// The fused connect_four step (6 x 7 board, 4 in a row) as one straight-line
// function over the two packed planes of a state — legality of the action,
// ApplyAction (connect_four.cc:130-145), the result of the game
// (connect_four.cc:138-142,163-209), the successor's LegalActions
// (connect_four.cc:147-156) and the status byte — written against the 32-bit
// halves the vector ALU works on.  Host + device so that a CPU test can drive
// exactly this code over random games (tests/test_c4_step_host.py); the kernel
// is k_step_c4std2 / k_step_c4std in osg_kernels.hip.
//
// Layout (osg_game_boards.h C4T<6,7,4>): bit = col*7 + row, row 6 of every
// column an always-empty sentinel; plane 0's top byte = result flags (bit 0
// terminal, bits 1-2 outcome: 0 x won, 1 o won, 2 draw).
#ifndef OSG_C4_STEP_H_
#define OSG_C4_STEP_H_

#include "osg_common.h"

namespace osg {

// run-of-four test on one player's stones: bit i of the result is set iff
// b[i], b[i+d], b[i+2d], b[i+3d] are all set.
OSG_HD uint64_t c4_fours(uint64_t b) {
  uint64_t hit, m;
  m = b & (b >> 1); hit = m & (m >> 2);
  m = b & (b >> 7); hit |= m & (m >> 14);
  m = b & (b >> 6); hit |= m & (m >> 12);
  m = b & (b >> 8); hit |= m & (m >> 16);
  return hit;
}

// Open columns (bit c = column c has room) from the occupied cells: the top cells sit at bits 5 + 7c.
// Columns 0-3 lie in the low word (bits 5, 12, 19, 26), columns 4-6 in the high word (bits 1, 8, 15);
// each group is gathered by one 24-bit multiply (full rate on the vector ALU, unlike the 64-bit product):
// bit 7j times 2^(T - 6k) lands on T + j for j == k and on distinct other bits otherwise, so there are no
// carries into the target field.
OSG_HD uint32_t c4_open_columns(uint32_t all_lo, uint32_t all_hi) {
  const uint32_t tl = (~all_lo >> 5) & 0x00204081u;                       // bits 0, 7, 14, 21
  const uint32_t th = (~all_hi >> 1) & 0x00004081u;                       // bits 0, 7, 14
  const uint32_t ml = tl * ((1u << 21) | (1u << 15) | (1u << 9) | (1u << 3));   // columns 0-3 at bits 21-24
  const uint32_t mh = th * ((1u << 14) | (1u << 8) | (1u << 2));                // columns 4-6 at bits 14-16
  return ((ml >> 21) & 0xFu) | (((mh >> 14) & 0x7u) << 4);
}

// One fused step.  x carries the flags byte on entry and on exit.  action 0xFF = "no action" (status /
// legal mask of the state as it is).  Returns mask | status << 8.
OSG_HD uint32_t c4_fused_step(uint64_t& x, uint64_t& o, uint32_t a) {
  const uint32_t flags = static_cast<uint32_t>(x >> 56);
  const uint64_t X = x & ((1ull << 56) - 1ull), O = o;
  const uint64_t all = X | O;
  const uint32_t stones = static_cast<uint32_t>(__builtin_popcountll(all));
  const uint32_t mover = stones & 1u;
  const bool valid = a < 7u;
  const uint32_t sh = valid ? a * 7u : 0u;
  const uint64_t cell = (all + (1ull << sh)) & (0x3Full << sh);             // lowest empty cell, 0 if the column is full
  const bool apply = valid & ((flags & 1u) == 0u) & (cell != 0ull);
  const uint64_t put = apply ? cell : 0ull;
  const uint64_t to_x = mover ? 0ull : ~0ull;                               // all ones when x moves
  const uint64_t nx = X | (put & to_x), no = O | (put & ~to_x);
  const uint64_t b = mover ? no : nx;                                       // only the mover can have a new line
  const bool win = c4_fours(b) != 0ull;
  const uint32_t stones_after = stones + (apply ? 1u : 0u);
  const bool done = win | (stones_after == 42u);                            // IsFull: all 42 cells taken
  const uint32_t fresh = done ? (((win ? mover : 2u) << 1) | 1u) : 0u;
  const uint32_t nflags = apply ? fresh : flags;
  const uint64_t nall = nx | no;
  const uint32_t open = c4_open_columns(static_cast<uint32_t>(nall), static_cast<uint32_t>(nall >> 32));
  const bool over = (nflags & 1u) != 0u;
  const uint32_t st_run = (stones_after & 1u) + 1u;                         // player to move + 1
  const uint32_t st_over = 0x80u | (nflags >> 1);
  const uint32_t st = (over ? st_over : st_run) | (((a != 0xFFu) & !apply) ? 0x40u : 0u);
  x = nx | (static_cast<uint64_t>(nflags) << 56);
  o = no;
  return (over ? 0u : open) | (st << 8);
}

}  // namespace osg
#endif  // OSG_C4_STEP_H_
