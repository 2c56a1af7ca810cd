// The fused tic_tac_toe step as one straight-line function over the packed HBM word of a state (bits 0-8 x
// stones, bits 16-24 o stones, osg_game_boards.h Ttt) — legality of the action, ApplyAction
// (tic_tac_toe.cc:128-136), IsTerminal / the outcome (tic_tac_toe.cc:109-120,215-227), the successor's
// LegalActions (tic_tac_toe.cc:138-148) and the status byte.  BOTH players' eight lines are tested in one pass
// over the packed word: the two 9-bit boards sit 16 bits apart and no shift below reaches from one into a
// tested bit of the other.  Host + device so that a CPU test can drive exactly this code over random games
// (tests/test_c4_step_host.py); the kernel is k_step_vec<Ttt, ...> in osg_kernels.hip.
#ifndef OSG_TTT_STEP_H_
#define OSG_TTT_STEP_H_

#include "osg_common.h"

namespace osg {

// bit p (x) / bit 16 + p (o) of the result is set iff that player holds a line anchored at cell p:
// rows at p = 0, 3, 6 (cells p, p+1, p+2), columns at p = 0, 1, 2 (p, p+3, p+6), the diagonal 0-4-8 and the
// anti-diagonal 2-4-6 at p = 0.  Non-zero half = BoardHasLine(player) (tic_tac_toe.cc:109-120).
OSG_HD uint32_t ttt_lines(uint32_t w) {
  const uint32_t rows = w & (w >> 1) & (w >> 2) & 0x00490049u;
  const uint32_t cols = w & (w >> 3) & (w >> 6) & 0x00070007u;
  const uint32_t dia = w & (w >> 4) & (w >> 8) & 0x00010001u;
  const uint32_t ant = (w >> 2) & (w >> 4) & (w >> 6) & 0x00010001u;
  return rows | cols | dia | ant;
}

// One fused step.  action 0xFF = "no action" (status / legal mask of the state as it is).
// Returns legal mask (9 bits) | status << 16.
OSG_HD uint32_t ttt_fused_step(uint32_t& word, uint32_t a) {
  const uint32_t w = word & 0x01FF01FFu;
  const uint32_t occ = (w | (w >> 16)) & 0x1FFu;
  const uint32_t plies = static_cast<uint32_t>(__builtin_popcount(w));  // the two boards are disjoint
  const bool over_before = (ttt_lines(w) != 0u) | (plies == 9u);
  const bool valid = a < 9u;
  const uint32_t cell = valid ? 1u << a : 0u;
  const bool apply = valid & !over_before & ((occ & cell) == 0u);
  const uint32_t mover = plies & 1u;  // player 0 (x) starts
  const uint32_t put = apply ? cell : 0u;
  const uint32_t nw = w | (put << (16u * mover));
  const uint32_t lines = ttt_lines(nw);
  const uint32_t nplies = plies + (apply ? 1u : 0u);
  const bool over = (lines != 0u) | (nplies == 9u);
  const uint32_t outcome = (lines & 0xFFFFu) ? 0u : ((lines >> 16) ? 1u : 2u);  // tic_tac_toe.cc:219-227
  const uint32_t mask = over ? 0u : (~(occ | put) & 0x1FFu);
  const uint32_t st = (over ? (0x80u | outcome) : ((nplies & 1u) + 1u)) | (((a != 0xFFu) & !apply) ? 0x40u : 0u);
  word = nw;
  return mask | (st << 16);
}

}  // namespace osg
#endif  // OSG_TTT_STEP_H_
