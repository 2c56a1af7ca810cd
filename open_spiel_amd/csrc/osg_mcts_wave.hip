// algorithms::MCTSBot (open_spiel/algorithms/mcts.{h,cc}), ONE WAVEFRONT PER ROOT.
//
// The 64 lanes of a wavefront cooperate on one search tree:
//   * selection   (mcts.cc:324-341)  children of a node are contiguous; lane l
//                 scores children l and l+64 (UCTValue, mcts.cc:90-101) and a
//                 6-step shuffle butterfly picks the arg-max
//   * expansion   (mcts.cc:281-299)  lane l initialises children l, l+64
//   * evaluation  (mcts.cc:43-72)    n_rollouts playouts spread over the lanes; a
//                 hex playout is ONE wave-parallel random fill of the board
//                 (lane = cell): ballot radix-select of the mover's half, then a
//                 wave-uniform bitboard flood fill decides the winner
//   * backup      (mcts.cc:383-395)  lane d updates the d-th node of the path
//   * MCTS-Solver (mcts.cc:398-434)  lanes scan the children, ballot / shuffle reduce
// The board state itself is wave-uniform (every lane holds the same bitboards), so
// the rule code runs without divergence.
//
// Random streams (shared with the oracle's replay, oracle MCTSBot mode 2):
//   sibling order  the reference shuffles a new node's children and lets the first
//                  maximum win (mcts.cc:294,336).  Here children stay in action order
//                  and ties go to the smallest order_key(seed, root, path, action) —
//                  the same thing as sorting the children by that key first.
//   chance nodes   CounterRng(seed ^ kTreeSalt, root, simulation).unit()
//   rollouts       generic games: rollout r of simulation s plays from
//                  CounterRng(seed, root, s * n_rollouts + r) like the lane layout.
//                  hex (no swap rule): the empty cells are ordered by fill_key; the
//                  player to move takes the ceil(m/2) smallest keys in turn, the
//                  opponent the others in turn — a uniformly random move sequence —
//                  and since a hex winner never changes once a side has connected, the
//                  playout's result is read off the filled board.
//
// Tree storage: node pool per root, ROOT-major (field[root * cap + node]) so a node's
// children are one coalesced load per field.
#include <cmath>

#include "osg_mcts_internal.h"

using namespace osg;

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kMaxPath = 160;

OSG_D int lane_id() { return static_cast<int>(threadIdx.x & 63u); }
template <class T>
OSG_D T uniform(T v) {  // move a wave-uniform 32-bit value into an SGPR
  return static_cast<T>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
}
OSG_D uint64_t uniform64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
OSG_D int wave_count(bool pred) { return __builtin_popcountll(__ballot(pred)); }

struct Cand {  // arg-max candidate: larger value wins, then smaller key
  double v;
  uint64_t key;
  int k;
};
OSG_D bool better(const Cand& a, const Cand& b) { return a.v > b.v || (a.v == b.v && a.key < b.key); }
OSG_D Cand wave_argmax(Cand c) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Cand o;
    o.v = __shfl_xor(c.v, off);
    o.key = __shfl_xor(c.key, off);
    o.k = __shfl_xor(c.k, off);
    if (better(o, c)) c = o;
  }
  return c;
}
OSG_D double wave_max(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}
OSG_D uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t o = __shfl_xor(v, off);
    v = o < v ? o : v;
  }
  return v;
}
OSG_D double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// BestChild ordering (mcts.cc:114-125): outcome, then visits, then total reward; ties -> smallest key.
struct Final {
  double out;
  uint32_t cnt;
  double tot;
  uint64_t key;
  int action;
};
OSG_D bool final_better(const Final& a, const Final& b) {  // a strictly preferred to b
  if (a.out != b.out) return a.out > b.out;
  if (a.cnt != b.cnt) return a.cnt > b.cnt;
  if (a.tot != b.tot) return a.tot > b.tot;
  return a.key < b.key;
}

// --- hex playout as a wave-parallel random fill --------------------------------------------
// Per-lane constants of the board geometry: lane l owns cells l and l + 64; for each it keeps the
// set of its (up to six) neighbours as a 128-bit mask, and whether it lies on black's two edges.
struct HexLane {
  uint64_t nb_lo[2], nb_hi[2];  // neighbours among cells 0-63 / 64-127
  bool first_row[2], last_row[2], on_board[2];
};
template <class G>
OSG_D HexLane hex_lane_setup(const typename G::Params& p) {
  HexLane hl;
  const int lane = lane_id();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cell = lane + 64 * j;
    hl.on_board[j] = cell < p.cells;
    const typename G::Bits nb = G::neighbours(p, G::single(hl.on_board[j] ? cell : 0));
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < static_cast<int>(sizeof(nb.w) / sizeof(nb.w[0])); ++i) w[i] = nb.w[i];
    hl.nb_lo[j] = hl.on_board[j] ? (static_cast<uint64_t>(w[1]) << 32 | w[0]) : 0ull;
    hl.nb_hi[j] = hl.on_board[j] ? (static_cast<uint64_t>(w[3]) << 32 | w[2]) : 0ull;
    hl.first_row[j] = hl.on_board[j] && G::test(p.row_first, cell);
    hl.last_row[j] = hl.on_board[j] && G::test(p.row_last, cell);
  }
  return hl;
}

template <class G>
OSG_D int hex_fill_winner(const typename G::Params& p, const typename G::State& s, uint64_t base, const HexLane& hl) {
  // Lane l owns cells l and l + 64.
  const int lane = lane_id();
  typename G::Bits occ = G::bor(s.black, s.white);
  bool cand[2], sel[2], empty[2];
  uint64_t key[2];
  int m = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cell = lane + 64 * j;
    empty[j] = hl.on_board[j] && !G::test(occ, cell);
    cand[j] = empty[j];
    key[j] = fill_key(base, cell);
    sel[j] = false;
    m += wave_count(cand[j]);
  }
  int remaining = (m + 1) >> 1;  // plies 0, 2, 4, ... belong to the player to move
  // MSB-first radix select of the `remaining` smallest keys among the candidates.
  for (int bit = 63; bit >= 0; --bit) {
    const int live = wave_count(cand[0]) + wave_count(cand[1]);
    if (remaining == 0) break;
    if (live == remaining) {  // everything still undecided is selected
      sel[0] |= cand[0];
      sel[1] |= cand[1];
      remaining = 0;
      break;
    }
    const bool z0 = cand[0] && !((key[0] >> bit) & 1ull), z1 = cand[1] && !((key[1] >> bit) & 1ull);
    const int zeros = wave_count(z0) + wave_count(z1);
    if (remaining <= zeros) {  // the threshold has a 0 here: keys with a 1 are too large
      cand[0] = z0;
      cand[1] = z1;
    } else {  // every 0-key is selected; keep looking among the 1-keys
      sel[0] |= z0;
      sel[1] |= z1;
      remaining -= zeros;
      cand[0] = cand[0] && !z0;
      cand[1] = cand[1] && !z1;
    }
  }
  // The filled board: the mover's new stones are `sel`, the opponent's the other empty cells.
  const int mover = G::to_move(s);
  bool blk[2], reached[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cell = lane + 64 * j;
    blk[j] = hl.on_board[j] && (G::test(s.black, cell) || (empty[j] && (mover == 0 ? sel[j] : !sel[j])));
    reached[j] = blk[j] && hl.first_row[j];
  }
  // Black wins iff its stones join the first row to the last row (hex.cc:108-171 edge labels).
  // Lane-parallel flood: a black cell joins the region when one of its neighbours is in it; the
  // region travels between lanes as two ballot masks.  Stops as soon as the last row is reached.
  for (int it = 0; it < 128; ++it) {
    const uint64_t r0 = __ballot(reached[0]), r1 = __ballot(reached[1]);
    if (__ballot((reached[0] && hl.last_row[0]) || (reached[1] && hl.last_row[1])) != 0ull) return 0;  // black
    const bool g0 = blk[0] && !reached[0] && ((hl.nb_lo[0] & r0) | (hl.nb_hi[0] & r1)) != 0ull;
    const bool g1 = blk[1] && !reached[1] && ((hl.nb_lo[1] & r0) | (hl.nb_hi[1] & r1)) != 0ull;
    if (__ballot(g0 || g1) == 0ull) break;
    reached[0] |= g0;
    reached[1] |= g1;
  }
  return 1;  // white: on a filled board exactly one side connects
}

template <class G, bool kBoard, bool kHexFill>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
k_mcts_wave(typename G::Params p, const typename G::word_t* base, int64_t n, int num_players, int num_actions,
            osg_mcts_cfg cfg, double max_utility, const double* __restrict__ log_table, Pool pool, MctsOut out) {
  __shared__ uint32_t s_path[kWavesPerBlock][kMaxPath];
  const int wave_in_block = static_cast<int>(threadIdx.x >> 6);
  const int64_t r = uniform(static_cast<int>(blockIdx.x * kWavesPerBlock + wave_in_block));
  if (r >= n) return;
  const int lane = lane_id();
  uint32_t* path = s_path[wave_in_block];
  const uint64_t gr = static_cast<uint64_t>(cfg.index_offset + r);
  const int cap = pool.cap;
  uint32_t* META = pool.meta + r * cap;
  uint32_t* FIRST = pool.first + r * cap;
  uint32_t* COUNT = pool.count + r * cap;
  double* TOTAL = pool.total + r * cap;
  const uint64_t obase = order_base(cfg.seed, gr);
  HexLane hl{};
  if constexpr (kHexFill) hl = hex_lane_setup<G>(p);

  const typename G::State root_state = G::load(p, base, n, r);
  const int root_player = G::current_player(p, root_state);
  if (lane == 0) {
    META[0] = make_meta(0xFF, root_player, 0);  // mcts.cc:356-357
    FIRST[0] = 0;
    COUNT[0] = 0;
    TOTAL[0] = 0.0;
  }
  __threadfence_block();
  uint32_t used = 1;
  int sims_done = 0;

  for (int sim = 0; sim < cfg.max_simulations; ++sim) {
    Rng trng(cfg.seed ^ kTreeSalt, gr, static_cast<uint64_t>(sim));
    // ---- ApplyTreePolicy (mcts.cc:273-351) ----
    typename G::State s = root_state;
    uint32_t node = 0;
    int depth = 0;
    uint64_t ph = path_hash_root();
    if (lane == 0) path[0] = 0;
    bool term;
    for (;;) {
      term = G::terminal(p, s);
      const uint32_t cnt = uniform(COUNT[node]);
      if (term || cnt == 0 || depth + 1 >= kMaxPath) break;
      uint32_t meta = uniform(META[node]);
      const int cur = G::current_player(p, s);
      const Mask legal = G::legal(p, s);
      if (m_nchild(meta) == 0) {  // expand: one child per Prior() entry, in action order
        const int c = legal.count();
        if (used + static_cast<uint32_t>(c) > static_cast<uint32_t>(cap)) break;  // pool exhausted: leaf evaluation
        const uint32_t first = used;
        used += c;
        // Children in action order.  Lane l looks at actions l and l + 64: a legal action's slot is its
        // rank among the legal ones (popcount of the mask below it) — the cheap direction of the
        // k <-> action mapping — so the writes are still one compacted, coalesced span.
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int a = lane + 64 * j;
          if (legal.test(a)) {
            int rank = 0;
#pragma unroll
            for (int w = 0; w < kMaskWords; ++w) {
              const int lo = 32 * w;
              if (a >= lo + 32) rank += __builtin_popcount(legal.w[w]);
              else if (a > lo) rank += __builtin_popcount(legal.w[w] & ((1u << (a - lo)) - 1u));
            }
            META[first + rank] = make_meta(a, cur, 0);
            FIRST[first + rank] = 0;
            COUNT[first + rank] = 0;
            TOTAL[first + rank] = 0.0;
          }
        }
        meta = make_meta(m_action(meta), m_player(meta), c) | (meta & 0x00F00000u);
        if (lane == 0) {
          META[node] = meta;
          FIRST[node] = first;
        }
        __threadfence_block();
      }
      const uint32_t first = uniform(FIRST[node]);
      const int c = m_nchild(meta);
      int chosen_k, action;
      if (cur == kChancePlayer) {  // mcts.cc:311-322; children are in outcome order
        action = sample_action_chance<G>(p, s, legal, trng);
        int below = 0;
#pragma unroll
        for (int w = 0; w < kMaskWords; ++w) {
          const int lo = 32 * w;
          if (action >= lo + 32) below += __builtin_popcount(legal.w[w]);
          else if (action > lo) below += __builtin_popcount(legal.w[w] & ((1u << (action - lo)) - 1u));
        }
        chosen_k = below;
      } else {  // arg-max of UCTValue (mcts.cc:90-101), ties to the smallest order key
        uint32_t cm2[2], cc2[2];
        bool unvisited[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = lane + 64 * j;
          cm2[j] = 0;
          cc2[j] = 0;
          unvisited[j] = false;
          if (k < c) {
            cm2[j] = META[first + k];
            cc2[j] = COUNT[first + k];
            unvisited[j] = cc2[j] == 0 && !m_has_outcome(cm2[j]);
          }
        }
        bool t0, t1;  // the candidates holding the maximum value
        const bool puct = cfg.child_selection_policy == 1;
        const uint64_t u0 = __ballot(unvisited[0]), u1 = __ballot(unvisited[1]);
        if (!puct && (u0 | u1) != 0ull) {
          // Some child has never been visited: its value is +infinity (mcts.cc:95), so the maximum is
          // +infinity whatever the others score — no UCT arithmetic needed at this node.
          t0 = unvisited[0];
          t1 = unvisited[1];
        } else {
          const double logn = log_table[cnt];
          const double prior = 1.0 / c, sqrt_n = sqrt(static_cast<double>(cnt));  // PUCT only
          double v2[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int k = lane + 64 * j;
            v2[j] = -INFINITY;
            if (k < c) {
              const double ct = TOTAL[first + k];
              if (m_has_outcome(cm2[j])) v2[j] = outcome_value<kBoard>(cm2[j], cc2[j], ct, m_player(cm2[j]));
              else if (puct) v2[j] = (cc2[j] != 0 ? ct / cc2[j] : 0.0) + cfg.uct_c * prior * sqrt_n / (cc2[j] + 1);
              else v2[j] = ct / cc2[j] + cfg.uct_c * sqrt(logn / cc2[j]);
            }
          }
          const double vmax = wave_max(v2[0] > v2[1] ? v2[0] : v2[1]);  // value only: 2 dwords per butterfly step
          t0 = lane < c && v2[0] == vmax;
          t1 = lane + 64 < c && v2[1] == vmax;
        }
        const uint64_t b0 = __ballot(t0), b1 = __ballot(t1);
        if (__builtin_popcountll(b0) + __builtin_popcountll(b1) == 1) {
          chosen_k = b0 ? __builtin_ctzll(b0) : 64 + __builtin_ctzll(b1);
        } else {  // several maxima: the smallest order key (= first in the shuffled order, mcts.cc:294,336)
          const uint32_t k0 = t0 ? order_key(obase, ph, static_cast<int>(m_action(cm2[0]))) : 0xFFFFFFFFu;
          const uint32_t k1 = t1 ? order_key(obase, ph, static_cast<int>(m_action(cm2[1]))) : 0xFFFFFFFFu;
          const uint32_t kmin = wave_min_u32(k0 < k1 ? k0 : k1);
          const uint64_t w0 = __ballot(t0 && k0 == kmin), w1 = __ballot(t1 && k1 == kmin);
          chosen_k = w0 ? __builtin_ctzll(w0) : 64 + __builtin_ctzll(w1);
        }
        chosen_k = uniform(chosen_k);
        action = static_cast<int>(m_action(uniform(META[first + chosen_k])));
      }
      G::apply(p, s, action);
      node = first + static_cast<uint32_t>(chosen_k);
      ph = path_hash_child(ph, action);
      ++depth;
      if (lane == 0) path[depth] = node;
    }
    // ---- evaluate (mcts.cc:372-381) ----
    double returns[kMaxPlayers];
    bool solved = false;
    if (term) {
      G::returns(p, s, returns);
      uint32_t meta = uniform(META[node]) | (1u << 20) | (1u << 23);
      if (kBoard) meta = (meta & ~(3u << 21)) | (static_cast<uint32_t>(static_cast<int>(returns[0]) + 1) << 21);
      if (lane == 0) META[node] = meta;
      solved = cfg.solve != 0;
    } else if constexpr (kHexFill) {
      double r0 = 0.0;
      for (int ro = 0; ro < cfg.n_rollouts; ++ro) {
        const uint64_t fb = fill_base(cfg.seed, gr, static_cast<uint64_t>(sim) * cfg.n_rollouts + ro);
        r0 += hex_fill_winner<G>(p, s, fb, hl) == 0 ? 1.0 : -1.0;
      }
      returns[0] = r0 / cfg.n_rollouts;
      returns[1] = -returns[0] + 0.0;
    } else {  // RandomRolloutEvaluator::Evaluate (mcts.cc:43-72), rollouts over the lanes
      for (int q = 0; q < num_players; ++q) returns[q] = 0.0;
      for (int ro0 = 0; ro0 < cfg.n_rollouts; ro0 += 64) {
        const int ro = ro0 + lane;
        double rr[kMaxPlayers];
        for (int q = 0; q < num_players; ++q) rr[q] = 0.0;
        if (ro < cfg.n_rollouts) {
          Rng rng(cfg.seed, gr, static_cast<uint64_t>(sim) * cfg.n_rollouts + ro);
          typename G::State w = s;
          while (!G::terminal(p, w)) {
            const Mask m = G::legal(p, w);
            G::apply(p, w, sample_action<G>(p, w, m, G::current_player(p, w), rng));
          }
          G::returns(p, w, rr);
        }
        // Returns() of these games are multiples of 0.5 with small magnitude: sums are exact in
        // any order, so the butterfly equals the reference's sequential accumulation.
        for (int q = 0; q < num_players; ++q) returns[q] += wave_sum(rr[q]);
      }
      for (int q = 0; q < num_players; ++q) returns[q] /= cfg.n_rollouts;
    }
    // ---- backup (mcts.cc:383-395): lane d owns the d-th node of the visit path ----
    for (int d = lane; d <= depth; d += 64) {
      const uint32_t v = path[d];
      int pl = m_player(META[v]);
      for (int up = d; pl == kChancePlayer;) {  // skip chance-player entries (poker trees)
        if (--up < 0) { pl = 0; break; }
        pl = m_player(META[path[up]]);
      }
      double rv = returns[0];
      for (int q = 1; q < num_players; ++q) rv = (pl == q) ? returns[q] : rv;
      TOTAL[v] += rv;
      COUNT[v] += 1;
    }
    __threadfence_block();
    // ---- MCTS-Solver (mcts.cc:398-434), leaf to root ----
    if (kBoard && solved) {
      for (int d = depth; d >= 0 && solved; --d) {
        const uint32_t v = path[d];
        const uint32_t meta = uniform(META[v]);
        const int c = m_nchild(meta);
        if (c == 0) continue;
        const uint32_t first = uniform(FIRST[v]);
        const int mover = m_player(uniform(META[first]));
        bool unsolved_here = false;
        Cand best{-INFINITY, ~0ull, 0};
        for (int k = lane; k < c; k += 64) {
          const uint32_t cm = META[first + k];
          if (!m_has_outcome(cm)) { unsolved_here = true; continue; }
          Cand me{outcome_value<true>(cm, 1, 0.0, mover), static_cast<uint64_t>(k), m_code(cm)};
          if (better(me, best)) best = me;
        }
        const bool all_solved = __ballot(unsolved_here) == 0ull;
        best = wave_argmax(best);
        const bool have = best.v > -INFINITY;
        if (have && (all_solved || best.v == max_utility)) {
          if (lane == 0) META[v] = (meta & ~(3u << 21)) | (1u << 20) | (static_cast<uint32_t>(uniform(best.k)) << 21);
        } else {
          solved = false;
        }
      }
      __threadfence_block();
    }
    ++sims_done;
    const uint32_t rm = uniform(META[0]);
    if (m_has_outcome(rm) || m_nchild(rm) == 1) break;  // mcts.cc:437-440 (a terminal root has an outcome too)
  }

  // ---- results: BestChild (mcts.cc:114-143) + per-action statistics ----
  const uint32_t rm = uniform(META[0]);
  const int c = m_nchild(rm);
  const uint32_t first = uniform(FIRST[0]);
  for (int a = lane; a < num_actions; a += 64) {
    if (out.child_visits) out.child_visits[r * num_actions + a] = 0;
    if (out.child_reward) out.child_reward[r * num_actions + a] = 0.0;
    if (out.child_outcome) out.child_outcome[r * num_actions + a] = 3;
  }
  __threadfence_block();
  Final best{-INFINITY, 0u, 0.0, ~0ull, -1};
  const uint64_t root_ph = path_hash_root();
  for (int k = lane; k < c; k += 64) {
    const uint32_t cm = META[first + k];
    const uint32_t cc = COUNT[first + k];
    const double ct = TOTAL[first + k];
    const int a = static_cast<int>(m_action(cm));
    const bool has = m_has_outcome(cm);
    const int pl = m_player(cm);
    const double o = (has && pl >= 0 && cc > 0) ? outcome_value<kBoard>(cm, cc, ct, pl)
                                                : ((has && kBoard && pl >= 0) ? outcome_value<true>(cm, 1, 0.0, pl) : 0.0);
    Final me{o, cc, ct, order_key(obase, root_ph, a), a};
    if (best.action < 0 || final_better(me, best)) best = me;
    if (a < num_actions) {
      if (out.child_visits) out.child_visits[r * num_actions + a] = static_cast<int32_t>(cc);
      if (out.child_reward) out.child_reward[r * num_actions + a] = ct;
      if (out.child_outcome) {
        int8_t code = 2;
        if (has && kBoard && root_player >= 0) code = static_cast<int8_t>(outcome_value<true>(cm, 1, 0.0, root_player));
        out.child_outcome[r * num_actions + a] = code;
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Final o;
    o.out = __shfl_xor(best.out, off);
    o.cnt = __shfl_xor(best.cnt, off);
    o.tot = __shfl_xor(best.tot, off);
    o.key = __shfl_xor(best.key, off);
    o.action = __shfl_xor(best.action, off);
    if (o.action >= 0 && (best.action < 0 || final_better(o, best))) best = o;
  }
  if (lane == 0) {
    if (out.best_action) out.best_action[r] = best.action;
    if (out.root_stats) {
      out.root_stats[r * 4 + 0] = static_cast<double>(COUNT[0]);
      out.root_stats[r * 4 + 1] = static_cast<double>(used);
      out.root_stats[r * 4 + 2] =
          (kBoard && m_has_outcome(rm) && root_player >= 0) ? outcome_value<true>(rm, 1, 0.0, root_player) : NAN;
      out.root_stats[r * 4 + 3] = static_cast<double>(sims_done);
    }
  }
}

template <class G, bool kBoard, bool kHexFill>
void launch(const typename G::Params& P, const osg_batch* roots, const osg_mcts_cfg& cfg, const double* d_logs,
            const Pool& pool, const MctsOut& out) {
  const osg_game_desc& d = roots->spec.desc;
  const unsigned grid = static_cast<unsigned>((roots->n + kWavesPerBlock - 1) / kWavesPerBlock);
  k_mcts_wave<G, kBoard, kHexFill><<<dim3(grid), dim3(64 * kWavesPerBlock), 0, roots->ctx->stream>>>(
      P, static_cast<const typename G::word_t*>(roots->d_words), roots->n, d.num_players, d.num_distinct_actions, cfg,
      d.max_utility, d_logs, pool, out);
}

}  // namespace

namespace osg {

int launch_mcts_wave(const osg_batch* roots, const osg_mcts_cfg& cfg, const double* d_logs, const Pool& pool,
                     const MctsOut& out) {
  const GameSpec& spec = roots->spec;
  switch (spec.desc.game_kind) {
    case kTtt: launch<Ttt, true, false>(spec.ttt, roots, cfg, d_logs, pool, out); break;
    case kC4:
      if (spec.c4_std) launch<C4Std, true, false>(spec.c4, roots, cfg, d_logs, pool, out);
      else launch<C4, true, false>(spec.c4, roots, cfg, d_logs, pool, out);
      break;
    case kKuhn: launch<Kuhn, false, false>(spec.kuhn, roots, cfg, d_logs, pool, out); break;
    case kLeduc: launch<Leduc, false, false>(spec.leduc, roots, cfg, d_logs, pool, out); break;
    case kHex: {
      // The random-fill playout needs "legal moves == empty cells": not with the swap rule.
#define OSG_HEX_CASE(NW, member)                                                                   \
  if (spec.member.swap) launch<HexT<NW>, true, false>(spec.member, roots, cfg, d_logs, pool, out); \
  else launch<HexT<NW>, true, true>(spec.member, roots, cfg, d_logs, pool, out)
      switch (spec.hex_nw) {
        case 1: OSG_HEX_CASE(1, hex1); break;
        case 2: OSG_HEX_CASE(2, hex2); break;
        case 3: OSG_HEX_CASE(3, hex3); break;
        default: OSG_HEX_CASE(4, hex4); break;
      }
#undef OSG_HEX_CASE
      break;
    }
    default: return set_error(OSG_ERR_INVALID, "bad game kind");
  }
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

}  // namespace osg
