// Declarations shared by the two MCTS layouts (osg_mcts.hip: one lane per root,
// osg_mcts_wave.hip: one wavefront per root).
#ifndef OSG_MCTS_INTERNAL_H_
#define OSG_MCTS_INTERNAL_H_

#include <type_traits>

#include "osg_internal.h"

namespace osg {

#ifndef OSG_MCTS_LANE_BLOCK
#define OSG_MCTS_LANE_BLOCK 64
#endif
constexpr int kBlockM = OSG_MCTS_LANE_BLOCK;  // one wave per block: searches differ in length, keep blocks small
constexpr uint64_t kTreeSalt = 0x7265655F73616C74ULL;  // stream separation for the tree-policy RNG
constexpr uint32_t kNoNode = 0xFFFFFFFFu;

struct Pool {
  uint32_t* meta;
  uint32_t* first;
  uint32_t* parent;
  uint32_t* count;
  double* total;
  uint32_t* remap;   // old -> new node index during garbage collection; null when GC cannot trigger
  int64_t n_roots;
  int cap;           // slots per root
  int gc_nodes;      // the reference's max_nodes_ (mcts.cc:214): garbage-collect when nodes_ >= gc_nodes; 0 = never
  int root_major = 1;  // lane-per-root kernel: a root's nodes contiguous (1) or field[node * n_roots + root] (0); the wave kernel is root-major
};
constexpr int kMinGcLimit = 5;  // mcts.cc:30 MIN_GC_LIMIT

// gc_limit_ *= (nodes_ > max_nodes_ / 2 ? 1.25 : 0.9), as an int, at least MIN_GC_LIMIT (mcts.cc:456-458)
OSG_D int next_gc_limit(int gc_limit, uint32_t nodes, int max_nodes) {
  const int g = static_cast<int>(gc_limit * (static_cast<int>(nodes) > max_nodes / 2 ? 1.25 : 0.9));
  return g > kMinGcLimit ? g : kMinGcLimit;
}

OSG_D uint32_t m_action(uint32_t m) { return m & 0xFFu; }
OSG_D int m_player(uint32_t m) { return static_cast<int>((m >> 8) & 15u) - 1; }
OSG_D int m_nchild(uint32_t m) { return static_cast<int>((m >> 12) & 0xFFu); }
OSG_D bool m_has_outcome(uint32_t m) { return (m >> 20) & 1u; }
OSG_D int m_code(uint32_t m) { return static_cast<int>((m >> 21) & 3u); }  // p0 value + 1
OSG_D bool m_terminal(uint32_t m) { return (m >> 23) & 1u; }
OSG_D uint32_t make_meta(int action, int player, int nchild) {
  return static_cast<uint32_t>(action & 0xFF) | ((static_cast<uint32_t>(player + 1) & 15u) << 8) |
         (static_cast<uint32_t>(nchild) << 12);
}

// Games with more than 255 actions (hex above 15 x 15: the lane-per-root kernels serve boards up to 19 x 19) keep the
// ninth bit of the action in bit 24 and the ninth bit of the child count in bit 25 of the same word; both are zero
// for every other game, so a decoder that always reads them (the host's) is right for all.  The wave-per-root
// kernel takes the narrow accessors above for boards of up to 128 cells and these for the larger hex boards (round 6).
template <bool kWide> OSG_HD uint32_t mw_action(uint32_t m) { return kWide ? ((m & 0xFFu) | ((m >> 16) & 0x100u)) : (m & 0xFFu); }
template <bool kWide> OSG_HD int mw_nchild(uint32_t m) {
  return static_cast<int>(kWide ? (((m >> 12) & 0xFFu) | ((m >> 17) & 0x100u)) : ((m >> 12) & 0xFFu));
}
template <bool kWide> OSG_HD uint32_t mw_make(int action, int player, int nchild) {
  uint32_t m = static_cast<uint32_t>(action & 0xFF) | ((static_cast<uint32_t>(player + 1) & 15u) << 8) |
               (static_cast<uint32_t>(nchild & 0xFF) << 12);
  if (kWide) m |= (static_cast<uint32_t>(action & 0x100) << 16) | (static_cast<uint32_t>(nchild & 0x100) << 17);
  return m;
}
template <bool kWide> OSG_HD uint32_t mw_clear_children(uint32_t m) { return m & ~((0xFFu << 12) | (kWide ? (1u << 25) : 0u)); }
constexpr uint32_t kMetaOutcomeBits = 0x00F00000u;   // has_outcome | code | terminal: what an expansion keeps
constexpr int kMaxSearchActions = 511;               // nine bits

// outcome[player] of a node that has one (mcts.cc:90-93): exact for the board
// games (code), total/N for terminal nodes of the poker games.
template <bool kBoard>
OSG_D double outcome_value(uint32_t meta, uint32_t count, double total, int player) {
  if (kBoard) {
    // negate as an integer: Returns() of a draw is {0, 0}, never -0.0 (mcts.cc:398-434 stores Returns())
    const int c0 = m_code(meta) - 1;
    return static_cast<double>(player == 0 ? c0 : -c0);
  }
  return total / static_cast<double>(count);
}


// Wave-wide maximum of an fp32 value as a DPP reduction (gfx9 row_shr 1, 2, 4, 8, row_bcast 15 / 31; identity -infinity),
// handed back wave-uniform.  All 64 lanes must be active.  (The fp32 filter of the UCT arg-max: k_mcts_wave's
// select_child, k_mcts_advance's lockstep search.)
#ifndef OSG_UCT_FILTER_STEP
#define OSG_UCT_FILTER_STEP 1   // k_mcts_advance's lockstep (one-root) search through the filter (1) or always fp64 (0)
#endif
#ifndef OSG_COOP_BACKUP
#define OSG_COOP_BACKUP 1       // the one-root search's backup by one lane per path node (1) or up the parent links (0)
#endif
template <int kCtrl, int kRowMask>
OSG_D float dpp_maxf_step_(float v) {
  const int o = __builtin_amdgcn_update_dpp(static_cast<int>(0xFF800000u), __float_as_int(v), kCtrl, kRowMask, 0xf, false);
  return fmaxf(__int_as_float(o), v);
}
OSG_D float wave_max_f32_dpp(float v) {
  v = dpp_maxf_step_<0x111, 0xf>(v);
  v = dpp_maxf_step_<0x112, 0xf>(v);
  v = dpp_maxf_step_<0x114, 0xf>(v);
  v = dpp_maxf_step_<0x118, 0xf>(v);
  v = dpp_maxf_step_<0x142, 0xa>(v);
  v = dpp_maxf_step_<0x143, 0xc>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// One playout of RandomRolloutEvaluator::Evaluate (mcts.cc:45-56) from `s` on `rng`: Returns() of the finished game in
// rr.  hex: the winner from the filled board (HexT::fill_playout_winner, round 6 — the same draws and moves without the
// edge labels; OSG_HEX_FILL_PLAYOUT=0 at build time keeps the move-by-move rules); the other games move by move.
#ifndef OSG_HEX_FILL_PLAYOUT
#define OSG_HEX_FILL_PLAYOUT 1
#endif
// The k-th (0-based) set bit of a 9-bit set; k < popcount(x).  (select32 without its two widest halvings.)
OSG_D int select9(uint32_t x, int k) {
  const int c8 = __builtin_popcount(x & 0xFFu);
  if (k >= c8) return 8;
  int pos = 0;
#pragma unroll
  for (int width = 4; width >= 1; width >>= 1) {
    const uint32_t low = x & ((1u << width) - 1u);
    const int c = __builtin_popcount(low);
    if (k >= c) { k -= c; x >>= width; pos += width; }
    else x = low;
  }
  return pos;
}
#ifndef OSG_TTT_PLAYOUT
#define OSG_TTT_PLAYOUT 1   // tic_tac_toe playouts test the mover's lines only (1) or run the generic rule calls (0)
#endif
template <class G>
OSG_D void playout_returns(const typename G::Params& p, const typename G::State& s, Rng& rng, double* rr) {
  if constexpr (std::is_same<G, Ttt>::value && OSG_TTT_PLAYOUT != 0) {
    // The same draws and moves as the generic loop below (rng.below(number of empty cells), the k-th empty cell), with
    // the rules' work cut to what a move can change: only the player who just moved can have completed a line, and the
    // board is full after nine stones (tic_tac_toe.cc:109-136, 215-227).
    typename G::State w = s;
    if (!G::terminal(p, w)) {
      uint32_t occ = w.x | w.o;
      int n = __builtin_popcount(occ);
      for (; n < 9; ++n) {
        const uint32_t empties = ~occ & 0x1FFu;
        const uint32_t bit = 1u << select9(empties, static_cast<int>(rng.below(static_cast<uint32_t>(9 - n))));
        occ |= bit;
        if (n & 1) { w.o |= bit; if (G::line(w.o)) break; }
        else { w.x |= bit; if (G::line(w.x)) break; }
      }
    }
    G::returns(p, w, rr);
  } else if constexpr (is_hex<G>::value && OSG_HEX_FILL_PLAYOUT) {
    const double r0 = G::fill_playout_winner(p, s, rng) == 0 ? 1.0 : -1.0;
    rr[0] = r0;
    rr[1] = -r0;
  } else {
    typename G::State w = s;
    for (int ply = 0; ply < kMaxPlayoutPlies && !G::terminal(p, w); ++ply) {
      const MaskT<G::kMaskW> m = G::legal(p, w);
      G::apply(p, w, sample_action<G>(p, w, m, G::current_player(p, w), rng));
    }
    G::returns(p, w, rr);
  }
}

// Device output pointers of a search (any may be null).
struct MctsOut {
  int32_t* best_action;
  int32_t* child_visits;
  double* child_reward;
  int8_t* child_outcome;
  double* root_stats;
};

// The wave-per-root search as a work queue: a persistent grid whose wavefronts take the next ticket from a counter
// and search root order[ticket] (order == nullptr: root = ticket).  Results stay keyed by the root's index.
constexpr int kQueueBuckets = 385;          // cost key = number of legal actions at the root (<= 384: twelve plane words)
constexpr int kQueueHeader = 512;           // ints ahead of the order array: the ticket, then the buckets
struct WaveQueue {
  int32_t* ticket;        // [1], zeroed before the launch
  const int32_t* order;   // [n] or nullptr
};

// Launches the wave-per-root kernel on the context's stream (osg_mcts_wave.hip).
int launch_mcts_wave(const osg_batch* roots, const osg_mcts_cfg& cfg, const double* d_log_table, const Pool& pool,
                     const MctsOut& out);

}  // namespace osg
#endif  // OSG_MCTS_INTERNAL_H_
