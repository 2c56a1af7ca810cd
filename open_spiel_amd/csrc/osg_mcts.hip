// algorithms::MCTSBot (open_spiel/algorithms/mcts.{h,cc}) for a batch of roots.
//
// Mapping: ONE LANE PER ROOT.  A search is a strictly sequential chain
// (simulation s+1 reads the statistics simulation s wrote), and with
// RandomRolloutEvaluator the chain is dominated by the random playout, which is
// itself sequential.  The SIMT-efficient unit of parallelism is therefore the
// root: 64 independent searches per wavefront, every lane running the same
// select / expand / rollout / backup code on its own tree.  (A wave-per-root
// layout with children spread over lanes keeps 63 of 64 lanes idle during the
// playout, which is >90 % of the work; see DESIGN.md.)
//
// Tree storage in HBM: a node pool of `cap` nodes per root, struct-of-arrays and — since round 6 — root-major
// (field[root * cap + node]): a lane scans and writes runs of its own tree (rounds 1-5: root-minor, which coalesces
// across lanes only near the root; see the macros in k_mcts).
//   meta   u32  action | (player+1)<<8 | nchild<<12 | has_outcome<<20 | code<<21 | terminal<<23
//   first  u32  index of the first child (children are contiguous)
//   parent u32
//   count  u32  explore_count
//   total  f64  total_reward
#include <algorithm>
#include <cmath>
#include <vector>

#include "osg_mcts_internal.h"

using namespace osg;

namespace {

template <class G, bool kBoard>
__global__ void __launch_bounds__(kBlockM)
k_mcts(typename G::Params p, const typename G::word_t* base, int64_t n, int num_players, int num_actions,
       osg_mcts_cfg cfg, double max_utility, const double* __restrict__ log_table, Pool pool, int32_t* best_action,
       int32_t* child_visits, double* child_reward, int8_t* child_outcome, double* root_stats) {
  // (games with more than 255 actions — hex above 15 x 15 — use the nine-bit action / child-count fields)
  constexpr bool kWide = G::kMaskW > kMaskWords;
  using LegalMask = MaskT<G::kMaskW>;
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlockM + threadIdx.x;
  if (r >= n) return;
#ifndef OSG_MCTS_LDS_SHUFFLE
#define OSG_MCTS_LDS_SHUFFLE 1
#endif
  // the lane's column of the shuffle stage: entry k at sh[k * kBlockM] (dynamic LDS: widest node x kBlockM entries)
  using act_t = uint8_t;   // (boards of up to six plane words: at most 193 actions)
  extern __shared__ unsigned char s_shuffle[];
  act_t* sh = reinterpret_cast<act_t*>(s_shuffle) + threadIdx.x;
  (void)sh;
  const uint64_t gr = static_cast<uint64_t>(cfg.index_offset + r);
  // Node i of this lane's root in every plane: i * NR + RB.  Round 6: ROOT-MAJOR (field[root * cap + node], the wave layout's
  // form: NR = 1, RB = root * cap) — a root's nodes are contiguous, so a lane's scan of a node's children and the stores of
  // an expansion walk ONE run of memory (32 children's counts per 128-byte line) instead of one line per child and field;
  // root-minor (field[node * n_roots + root]: NR = n_roots, RB = root; rounds 1-5) coalesces across lanes only while the
  // lanes stand on the same node index, i.e. near the root.  2^16 roots (profiles/r06zf_*, r06zi_*): hex(9) x 512
  // simulations 4.5e8 -> 7.2e8, 11 x 11 2.2e8 -> 4.4e8, 13 x 13 1.2e8 -> 2.9e8, tic_tac_toe x 1000 8.2e8 -> 9.9e8,
  // connect_four x 256 1.35e9 -> 1.42e9 simulations/s; SHORT searches of narrow games stay near the root and keep
  // root-minor (connect_four x 32: 2.78e9 against 2.27e9, tic_tac_toe x 100: 2.09e9 against 1.89e9): the host picks
  // (pool.root_major; OSG_MCTS_ROOT_MAJOR=0/1 forces one).
  const int64_t NR = pool.root_major ? 1 : pool.n_roots, RB = pool.root_major ? r * static_cast<int64_t>(pool.cap) : r;
#define META(i) pool.meta[static_cast<int64_t>(i) * NR + RB]
#define FIRST(i) pool.first[static_cast<int64_t>(i) * NR + RB]
#define PARENT(i) pool.parent[static_cast<int64_t>(i) * NR + RB]
#define COUNT(i) pool.count[static_cast<int64_t>(i) * NR + RB]
#define TOTAL(i) pool.total[static_cast<int64_t>(i) * NR + RB]

  const typename G::State root_state = G::load(p, base, n, r);
  const int root_player = G::current_player(p, root_state);
  META(0) = mw_make<kWide>(0xFF, root_player, 0);  // mcts.cc:356-357: root = (kInvalidAction, CurrentPlayer(), 1)
  FIRST(0) = 0; PARENT(0) = kNoNode; COUNT(0) = 0; TOTAL(0) = 0.0;
  uint32_t used = 1;       // = the reference's nodes_: 1 + the children blocks allocated (mcts.cc:299,354)
  int gc_limit = kMinGcLimit;
  int sims_done = 0;
#define REMAP(i) pool.remap[static_cast<int64_t>(i) * NR + RB]

  for (int sim = 0; sim < cfg.max_simulations; ++sim) {
    Rng trng(cfg.seed ^ kTreeSalt, gr, static_cast<uint64_t>(sim));
    // ---- ApplyTreePolicy (mcts.cc:273-351) ----
    typename G::State s = root_state;
    uint32_t node = 0;
    bool term;
    for (;;) {
      term = G::terminal(p, s);
      const uint32_t cnt = COUNT(node);
      if (term || cnt == 0) break;
      uint32_t meta = META(node);
      const int cur = G::current_player(p, s);
      if (mw_nchild<kWide>(meta) == 0) {  // expand: children = Prior(state), shuffled (mcts.cc:281-299)
        const LegalMask legal = G::legal(p, s);
        const int c = legal.count();
        // slots exhausted (unreachable unless the caller's HBM could not hold max_nodes + slack), or a record that is
        // not terminal and has no legal action (only an uploaded inconsistent one): evaluate as a leaf
        if (c == 0 || used + static_cast<uint32_t>(c) > static_cast<uint32_t>(pool.cap)) break;
        const uint32_t first = used;
        used += c;
        if constexpr (OSG_MCTS_LDS_SHUFFLE && G::kMaskW < 8) {
        // Round 6: the children's order is formed in LDS and every header written ONCE.  The shuffle used to run on
        // the pool — per child two loads and two stores of the lane's own, scattered node words on top of the five
        // initialising stores — which is most of what an expansion of a wide node (hex: ~100 children) costs.
        // Same draws, same swaps, same order.
        {
          int k = 0;
#pragma unroll
          for (int w = 0; w < G::kMaskW; ++w)
            for (uint32_t bits = legal.w[w]; bits != 0u; bits &= bits - 1u)
              sh[(k++) * kBlockM] = static_cast<act_t>(32 * w + __builtin_ctz(bits));
        }
        for (int i = c - 1; i >= 1; --i) {  // Fisher-Yates == std::shuffle's role (order only)
          const int j = static_cast<int>(trng.below(static_cast<uint32_t>(i + 1)));
          const act_t ai = sh[i * kBlockM], aj = sh[j * kBlockM];
          sh[i * kBlockM] = aj;
          sh[j * kBlockM] = ai;
        }
        for (int k = 0; k < c; ++k) {
          META(first + k) = mw_make<kWide>(static_cast<int>(sh[k * kBlockM]), cur, 0);
          FIRST(first + k) = 0; PARENT(first + k) = node; COUNT(first + k) = 0; TOTAL(first + k) = 0.0;
        }
        } else {   // (boards from 16 x 16 on: two-byte entries x 362 children x 64 lanes would cost the kernel a wavefront per CU — 19 x 19: 2.9e8 -> 1.5e8)
        for (int k = 0; k < c; ++k) {
          META(first + k) = mw_make<kWide>(select_action(legal, k), cur, 0);
          FIRST(first + k) = 0; PARENT(first + k) = node; COUNT(first + k) = 0; TOTAL(first + k) = 0.0;
        }
        for (int i = c - 1; i >= 1; --i) {  // Fisher-Yates == std::shuffle's role (order only)
          const int j = static_cast<int>(trng.below(static_cast<uint32_t>(i + 1)));
          const uint32_t mi = META(first + i), mj = META(first + j);
          META(first + i) = mj;
          META(first + j) = mi;
        }
        }
        meta = mw_make<kWide>(static_cast<int>(mw_action<kWide>(meta)), m_player(meta), c) | (meta & kMetaOutcomeBits);
        META(node) = meta;
        FIRST(node) = first;
      }
      const uint32_t first = FIRST(node);
      const int c = mw_nchild<kWide>(meta);
      uint32_t chosen = first, chosen_meta = 0;
      bool have_meta = false;
      if (cur == kChancePlayer) {  // mcts.cc:311-322
        const LegalMask legal = G::legal(p, s);
        const int a = sample_action_chance<G>(p, s, legal, trng);
        for (int k = 0; k < c; ++k)
          if (static_cast<int>(mw_action<kWide>(META(first + k))) == a) { chosen = first + k; break; }
      } else {  // arg-max of UCTValue, first maximum wins (mcts.cc:324-341, 90-101)
        double best = -INFINITY;
        const double logn = log_table[cnt];
        const bool puct = cfg.child_selection_policy == 1;
        const double prior = 1.0 / c;                              // Prior(): uniform over the legal actions
        const double sqrt_n = sqrt(static_cast<double>(cnt));
        // a search is ONE chain of dependent, scattered loads: the children's statistics are requested eight children at a
        // time with clamped indices (24 independent loads, one round trip) instead of child by child behind the
        // branches of the value formula (see k_mcts_advance, osg_mcts_step.hip)
        constexpr int kChunk = 8;
        // Round 6.  Under UCT a child that was never visited scores +infinity (mcts.cc:95) and the FIRST such child wins:
        // nothing behind it can score higher, nothing before it needs its value formed.  While a node still has
        // unvisited children — most visits of a wide node (hex: ~100 children, visited in order) — the scan therefore
        // only looks for the first one: counts and headers, no totals, none of the ~60 fp64 instructions per child.
        // Whether that is the case is read off the LAST child (the children are first visited in order, so it is the
        // last to go); if it was visited the full scan below runs, which is exact whatever the order was.
        bool scan_only = false;
        if (!puct) {
          const uint32_t at = first + static_cast<uint32_t>(c - 1);
          scan_only = COUNT(at) == 0 && !m_has_outcome(META(at));
        }
        bool settled = false;   // an unvisited child was found: the arg-max is decided
        for (int k0 = 0; k0 < c && !settled; k0 += kChunk) {
          uint32_t cm[kChunk], cc[kChunk];
          double ct[kChunk];
#pragma unroll
          for (int j = 0; j < kChunk; ++j) {
            const uint32_t at = first + static_cast<uint32_t>(k0 + j < c ? k0 + j : c - 1);
            cm[j] = META(at);
            cc[j] = COUNT(at);
            ct[j] = scan_only ? 0.0 : TOTAL(at);
          }
#pragma unroll
          for (int j = 0; j < kChunk; ++j) {
            if (k0 + j >= c || settled) continue;
            const bool unvisited = !puct && cc[j] == 0 && !m_has_outcome(cm[j]);
            if (scan_only && !unvisited) continue;
            double v;
            if (m_has_outcome(cm[j])) v = outcome_value<kBoard>(cm[j], cc[j], ct[j], m_player(cm[j]));
            else if (puct) v = (cc[j] != 0 ? ct[j] / cc[j] : 0.0) + cfg.uct_c * prior * sqrt_n / (cc[j] + 1);  // mcts.cc:103-112
            else if (cc[j] == 0) v = INFINITY;
            else v = ct[j] / cc[j] + cfg.uct_c * sqrt(logn / cc[j]);
            if (v > best) { best = v; chosen = first + static_cast<uint32_t>(k0 + j); chosen_meta = cm[j]; have_meta = true; }
            settled = unvisited;
          }
        }
      }
      G::apply(p, s, static_cast<int>(mw_action<kWide>(have_meta ? chosen_meta : META(chosen))));
      node = chosen;
    }
    // ---- evaluate (mcts.cc:372-381) ----
    double returns[kMaxPlayers];
    bool solved = false;
    if (term) {
      G::returns(p, s, returns);
      uint32_t meta = META(node) | (1u << 20) | (1u << 23);
      if (kBoard) meta = (meta & ~(3u << 21)) | (static_cast<uint32_t>(static_cast<int>(returns[0]) + 1) << 21);
      META(node) = meta;
      solved = cfg.solve != 0;
    } else {  // RandomRolloutEvaluator::Evaluate (mcts.cc:43-72)
      for (int q = 0; q < num_players; ++q) returns[q] = 0.0;
      for (int ro = 0; ro < cfg.n_rollouts; ++ro) {
        Rng rng(cfg.seed, gr, static_cast<uint64_t>(sim) * cfg.n_rollouts + ro);
        double rr[kMaxPlayers];
        playout_returns<G>(p, s, rng, rr);
        for (int q = 0; q < num_players; ++q) returns[q] += rr[q];
      }
      for (int q = 0; q < num_players; ++q) returns[q] /= cfg.n_rollouts;
    }
    // ---- backup (mcts.cc:383-435) ----
    for (uint32_t v = node; v != kNoNode; v = PARENT(v)) {
      uint32_t meta = META(v);
      int pl = m_player(meta);
      for (uint32_t up = v; pl == kChancePlayer;) {  // chance node: use the parent decision player
        up = PARENT(up);
        if (up == kNoNode) { pl = 0; break; }
        pl = m_player(META(up));
      }
      TOTAL(v) += returns[(pl < 0 || pl >= num_players) ? 0 : pl];  // (a terminal root has no player)
      COUNT(v) += 1;
      if (kBoard && solved && mw_nchild<kWide>(meta) > 0) {  // MCTS-Solver, max^n over proven children
        const uint32_t first = FIRST(v);
        const int c = mw_nchild<kWide>(meta);
        const int mover = m_player(META(first));
        bool all_solved = true, have = false;
        double best = 0.0;
        int best_code = 0;
        for (int k = 0; k < c; ++k) {
          const uint32_t cm = META(first + k);
          if (!m_has_outcome(cm)) { all_solved = false; continue; }
          const double val = outcome_value<true>(cm, 1, 0.0, mover);
          if (!have || val > best) { have = true; best = val; best_code = m_code(cm); }
        }
        if (have && (all_solved || best == max_utility)) {
          META(v) = (meta & ~(3u << 21)) | (1u << 20) | (static_cast<uint32_t>(best_code) << 21);
        } else {
          solved = false;
        }
      } else if (!kBoard) {
        solved = false;
      }
    }
    ++sims_done;
    const uint32_t rm = META(0);
    if ((m_has_outcome(rm) && !m_terminal(rm)) || mw_nchild<kWide>(rm) == 1) break;  // mcts.cc:437-440
    if (m_terminal(rm)) break;  // a terminal root: nothing to search
    // ---- GarbageCollect (mcts.cc:441-482): when nodes_ >= max_nodes_, every node with explore_count <
    // gc_limit_ loses its children.  Visit counts never grow from parent to child, so a node survives
    // exactly when its parent's count reaches the limit; the survivors are compacted in index order
    // (children blocks stay contiguous, parents stay below their children).
    if (pool.gc_nodes > 1 && used >= static_cast<uint32_t>(pool.gc_nodes)) {
      const uint32_t limit = static_cast<uint32_t>(gc_limit);
      uint32_t w = 1;
      REMAP(0) = 0;
      for (uint32_t i = 1; i < used; ++i) {
        const bool alive = COUNT(PARENT(i)) >= limit;
        REMAP(i) = alive ? w : kNoNode;
        w += alive ? 1u : 0u;
      }
      for (uint32_t i = 0; i < used; ++i) {
        const uint32_t to = REMAP(i);
        if (to == kNoNode) continue;
        uint32_t meta = META(i), first = FIRST(i);
        const uint32_t cnt = COUNT(i), par = PARENT(i);
        const double tot = TOTAL(i);
        if (mw_nchild<kWide>(meta) > 0) {
          if (cnt < limit) { meta = mw_clear_children<kWide>(meta); first = 0; }   // children.clear(); the outcome stays
          else first = REMAP(first);
        }
        META(to) = meta; FIRST(to) = first; COUNT(to) = cnt; TOTAL(to) = tot;
        PARENT(to) = i == 0 ? kNoNode : REMAP(par);
      }
      used = w;
      gc_limit = next_gc_limit(gc_limit, used, pool.gc_nodes);
    }
  }

  // ---- results: BestChild by CompareFinal (mcts.cc:114-143) + per-action statistics ----
  const uint32_t rm = META(0);
  const int c = mw_nchild<kWide>(rm);
  const uint32_t first = FIRST(0);
  if (child_visits) for (int a = 0; a < num_actions; ++a) child_visits[r * num_actions + a] = 0;
  if (child_reward) for (int a = 0; a < num_actions; ++a) child_reward[r * num_actions + a] = 0.0;
  if (child_outcome) for (int a = 0; a < num_actions; ++a) child_outcome[r * num_actions + a] = 3;
  int best = -1;
  double b_out = 0.0, b_tot = 0.0;
  uint32_t b_cnt = 0;
  for (int k = 0; k < c; ++k) {
    const uint32_t cm = META(first + k);
    const uint32_t cc = COUNT(first + k);
    const double ct = TOTAL(first + k);
    const int a = static_cast<int>(mw_action<kWide>(cm));
    const bool has = m_has_outcome(cm);
    const int pl = m_player(cm);
    const double out = (has && pl >= 0 && cc > 0) ? outcome_value<kBoard>(cm, cc, ct, pl)
                                                  : ((has && kBoard && pl >= 0) ? outcome_value<true>(cm, 1, 0.0, pl) : 0.0);
    // strict "a < b" ordering, first maximum kept (std::max_element)
    const bool better = best < 0 || (b_out != out ? b_out < out : (b_cnt != cc ? b_cnt < cc : b_tot < ct));
    if (better) { best = a; b_out = out; b_cnt = cc; b_tot = ct; }
    if (a < num_actions) {
      if (child_visits) child_visits[r * num_actions + a] = static_cast<int32_t>(cc);
      if (child_reward) child_reward[r * num_actions + a] = ct;
      if (child_outcome) {
        int8_t code = 2;
        if (has && kBoard && root_player >= 0) code = static_cast<int8_t>(outcome_value<true>(cm, 1, 0.0, root_player));
        child_outcome[r * num_actions + a] = code;
      }
    }
  }
  if (best_action) best_action[r] = best;
  if (root_stats) {
    root_stats[r * 4 + 0] = static_cast<double>(COUNT(0));
    root_stats[r * 4 + 1] = static_cast<double>(used);
    root_stats[r * 4 + 2] = (kBoard && m_has_outcome(rm) && root_player >= 0) ? outcome_value<true>(rm, 1, 0.0, root_player)
                                                                              : NAN;
    root_stats[r * 4 + 3] = static_cast<double>(sims_done);
  }
#undef REMAP
#undef META
#undef FIRST
#undef PARENT
#undef COUNT
#undef TOTAL
}

}  // namespace

extern "C" int osg_mcts_search(const osg_batch* roots, const osg_mcts_cfg* cfg_in, int32_t* best_action,
                               int32_t* child_visits, double* child_reward, int8_t* child_outcome,
                               double* root_stats, int on_host) {
  if (!roots || !cfg_in) return set_error(OSG_ERR_INVALID, "osg_mcts_search: null argument");
  osg_ctx* ctx = roots->ctx;
  osg_mcts_cfg cfg = *cfg_in;
  const osg_game_desc& d = roots->spec.desc;
  const bool board = d.game_kind <= kHex;
  if (cfg.max_simulations < 1 || cfg.n_rollouts < 1) return set_error(OSG_ERR_INVALID, "max_simulations and n_rollouts must be >= 1");
  if (int rc = refuse_endless_playouts(roots->spec, "osg_mcts_search")) return rc;
  if (roots->spec.desc.num_distinct_actions > kMaxSearchActions)
    return set_error(OSG_ERR_UNSUPPORTED, "osg_mcts_search: a node holds up to 511 actions");
  // the games beyond the 4-word mask / the two-word records (hex above 11 x 11 with the swap rule, connect_four above 64
  // board bits, leduc_poker with 4+ players) are searched by the lane-per-root kernel; the wave-per-root kernel keeps a hex
  // position in scalar registers (two 64-cell sets per colour up to 128 cells, three / four / six on the boards above)
  const bool wide_game = roots->spec.desc.num_distinct_actions > 32 * kMaskWords || roots->spec.c4_wide || roots->spec.leduc_big;
  if (cfg.solve && !board)
    return set_error(OSG_ERR_UNSUPPORTED, "solve=true needs win/draw/loss outcomes (tic_tac_toe, connect_four, hex)");
  int layout = cfg.layout;
  if (layout != 0 && layout != 1 && layout != 2) return set_error(OSG_ERR_INVALID, "osg_mcts_cfg.layout must be 0, 1 or 2");
  // (round 6: the hex boards above 128 cells too — kS = 3 / 4 / 6 cell sets per colour in scalar registers and as many
  // child slots per lane — in the random-fill form only: no swap rule, and at most 64 columns so that a cell's neighbours
  // lie in the adjacent cell sets)
  bool wide_hex_fill = false;
  if (wide_game && d.game_kind == kHex && d.num_distinct_actions == d.obs_shape[1] * d.obs_shape[2]) {
    int rows = 0, cols = 0, cells = 0;
    hex_dims(roots->spec, &rows, &cols, &cells);
    wide_hex_fill = rows >= 2 && cols >= 2 && cols <= 64;
  }
  if (layout == 0)  // auto: the wave layout where its parallel playout applies (hex without the swap rule; measured on the
                    // boards above 128 cells too: 2.1-2.6 x the lane layout on 12 x 12 ... 16 x 16, profiles/r06zn_*)
    layout = ((!wide_game || wide_hex_fill) && d.game_kind == kHex && d.num_distinct_actions == d.obs_shape[1] * d.obs_shape[2]) ? 2 : 1;
  if (layout == 2 && wide_game && !wide_hex_fill)
    return set_error(OSG_ERR_UNSUPPORTED, "osg_mcts_search: layout 2 (a wavefront per root) serves games of up to 128 actions and "
                                          "the larger hex boards without the swap rule (2 ... 64 columns); "
                                          "this game is searched with layout 1 (or 0 = automatic)");
  if (cfg.child_selection_policy != 0 && cfg.child_selection_policy != 1)
    return set_error(OSG_ERR_INVALID, "osg_mcts_cfg.child_selection_policy must be 0 (UCT) or 1 (PUCT)");
  const int64_t n = roots->n;
  const int A = d.num_distinct_actions;
  const int widest = A > d.max_chance_outcomes ? A : d.max_chance_outcomes;
  // Node pool: `cap` slots per root.  A simulation expands at most one node (<= widest children), so
  // 1 + max_simulations * widest slots can never run out.  cfg.max_nodes > 0 is the reference's max_nodes_
  // (mcts.cc:214: (max_memory_mb << 20) / sizeof(SearchNode) + 1): the search garbage-collects when
  // nodes_ >= max_nodes_ (mcts.cc:441-482), and the pool gets the slots one search can hold between two
  // collections (a collection that frees nothing is followed by ever stricter ones: gc_limit_ grows by a
  // quarter each time, so at most ~log_1.25(max_simulations / 5) consecutive collections fail).
  // cfg.max_nodes <= 0: no limit from the caller: the pool gets the slots that can never run out when they fit
  // in 60 % of the free HBM (288 GB: 2^16 hex(9) roots x 1024 simulations need 130 GB of address space and
  // touch a twentieth of it), otherwise what fits, and a tree that outgrows it is collected like the
  // reference collects at that size.
  // (node indices are 32-bit with 0xFFFFFFFF reserved: a tree never holds more than 2^30 slots)
  const int64_t never = std::min<int64_t>(1 + static_cast<int64_t>(cfg.max_simulations) * widest, int64_t{1} << 30);
  const int64_t slack = 32 * static_cast<int64_t>(widest);
  int64_t cap, gc_nodes = 0;
  if (cfg.max_nodes > 0 && cfg.max_nodes < never) {
    gc_nodes = cfg.max_nodes > 2 ? cfg.max_nodes : 2;
    cap = std::min<int64_t>(never, gc_nodes + slack);
  } else {
    cap = never;
  }
  if (cap < 1 + widest) cap = 1 + widest;
  if (static_cast<size_t>(cap) * n * (gc_nodes > 0 ? 28 : 24) > ctx->mcts_pool_bytes) {
    size_t free_b = 0, total_b = 0;
    OSG_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += ctx->mcts_pool_bytes;  // the old pool is released before the new one is allocated
    if (gc_nodes > 0) {
      if (static_cast<size_t>(cap) * n * 28 > free_b * 9 / 10)
        return set_error(OSG_ERR_NOMEM, "osg_mcts_search: max_nodes slots per root do not fit the free HBM");
    } else if (static_cast<size_t>(cap) * n * 24 > free_b * 6 / 10) {
      cap = static_cast<int64_t>(free_b * 6 / 10 / (static_cast<size_t>(n) * 28));
      if (cap < 2 + 2 * widest) cap = 2 + 2 * widest;
      gc_nodes = cap - widest;  // collect before a simulation's expansion could overrun the slots
    }
  }
  const size_t per_node = gc_nodes > 0 ? 28 : 24;
  cfg.max_nodes = static_cast<int32_t>(cap);
  const size_t slots = static_cast<size_t>(cap) * n;
  const size_t pool_bytes = slots * per_node;
  if (pool_bytes > ctx->mcts_pool_bytes) {
    OSG_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_mcts_pool) OSG_HIP(hipFree(ctx->d_mcts_pool));
    ctx->d_mcts_pool = nullptr;
    ctx->mcts_pool_bytes = 0;
    hipError_t e = hipMalloc(&ctx->d_mcts_pool, pool_bytes);
    if (e != hipSuccess) return set_error(OSG_ERR_NOMEM, std::string("MCTS node pool: ") + hipGetErrorString(e));
    ctx->mcts_pool_bytes = pool_bytes;
  }
  char* pool_mem = static_cast<char*>(ctx->d_mcts_pool);
  Pool pool;
  pool.total = reinterpret_cast<double*>(pool_mem);
  pool.meta = reinterpret_cast<uint32_t*>(pool_mem + slots * 8);
  pool.first = pool.meta + slots;
  pool.parent = pool.first + slots;
  pool.count = pool.parent + slots;
  pool.remap = gc_nodes > 0 ? pool.count + slots : nullptr;
  pool.n_roots = n;
  {
    const char* e = std::getenv("OSG_MCTS_ROOT_MAJOR");
    pool.root_major = e ? (e[0] == '1' ? 1 : 0) : ((widest <= 16 && cfg.max_simulations <= 128) ? 0 : 1);
  }
  pool.cap = static_cast<int>(cap);
  pool.gc_nodes = static_cast<int>(gc_nodes);

  // log(parent explore_count) from the host libm: the CPU oracle (and the reference) call
  // std::log, so sharing the table makes UCT values bit-equal.  Cached in the context.
  if (cfg.max_simulations + 2 > ctx->mcts_logs_n) {
    const int want = cfg.max_simulations + 2;
    std::vector<double> logs(want);
    logs[0] = 0.0;
    for (int i = 1; i < want; ++i) logs[i] = std::log(static_cast<double>(i));
    OSG_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_mcts_logs) OSG_HIP(hipFree(ctx->d_mcts_logs));
    ctx->d_mcts_logs = nullptr;
    ctx->mcts_logs_n = 0;
    OSG_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_mcts_logs), want * sizeof(double)));
    OSG_HIP(hipMemcpy(ctx->d_mcts_logs, logs.data(), want * sizeof(double), hipMemcpyHostToDevice));
    ctx->mcts_logs_n = want;
  }
  const double* d_logs = ctx->d_mcts_logs;

  // host-side outputs are staged through scratch
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~static_cast<size_t>(255); return o; };
  const size_t o_best = carve(sizeof(int32_t) * n), o_vis = carve(sizeof(int32_t) * n * A),
               o_rew = carve(sizeof(double) * n * A), o_out = carve(static_cast<size_t>(n) * A),
               o_stats = carve(sizeof(double) * n * 4);
  int32_t* d_best = best_action; int32_t* d_vis = child_visits; double* d_rew = child_reward;
  int8_t* d_out = child_outcome; double* d_stats = root_stats;
  if (on_host) {
    void* scratch = nullptr;
    int rc = osg_ctx_scratch(ctx, off, &scratch);
    if (rc) return rc;
    char* sc = static_cast<char*>(scratch);
    d_best = best_action ? reinterpret_cast<int32_t*>(sc + o_best) : nullptr;
    d_vis = child_visits ? reinterpret_cast<int32_t*>(sc + o_vis) : nullptr;
    d_rew = child_reward ? reinterpret_cast<double*>(sc + o_rew) : nullptr;
    d_out = child_outcome ? reinterpret_cast<int8_t*>(sc + o_out) : nullptr;
    d_stats = root_stats ? reinterpret_cast<double*>(sc + o_stats) : nullptr;
  }
  if (layout == 2) {
    MctsOut out{d_best, d_vis, d_rew, d_out, d_stats};
    int rc = launch_mcts_wave(roots, cfg, d_logs, pool, out);
    if (rc) return rc;
  } else {
    const unsigned grid = static_cast<unsigned>((n + kBlockM - 1) / kBlockM);
    // the expansion's shuffle stage: one entry per child of the widest node and lane (two bytes above 255 actions)
    const size_t shuffle_lds = A > 32 * 6 ? 0 : static_cast<size_t>(widest) * kBlockM;   // (HexT<8>, HexT<12> shuffle on the pool)
    if (board) {
      OSG_DISPATCH_WIDE(roots->spec, k_mcts<G, true><<<dim3(grid), dim3(kBlockM), shuffle_lds, ctx->stream>>>(
                                    P, static_cast<const typename G::word_t*>(roots->d_words), n, d.num_players, A, cfg,
                                    d.max_utility, d_logs, pool, d_best, d_vis, d_rew, d_out, d_stats));
    } else {
      OSG_DISPATCH_WIDE(roots->spec, k_mcts<G, false><<<dim3(grid), dim3(kBlockM), shuffle_lds, ctx->stream>>>(
                                    P, static_cast<const typename G::word_t*>(roots->d_words), n, d.num_players, A, cfg,
                                    d.max_utility, d_logs, pool, d_best, d_vis, d_rew, d_out, d_stats));
    }
    OSG_HIP(hipGetLastError());
  }
  if (on_host) {
    if (best_action) OSG_HIP(hipMemcpyAsync(best_action, d_best, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (child_visits) OSG_HIP(hipMemcpyAsync(child_visits, d_vis, sizeof(int32_t) * n * A, hipMemcpyDeviceToHost, ctx->stream));
    if (child_reward) OSG_HIP(hipMemcpyAsync(child_reward, d_rew, sizeof(double) * n * A, hipMemcpyDeviceToHost, ctx->stream));
    if (child_outcome) OSG_HIP(hipMemcpyAsync(child_outcome, d_out, static_cast<size_t>(n) * A, hipMemcpyDeviceToHost, ctx->stream));
    if (root_stats) OSG_HIP(hipMemcpyAsync(root_stats, d_stats, sizeof(double) * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    OSG_HIP(hipStreamSynchronize(ctx->stream));
  }
  return OSG_OK;
}
