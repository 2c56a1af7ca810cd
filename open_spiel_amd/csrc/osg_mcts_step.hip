// algorithms::MCTSBot (open_spiel/algorithms/mcts.{h,cc}) with the Evaluator OUTSIDE the kernel: search
// trees that persist between launches, advanced until every search needs its evaluator — Prior(state) of
// the node it is about to expand (mcts.cc:281-283) or Evaluate(state) of the leaf it has reached
// (mcts.cc:377-380) — and resumed with the answers.  This is the shape of alpha_zero_torch/vpevaluator.{h,cc}
// turned around for a batch: thousands of searches park their leaves in one [n] batch of states, ONE
// network forward over osg_observation of that batch answers all of them, the searches resume.
//
// One lane per root (the layout of osg_mcts.hip: pool node-major / root-minor, the same counter streams —
// tree-policy draws from Rng(seed ^ kTreeSalt, root, simulation): the sibling shuffle and the chance
// sampling — so a search driven through this file with the rollout evaluator is, draw for draw, the fused
// kernel's search).  Everything else follows mcts.cc as osg_mcts.hip does: lazily expanded children with
// their priors, UCT / PUCT, chance nodes sampled, backup with the stored parent player, MCTS-Solver,
// early exit, node budget with GarbageCollect.  Extras of MCTSBot's constructor served here:
// dont_return_chance_node (mcts.cc:279), priors with Dirichlet noise at the root (the host mixes the noise
// into the root's prior, mcts.cc:284-292), max_wall_clock_time (the host checks its clock between rounds).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "osg_mcts_internal.h"

using namespace osg;

struct osg_mcts_tree {
  osg_ctx* ctx = nullptr;
  osg_batch* roots = nullptr;   // private copy of the root states
  osg_mcts_cfg cfg{};
  int flags = 0;
  int64_t n = 0;
  int cap = 0, gc_nodes = 0, A = 0, P = 0, widest = 0;
  double max_utility = 0;
  bool board = false;
  char* d_mem = nullptr;        // pool planes + per-root search state
  size_t bytes = 0;
  double* d_logs = nullptr;
  int logs_n = 0;
  // the tree's own request / answer buffers, for hosts that hold no device memory (osg_mcts_tree_advance_host)
  double* d_own_prior = nullptr;   // [n, A]
  double* d_own_value = nullptr;   // [n, P]
  uint8_t* d_own_request = nullptr;
  // flag 8: the evaluator answers a value request with the leaf's PRIOR as well (one network forward gives both,
  // vpevaluator.cc:60-85 caches them per state): the prior is kept per (root, simulation) until the leaf is expanded
  double* d_stash = nullptr;       // [n, stash_slots, A]
  int stash_slots = 0;
};

namespace {

enum Phase : uint8_t { kNewSimulation = 0, kWantPrior = 1, kWantValue = 2, kFinished = 3 };
constexpr int kScanChunk = 8;  // children whose statistics are requested together in the descent

struct StepPool {
  double* total;    // [cap, n]
  double* prior;    // [cap, n]
  uint32_t* meta;
  uint32_t* first;
  uint32_t* parent;
  uint32_t* count;
  uint32_t* remap;
  // per-root search state
  uint64_t* rng;    // tree-policy stream position of the running simulation
  uint32_t* used;
  uint32_t* node;   // the node a parked search waits at
  int32_t* gc_limit;
  int32_t* sims;
  uint8_t* phase;
  int64_t n;
  int cap, gc_nodes;
  // where node i of root r lives in every plane: root-minor (field[node * n + root]) by default — this kernel runs ONE
  // simulation per launch for every root, the lanes stand on the same node indices near the root most of the time and
  // root-minor coalesces those accesses; root-major (a root's nodes contiguous: what k_mcts gained a factor 2 from in
  // round 6, whole searches in one launch) measured SLOWER here — hex(9) 1.39e8 -> 1.1e8, connect_four with a network
  // 1.97e8 -> 1.70e8 simulations/s (profiles/r06zg_*, r06zh_*).  OSG_STEP_ROOT_MAJOR=1 selects it.
  int root_major;
  OSG_HD int64_t at(uint32_t i, int64_t r) const {
    return root_major ? r * static_cast<int64_t>(cap) + static_cast<int64_t>(i) : static_cast<int64_t>(i) * n + r;
  }
  double* stash;      // [n, stash_slots, A] or null (flag 8)
  int stash_slots;
};

// ONE root (what MCTSBot::Step / MCTSearch always ask for) with RandomRolloutEvaluator in the launch: the search
// itself is one lane's chain of dependent steps, but its leaf evaluation — n_rollouts independent playouts, each on
// its own counter stream Rng(seed, root, s * n_rollouts + ro) — is not.  kCoop launches one workgroup of TWO
// wavefronts: lane 0 of the first walks the tree as in the batch form, the second plays the playouts of every leaf,
// one per lane, and hands the sum back through LDS (returns of the five games are integers: the sum does not depend
// on the order).  Same streams, same values as the sequential form — 20 playouts in the time of one.
template <class G>
struct CoopBox {
  typename G::State state;     // the leaf to evaluate
  uint32_t seq, done, exit, sim;
  uint32_t used;               // nodes of the tree when the search left (the write-back's extent)
  double sum[kMaxPlayers];
};
OSG_D uint32_t coop_load(const uint32_t* at) { return __hip_atomic_load(at, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
OSG_D void coop_store(uint32_t* at, uint32_t v) { __hip_atomic_store(at, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
struct CoopLeave {  // whichever way the searching lane leaves the kernel, the playout wavefront is told
  uint32_t* seq;
  uint32_t* exit;
  uint32_t* used_out;      // where the playout wavefront finds how many nodes to write back from LDS
  const uint32_t* used;    // the searching lane's node count
  OSG_D ~CoopLeave() {
    if (seq) {
      *used_out = *used;
      coop_store(exit, 1u);
      coop_store(seq, coop_load(seq) + 1u);
    }
  }
};
// The tree of a ONE-ROOT search in LDS: a simulation of such a search is a chain of dependent node reads and
// read-modify-writes — the descent's child scans, the backup — and from the HBM pool every link is an L2 round trip
// (~10 us of them per simulation: 7.3e4 simulations/s against the 1.55e5 of one host core).  The first kLdsNodes
// nodes (24 B each: meta, first child, parent, count, total reward; the prior is never read back in this form) live in LDS for the length of the launch;
// nodes beyond that stay in the pool (the accessors pick by index: both are flat addresses), so trees of any size
// work.  All 128 threads copy the existing nodes in at the start; the playout wavefront writes them back when the
// searching lane leaves.
constexpr int kLdsNodes = 6144;   // 24 B each: 144 KiB
struct LdsTree {
  double* total;
  uint32_t* meta;
  uint32_t* first;
  uint32_t* parent;
  uint32_t* count;
};
OSG_D LdsTree lds_tree(void* base) {
  LdsTree t;
  t.total = static_cast<double*>(base);
  t.meta = reinterpret_cast<uint32_t*>(t.total + kLdsNodes);
  t.first = t.meta + kLdsNodes;
  t.parent = t.first + kLdsNodes;
  t.count = t.parent + kLdsNodes;
  return t;
}
constexpr size_t kLdsTreeBytes = static_cast<size_t>(kLdsNodes) * (sizeof(double) + 4 * sizeof(uint32_t));
// One field of one node: in LDS (kCoop and a low index) or in the pool.
#ifndef OSG_NODEREF_MODE
#define OSG_NODEREF_MODE 1   // round 6: 12.39 -> 11.68 ms per 1000-simulation tic_tac_toe search (profiles/r06d_one_root_noderef_ab.txt), parity green
#endif
template <class T, bool kCoop>
struct NodeRef {
  T* l;
  T* g;
  bool in_lds;
  // kCoop: lane 0 alone stores and all 64 lanes read the field back, so the accesses are VOLATILE there: with plain
  // ones the compiler may serve a lane that did not store from a value it loaded earlier (e.g. after COUNT(v) += 1),
  // and the replicated search state of the lanes would part ways.  (The other form has one lane per search: plain.)
  // OSG_NODEREF_MODE 1 (A/B, round 6): plain accesses, and a compiler-level memory barrier after lane 0's store — every
  // later read is loaded again (the hazard above), while reads BETWEEN two stores may still be shared, which volatile forbids.
  OSG_D operator T() const {
#if OSG_NODEREF_MODE == 1
    if (kCoop) return in_lds ? *l : *g;
#else
    if (kCoop) return in_lds ? *static_cast<volatile T*>(l) : *static_cast<volatile T*>(g);
#endif
    return *g;
  }
  OSG_D T operator=(T v) const {
    // kCoop: the whole first wavefront runs the search in lockstep (every lane holds the same search state, so that
    // the lanes can share out a node's children); one lane's store is enough — 64 stores to one LDS address serialise
    if (kCoop) {
#if OSG_NODEREF_MODE == 1
      if (threadIdx.x == 0) {
        if (in_lds) *l = v; else *g = v;
      }
      asm volatile("" ::: "memory");
#else
      if (threadIdx.x == 0) {
        if (in_lds) *static_cast<volatile T*>(l) = v; else *static_cast<volatile T*>(g) = v;
      }
#endif
      return v;
    }
    *g = v;
    return v;
  }
  OSG_D T operator+=(T v) const { return *this = static_cast<T>(*this) + v; }
};
// A store by the CALLING lane (the reference object's assignment is lane 0's alone): where lanes write different nodes.
template <bool kCoop, class T>
OSG_D void node_store(T* lds_plane, T* pool_plane, uint32_t i, int64_t NR, int64_t RB, T v) {   // (NR, RB: StepPool::at as stride and offset)
  if (kCoop && i < static_cast<uint32_t>(kLdsNodes)) lds_plane[i] = v;
  else pool_plane[static_cast<int64_t>(i) * NR + RB] = v;
}
template <bool kCoop, class T>
OSG_D NodeRef<T, kCoop> node_ref(T* lds_plane, T* pool_plane, uint32_t i, int64_t NR, int64_t RB) {
  const bool in_lds = kCoop && i < static_cast<uint32_t>(kLdsNodes);
  return {lds_plane + (in_lds ? i : 0u), pool_plane + static_cast<int64_t>(i) * NR + RB, in_lds};
}
template <class G>
OSG_D void coop_playouts(const typename G::Params& p, const osg_mcts_cfg& cfg, int num_players, uint64_t gr, CoopBox<G>* box,
                         const StepPool& pool, const LdsTree& lt) {
  const int lane = threadIdx.x & 63;
  uint32_t last = 0;
  for (;;) {
    uint32_t now;
    while ((now = coop_load(&box->seq)) == last) __builtin_amdgcn_s_sleep(1);
    last = now;
    if (coop_load(&box->exit)) {   // the search has left: the LDS part of the tree goes back to the pool
      const uint32_t keep = box->used < static_cast<uint32_t>(kLdsNodes) ? box->used : static_cast<uint32_t>(kLdsNodes);
      for (uint32_t i = lane; i < keep; i += 64) {   // (one root: plane element i of root 0 is pool plane[i])
        pool.total[i] = lt.total[i]; pool.meta[i] = lt.meta[i];
        pool.first[i] = lt.first[i]; pool.parent[i] = lt.parent[i]; pool.count[i] = lt.count[i];
      }
      return;
    }
    const typename G::State s = box->state;
    const uint64_t sim = box->sim;
    double sum[kMaxPlayers];
    for (int q = 0; q < num_players; ++q) sum[q] = 0.0;
    for (int ro = lane; ro < cfg.n_rollouts; ro += 64) {
      Rng rng(cfg.seed, gr, sim * cfg.n_rollouts + ro);
      double rr[kMaxPlayers];
      playout_returns<G>(p, s, rng, rr);
      for (int q = 0; q < num_players; ++q) sum[q] += rr[q];
    }
    for (int q = 0; q < num_players; ++q) {
      double v = sum[q];
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) box->sum[q] = v;
    }
    if (lane == 0) coop_store(&box->done, now);
  }
}

template <class G, bool kBoard, bool kCoop = false>
__global__ void __launch_bounds__(kCoop ? 2 * kBlockM : kBlockM)
k_mcts_advance(typename G::Params p, const typename G::word_t* root_words, typename G::word_t* leaf_words, int64_t n,
               int num_players, int num_actions, osg_mcts_cfg cfg, int flags, double max_utility,
               const double* __restrict__ log_table, StepPool pool, const double* __restrict__ prior_in,
               const double* __restrict__ value_in, uint8_t* __restrict__ request, int max_new_simulations,
               int lane_stride) {
  constexpr bool kWide = G::kMaskW > kMaskWords;   // nine-bit action / child-count fields (osg_mcts_internal.h)
  // lane_stride > 1: only every lane_stride-th lane of a wavefront carries a search.  A search is a chain of
  // dependent, scattered loads (its own tree); with one search per lane 2^16 roots are 1 024 wavefronts — one per
  // SIMD, nothing to hide that latency behind.  Spreading the same searches over lane_stride times as many
  // wavefronts (fewer active lanes each) gives every SIMD several chains to interleave.
  __shared__ CoopBox<G> coop_box;  // (kCoop only)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // (kCoop only: the LDS part of the tree)
  LdsTree lt{};
  uint32_t used = 0;
  if (kCoop) {  // one root: lane 0 searches, the second wavefront plays its leaves' playouts
    lt = lds_tree(lds_raw);
    const uint32_t have = pool.used[0] < static_cast<uint32_t>(kLdsNodes) ? pool.used[0] : static_cast<uint32_t>(kLdsNodes);
    for (uint32_t i = threadIdx.x; i < have; i += 2 * kBlockM) {   // the nodes an earlier launch left in the pool
      lt.total[i] = pool.total[i]; lt.meta[i] = pool.meta[i];
      lt.first[i] = pool.first[i]; lt.parent[i] = pool.parent[i]; lt.count[i] = pool.count[i];
    }
    if (threadIdx.x == 0) { coop_box.seq = 0; coop_box.done = 0; coop_box.exit = 0; coop_box.used = have; }
    __syncthreads();
    if (threadIdx.x >= 64) {
      coop_playouts<G>(p, cfg, num_players, static_cast<uint64_t>(cfg.index_offset), &coop_box, pool, lt);
      return;
    }
    // (all 64 lanes of the first wavefront go on, in lockstep: one search, its state replicated over the lanes)
  }
  CoopLeave coop_leave{kCoop ? &coop_box.seq : nullptr, kCoop ? &coop_box.exit : nullptr, &coop_box.used, &used};
  uint32_t coop_seq = 0;
  const int64_t slot = static_cast<int64_t>(blockIdx.x) * kBlockM + threadIdx.x;
  if (!kCoop && lane_stride > 1 && (threadIdx.x % lane_stride) != 0) return;
  const int64_t r = kCoop ? 0 : slot / lane_stride;
  if (r >= n) return;
  const uint64_t gr = static_cast<uint64_t>(cfg.index_offset + r);
  const int64_t NR = pool.root_major ? 1 : pool.n, RB = pool.root_major ? r * static_cast<int64_t>(pool.cap) : r;   // node i at i * NR + RB (StepPool::at)
  const bool host_priors = (flags & 1) != 0, through_chance = (flags & 2) != 0, own_rollouts = (flags & 4) != 0;
  const bool stashing = pool.stash != nullptr;
// (kCoop: node i < kLdsNodes lives in LDS, the others — and every node of the batch form — in the pool.  A reference
// object that branches on the index: a select between the two ADDRESSES made every access a flat one, slower than the
// pool itself — 5.1e4 against 7.3e4 simulations/s; with the branch the LDS side is a ds_read / ds_write.)
#define META(i) node_ref<kCoop>(lt.meta, pool.meta, static_cast<uint32_t>(i), NR, RB)
#define FIRST(i) node_ref<kCoop>(lt.first, pool.first, static_cast<uint32_t>(i), NR, RB)
#define PARENT(i) node_ref<kCoop>(lt.parent, pool.parent, static_cast<uint32_t>(i), NR, RB)
#define COUNT(i) node_ref<kCoop>(lt.count, pool.count, static_cast<uint32_t>(i), NR, RB)
#define TOTAL(i) node_ref<kCoop>(lt.total, pool.total, static_cast<uint32_t>(i), NR, RB)
#define PRIOR(i) pool.prior[static_cast<int64_t>(i) * NR + RB]   /* (the pool only: PUCT reads it, UCT with playouts never) */
#define REMAP(i) pool.remap[static_cast<int64_t>(i) * NR + RB]
  uint8_t phase = pool.phase[r];
  if (phase == kFinished) { request[r] = 0; return; }
  // A search parked on a request whose answer the caller did not bring (NULL prior / value pointer) stays
  // parked and asks again: never dereference a missing answer.
  if (phase == kWantPrior && prior_in == nullptr) { request[r] = pool.node[r] == 0 ? 5 : 1; return; }
  if (phase == kWantValue && value_in == nullptr) { request[r] = 2; return; }
  used = pool.used[r];
  int gc_limit = pool.gc_limit[r];
  int sims_done = pool.sims[r];
  uint32_t node = pool.node[r];
  Rng trng(0, 0, 0);
  trng.s = pool.rng[r];
  const typename G::State root_state = G::load(p, root_words, n, r);
  typename G::State s = root_state;
  double returns[kMaxPlayers];
  bool solved = false;
  int started = 0;
  // kCoop (one root, lockstep): LANE d remembers the d-th node of the running simulation's visit path, so that the backup
  // is one read-modify-write per lane side by side instead of a chain of them up the parent links.  Valid while the
  // simulation ran inside this launch without parking (a resumed one walks the parent links as before).
  uint32_t path_node = 0;
  int path_depth = 0;
  bool path_ok = false;
  auto park = [&](uint8_t ph, uint8_t req) {
    pool.phase[r] = ph; pool.used[r] = used; pool.gc_limit[r] = gc_limit; pool.sims[r] = sims_done;
    pool.node[r] = node; pool.rng[r] = trng.s;
    request[r] = req;
  };
  // expand `node` (mcts.cc:281-299): one child per prior entry, shuffled
  auto expand = [&](const MaskT<G::kMaskW>& legal, int cur, bool from_host, const double* stashed) -> bool {
    const int c = legal.count();
    if (c == 0 || used + static_cast<uint32_t>(c) > static_cast<uint32_t>(pool.cap)) return false;  // nothing to expand / slots exhausted (see osg_mcts.hip)
    const uint32_t first = used;
    used += c;
    if (kCoop && c <= 64) {   // (wider nodes — hex 9 x 9 and up — take the sequential form below, in lockstep)
      // lane k makes child k; the shuffle (mcts.cc:294, Fisher-Yates on the tree-policy stream, the same draws in the
      // same order) runs on the lanes' registers — lane L tracks which child ends in slot L — and every lane then
      // writes one node: no read-modify-write chain through the tree's memory
      const int L = static_cast<int>(threadIdx.x);
      int mine = L;                                   // the child that sits in slot L
      for (int i = c - 1; i >= 1; --i) {
        const int j = static_cast<int>(trng.below(static_cast<uint32_t>(i + 1)));   // (wave-uniform)
        const int at_i = __builtin_amdgcn_readlane(mine, i), at_j = __builtin_amdgcn_readlane(mine, j);
        mine = L == i ? at_j : (L == j ? at_i : mine);
      }
      if (L < c) {
        const int a = select_action(legal, mine);
        double pr;
        if (cur == kChancePlayer) pr = G::chance_prob(p, s, a);
        else if (stashed) pr = stashed[a];
        else if (from_host) pr = prior_in[r * num_actions + a];
        else pr = 1.0 / c;
        const uint32_t at = first + static_cast<uint32_t>(L);
        node_store<kCoop>(lt.meta, pool.meta, at, NR, r, mw_make<kWide>(a, cur, 0));
        node_store<kCoop>(lt.first, pool.first, at, NR, r, 0u);
        node_store<kCoop>(lt.parent, pool.parent, at, NR, r, node);
        node_store<kCoop>(lt.count, pool.count, at, NR, r, 0u);
        node_store<kCoop>(lt.total, pool.total, at, NR, r, 0.0);
        PRIOR(at) = pr;
      }
      const uint32_t meta = META(node);
      META(node) = mw_make<kWide>(static_cast<int>(mw_action<kWide>(meta)), m_player(meta), c) | (meta & kMetaOutcomeBits);
      FIRST(node) = first;
      return true;
    }
    for (int k = 0; k < c; ++k) {
      const int a = select_action(legal, k);
      double pr;
      if (cur == kChancePlayer) pr = G::chance_prob(p, s, a);       // Prior() of a chance node: ChanceOutcomes()
      else if (stashed) pr = stashed[a];                            // the prior that came with the node's evaluation
      else if (from_host) pr = prior_in[r * num_actions + a];
      else pr = 1.0 / c;                                            // RandomRolloutEvaluator::Prior (mcts.cc:74-87)
      META(first + k) = mw_make<kWide>(a, cur, 0);
      FIRST(first + k) = 0; PARENT(first + k) = node; COUNT(first + k) = 0; TOTAL(first + k) = 0.0; PRIOR(first + k) = pr;
    }
    for (int i = c - 1; i >= 1; --i) {  // the shuffle (mcts.cc:294), Fisher-Yates on the tree-policy stream
      const int j = static_cast<int>(trng.below(static_cast<uint32_t>(i + 1)));
      const uint32_t mi = META(first + i), mj = META(first + j);
      const double pi = PRIOR(first + i), pj = PRIOR(first + j);
      META(first + i) = mj; META(first + j) = mi;
      PRIOR(first + i) = pj; PRIOR(first + j) = pi;
    }
    const uint32_t meta = META(node);
    META(node) = mw_make<kWide>(static_cast<int>(mw_action<kWide>(meta)), m_player(meta), c) | (meta & kMetaOutcomeBits);
    FIRST(node) = first;
    return true;
  };

#ifdef OSG_MCTS_PROFILE
  unsigned long long prof[5] = {0, 0, 0, 0, 0}, pt = wall_clock64();   // descent, expansion (inside descent), playout wait, backup, rest
#define OSG_PROF(slot) do { const unsigned long long now_ = wall_clock64(); prof[slot] += now_ - pt; pt = now_; } while (0)
#else
#define OSG_PROF(slot) do {} while (0)
#endif
  for (;;) {
    bool term = false;
    OSG_PROF(4);
    if (phase == kNewSimulation) {
      if (sims_done >= cfg.max_simulations || started >= max_new_simulations) {
#ifdef OSG_MCTS_PROFILE
        if (kCoop && threadIdx.x == 0)
          printf("k_mcts_advance (one root, us per simulation over %d): descent %.2f  playouts (hand-off + wait) %.2f  backup %.2f  rest %.2f\n",
                 sims_done, prof[0] / 100.0 / sims_done, prof[2] / 100.0 / sims_done, prof[3] / 100.0 / sims_done, prof[4] / 100.0 / sims_done);
#endif
        const bool fin = sims_done >= cfg.max_simulations;
        park(fin ? kFinished : kNewSimulation, fin ? 0 : 3);
        return;
      }
      ++started;
      trng = Rng(cfg.seed ^ kTreeSalt, gr, static_cast<uint64_t>(sims_done));
      s = root_state;
      node = 0;
      path_node = 0; path_depth = 0; path_ok = kCoop && OSG_COOP_BACKUP != 0;
    } else {  // resume at the parked node: its state is in the leaf batch
      s = G::load(p, leaf_words, n, r);
      path_ok = false;
    }
    if (phase == kWantValue) {
      for (int q = 0; q < num_players; ++q) returns[q] = value_in[r * num_players + q];
      if (stashing && prior_in != nullptr && sims_done < pool.stash_slots) {
        // keep the leaf's prior for its expansion (a later simulation's second visit, mcts.cc:281-299); an
        // unexpanded node's first-child field is free: it holds the stash slot + 1
        double* slot = pool.stash + (static_cast<size_t>(r) * pool.stash_slots + sims_done) * num_actions;
        for (int a = 0; a < num_actions; ++a) slot[a] = prior_in[r * num_actions + a];
        FIRST(node) = static_cast<uint32_t>(sims_done) + 1u;
      }
    } else {
      // ---- ApplyTreePolicy (mcts.cc:273-351) ----
      bool resume_expand = phase == kWantPrior;
      // the header of the node the walk stands on (count, meta, first child) comes down with the parent's child scan
      bool carried = false;
      uint32_t n_cnt = 0, n_meta = 0, n_first = 0;
      for (;;) {
        term = G::terminal(p, s);
        const uint32_t cnt = carried ? n_cnt : COUNT(node);
        const int cur = term ? kTerminalPlayer : G::current_player(p, s);
        if (!resume_expand && !((!term && cnt > 0) || (!term && cur == kChancePlayer && through_chance))) break;
        uint32_t meta = carried ? n_meta : META(node);
        bool expanded_now = false;
        if (mw_nchild<kWide>(meta) == 0) {
          const MaskT<G::kMaskW> legal = G::legal(p, s);
          const uint32_t slot1 = (stashing && cur != kChancePlayer && !resume_expand) ? (carried ? n_first : FIRST(node)) : 0u;
          const double* stashed =
              slot1 ? pool.stash + (static_cast<size_t>(r) * pool.stash_slots + (slot1 - 1u)) * num_actions : nullptr;
          if (cur != kChancePlayer && host_priors && !resume_expand && !stashed) {  // Prior(state) comes from the host
            G::store(p, leaf_words, n, r, s);
            park(kWantPrior, node == 0 ? 5 : 1);  // 5 = the ROOT's prior (where Dirichlet noise goes, mcts.cc:284)
            return;
          }
          const bool ok = expand(legal, cur, resume_expand, stashed);
          resume_expand = false;
          if (!ok) break;
          meta = META(node);
          expanded_now = true;
        }
        resume_expand = false;
        const uint32_t first = (carried && !expanded_now) ? n_first : FIRST(node);
        const int c = mw_nchild<kWide>(meta);
        uint32_t chosen = first, chosen_meta = 0, chosen_cnt = 0, chosen_first = 0;
        bool have_chosen_meta = false;
        if (cur == kChancePlayer) {  // mcts.cc:311-322
          const MaskT<G::kMaskW> legal = G::legal(p, s);
          const int a = sample_action_chance<G>(p, s, legal, trng);
          for (int k = 0; k < c; ++k)
            if (static_cast<int>(mw_action<kWide>(META(first + k))) == a) { chosen = first + k; break; }
        } else {  // arg-max of UCTValue / PUCTValue, first maximum wins (mcts.cc:324-341, 90-112)
          double best = -INFINITY;
          const double logn = log_table[cnt];
          const bool puct = cfg.child_selection_policy == 1;
          const double sqrt_n = sqrt(static_cast<double>(cnt));
          // A search is ONE chain of dependent, scattered loads (its own tree), and that chain — not bandwidth, not
          // instruction issue — is what a launch lasts (2^16 roots, one simulation per launch: 370 us, unchanged
          // with 2x / 4x the wavefronts per SIMD).  So the children's four planes are requested eight children at a
          // time with clamped indices (32 independent loads, ONE round trip) instead of child by child behind the
          // branches of the value formula (two round trips per child).
          if constexpr (kCoop) {
            // one child per lane: the values — a division, a square root, a division in fp64, ~150 dependent
            // instructions each — are formed side by side, then a butterfly picks the winner: the highest value, the
            // lowest index among equals, i.e. the sequential scan's "first maximum wins" (a NaN or -inf value never wins)
            double bv = -INFINITY;
            uint32_t bk = 0xFFFFFFFFu, bm = 0, bc = 0, bf = 0;
            // (round 6) under UCT the first never-visited child wins with +infinity: where one exists — a ballot — no
            // value is formed at all (nodes of up to 64 children: one child per lane in one pass)
            bool decided = false;
            if (!puct && c <= 64) {
              const int k = static_cast<int>(threadIdx.x);
              const uint32_t at = first + static_cast<uint32_t>(k < c ? k : c - 1);
              const uint32_t cm = META(at), cc = COUNT(at), cf = FIRST(at);
              const unsigned long long unvisited = __ballot(k < c && cc == 0 && !m_has_outcome(cm));
              if (unvisited != 0ull) {
                const int src = __builtin_ctzll(unvisited);
                bk = static_cast<uint32_t>(src);
                bm = static_cast<uint32_t>(__shfl(static_cast<int>(cm), src, 64));
                bc = 0;
                bf = static_cast<uint32_t>(__shfl(static_cast<int>(cf), src, 64));
                decided = true;
              } else if constexpr (kBoard && OSG_UCT_FILTER_STEP != 0) {
                // (round 6, as in k_mcts_wave's select_child) the arg-max through an fp32 filter: every child's value in
                // single precision (a proven outcome is a small integer, exact; the others within 2^-20 (1 + |c| sqrt(log
                // n)) of the fp64 value: returns in [-1, 1]); only children within 2^-18 of the largest can hold the exact
                // maximum.  One such child: the arg-max.  Several without a proven outcome and with IDENTICAL (count,
                // total): their fp64 values are one number, the lowest index wins as in the scan.  Anything else takes the
                // fp64 butterfly below — so the ~150 dependent fp64 instructions per level are rarely run.
                const double ct = TOTAL(at);
                const bool has = m_has_outcome(cm);
                const float lf = static_cast<float>(logn), cf32 = static_cast<float>(cfg.uct_c);
                const float rc = __builtin_amdgcn_rcpf(static_cast<float>(cc));
                float a = has ? static_cast<float>(outcome_value<true>(cm, cc, ct, m_player(cm)))
                              : static_cast<float>(ct) * rc + cf32 * __builtin_amdgcn_sqrtf(lf * rc);
                a = k < c ? a : -INFINITY;
                const float top = wave_max_f32_dpp(a);
                const float floor_v = top - 0x1p-18f * (1.0f + fabsf(cf32) * __builtin_amdgcn_sqrtf(lf));
                const unsigned long long near = __ballot(a >= floor_v);
                const int n_near = __builtin_popcountll(near);
                bool ok = n_near == 1;
                const int src = near != 0ull ? __builtin_ctzll(near) : 0;
                if (n_near > 1 && (__ballot(has) & near) == 0ull) {
                  const uint32_t c0 = static_cast<uint32_t>(__shfl(static_cast<int>(cc), src, 64));
                  const double t0 = __shfl(ct, src, 64);
                  ok = (near & ~__ballot(cc == c0 && __double_as_longlong(ct) == __double_as_longlong(t0))) == 0ull;
                }
                if (ok) {
                  bk = static_cast<uint32_t>(src);
                  bm = static_cast<uint32_t>(__shfl(static_cast<int>(cm), src, 64));
                  bc = static_cast<uint32_t>(__shfl(static_cast<int>(cc), src, 64));
                  bf = static_cast<uint32_t>(__shfl(static_cast<int>(cf), src, 64));
                  decided = true;
                }
              }
            }
            for (int k0 = 0; !decided && k0 < c; k0 += 64) {
              const int k = k0 + static_cast<int>(threadIdx.x);
              const uint32_t at = first + static_cast<uint32_t>(k < c ? k : c - 1);
              const uint32_t cm = META(at), cc = COUNT(at), cf = FIRST(at);
              const double ct = TOTAL(at);
              double v;
              if (m_has_outcome(cm)) v = outcome_value<kBoard>(cm, cc, ct, m_player(cm));
              else if (puct) v = (cc != 0 ? ct / cc : 0.0) + cfg.uct_c * PRIOR(at) * sqrt_n / (cc + 1);
              else if (cc == 0) v = INFINITY;
              else v = ct / cc + cfg.uct_c * sqrt(logn / cc);
              if (k < c && v > bv) { bv = v; bk = static_cast<uint32_t>(k); bm = cm; bc = cc; bf = cf; }
            }
#pragma unroll
            for (int off = 32; !decided && off >= 1; off >>= 1) {
              const double ov = __shfl_xor(bv, off, 64);
              const uint32_t ok = __shfl_xor(bk, off, 64), om = __shfl_xor(bm, off, 64), oc = __shfl_xor(bc, off, 64),
                             of = __shfl_xor(bf, off, 64);
              if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; bm = om; bc = oc; bf = of; }
            }
            if (bk != 0xFFFFFFFFu) {
              chosen = first + bk; chosen_meta = bm; chosen_cnt = bc; chosen_first = bf;
              have_chosen_meta = true;
            }
          }
          // (round 6, as in k_mcts) under UCT the first never-visited child wins with +infinity: while the LAST child is
          // unvisited the scan only looks for the first such child, and any scan stops at one
          bool scan_only = false, settled = false;
          if (!kCoop && !puct) {
            const uint32_t at = first + static_cast<uint32_t>(c - 1);
            scan_only = COUNT(at) == 0 && !m_has_outcome(META(at));
          }
          for (int k0 = 0; !kCoop && k0 < c && !settled; k0 += kScanChunk) {
            uint32_t cm[kScanChunk], cc[kScanChunk], cf[kScanChunk];
            double ct[kScanChunk], cp[kScanChunk];
#pragma unroll
            for (int j = 0; j < kScanChunk; ++j) {
              const uint32_t at = first + static_cast<uint32_t>(k0 + j < c ? k0 + j : c - 1);
              cm[j] = META(at);
              cc[j] = COUNT(at);
              cf[j] = FIRST(at);
              ct[j] = scan_only ? 0.0 : TOTAL(at);
              cp[j] = scan_only ? 0.0 : PRIOR(at);
            }
#pragma unroll
            for (int j = 0; j < kScanChunk; ++j) {
              if (k0 + j >= c || settled) continue;
              const bool unvisited = !puct && cc[j] == 0 && !m_has_outcome(cm[j]);
              if (scan_only && !unvisited) continue;
              settled = unvisited;
              double v;
              if (m_has_outcome(cm[j])) v = outcome_value<kBoard>(cm[j], cc[j], ct[j], m_player(cm[j]));
              else if (puct) v = (cc[j] != 0 ? ct[j] / cc[j] : 0.0) + cfg.uct_c * cp[j] * sqrt_n / (cc[j] + 1);
              else if (cc[j] == 0) v = INFINITY;
              else v = ct[j] / cc[j] + cfg.uct_c * sqrt(logn / cc[j]);
              if (v > best) {
                best = v; chosen = first + static_cast<uint32_t>(k0 + j);
                chosen_meta = cm[j]; chosen_cnt = cc[j]; chosen_first = cf[j];
                have_chosen_meta = true;  // only a child that won the comparison carries its header (NaN values: none does)
              }
            }
          }
        }
        G::apply(p, s, static_cast<int>(mw_action<kWide>(have_chosen_meta ? chosen_meta : META(chosen))));
        node = chosen;
        if (kCoop) {
          ++path_depth;
          path_node = static_cast<int>(threadIdx.x) == path_depth ? chosen : path_node;
          path_ok = path_ok && path_depth < 64;
        }
        carried = have_chosen_meta;
        if (have_chosen_meta) { n_cnt = chosen_cnt; n_meta = chosen_meta; n_first = chosen_first; }
      }
      OSG_PROF(0);
      // ---- evaluate (mcts.cc:372-381) ----
      if (term) {
        G::returns(p, s, returns);
        uint32_t meta = META(node) | (1u << 20) | (1u << 23);
        if (kBoard) meta = (meta & ~(3u << 21)) | (static_cast<uint32_t>(static_cast<int>(returns[0]) + 1) << 21);
        META(node) = meta;
        solved = cfg.solve != 0;
      } else if (own_rollouts) {
        // RandomRolloutEvaluator::Evaluate (mcts.cc:43-72) right here, on the streams k_mcts_tree_rollout draws
        // from (rollout ro of simulation s of root r: Rng(seed, root, s * n_rollouts + ro)): the same values as the
        // park / rollout-kernel / resume round trip, without leaving the launch
        for (int q = 0; q < num_players; ++q) returns[q] = 0.0;
        if (kCoop) {
          // the playouts on the lanes of THIS wavefront (it runs in lockstep, so every lane holds the leaf): lane ro
          // plays rollout ro on its own stream, a butterfly sums the returns (integers: any order) and leaves the sum
          // in every lane — no hand-off to a second wavefront (that cost ~1.2 us of polling per simulation)
          double sum[kMaxPlayers];
          for (int q = 0; q < num_players; ++q) sum[q] = 0.0;
          for (int ro = static_cast<int>(threadIdx.x); ro < cfg.n_rollouts; ro += 64) {
            Rng rng(cfg.seed, gr, static_cast<uint64_t>(sims_done) * cfg.n_rollouts + ro);
            double rr[kMaxPlayers];
            playout_returns<G>(p, s, rng, rr);
            for (int q = 0; q < num_players; ++q) sum[q] += rr[q];
          }
          if (kBoard && cfg.n_rollouts <= 64 && OSG_COOP_BACKUP != 0) {
            // win / draw / loss games, one playout per lane: the sum of player 0's returns (each -1, 0 or +1) is wins
            // minus losses — two ballots instead of two 64-bit butterflies through the LDS crossbar — and player 1's
            // is its negation (Returns() of these games: {r, -r + 0.0}; small integers, exact in any order)
            const int wins = __builtin_popcountll(__ballot(sum[0] > 0.0)), losses = __builtin_popcountll(__ballot(sum[0] < 0.0));
            returns[0] = static_cast<double>(wins - losses);
            returns[1] = -returns[0] + 0.0;
          } else {
            for (int q = 0; q < num_players; ++q) {
              double v = sum[q];
              for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
              returns[q] = v;
            }
          }
        }
        for (int ro = 0; !kCoop && ro < cfg.n_rollouts; ++ro) {
          Rng rng(cfg.seed, gr, static_cast<uint64_t>(sims_done) * cfg.n_rollouts + ro);
          double rr[kMaxPlayers];
          playout_returns<G>(p, s, rng, rr);
          for (int q = 0; q < num_players; ++q) returns[q] += rr[q];
        }
        for (int q = 0; q < num_players; ++q) returns[q] = returns[q] / cfg.n_rollouts;
      } else {  // Evaluate(state) comes from outside
        G::store(p, leaf_words, n, r, s);
        park(kWantValue, 2);
        return;
      }
    }
    OSG_PROF(2);
    // ---- backup (mcts.cc:383-435) ----
    bool counted = false;
    if (kCoop && kBoard && path_ok) {   // two players, no chance nodes: a node's player is in its own header
      if (static_cast<int>(threadIdx.x) <= path_depth) {
        const uint32_t v = path_node;
        const int pl = m_player(static_cast<uint32_t>(META(v)));
        const double t = static_cast<double>(TOTAL(v)) + returns[(pl < 0 || pl >= num_players) ? 0 : pl];  // (a terminal root has no player)
        const uint32_t cn = static_cast<uint32_t>(COUNT(v)) + 1u;
        node_store<kCoop>(lt.total, pool.total, v, NR, RB, t);
        node_store<kCoop>(lt.count, pool.count, v, NR, RB, cn);
      }
      asm volatile("" ::: "memory");
      counted = true;
    }
    for (uint32_t v = node; v != kNoNode && !(counted && !solved); v = PARENT(v)) {
      uint32_t meta = META(v);
      int pl = m_player(meta);
      for (uint32_t up = v; pl == kChancePlayer;) {
        up = PARENT(up);
        if (up == kNoNode) { pl = 0; break; }
        pl = m_player(META(up));
      }
      if (!counted) {
        TOTAL(v) += returns[(pl < 0 || pl >= num_players) ? 0 : pl];  // (a terminal root has no player)
        COUNT(v) += 1;
      }
      if (kBoard && solved && mw_nchild<kWide>(meta) > 0) {
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
    solved = false;
    ++sims_done;
    phase = kNewSimulation;
    OSG_PROF(3);
    const uint32_t rm = META(0);
    if ((m_has_outcome(rm) && !m_terminal(rm)) || mw_nchild<kWide>(rm) == 1 || m_terminal(rm)) {  // mcts.cc:437-440
      park(kFinished, 0);
      return;
    }
    if (pool.gc_nodes > 1 && used >= static_cast<uint32_t>(pool.gc_nodes)) {  // GarbageCollect (see osg_mcts.hip)
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
        const double tot = TOTAL(i), pri = PRIOR(i);
        if (mw_nchild<kWide>(meta) > 0) {
          if (cnt < limit) { meta = mw_clear_children<kWide>(meta); first = 0; }
          else first = REMAP(first);
        }
        META(to) = meta; FIRST(to) = first; COUNT(to) = cnt; TOTAL(to) = tot; PRIOR(to) = pri;
        PARENT(to) = i == 0 ? kNoNode : REMAP(par);
      }
      used = w;
      gc_limit = next_gc_limit(gc_limit, used, pool.gc_nodes);
    }
  }
#undef META
#undef FIRST
#undef PARENT
#undef COUNT
#undef TOTAL
#undef PRIOR
#undef REMAP
}

// RandomRolloutEvaluator::Evaluate (mcts.cc:43-72) for the parked leaves, on the fused kernel's streams:
// rollout ro of simulation s of root r draws from Rng(seed, root, s * n_rollouts + ro).
template <class G>
__global__ void __launch_bounds__(kBlockM)
k_mcts_tree_rollout(typename G::Params p, const typename G::word_t* leaf_words, int64_t n, int num_players,
                    osg_mcts_cfg cfg, const uint8_t* __restrict__ phase, const int32_t* __restrict__ sims, double* value,
                    int lane_stride) {
  const int64_t slot = static_cast<int64_t>(blockIdx.x) * kBlockM + threadIdx.x;
  if (lane_stride > 1 && (threadIdx.x % lane_stride) != 0) return;
  const int64_t r = slot / lane_stride;
  if (r >= n || phase[r] != kWantValue) return;
  const uint64_t gr = static_cast<uint64_t>(cfg.index_offset + r);
  const typename G::State s = G::load(p, leaf_words, n, r);
  double sum[kMaxPlayers];
  for (int q = 0; q < num_players; ++q) sum[q] = 0.0;
  for (int ro = 0; ro < cfg.n_rollouts; ++ro) {
    Rng rng(cfg.seed, gr, static_cast<uint64_t>(sims[r]) * cfg.n_rollouts + ro);
    double rr[kMaxPlayers];
    playout_returns<G>(p, s, rng, rr);
    for (int q = 0; q < num_players; ++q) sum[q] += rr[q];
  }
  for (int q = 0; q < num_players; ++q) value[r * num_players + q] = sum[q] / cfg.n_rollouts;
}

// Root statistics in the layout of osg_mcts_search's outputs, plus the children's priors.
template <bool kBoard>
__global__ void __launch_bounds__(kBlockM)
k_mcts_tree_results(StepPool pool, int64_t n, int num_actions, int32_t* best_action, int32_t* child_visits,
                    double* child_reward, int8_t* child_outcome, double* child_prior, double* root_stats) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlockM + threadIdx.x;
  if (r >= n) return;
  const int64_t NR = pool.root_major ? 1 : pool.n, RB = pool.root_major ? r * static_cast<int64_t>(pool.cap) : r;   // node i at i * NR + RB (StepPool::at)
#define META(i) pool.meta[static_cast<int64_t>(i) * NR + RB]
#define FIRST(i) pool.first[static_cast<int64_t>(i) * NR + RB]
#define COUNT(i) pool.count[static_cast<int64_t>(i) * NR + RB]
#define TOTAL(i) pool.total[static_cast<int64_t>(i) * NR + RB]
#define PRIOR(i) pool.prior[static_cast<int64_t>(i) * NR + RB]
  const uint32_t rm = META(0);
  const int root_player = m_terminal(rm) ? -1 : m_player(rm);  // a terminal root has no player to move
  const int c = mw_nchild<true>(rm);
  const uint32_t first = FIRST(0);
  for (int a = 0; a < num_actions; ++a) {
    if (child_visits) child_visits[r * num_actions + a] = 0;
    if (child_reward) child_reward[r * num_actions + a] = 0.0;
    if (child_outcome) child_outcome[r * num_actions + a] = 3;
    if (child_prior) child_prior[r * num_actions + a] = 0.0;
  }
  int best = -1;
  double b_out = 0.0, b_tot = 0.0;
  uint32_t b_cnt = 0;
  for (int k = 0; k < c; ++k) {
    const uint32_t cm = META(first + k);
    const uint32_t cc = COUNT(first + k);
    const double ct = TOTAL(first + k);
    const int a = static_cast<int>(mw_action<true>(cm));
    const bool has = m_has_outcome(cm);
    const int pl = m_player(cm);
    const double out = (has && pl >= 0 && cc > 0) ? outcome_value<kBoard>(cm, cc, ct, pl)
                                                  : ((has && kBoard && pl >= 0) ? outcome_value<true>(cm, 1, 0.0, pl) : 0.0);
    const bool better = best < 0 || (b_out != out ? b_out < out : (b_cnt != cc ? b_cnt < cc : b_tot < ct));
    if (better) { best = a; b_out = out; b_cnt = cc; b_tot = ct; }
    if (a < num_actions) {
      if (child_visits) child_visits[r * num_actions + a] = static_cast<int32_t>(cc);
      if (child_reward) child_reward[r * num_actions + a] = ct;
      if (child_prior) child_prior[r * num_actions + a] = PRIOR(first + k);
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
    root_stats[r * 4 + 1] = static_cast<double>(pool.used[r]);
    root_stats[r * 4 + 2] = (kBoard && m_has_outcome(rm) && root_player >= 0) ? outcome_value<true>(rm, 1, 0.0, root_player) : NAN;
    root_stats[r * 4 + 3] = static_cast<double>(pool.sims[r]);
  }
#undef META
#undef FIRST
#undef COUNT
#undef TOTAL
#undef PRIOR
}

__global__ void __launch_bounds__(256) k_mcts_tree_init(StepPool pool, const int8_t* root_player, int64_t n) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (r >= n) return;
  const int64_t at0 = pool.at(0, r);
  pool.meta[at0] = make_meta(0xFF, root_player[r], 0);  // mcts.cc:356-357: root = (kInvalidAction, CurrentPlayer(), 1)
  pool.first[at0] = 0; pool.parent[at0] = kNoNode; pool.count[at0] = 0; pool.total[at0] = 0.0; pool.prior[at0] = 1.0;
  pool.rng[r] = 0; pool.used[r] = 1; pool.node[r] = 0; pool.gc_limit[r] = kMinGcLimit; pool.sims[r] = 0;
  pool.phase[r] = kNewSimulation;
}

// One root's tree, flattened for the host (SearchNode, mcts.h:114-146).
__global__ void k_mcts_tree_extract(StepPool pool, int64_t r, uint32_t* meta, uint32_t* first, uint32_t* count, double* total,
                                    double* prior) {
  const uint32_t used = pool.used[r];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < used; i += gridDim.x * blockDim.x) {
    const int64_t at = pool.at(i, r);
    meta[i] = pool.meta[at]; first[i] = pool.first[at]; count[i] = pool.count[at]; total[i] = pool.total[at];
    prior[i] = pool.prior[at];
  }
}

StepPool make_pool(const osg_mcts_tree* t) {
  StepPool pool;
  const size_t slots = static_cast<size_t>(t->cap) * t->n;
  char* m = t->d_mem;
  pool.total = reinterpret_cast<double*>(m); m += slots * 8;
  pool.prior = reinterpret_cast<double*>(m); m += slots * 8;
  pool.rng = reinterpret_cast<uint64_t*>(m); m += static_cast<size_t>(t->n) * 8;
  pool.meta = reinterpret_cast<uint32_t*>(m); m += slots * 4;
  pool.first = reinterpret_cast<uint32_t*>(m); m += slots * 4;
  pool.parent = reinterpret_cast<uint32_t*>(m); m += slots * 4;
  pool.count = reinterpret_cast<uint32_t*>(m); m += slots * 4;
  pool.remap = reinterpret_cast<uint32_t*>(m); m += slots * 4;
  pool.used = reinterpret_cast<uint32_t*>(m); m += static_cast<size_t>(t->n) * 4;
  pool.node = reinterpret_cast<uint32_t*>(m); m += static_cast<size_t>(t->n) * 4;
  pool.gc_limit = reinterpret_cast<int32_t*>(m); m += static_cast<size_t>(t->n) * 4;
  pool.sims = reinterpret_cast<int32_t*>(m); m += static_cast<size_t>(t->n) * 4;
  pool.phase = reinterpret_cast<uint8_t*>(m);
  pool.n = t->n;
  pool.cap = t->cap;
  {
    const char* e = std::getenv("OSG_STEP_ROOT_MAJOR");
    pool.root_major = (e && e[0] == '1') ? 1 : 0;
  }
  pool.gc_nodes = t->gc_nodes;
  pool.stash = t->d_stash;
  pool.stash_slots = t->stash_slots;
  return pool;
}

size_t pool_bytes(int64_t cap, int64_t n) {
  return static_cast<size_t>(cap) * n * 36 + static_cast<size_t>(n) * (8 + 4 * 4 + 1) + 256;
}

// Active lanes per wavefront for the lane-per-root kernels.  1 = every lane carries a search (the default).  Spreading
// the searches over more wavefronts (OSG_MCTS_LANE_STRIDE = 2 ... 16: every k-th lane active) was measured on 2^16
// connect_four roots, one simulation per launch (tools/probe_advance_stride.py, profiles/r03_advance_stride.log):
// 370 / 377 / 418 / 498 / 633 us for k = 1 / 2 / 4 / 8 / 16 — a launch lasts as long as ONE search's chain of
// dependent loads, which more wavefronts do not shorten (they only add instruction issue); what shortens it is
// fewer round trips per tree level (the chunked child scan in k_mcts_advance).
int lane_stride_for(osg_ctx*, int64_t) {
  if (const char* e = std::getenv("OSG_MCTS_LANE_STRIDE")) {
    const int v = std::atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) return v;
  }
  return 1;
}

bool same_game(const osg_batch* a, const osg_batch* b) {
  return a && b && a->n == b->n && std::strcmp(a->spec.desc.canonical, b->spec.desc.canonical) == 0;
}

}  // namespace

extern "C" {

int osg_mcts_tree_create(const osg_batch* roots, const osg_mcts_cfg* cfg_in, int flags, osg_mcts_tree** out) {
  if (!roots || !cfg_in || !out) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_create: null argument");
  osg_ctx* ctx = roots->ctx;
  if (ctx->closed) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_create: the context was destroyed");
  const osg_game_desc& d = roots->spec.desc;
  const bool board = d.game_kind <= kHex;
  if (cfg_in->max_simulations < 1 || cfg_in->n_rollouts < 1)
    return set_error(OSG_ERR_INVALID, "max_simulations and n_rollouts must be >= 1");
  if (cfg_in->solve && !board)
    return set_error(OSG_ERR_UNSUPPORTED, "solve=true needs win/draw/loss outcomes (tic_tac_toe, connect_four, hex)");
  if (cfg_in->child_selection_policy != 0 && cfg_in->child_selection_policy != 1)
    return set_error(OSG_ERR_INVALID, "osg_mcts_cfg.child_selection_policy must be 0 (UCT) or 1 (PUCT)");
  if (flags & ~15) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_create: unknown flag");
  if ((flags & 8) && !(flags & 1)) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_create: flag 8 (priors arrive with the values) needs flag 1");
  if ((flags & 8) && (flags & 4)) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_create: flags 4 and 8 exclude each other");
  if (int rc = refuse_endless_playouts(roots->spec, "osg_mcts_tree_create")) return rc;
  if (d.num_distinct_actions > kMaxSearchActions)
    return set_error(OSG_ERR_UNSUPPORTED, "osg_mcts_tree_create: a node holds up to 511 actions");
  osg_mcts_tree* t = new osg_mcts_tree;
  t->ctx = ctx;
  t->cfg = *cfg_in;
  t->flags = flags;
  t->n = roots->n;
  t->A = d.num_distinct_actions;
  t->P = d.num_players;
  t->widest = std::max(d.num_distinct_actions, d.max_chance_outcomes);
  t->max_utility = d.max_utility;
  t->board = board;
  // with dont_return_chance_node a simulation can expand a chain of chance nodes besides its decision node
  const int64_t per_sim = static_cast<int64_t>(t->widest) * ((flags & 2) ? 1 + std::max(d.max_chance_nodes, 1) : 1);
  const int64_t never = std::min<int64_t>(1 + static_cast<int64_t>(cfg_in->max_simulations) * per_sim, int64_t{1} << 30);
  int64_t cap = never, gc_nodes = 0;
  if (cfg_in->max_nodes > 0 && cfg_in->max_nodes < never) {
    gc_nodes = std::max(2, cfg_in->max_nodes);
    cap = std::min<int64_t>(never, gc_nodes + 32 * per_sim);
  }
  // every failure path below goes through here: what has been allocated so far goes with the handle
  auto fail = [&](int code) {
    if (t->roots) osg_batch_destroy(t->roots);
    if (t->d_mem) (void)hipFree(t->d_mem);
    if (t->d_stash) (void)hipFree(t->d_stash);
    if (t->d_logs) (void)hipFree(t->d_logs);
    delete t;
    return code;
  };
  size_t free_b = 0, total_b = 0;
  hipError_t e = hipMemGetInfo(&free_b, &total_b);
  if (e != hipSuccess) return fail(set_error(OSG_ERR_HIP, hipGetErrorString(e)));
  // 60 % of the free HBM for the trees AND the prior stash (flag 8: one row of A priors per unexpanded leaf,
  // at most max_simulations + 1 per root); the stash has a fixed size, the node pool takes what is left
  size_t budget = free_b * 6 / 10;
  size_t stash_bytes = 0;
  if (flags & 8) {
    t->stash_slots = cfg_in->max_simulations + 1;
    stash_bytes = sizeof(double) * static_cast<size_t>(t->n) * t->stash_slots * t->A;
    if (stash_bytes > budget / 2)
      return fail(set_error(OSG_ERR_NOMEM, "osg_mcts_tree_create: the prior stash (roots x (max_simulations + 1) x actions fp64) "
                                           "does not fit the free HBM; search fewer roots per tree or fewer simulations"));
    budget -= stash_bytes;
  }
  if (pool_bytes(cap, t->n) > budget) {
    if (gc_nodes > 0) return fail(set_error(OSG_ERR_NOMEM, "osg_mcts_tree_create: max_nodes slots per root do not fit the free HBM"));
    cap = static_cast<int64_t>((budget - static_cast<size_t>(t->n) * 32) / (static_cast<size_t>(t->n) * 36));
    if (cap < 2 + 2 * per_sim) return fail(set_error(OSG_ERR_NOMEM, "osg_mcts_tree_create: too many roots for the free HBM"));
    gc_nodes = cap - per_sim;
  }
  t->cap = static_cast<int>(cap);
  t->gc_nodes = static_cast<int>(gc_nodes);
  t->bytes = pool_bytes(cap, t->n);
  e = hipMalloc(reinterpret_cast<void**>(&t->d_mem), t->bytes);
  if (e != hipSuccess) { t->d_mem = nullptr; return fail(set_error(OSG_ERR_NOMEM, std::string("MCTS trees: ") + hipGetErrorString(e))); }
  if (flags & 8) {
    e = hipMalloc(reinterpret_cast<void**>(&t->d_stash), stash_bytes);
    if (e != hipSuccess) { t->d_stash = nullptr; return fail(set_error(OSG_ERR_NOMEM, std::string("MCTS prior stash: ") + hipGetErrorString(e))); }
  }
  int rc = osg_batch_create(ctx, d.canonical, roots->n, &t->roots);
  if (rc == OSG_OK) rc = osg_batch_copy(t->roots, roots);
  if (rc) return fail(rc);
  // log(parent explore_count) from the host libm, like osg_mcts_search
  t->logs_n = cfg_in->max_simulations + 2;
  std::vector<double> logs(t->logs_n);
  logs[0] = 0.0;
  for (int i = 1; i < t->logs_n; ++i) logs[i] = std::log(static_cast<double>(i));
  e = hipMalloc(reinterpret_cast<void**>(&t->d_logs), sizeof(double) * t->logs_n);
  if (e != hipSuccess) t->d_logs = nullptr;
  if (e == hipSuccess) e = hipMemcpy(t->d_logs, logs.data(), sizeof(double) * t->logs_n, hipMemcpyHostToDevice);
  // the roots' players (status query on the copy)
  int8_t* d_cur = nullptr;
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_cur), static_cast<size_t>(t->n));
  if (e != hipSuccess) return fail(set_error(OSG_ERR_NOMEM, hipGetErrorString(e)));
  rc = osg_status_query(t->roots, d_cur, nullptr, nullptr, 0);
  if (rc == OSG_OK) {
    k_mcts_tree_init<<<dim3(static_cast<unsigned>((t->n + 255) / 256)), dim3(256), 0, ctx->stream>>>(make_pool(t), d_cur, t->n);
    if (hipGetLastError() != hipSuccess) rc = set_error(OSG_ERR_HIP, "k_mcts_tree_init");
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_cur);
  if (rc) return fail(rc);
  osg::ctx_retain(ctx);
  *out = t;
  return OSG_OK;
}

int osg_mcts_tree_destroy(osg_mcts_tree* t) {
  if (!t) return OSG_OK;
  (void)hipStreamSynchronize(t->ctx->stream);
  if (t->roots) osg_batch_destroy(t->roots);
  (void)hipFree(t->d_mem);
  (void)hipFree(t->d_logs);
  if (t->d_own_prior) (void)hipFree(t->d_own_prior);
  if (t->d_stash) (void)hipFree(t->d_stash);
  osg::ctx_release(t->ctx);
  delete t;
  return OSG_OK;
}

int osg_mcts_tree_advance(osg_mcts_tree* t, osg_batch* leaf, const double* d_prior, const double* d_value,
                          uint8_t* d_request, int max_new_simulations, int64_t* h_counts) {
  if (!t || !leaf || !d_request) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_advance: null argument");
  if (!same_game(t->roots, leaf)) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_advance: the leaf batch must have the roots' game and size");
  if (max_new_simulations < 0) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_advance: max_new_simulations < 0");
  const int lane_stride = lane_stride_for(t->ctx, t->n);
  const unsigned grid = static_cast<unsigned>((t->n * lane_stride + kBlockM - 1) / kBlockM);
  const osg_game_desc& d = t->roots->spec.desc;
  const StepPool pool = make_pool(t);
  hipStream_t st = t->ctx->stream;
  // one root, playouts in the launch: the two-wavefront form (OSG_MCTS_COOP=0 keeps the one-lane form, for A/B)
  static const bool coop_on = !(std::getenv("OSG_MCTS_COOP") && std::atoi(std::getenv("OSG_MCTS_COOP")) == 0);
  bool coop = t->n == 1 && (t->flags & 4) && t->cfg.n_rollouts > 1 && coop_on;
  if (coop) {
    // the LDS part of the tree is 144 KiB of dynamic shared memory: above the default limit, asked for once per
    // (device, kernel) (raise_lds_cap); a device that cannot grant it keeps the one-lane form below
    hipError_t e = hipSuccess;
    if (t->board) OSG_DISPATCH_WIDE(t->roots->spec, e = raise_lds_cap(reinterpret_cast<const void*>(&k_mcts_advance<G, true, true>),
                                                                      static_cast<int>(kLdsTreeBytes)));
    else OSG_DISPATCH_WIDE(t->roots->spec, e = raise_lds_cap(reinterpret_cast<const void*>(&k_mcts_advance<G, false, true>),
                                                             static_cast<int>(kLdsTreeBytes)));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      coop = false;
    }
  }
  if (coop) {
    if (t->board) {
      OSG_DISPATCH_WIDE(t->roots->spec, k_mcts_advance<G, true, true><<<dim3(1), dim3(2 * kBlockM), kLdsTreeBytes, st>>>(
                                       P, static_cast<const typename G::word_t*>(t->roots->d_words),
                                       static_cast<typename G::word_t*>(leaf->d_words), t->n, d.num_players, t->A, t->cfg,
                                       t->flags, t->max_utility, t->d_logs, pool, d_prior, d_value, d_request,
                                       max_new_simulations, 1));
    } else {
      OSG_DISPATCH_WIDE(t->roots->spec, k_mcts_advance<G, false, true><<<dim3(1), dim3(2 * kBlockM), kLdsTreeBytes, st>>>(
                                       P, static_cast<const typename G::word_t*>(t->roots->d_words),
                                       static_cast<typename G::word_t*>(leaf->d_words), t->n, d.num_players, t->A, t->cfg,
                                       t->flags, t->max_utility, t->d_logs, pool, d_prior, d_value, d_request,
                                       max_new_simulations, 1));
    }
  } else if (t->board) {
    OSG_DISPATCH_WIDE(t->roots->spec, k_mcts_advance<G, true><<<dim3(grid), dim3(kBlockM), 0, st>>>(
                                     P, static_cast<const typename G::word_t*>(t->roots->d_words),
                                     static_cast<typename G::word_t*>(leaf->d_words), t->n, d.num_players, t->A, t->cfg,
                                     t->flags, t->max_utility, t->d_logs, pool, d_prior, d_value, d_request,
                                     max_new_simulations, lane_stride));
  } else {
    OSG_DISPATCH_WIDE(t->roots->spec, k_mcts_advance<G, false><<<dim3(grid), dim3(kBlockM), 0, st>>>(
                                     P, static_cast<const typename G::word_t*>(t->roots->d_words),
                                     static_cast<typename G::word_t*>(leaf->d_words), t->n, d.num_players, t->A, t->cfg,
                                     t->flags, t->max_utility, t->d_logs, pool, d_prior, d_value, d_request,
                                     max_new_simulations, lane_stride));
  }
  OSG_HIP(hipGetLastError());
  if (h_counts) {  // how many searches are finished / want a prior / want a value / were paused by max_new_simulations
    std::vector<uint8_t> req(static_cast<size_t>(t->n));
    OSG_HIP(hipMemcpyAsync(req.data(), d_request, req.size(), hipMemcpyDeviceToHost, st));
    OSG_HIP(hipStreamSynchronize(st));
    h_counts[0] = h_counts[1] = h_counts[2] = h_counts[3] = 0;
    for (uint8_t q : req) ++h_counts[q & 3];
  }
  return OSG_OK;
}

static int own_buffers(osg_mcts_tree* t) {
  if (t->d_own_prior) return OSG_OK;
  const size_t bytes = sizeof(double) * static_cast<size_t>(t->n) * (t->A + t->P) + static_cast<size_t>(t->n) + 64;
  char* m = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&m), bytes);
  if (e != hipSuccess) return set_error(OSG_ERR_NOMEM, hipGetErrorString(e));
  t->d_own_prior = reinterpret_cast<double*>(m);
  t->d_own_value = t->d_own_prior + static_cast<size_t>(t->n) * t->A;
  t->d_own_request = reinterpret_cast<uint8_t*>(t->d_own_value + static_cast<size_t>(t->n) * t->P);
  return OSG_OK;
}

int osg_mcts_tree_advance_host(osg_mcts_tree* t, osg_batch* leaf, const double* h_prior, const double* h_value,
                               int values_on_device, uint8_t* h_request, int max_new_simulations, int64_t* h_counts) {
  if (!t || !leaf) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_advance_host: null argument");
  int rc = own_buffers(t);
  if (rc) return rc;
  hipStream_t st = t->ctx->stream;
  if (h_prior) OSG_HIP(hipMemcpyAsync(t->d_own_prior, h_prior, sizeof(double) * t->n * t->A, hipMemcpyHostToDevice, st));
  if (h_value) OSG_HIP(hipMemcpyAsync(t->d_own_value, h_value, sizeof(double) * t->n * t->P, hipMemcpyHostToDevice, st));
  int64_t counts[4];
  rc = osg_mcts_tree_advance(t, leaf, h_prior ? t->d_own_prior : nullptr,
                             (h_value || values_on_device) ? t->d_own_value : nullptr, t->d_own_request,
                             max_new_simulations, counts);
  if (rc) return rc;
  if (h_counts) for (int k = 0; k < 4; ++k) h_counts[k] = counts[k];
  if (h_request) {
    OSG_HIP(hipMemcpyAsync(h_request, t->d_own_request, static_cast<size_t>(t->n), hipMemcpyDeviceToHost, st));
    OSG_HIP(hipStreamSynchronize(st));
  }
  return OSG_OK;
}

int osg_mcts_tree_rollout_values(osg_mcts_tree* t, const osg_batch* leaf, double* d_value) {
  if (!t || !leaf) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_rollout_values: null argument");
  if (int rc = refuse_endless_playouts(leaf->spec, "osg_mcts_tree_rollout_values")) return rc;
  if (!d_value) {  // into the tree's own value buffer (the next osg_mcts_tree_advance_host reads it with values_on_device)
    int rc = own_buffers(t);
    if (rc) return rc;
    d_value = t->d_own_value;
  }
  if (!same_game(t->roots, leaf)) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_rollout_values: the leaf batch must have the roots' game and size");
  const int lane_stride = lane_stride_for(t->ctx, t->n);
  const unsigned grid = static_cast<unsigned>((t->n * lane_stride + kBlockM - 1) / kBlockM);
  const StepPool pool = make_pool(t);
  OSG_DISPATCH_WIDE(t->roots->spec, k_mcts_tree_rollout<G><<<dim3(grid), dim3(kBlockM), 0, t->ctx->stream>>>(
                                   P, static_cast<const typename G::word_t*>(leaf->d_words), t->n, t->P, t->cfg, pool.phase,
                                   pool.sims, d_value, lane_stride));
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int osg_mcts_tree_results(osg_mcts_tree* t, int32_t* best_action, int32_t* child_visits, double* child_reward,
                          int8_t* child_outcome, double* child_prior, double* root_stats) {
  if (!t) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_results: null argument");
  const unsigned grid = static_cast<unsigned>((t->n + kBlockM - 1) / kBlockM);
  const StepPool pool = make_pool(t);
  if (t->board)
    k_mcts_tree_results<true><<<dim3(grid), dim3(kBlockM), 0, t->ctx->stream>>>(pool, t->n, t->A, best_action, child_visits,
                                                                                child_reward, child_outcome, child_prior, root_stats);
  else
    k_mcts_tree_results<false><<<dim3(grid), dim3(kBlockM), 0, t->ctx->stream>>>(pool, t->n, t->A, best_action, child_visits,
                                                                                 child_reward, child_outcome, child_prior, root_stats);
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int64_t osg_mcts_tree_nodes(osg_mcts_tree* t, int64_t root) {
  if (!t || root < 0 || root >= t->n) return -1;
  uint32_t used = 0;
  const StepPool pool = make_pool(t);
  if (hipMemcpyAsync(&used, pool.used + root, sizeof(used), hipMemcpyDeviceToHost, t->ctx->stream) != hipSuccess) return -1;
  if (hipStreamSynchronize(t->ctx->stream) != hipSuccess) return -1;
  return used;
}

int osg_mcts_tree_leaf_path(osg_mcts_tree* t, int64_t root, int32_t* h_actions, int cap) {
  if (!t || root < 0 || root >= t->n || (!h_actions && cap > 0)) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_leaf_path: bad argument");
  // the parked node and its ancestors: a handful of 4-byte reads per level (host-driven evaluators only)
  const StepPool pool = make_pool(t);
  hipStream_t st = t->ctx->stream;
  uint32_t node = 0;
  OSG_HIP(hipMemcpyAsync(&node, pool.node + root, 4, hipMemcpyDeviceToHost, st));
  OSG_HIP(hipStreamSynchronize(st));
  std::vector<int32_t> rev;
  while (node != 0 && node != kNoNode) {
    uint32_t meta = 0, parent = 0;
    OSG_HIP(hipMemcpyAsync(&meta, pool.meta + pool.at(node, root), 4, hipMemcpyDeviceToHost, st));
    OSG_HIP(hipMemcpyAsync(&parent, pool.parent + pool.at(node, root), 4, hipMemcpyDeviceToHost, st));
    OSG_HIP(hipStreamSynchronize(st));
    rev.push_back(static_cast<int32_t>(mw_action<true>(meta)));
    node = parent;
  }
  const int len = static_cast<int>(rev.size());
  if (len > cap) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_leaf_path: buffer too small");
  for (int i = 0; i < len; ++i) h_actions[i] = rev[len - 1 - i];
  return len;
}

int osg_mcts_tree_download(osg_mcts_tree* t, int64_t root, int64_t cap, uint32_t* h_meta, uint32_t* h_first, uint32_t* h_count,
                           double* h_total, double* h_prior) {
  if (!t || !h_meta || !h_first || !h_count || !h_total || !h_prior) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_download: null argument");
  const int64_t used = osg_mcts_tree_nodes(t, root);
  if (used < 0) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_download: no such root");
  if (used > cap) return set_error(OSG_ERR_INVALID, "osg_mcts_tree_download: buffers too small (see osg_mcts_tree_nodes)");
  void* scratch = nullptr;
  const size_t u = static_cast<size_t>(used);
  int rc = osg_ctx_scratch(t->ctx, u * 28 + 64, &scratch);
  if (rc) return rc;
  double* d_total = static_cast<double*>(scratch);
  double* d_prior = d_total + u;
  uint32_t* d_meta = reinterpret_cast<uint32_t*>(d_prior + u);
  uint32_t* d_first = d_meta + u;
  uint32_t* d_count = d_first + u;
  hipStream_t st = t->ctx->stream;
  k_mcts_tree_extract<<<dim3(64), dim3(256), 0, st>>>(make_pool(t), root, d_meta, d_first, d_count, d_total, d_prior);
  OSG_HIP(hipGetLastError());
  OSG_HIP(hipMemcpyAsync(h_meta, d_meta, u * 4, hipMemcpyDeviceToHost, st));
  OSG_HIP(hipMemcpyAsync(h_first, d_first, u * 4, hipMemcpyDeviceToHost, st));
  OSG_HIP(hipMemcpyAsync(h_count, d_count, u * 4, hipMemcpyDeviceToHost, st));
  OSG_HIP(hipMemcpyAsync(h_total, d_total, u * 8, hipMemcpyDeviceToHost, st));
  OSG_HIP(hipMemcpyAsync(h_prior, d_prior, u * 8, hipMemcpyDeviceToHost, st));
  OSG_HIP(hipStreamSynchronize(st));
  return OSG_OK;
}

}  // extern "C"
