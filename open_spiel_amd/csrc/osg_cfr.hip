// Tabular CFR / CFR+ (open_spiel/algorithms/cfr.{h,cc}) and external-sampling
// MCCFR (open_spiel/algorithms/external_sampling_mccfr.{h,cc}) on the device.
//
// The reference walks the game tree recursively, cloning a State per edge and
// looking infostates up by string (cfr.cc:331-408,443-469).  Here the tree is
// flattened ONCE, level by level, by the batched step kernels themselves
// (osg_legal_mask / osg_status_query / osg_batch_gather / osg_apply), and every
// iteration is arithmetic over flat arrays:
//
//   histories   level-ordered (BFS); children of a node are contiguous
//   per node    kind (chance / decision / terminal), actor, parent, first_child,
//               nchild, edge index within the parent, infostate id,
//               chance probability of the incoming edge, terminal returns [P]
//   infostates  [I, Amax] fp64 tables: cumulative regrets, cumulative policy,
//               current policy (CFRInfoStateValues, cfr.h:42-98); member
//               histories listed in the reference's DFS visiting order
//
// k_cfr runs a whole batch of iterations in ONE launch by ONE workgroup:
// top-down reach pass, bottom-up value pass (one __syncthreads per tree level),
// then one thread per infostate folds its member histories' regret / average
// policy terms in DFS order — the same additions in the same order as the
// reference's recursion, so the tables are bit-comparable with the CPU oracle
// (both sides are compiled with -ffp-contract=off).  Small trees (kuhn) keep
// reach / values / tables in LDS; larger ones (leduc: 9 457 histories) keep
// them in global memory, which is L2-resident at these sizes.
//
// k_mccfr runs one external-sampling traversal per thread with an explicit
// stack over the same flat tree; regret / average-policy deltas are accumulated
// per workgroup in LDS and flushed with fp64 atomics, so that a multi-GPU job
// can all-reduce the [I, Amax] delta tables once per mini-batch (RCCL) before
// every rank folds them in.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "osg_internal.h"

using namespace osg;

namespace {

constexpr int kMaxA = 4;        // widest decision node the MCCFR frame holds (kuhn 2, leduc 3)
constexpr int kMaxPolicyRow = 8;  // widest policy row a thread regret-matches in registers (kuhn 2, leduc 3)
constexpr int kMaxFrames = 24;  // traverser decision nodes on one path
#ifndef OSG_MCCFR_FRAMES2
#define OSG_MCCFR_FRAMES2 1     // the flat ES-MCCFR kernel keeps the two upper frames of the traverser's stack in registers (0: A/B)
#endif
#ifndef OSG_MCCFR_PEEK
#define OSG_MCCFR_PEEK 0        // 1: the flat ES-MCCFR kernel forms the next draw while the node record is in flight — measured
                                // 2.7 % SLOWER (profiles/r05_ab_solvers.txt: the draws of traverser / terminal visits are wasted
                                // vector work on a SIMD that is already two thirds busy issuing); kept as a switch
#endif
constexpr double kMccfrInit = 0.000001;  // external_sampling_mccfr.h:59 kInitialTableValues

enum NodeKind : uint8_t { kChanceNode = 0, kDecisionNode = 1, kTerminalNode = 2 };

struct Tree {  // device pointers
  int H, I, A, P, D;
  const int32_t* level_off;    // [D+1]
  const int32_t* parent;       // [H]
  const int32_t* first_child;  // [H]
  const uint8_t* kind;         // [H]
  const uint8_t* nchild;       // [H]
  const uint8_t* aidx;         // [H] index of the incoming edge among the parent's children
  const int8_t* actor;         // [H] acting player, -1 at chance / terminal nodes
  const int32_t* info;         // [H] infostate id of a decision node, else -1
  const double* edge_prob;     // [H] chance probability of the incoming edge (parent is chance)
  const double* term_ret;      // [H, P] Returns() of terminal nodes
  const int32_t* mem_off;      // [I+1]
  const int32_t* mem;          // member histories of every infostate, DFS order
  const int32_t* nact;         // [I]
  const int8_t* info_player;   // [I]
};

struct Tables {  // [I, A] fp64
  double* regrets;
  double* cum;
  double* cur;
};

// ---------------------------------------------------------------------------
// CFRInfoStateValues::ApplyRegretMatching (cfr.cc:596-615) on one row.
// ---------------------------------------------------------------------------
OSG_D void regret_match_row(const double* regrets, double* policy, int n) {
  double sum_pos = 0.0;
  for (int a = 0; a < n; ++a)
    if (regrets[a] > 0) sum_pos += regrets[a];
  for (int a = 0; a < n; ++a) {
    if (sum_pos > 0) policy[a] = regrets[a] > 0 ? regrets[a] / sum_pos : 0.0;
    else policy[a] = 1.0 / n;
  }
}

// ---------------------------------------------------------------------------
// CFRSolverBase::EvaluateAndUpdatePolicy x iters (cfr.cc:263-282), one workgroup.
// ---------------------------------------------------------------------------
// kBr: one CFRBRSolver::EvaluateAndUpdatePolicy pass set (cfr_br.cc:48-83): P passes, pass p updates
// player p while every other player follows its best-response action best[i] (policy_overrides,
// cfr.cc:365-372) instead of the current policy.
template <bool kLds, bool kBr = false>
__global__ void __launch_bounds__(1024)
k_cfr(Tree t, Tables tb, double* g_reach, double* g_value, int iters, int iteration0, osg_cfr_cfg cfg,
      const int32_t* __restrict__ best = nullptr) {
  extern __shared__ double smem[];
  const int P = t.P, S = t.P + 1, A = t.A;
  double* reach = kLds ? smem : g_reach;                       // [H, P+1], chance last (cfr.cc:196,201)
  double* value = kLds ? smem + static_cast<size_t>(t.H) * S : g_value;  // [H, P]
  double* regrets = tb.regrets;
  double* cum = tb.cum;
  double* cur = tb.cur;
  const int tid = threadIdx.x, nt = blockDim.x;
  if (kLds) {  // stage the tables too: the whole solver state lives in LDS for the launch
    double* base = smem + static_cast<size_t>(t.H) * (S + P);
    regrets = base;
    cum = base + t.I * A;
    cur = base + 2 * t.I * A;
    for (int k = tid; k < t.I * A; k += nt) {
      regrets[k] = tb.regrets[k];
      cum[k] = tb.cum[k];
      cur[k] = tb.cur[k];
    }
    __syncthreads();
  }
  const int passes = (kBr || cfg.alternating_updates) ? P : 1;
  for (int it = 0; it < iters; ++it) {
    const int iteration = iteration0 + it + 1;  // ++iteration_ (cfr.cc:264)
    for (int pass = 0; pass < passes; ++pass) {
      const int upd = (kBr || cfg.alternating_updates) ? pass : -1;
      // probability of action index a at infostate i in this pass
      auto pol_at = [&](int i, int a) -> double {
        if (kBr && t.info_player[i] != upd) return a == best[i] ? 1.0 : 0.0;
        return cur[i * A + a];
      };
      // ---- reach probabilities, top-down (cfr.cc:452-454: new_reach[current_player] *= prob) ----
      for (int l = 0; l < t.D; ++l) {
        for (int h = t.level_off[l] + tid; h < t.level_off[l + 1]; h += nt) {
          if (l == 0) {
            for (int q = 0; q < S; ++q) reach[h * S + q] = 1.0;
            continue;
          }
          const int par = t.parent[h];
          const int pa = t.actor[par];
          const int slot = pa < 0 ? P : pa;
          const double pr = t.kind[par] == kChanceNode ? t.edge_prob[h] : pol_at(t.info[par], t.aidx[h]);
          for (int q = 0; q < S; ++q) {
            const double r = reach[par * S + q];
            reach[h * S + q] = (q == slot) ? r * pr : r;
          }
        }
        __syncthreads();
      }
      // ---- state values, bottom-up (cfr.cc:443-469) ----
      for (int l = t.D - 1; l >= 0; --l) {
        for (int h = t.level_off[l] + tid; h < t.level_off[l + 1]; h += nt) {
          const int k = t.kind[h];
          if (k == kTerminalNode) {
            for (int q = 0; q < P; ++q) value[h * P + q] = t.term_ret[h * P + q];
            continue;
          }
          bool pruned = false;
          if (k == kDecisionNode) {  // AllPlayersHaveZeroReachProb (cfr.cc:350-355,471-479)
            pruned = true;
            for (int q = 0; q < P; ++q) pruned &= (reach[h * S + q] == 0.0);
          }
          const int fc = t.first_child[h], nc = t.nchild[h];
          const int row = k == kDecisionNode ? t.info[h] : 0;
          for (int q = 0; q < P; ++q) {
            double v = 0.0;
            if (!pruned) {
              for (int a = 0; a < nc; ++a) {
                const double pr = k == kChanceNode ? t.edge_prob[fc + a] : pol_at(row, a);
                v += pr * value[(fc + a) * P + q];
              }
            }
            value[h * P + q] = v;
          }
        }
        __syncthreads();
      }
      // ---- regret / average-policy updates (cfr.cc:379-405), then RM+ reset and regret
      //      matching (cfr.cc:683-697).  Rows of the other players are unchanged in an
      //      alternating pass, so re-matching them (as the reference does) is a no-op. ----
      for (int i = tid; i < t.I; i += nt) {
        const int pl = t.info_player[i];
        if (upd >= 0 && pl != upd) continue;
        const int n = t.nact[i];
        for (int m = t.mem_off[i]; m < t.mem_off[i + 1]; ++m) {
          const int h = t.mem[m];
          bool pruned = true;
          for (int q = 0; q < P; ++q) pruned &= (reach[h * S + q] == 0.0);
          if (pruned) continue;
          const double self_reach = reach[h * S + pl];
          double cf_reach = 1.0;  // CounterFactualReachProb (cfr.cc:309-318)
          for (int q = 0; q < S; ++q)
            if (q != pl) cf_reach *= reach[h * S + q];
          const double vh = value[h * P + pl];
          const int fc = t.first_child[h];
          for (int a = 0; a < n; ++a) {
            const double cfr_regret = cf_reach * (value[(fc + a) * P + pl] - vh);
            regrets[i * A + a] += cfr_regret;
            const double pol = cur[i * A + a];
            if (cfg.linear_averaging) cum[i * A + a] += iteration * self_reach * pol;
            else cum[i * A + a] += self_reach * pol;
          }
        }
        if (cfg.regret_matching_plus)
          for (int a = 0; a < n; ++a)
            if (regrets[i * A + a] < 0) regrets[i * A + a] = 0;
        regret_match_row(regrets + i * A, cur + i * A, n);
      }
      __syncthreads();
    }
  }
  if (kLds) {
    for (int k = tid; k < t.I * A; k += nt) {
      tb.regrets[k] = regrets[k];
      tb.cum[k] = cum[k];
      tb.cur[k] = cur[k];
    }
  }
}

// ---------------------------------------------------------------------------
// Small trees (kuhn_poker: 58 histories): the WHOLE problem — tree structure,
// values, tables — lives in LDS for the launch; nothing but LDS traffic inside
// the iteration loop.  Per player pass:
//   A  values bottom-up, one level per step (terminal values are staged once)
//   B  one thread per decision history: reach probabilities from its root path
//      (host-precomputed, root-to-leaf order so the products round like the
//      reference's top-down recursion), then its regret / average-policy terms
//   C  one thread per infostate: fold its members' terms in DFS order, RM+ clamp,
//      regret matching
// The reference's zero-reach prune (cfr.cc:350-355) only ever changes values that
// are multiplied by an exact zero afterwards (the first all-zero node on a path
// hangs off a probability-0 edge, and an unpruned parent of a pruned child has
// counterfactual reach 0), so phase A does not need reach probabilities; phase B
// applies the prune where it is observable (no update at pruned histories).
// ---------------------------------------------------------------------------
struct SmallTree {  // device pointers to the extra host-built arrays
  const int32_t* path_off;    // [M+1] per decision history (member order)
  const int32_t* path;        // entries: slot << 24 | is_chance << 23 | index
  int M;                      // decision histories
  int n_path;
  int L0 = 0;                 // the first level that holds a decision history: the sweep of k_cfr_small stops there — the
                              // values of the chance levels above (the deals) are read by nobody (phase B reads a decision
                              // history's own value and its children's), and for kuhn_poker they were 2 of its 5 level steps
};

struct SmallGlobal {  // global-memory homes of the same arrays, for trees too big for LDS (leduc)
  double* value;         // [H, P]
  double* dreg;          // [M, A]
  double* dpol;          // [M, A]
  int32_t* skip;         // [M]
  const int32_t* meta;   // [H] kind | nchild << 2 | (actor + 1) << 10
  const int32_t* info_player;  // [I]
};

// kPath: decision entries of a root path the owner form keeps in registers.  The reach block is straight-line code
// over kPath entries x kSlots reach slots; kuhn_poker's paths hold at most 2 decisions, and with the generic 8 the
// block was 180 of the ~790 instructions a player pass issues (round 5: one wavefront runs at its instruction issue
// rate, profiles/r05a_pmc_solvers.json — fewer instructions is the only lever).  The host picks the instantiation from
// the longest path of the tree (osg_cfr::max_path_decisions).
// kW > 0 (owner form, alternating updates): every decision node has at most kW actions — the loops over a row's
// actions are unrolled and predicated instead of running as lane-masked loops (a third of the kernel's instructions
// were loop control: scalar mask bookkeeping and branches, which a lone wavefront issues one at a time like any other).
template <bool kLds, bool kOwner, int kSlots, int kPath = 8, int kW = 0>  // kSlots >= P + 1 reach slots kept in registers
__global__ void __launch_bounds__(1024)
k_cfr_small(Tree t, SmallTree st, SmallGlobal sg, Tables tb, int iters, int iteration0, osg_cfr_cfg cfg) {
  extern __shared__ double smem[];
  const int P = t.P, A = t.A, H = t.H, I = t.I, M = st.M, IA = t.I * t.A;
  const int tid = threadIdx.x, nt = blockDim.x;
  // replica = workgroup: tables of replica b live 5 * IA doubles further on (regrets | cum | cur | 2 deltas)
  tb.regrets += static_cast<size_t>(blockIdx.x) * 5 * IA;
  tb.cum += static_cast<size_t>(blockIdx.x) * 5 * IA;
  tb.cur += static_cast<size_t>(blockIdx.x) * 5 * IA;
  double *value, *regrets, *cum, *cur, *dreg, *dpol;
  const double* edge_prob;
  const int32_t *first_child, *info, *meta, *mem, *mem_off, *path_off, *path, *nact, *info_player, *level_off;
  int32_t* skip;
  if (kLds) {
    // ---- carve LDS (doubles first, then 32-bit) and stage everything once ----
    double* l_value = smem;                      // [H, P]
    double* l_edge = l_value + H * P;            // [H]
    regrets = l_edge + H;                        // [I, A]
    cum = regrets + IA;
    cur = cum + IA;
    dreg = cur + IA;                             // [M, A]
    dpol = dreg + M * A;                         // [M, A]
    int32_t* l_first = reinterpret_cast<int32_t*>(dpol + M * A);  // [H]
    int32_t* l_info = l_first + H;               // [H]
    int32_t* l_meta = l_info + H;                // [H]
    int32_t* l_mem = l_meta + H;                 // [M]
    int32_t* l_mem_off = l_mem + M;              // [I+1]
    int32_t* l_path_off = l_mem_off + (I + 1);   // [M+1]
    int32_t* l_path = l_path_off + (M + 1);      // [n_path]
    int32_t* l_nact = l_path + st.n_path;        // [I]
    int32_t* l_info_player = l_nact + I;         // [I]
    skip = l_info_player + I;                    // [M] 1 = pruned / not updated this pass
    int32_t* l_level_off = skip + M;             // [D+1]
    for (int h = tid; h < H; h += nt) {
      l_first[h] = t.first_child[h];
      l_info[h] = t.info[h];
      l_meta[h] = sg.meta[h];
      l_edge[h] = t.edge_prob[h];
      for (int q = 0; q < P; ++q) l_value[h * P + q] = t.term_ret[h * P + q];  // terminals keep these forever
    }
    for (int k = tid; k < IA; k += nt) {
      regrets[k] = tb.regrets[k];
      cum[k] = tb.cum[k];
      cur[k] = tb.cur[k];
    }
    for (int k = tid; k < M; k += nt) l_mem[k] = t.mem[k];
    for (int k = tid; k <= M; k += nt) l_path_off[k] = st.path_off[k];
    for (int k = tid; k < st.n_path; k += nt) l_path[k] = st.path[k];
    for (int k = tid; k <= I; k += nt) l_mem_off[k] = t.mem_off[k];
    for (int k = tid; k < I; k += nt) {
      l_nact[k] = t.nact[k];
      l_info_player[k] = sg.info_player[k];
    }
    for (int k = tid; k <= t.D; k += nt) l_level_off[k] = t.level_off[k];
    value = l_value; edge_prob = l_edge; first_child = l_first; info = l_info; meta = l_meta; mem = l_mem;
    mem_off = l_mem_off; path_off = l_path_off; path = l_path; nact = l_nact; info_player = l_info_player;
    level_off = l_level_off;
  } else {
    value = sg.value; regrets = tb.regrets; cum = tb.cum; cur = tb.cur; dreg = sg.dreg; dpol = sg.dpol;
    edge_prob = t.edge_prob; first_child = t.first_child; info = t.info; meta = sg.meta; mem = t.mem;
    mem_off = t.mem_off; path_off = st.path_off; path = st.path; nact = t.nact; info_player = sg.info_player;
    level_off = t.level_off; skip = sg.skip;
    for (int h = tid; h < H; h += nt)
      for (int q = 0; q < P; ++q) value[h * P + q] = t.term_ret[h * P + q];
  }
  __syncthreads();

  // kOwner (whole tree no larger than the workgroup): thread t owns history t, decision history
  // (member) t and infostate t for the whole launch; their descriptors are hoisted into registers so
  // that inside the iteration loop every phase is one LDS round trip instead of a chain of four.
  int o_k = kTerminalNode, o_fc = 0, o_nc = 0, o_row = 0, o_lvl = -1;
  int b_h = 0, b_pl = -1, b_i = 0, b_n = 0, b_fc = 0, b_e0 = 0, b_e1 = 0;
  int c_n = 0, c_pl = -1, c_m0 = 0, c_m1 = 0;
  // The root path of the owned decision history: its chance factors never change, so their product (same
  // order as the walk) is taken once; the decision entries (slot << 24 | policy index) stay in registers,
  // which turns the per-iteration reach computation into independent LDS reads instead of a
  // load -> decode -> load chain per path entry.
  constexpr int kOwnerPath = kPath;
  int b_code[kOwnerPath];
  double b_chance = 1.0;
  bool b_fast = false;
#pragma unroll
  for (int j = 0; j < kOwnerPath; ++j) b_code[j] = -1;
  if (kOwner) {
    if (tid < H) {
      const int mt = meta[tid];
      o_k = mt & 3;
      o_fc = first_child[tid];
      o_nc = (mt >> 2) & 0xFF;
      o_row = o_k == kDecisionNode ? info[tid] * A : 0;
      for (int l = 0; l < t.D; ++l)
        if (tid >= level_off[l] && tid < level_off[l + 1]) o_lvl = l;
    }
    if (tid < M) {
      b_h = mem[tid];
      b_pl = ((meta[b_h] >> 10) & 15) - 1;
      b_i = info[b_h];
      b_n = nact[b_i];
      b_fc = first_child[b_h];
      b_e0 = path_off[tid];
      b_e1 = path_off[tid + 1];
      int np = 0;
      b_fast = true;
      for (int e = b_e0; e < b_e1; ++e) {
        const int code = path[e];
        if ((code >> 23) & 1) {
          b_chance *= edge_prob[code & 0x7FFFFF];
        } else {
          if (np >= kOwnerPath) b_fast = false;
#pragma unroll
          for (int j = 0; j < kOwnerPath; ++j)
            if (j == np) b_code[j] = code & 0x0F7FFFFF;
          ++np;
        }
      }
    }
    if (tid < I) {
      c_n = nact[tid];
      c_pl = info_player[tid];
      c_m0 = mem_off[tid];
      c_m1 = mem_off[tid + 1];
    }
  }

  const int passes = (kW > 0 || cfg.alternating_updates) ? P : 1;
  for (int it = 0; it < iters; ++it) {
    const int iteration = iteration0 + it + 1;
    for (int pass = 0; pass < passes; ++pass) {
      const int upd = (kW > 0 || cfg.alternating_updates) ? pass : -1;
      const int q0 = upd >= 0 ? upd : 0, q1 = upd >= 0 ? upd + 1 : P;
      // value of one non-terminal history from its children (cfr.cc:443-469)
      auto do_node = [&](int h, int k, int fc, int nc, int row) {
        if constexpr (kW > 0) {   // (launched for alternating updates only: upd >= 0) one value per history, the
          double v = 0.0;         // updating player's; a decision row is walked unrolled
          if (k == kChanceNode) {
            for (int a = 0; a < nc; ++a) v += edge_prob[fc + a] * value[(fc + a) * P + upd];
          } else {
#pragma unroll
            for (int a = 0; a < kW; ++a) {
              const int aa = a < nc ? a : 0;
              const double term = cur[row + aa] * value[(fc + aa) * P + upd];
              v = a < nc ? v + term : v;
            }
          }
          value[h * P + upd] = v;
          return;
        }
        for (int q = q0; q < q1; ++q) {
          double v = 0.0;
          for (int a = 0; a < nc; ++a) {
            const double pr = k == kChanceNode ? edge_prob[fc + a] : cur[row + a];
            v += pr * value[(fc + a) * P + q];
          }
          value[h * P + q] = v;
        }
      };
      // one decision history: reach from the root path, then its regret / average-policy terms
      auto do_member = [&](int m, int h, int pl, int i, int n, int fc, int e0, int e1) {
        if (upd >= 0 && pl != upd) { skip[m] = 1; return; }
        double reach[kSlots];
#pragma unroll
        for (int q = 0; q < kSlots; ++q) reach[q] = 1.0;
        if (kOwner && b_fast) {
          // (opaque per pass: otherwise every `slot == q` comparison is hoisted out of the iteration loop as a lane mask
          // in a scalar register pair and spilled to vector lanes — see k_cfr_split)
#ifndef OSG_AB_R4_REGS
#pragma unroll
          for (int j = 0; j < kOwnerPath; ++j) asm volatile("" : "+v"(b_code[j]));
#endif
          double pr[kOwnerPath];
#pragma unroll
          for (int j = 0; j < kOwnerPath; ++j) pr[j] = cur[b_code[j] >= 0 ? (b_code[j] & 0x7FFFFF) : 0];
#pragma unroll
          for (int q = 0; q < kSlots; ++q) reach[q] = (q == P) ? b_chance : 1.0;
#pragma unroll
          for (int j = 0; j < kOwnerPath; ++j) {
            const int slot = b_code[j] >= 0 ? (b_code[j] >> 24) & 0xF : -1;
#pragma unroll
            for (int q = 0; q < kSlots; ++q) reach[q] = (q == slot) ? reach[q] * pr[j] : reach[q];
          }
        } else {
          for (int e = e0; e < e1; ++e) {
            const int code = path[e];
            const int slot = (code >> 24) & 0xF, idx = code & 0x7FFFFF;
            const double pr = ((code >> 23) & 1) ? edge_prob[idx] : cur[idx];
#pragma unroll
            for (int q = 0; q < kSlots; ++q) reach[q] = (q == slot) ? reach[q] * pr : reach[q];
          }
        }
        bool pruned = true;  // AllPlayersHaveZeroReachProb (cfr.cc:471-479)
        double self_reach = 0.0, cf_reach = 1.0;
#pragma unroll
        for (int q = 0; q < kSlots; ++q) {
          if (q < P) pruned &= (reach[q] == 0.0);
          if (q == pl) self_reach = reach[q];
          else if (q <= P) cf_reach *= reach[q];  // CounterFactualReachProb (cfr.cc:309-318), chance slot = P
        }
        skip[m] = pruned ? 1 : 0;
        if (pruned) return;
        const double vh = value[h * P + pl];
        if constexpr (kW > 0) {
          double cv[kW], pol[kW];
#pragma unroll
          for (int a = 0; a < kW; ++a) {   // every operand requested before the first is used
            const int aa = a < n ? a : 0;
            cv[a] = value[(fc + aa) * P + pl];
            pol[a] = cur[i * A + aa];
          }
#pragma unroll
          for (int a = 0; a < kW; ++a)
            if (a < n) {
              dreg[m * A + a] = cf_reach * (cv[a] - vh);
              dpol[m * A + a] = cfg.linear_averaging ? iteration * self_reach * pol[a] : self_reach * pol[a];
            }
          return;
        }
        for (int a = 0; a < n; ++a) {
          dreg[m * A + a] = cf_reach * (value[(fc + a) * P + pl] - vh);
          const double pol = cur[i * A + a];
          dpol[m * A + a] = cfg.linear_averaging ? iteration * self_reach * pol : self_reach * pol;
        }
      };
      // one infostate: fold its members' terms in DFS order, RM+ clamp, regret matching
      auto do_info = [&](int i, int n, int pl, int m0, int m1) {
        if (upd >= 0 && pl != upd) return;
        for (int m = m0; m < m1; ++m) {
          if (skip[m]) continue;
          for (int a = 0; a < n; ++a) {
            regrets[i * A + a] += dreg[m * A + a];
            cum[i * A + a] += dpol[m * A + a];
          }
        }
        if (cfg.regret_matching_plus)
          for (int a = 0; a < n; ++a)
            if (regrets[i * A + a] < 0) regrets[i * A + a] = 0;
        regret_match_row(regrets + i * A, cur + i * A, n);
      };
      // the same for the owner form with rows of up to kMaxA actions: the row in registers for the whole fold (one LDS
      // read and one write-back instead of a read-modify-write per member and action) and 1 / n as an exact constant
      // (a correctly rounded quotient either way) instead of a division sequence — the same additions in the same order
      auto do_info_owner = [&](int i, int n, int pl, int m0, int m1) {
        if (upd >= 0 && pl != upd) return;
        constexpr int kFW = kW > 0 ? kW : kMaxA;   // the widest row this instantiation meets
        double r_reg[kFW], r_cum[kFW];
#pragma unroll
        for (int a = 0; a < kFW; ++a) {
          const int k = i * A + (a < n ? a : 0);
          r_reg[a] = regrets[k];
          r_cum[a] = cum[k];
        }
        if constexpr (kW > 0) {
          for (int m = m0; m < m1; m += 2) {   // two members per step: their records are requested together, added in order
            const int mb = m + 1 < m1 ? m + 1 : m;
            const int sa = skip[m], sb = skip[mb];
            double ta[kFW], ua[kFW], tb2[kFW], ub[kFW];
#pragma unroll
            for (int a = 0; a < kFW; ++a) {
              const int aa = a < n ? a : 0;
              ta[a] = dreg[m * A + aa]; ua[a] = dpol[m * A + aa];
              tb2[a] = dreg[mb * A + aa]; ub[a] = dpol[mb * A + aa];
            }
#pragma unroll
            for (int a = 0; a < kFW; ++a)
              if (a < n && !sa) { r_reg[a] += ta[a]; r_cum[a] += ua[a]; }
#pragma unroll
            for (int a = 0; a < kFW; ++a)
              if (a < n && !sb && mb != m) { r_reg[a] += tb2[a]; r_cum[a] += ub[a]; }
          }
        } else {
          for (int m = m0; m < m1; ++m) {
            if (skip[m]) continue;
#pragma unroll
            for (int a = 0; a < kFW; ++a)
              if (a < n) {
                r_reg[a] += dreg[m * A + a];
                r_cum[a] += dpol[m * A + a];
              }
          }
        }
        double sum_pos = 0.0;
#pragma unroll
        for (int a = 0; a < kFW; ++a) {
          if (cfg.regret_matching_plus && r_reg[a] < 0) r_reg[a] = 0;
          if (a < n && r_reg[a] > 0) sum_pos += r_reg[a];
        }
        const double inv_n = n == 1 ? 1.0 : (n == 2 ? 0.5 : (n == 3 ? 1.0 / 3.0 : 0.25));
#pragma unroll
        for (int a = 0; a < kFW; ++a)
          if (a < n) {
            regrets[i * A + a] = r_reg[a];
            cum[i * A + a] = r_cum[a];
            cur[i * A + a] = sum_pos > 0 ? (r_reg[a] > 0 ? r_reg[a] / sum_pos : 0.0) : inv_n;
          }
      };
      // ---- A: values, bottom-up.  Alternating passes only need the updating player's value. ----
      for (int l = t.D - 2; l >= st.L0; --l) {  // the last level holds terminals only
        if (kOwner) {
          if (o_lvl == l && o_k != kTerminalNode) do_node(tid, o_k, o_fc, o_nc, o_row);
        } else {
          for (int h = level_off[l] + tid; h < level_off[l + 1]; h += nt) {
            const int mt = meta[h];
            const int k = mt & 3;
            if (k == kTerminalNode) continue;
            do_node(h, k, first_child[h], (mt >> 2) & 0xFF, k == kDecisionNode ? info[h] * A : 0);
          }
        }
        __syncthreads();
      }
      // ---- B: per decision history ----
      if (kOwner) {
        if (tid < M) do_member(tid, b_h, b_pl, b_i, b_n, b_fc, b_e0, b_e1);
      } else {
        for (int m = tid; m < M; m += nt) {
          const int h = mem[m], i = info[h];
          do_member(m, h, ((meta[h] >> 10) & 15) - 1, i, nact[i], first_child[h], path_off[m], path_off[m + 1]);
        }
      }
      __syncthreads();
      // ---- C: per infostate ----
      if (kOwner) {
#ifdef OSG_AB_R4_REGS
        if (tid < I) do_info(tid, c_n, c_pl, c_m0, c_m1);
#else
        if (tid < I) do_info_owner(tid, c_n, c_pl, c_m0, c_m1);   // (the host launches the owner form for A <= kMaxA only)
#endif
      } else {
        for (int i = tid; i < I; i += nt) do_info(i, nact[i], info_player[i], mem_off[i], mem_off[i + 1]);
      }
      __syncthreads();
    }
  }
  if (kLds) {
    for (int k = tid; k < IA; k += nt) {
      tb.regrets[k] = regrets[k];
      tb.cum[k] = cum[k];
      tb.cur[k] = cur[k];
    }
  }
}

// ---------------------------------------------------------------------------
// Trees that start with their chance deals (leduc_poker: 9 457 histories, two deal levels, then 30 subtrees of
// 314 histories): ONE WORKGROUP PER DEAL SUBTREE instead of one workgroup for the whole tree.  Everything a pass
// touches inside a subtree — values, the regret / policy rows of the subtree's infostates, the descriptors of the
// thread's own history, member and infostate — lives in LDS and registers (the kOwner form of k_cfr_small), so a
// tree level costs an LDS round trip instead of an L2 one.  What crosses subtrees is exactly what the reference's
// recursion adds up across deals (cfr.cc:379-405): an infostate's regret / average-policy terms come from member
// histories in several subtrees.  Per player pass:
//   A  values bottom-up inside the subtree (one __syncthreads per level);
//   B  one thread per decision history of the subtree: reach from the root path, then its record — own reach (-1 when
//      every player's reach is zero: nothing to add, cfr.cc:471-479) and the A regret terms — written THROUGH to
//      memory (agent-scope stores) into the pass's buffer (two buffers, by pass parity).  The average-policy term is
//      own reach x policy: every reader forms it from its own bit-identical copy of the row (cfr.cc:398-404), so it
//      does not travel;
//   -- one grid barrier: a counter every workgroup bumps once its stores have drained, polled by one lane --
//   C  every workgroup folds, for each infostate that has a member in ITS subtree, ALL that infostate's members'
//      records (agent-scope loads: they bypass the caches that may hold the previous pass's lines; up to kSplitChunk
//      members per round trip) in DFS order into the row held in registers — the same additions in the same order in
//      every workgroup that keeps the row, so the copies stay bit-identical and equal to the single-workgroup
//      kernels' tables — then RM+ clamp and regret matching back into its LDS rows.
// Where a pass's 8.3 us go (leduc, wall_clock64 of workgroup 0): A 2.4 (nine levels of LDS round trip + barrier), B 1.3,
// drain 0.4, counter barrier 1.8, C 1.85 (one memory round trip + fold), regret matching + barrier 0.6.
// One barrier per pass, no second one: the rows a subtree needs next are the rows it has just folded itself.
// The grid (one workgroup per subtree, <= the number of CUs, ~100 KB of LDS each) is launched COOPERATIVELY: the runtime
// starts it only when all its workgroups fit the device at once, so the barrier cannot starve behind another stream's
// kernels (tests/test_gpu_cfr.py runs it beside a matmul loop); the spin keeps a wall-clock bound against a hung device.
// ---------------------------------------------------------------------------
struct SplitTree {
  int G, L, NL, NM, NI;          // subtrees, cut level, padded histories / members / infostates per subtree
  const int32_t* nloc;           // [G] histories of the subtree
  const int32_t* hist_desc;      // [G, NL] kind | nchild << 2 | level << 10 | (actor + 1) << 16
  const int32_t* hist_fc;        // [G, NL] LOCAL index of the first child
  const int32_t* hist_row;       // [G, NL] info * A of a decision node
  const int32_t* hist_glob;      // [G, NL] the history's index in the whole tree
  const int32_t* mem_m;          // [G, NM] member index (position in Tree::mem), -1 = padding
  const int32_t* mem_hloc;       // [G, NM] its history, local index
  const int32_t* info_list;      // [G, NI] infostates with a member in the subtree, -1 = padding
  double* terms;                 // [2][M][kSplitRec]: buffer (pass parity) x {own reach or -1, A regret terms} per member
  unsigned int* bar;             // [0] arrival counter (zero between launches), [1] error flag, [2] sticky error, [3] exit counter
  unsigned int* host_err;        // pinned host word raised on a timeout: the host's next call reads it without a copy
};

OSG_D void store_through(double* p, double v) {   // agent scope: written through to memory, visible to every CU
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(__double_as_longlong(v)),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
OSG_D double load_through(const double* p) {      // agent scope: never served from a stale L1 / L2 line
  return __longlong_as_double(static_cast<long long>(__hip_atomic_load(
      reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
}

constexpr int kSplitMaxA = 4;       // widest policy row the split kernel folds
constexpr int kSplitOwnerPath = 10;  // decision entries of a root path kept in registers
constexpr int kSplitRec = 1 + kSplitMaxA;  // doubles per member and buffer: own reach, regret terms
constexpr int kSplitChunk = 6;       // members whose records are requested together
// kBr: the pass set of CFRBRSolver::EvaluateAndUpdatePolicy (cfr_br.cc:70-81) — P passes, pass p updates player p while
// every other player follows best[i] (k_eval_jobs wrote it): the pass reads an effective policy `eff` (the updating
// player's rows of `cur`, one-hot rows for the others) that is rebuilt in LDS at the start of every pass.
// kBound: the launch bound the instantiation is compiled for.  A subtree of leduc is 314 histories = 320 threads = 5
// wavefronts, at most 2 per SIMD: compiled for 1024 threads the kernel was capped at 128 VGPRs and spilled (24 vector +
// 69 scalar registers, 84 B of scratch per lane — round 4's code object); compiled for 512 it has 256 and keeps
// everything in registers.  split_kernel() picks the instantiation by the launch size.
// kW > 0 (alternating updates or kBr: one value per history): decision rows of at most kW actions are walked unrolled
// and predicated instead of as lane-masked loops (as in k_cfr_small).
template <int kSlots, bool kBr = false, int kBound = 1024, int kW = 0>  // kSlots >= P + 1
__global__ void __launch_bounds__(kBound)
k_cfr_split(Tree t, SmallTree st, SplitTree sp, Tables tb, int iters, int iteration0, osg_cfr_cfg cfg,
            const int32_t* __restrict__ best = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int P = t.P, A = t.A, IA = t.I * t.A, M = st.M;
  const int tid = threadIdx.x, g = blockIdx.x;
  double* value = smem;                         // [NL, P]
  double* l_edge = value + sp.NL * P;           // [NL] chance probability of the incoming edge
  double* regrets = l_edge + sp.NL;             // [I, A] (only the rows of this subtree's infostates are kept current)
  double* cum = regrets + IA;
  double* cur = cum + IA;
  int* s_ok = reinterpret_cast<int*>(cur + IA);  // (in the dynamic region: a static would shift its 16-byte base)
  double* eff = cur + IA + 2;                    // [I, A], kBr only
  const double* pol = kBr ? eff : cur;
  for (int k = tid; k < IA; k += blockDim.x) {
    regrets[k] = tb.regrets[k];
    cum[k] = tb.cum[k];
    cur[k] = tb.cur[k];
  }
  // ---- the thread's own history, member and infostate: descriptors in registers for the whole launch ----
  const int nloc = sp.nloc[g];
  int o_k = kTerminalNode, o_fc = 0, o_nc = 0, o_row = 0, o_lvl = -1;
  if (tid < nloc) {
    const int d = sp.hist_desc[g * sp.NL + tid];
    o_k = d & 3; o_nc = (d >> 2) & 0xFF; o_lvl = (d >> 10) & 0x3F;
    o_fc = sp.hist_fc[g * sp.NL + tid];
    o_row = sp.hist_row[g * sp.NL + tid];
    const int hg = sp.hist_glob[g * sp.NL + tid];
    l_edge[tid] = t.edge_prob[hg];
    for (int q = 0; q < P; ++q) value[tid * P + q] = o_k == kTerminalNode ? t.term_ret[hg * P + q] : 0.0;
  }
  constexpr int kOwnerPath = kSplitOwnerPath;
  int b_m = -1, b_h = 0, b_pl = -1, b_i = 0, b_n = 0, b_fc = 0;
  int b_code[kOwnerPath];
  double b_chance = 1.0;
#pragma unroll
  for (int j = 0; j < kOwnerPath; ++j) b_code[j] = -1;
  if (tid < sp.NM) {
    b_m = sp.mem_m[g * sp.NM + tid];
    if (b_m >= 0) {
      b_h = sp.mem_hloc[g * sp.NM + tid];
      const int d = sp.hist_desc[g * sp.NL + b_h];
      b_pl = ((d >> 16) & 15) - 1;
      b_i = sp.hist_row[g * sp.NL + b_h] / A;
      b_n = t.nact[b_i];
      b_fc = sp.hist_fc[g * sp.NL + b_h];
      int np = 0;
      for (int e = st.path_off[b_m]; e < st.path_off[b_m + 1]; ++e) {  // (the host checked: <= kOwnerPath decisions)
        const int code = st.path[e];
        if ((code >> 23) & 1) {
          b_chance *= t.edge_prob[code & 0x7FFFFF];
        } else {
#pragma unroll
          for (int j = 0; j < kOwnerPath; ++j)
            if (j == np) b_code[j] = code & 0x0F7FFFFF;
          ++np;
        }
      }
    }
  }
  int c_i = -1, c_n = 0, c_pl = -1, c_m0 = 0, c_m1 = 0;
  if (tid < sp.NI) {
    c_i = sp.info_list[g * sp.NI + tid];
    if (c_i >= 0) {
      c_n = t.nact[c_i];
      c_pl = t.info_player[c_i];
      c_m0 = t.mem_off[c_i];
      c_m1 = t.mem_off[c_i + 1];
    }
  }
  __syncthreads();

  const int passes = (kW > 0 || kBr || cfg.alternating_updates) ? P : 1;
  unsigned int epoch = 0;
  for (int it = 0; it < iters; ++it) {
    const int iteration = iteration0 + it + 1;
    for (int pass = 0; pass < passes; ++pass) {
      const int upd = (kW > 0 || kBr || cfg.alternating_updates) ? pass : -1;
      const int q0 = upd >= 0 ? upd : 0, q1 = upd >= 0 ? upd + 1 : P;
      if (kBr) {  // policy_overrides (cfr.cc:365-372)
        for (int i = tid; i < t.I; i += blockDim.x) {
          const bool mine = t.info_player[i] == upd;
          const int bi = best[i];
          for (int a = 0; a < A; ++a) eff[i * A + a] = mine ? cur[i * A + a] : (a == bi ? 1.0 : 0.0);
        }
        __syncthreads();
      }
      // ---- A: values, bottom-up inside the subtree (cfr.cc:443-469) ----
      for (int l = t.D - 2; l >= sp.L; --l) {
        if (kW > 0 && o_lvl == l && o_k != kTerminalNode) {   // (kW > 0 is launched with upd >= 0 only)
          double v = 0.0;
          if (o_k == kChanceNode) {
            for (int a = 0; a < o_nc; ++a) v += l_edge[o_fc + a] * value[(o_fc + a) * P + upd];
          } else {
#pragma unroll
            for (int a = 0; a < (kW > 0 ? kW : 1); ++a) {
              const int aa = a < o_nc ? a : 0;
              const double term = pol[o_row + aa] * value[(o_fc + aa) * P + upd];
              v = a < o_nc ? v + term : v;
            }
          }
          value[tid * P + upd] = v;
        } else if (kW == 0 && o_lvl == l && o_k != kTerminalNode) {
          for (int q = q0; q < q1; ++q) {
            double v = 0.0;
            for (int a = 0; a < o_nc; ++a) {   // (six children per round trip with clamped indices: 18.6 vs 16.9 us per
              // iteration — the sweep is bound by the instructions of a lone wavefront, not by LDS round trips)
              const double pr = o_k == kChanceNode ? l_edge[o_fc + a] : pol[o_row + a];
              v += pr * value[(o_fc + a) * P + q];
            }
            value[tid * P + q] = v;
          }
        }
        __syncthreads();
      }
      // ---- B: the thread's decision history: reach from its root path, regret / average-policy terms ----
      double* terms = sp.terms + static_cast<size_t>(epoch & 1u) * M * kSplitRec;  // the pass's buffer: [M][1 + kSplitMaxA]
      if (b_m >= 0 && (upd < 0 || b_pl == upd)) {
        // (the path codes are loop-invariant per thread: left alone, the compiler hoists every `slot == q` comparison out
        // of the iteration loop as a 64-bit lane mask in a scalar register pair — 10 entries x kSlots masks = 60+ scalar
        // registers held across the loop and spilled to vector lanes.  An empty asm makes the codes opaque per pass, so
        // the comparisons are formed where they are used: ~30 vector compares per pass instead of 69 spilled registers)
#ifndef OSG_AB_R4_REGS
#pragma unroll
        for (int j = 0; j < kOwnerPath; ++j) asm volatile("" : "+v"(b_code[j]));
#endif
        double pr[kOwnerPath];
#pragma unroll
        for (int j = 0; j < kOwnerPath; ++j) pr[j] = pol[b_code[j] >= 0 ? (b_code[j] & 0x7FFFFF) : 0];
        double reach[kSlots];
#pragma unroll
        for (int q = 0; q < kSlots; ++q) reach[q] = (q == P) ? b_chance : 1.0;
#pragma unroll
        for (int j = 0; j < kOwnerPath; ++j) {
          const int slot = b_code[j] >= 0 ? (b_code[j] >> 24) & 0xF : -1;
#pragma unroll
          for (int q = 0; q < kSlots; ++q) reach[q] = (q == slot) ? reach[q] * pr[j] : reach[q];
        }
        bool pruned = true;  // AllPlayersHaveZeroReachProb (cfr.cc:471-479)
        double self_reach = 0.0, cf_reach = 1.0;
#pragma unroll
        for (int q = 0; q < kSlots; ++q) {
          if (q < P) pruned &= (reach[q] == 0.0);
          if (q == b_pl) self_reach = reach[q];
          else if (q <= P) cf_reach *= reach[q];  // CounterFactualReachProb (cfr.cc:309-318), chance slot = P
        }
        // own reach first (-1: pruned, nothing to add), then the A regret terms; the average-policy term is
        // own reach x policy, which every reader forms from its own bit-identical copy of the row (cfr.cc:398-404)
        double* rec = terms + static_cast<size_t>(b_m) * kSplitRec;
        if (pruned) {
          store_through(rec, -1.0);
        } else {
          store_through(rec, cfg.linear_averaging ? iteration * self_reach : self_reach);
          const double vh = value[b_h * P + b_pl];
          for (int a = 0; a < b_n; ++a) store_through(rec + 1 + a, cf_reach * (value[(b_fc + a) * P + b_pl] - vh));
        }
      }
      // ---- the grid barrier: every storing wave drains, one lane signals, one lane polls ----
      ++epoch;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(&sp.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int want = epoch * static_cast<unsigned int>(sp.G);
        int ok = 0;
        // (the launch is cooperative: every workgroup IS resident; the bound — 4 s of the 100 MHz wall clock — only
        // keeps a broken device from spinning for ever)
        const unsigned long long t0 = wall_clock64();
        for (;;) {
          if (__hip_atomic_load(&sp.bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = 1; break; }
          if (__hip_atomic_load(&sp.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          if (wall_clock64() - t0 > 400000000ull) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) {
          __hip_atomic_store(&sp.bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&sp.bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sticky: read by the host
          __hip_atomic_store(sp.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        *s_ok = ok;
      }
      __syncthreads();
      if (!*s_ok) return;  // a workgroup never arrived within seconds (a hung device): leave, the host reports it
      // ---- C: the thread's infostate: fold ALL its members' terms in DFS order, RM+ clamp, regret matching ----
      if (c_i >= 0 && (upd < 0 || c_pl == upd)) {
        // the row in registers for the whole fold (one LDS read, one write-back); kSplitChunk members' records are
        // requested together (clamped indices: independent loads, one round trip per chunk), then added in member order
        double r_reg[kSplitMaxA], r_cum[kSplitMaxA], r_cur[kSplitMaxA];
#pragma unroll
        for (int a = 0; a < kSplitMaxA; ++a) {
          const int k = c_i * A + (a < c_n ? a : 0);
          r_reg[a] = regrets[k]; r_cum[a] = cum[k]; r_cur[a] = cur[k];
        }
        for (int m0 = c_m0; m0 < c_m1; m0 += kSplitChunk) {
          double own[kSplitChunk], rt[kSplitChunk][kSplitMaxA];
#pragma unroll
          for (int j = 0; j < kSplitChunk; ++j) {
            const int m = m0 + j < c_m1 ? m0 + j : c_m1 - 1;
            const double* rec = terms + static_cast<size_t>(m) * kSplitRec;
            own[j] = load_through(rec);
#pragma unroll
            for (int a = 0; a < kSplitMaxA; ++a) rt[j][a] = a < c_n ? load_through(rec + 1 + a) : 0.0;
          }
#pragma unroll
          for (int j = 0; j < kSplitChunk; ++j) {
            if (m0 + j >= c_m1 || own[j] < 0.0) continue;
#pragma unroll
            for (int a = 0; a < kSplitMaxA; ++a) {
              r_reg[a] += rt[j][a];
              r_cum[a] += own[j] * r_cur[a];
            }
          }
        }
        // RM+ clamp (cfr.cc:265-273 with regret_matching_plus) and regret matching (regret_match_row, unrolled; 1 / n as
        // an exact constant — a correctly rounded quotient either way — instead of a division sequence)
        const double inv_n = c_n == 1 ? 1.0 : (c_n == 2 ? 0.5 : (c_n == 3 ? 1.0 / 3.0 : 0.25));
        double sum_pos = 0.0;
#pragma unroll
        for (int a = 0; a < kSplitMaxA; ++a) {
          if (cfg.regret_matching_plus && r_reg[a] < 0) r_reg[a] = 0;
          if (a < c_n && r_reg[a] > 0) sum_pos += r_reg[a];
        }
#pragma unroll
        for (int a = 0; a < kSplitMaxA; ++a) {
          if (a < c_n) {
            const double matched = sum_pos > 0 ? (r_reg[a] > 0 ? r_reg[a] / sum_pos : 0.0) : inv_n;
            regrets[c_i * A + a] = r_reg[a];
            cum[c_i * A + a] = r_cum[a];
            cur[c_i * A + a] = matched;
          }
        }
      }
      __syncthreads();
    }
  }
  // every workgroup writes the rows it kept (copies of one row are bit-identical: the same additions in the same order)
  if (c_i >= 0) {
    for (int a = 0; a < A; ++a) {
      tb.regrets[c_i * A + a] = regrets[c_i * A + a];
      tb.cum[c_i * A + a] = cum[c_i * A + a];
      tb.cur[c_i * A + a] = cur[c_i * A + a];
    }
  }
  // the last workgroup to leave zeroes the barrier's counters for the next launch (every workgroup has passed the last
  // barrier by then): no fill launch per call — 5 us of a one-iteration launch's ~50
  if (tid == 0) {
    const unsigned int left = __hip_atomic_fetch_add(&sp.bar[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == static_cast<unsigned int>(sp.G) - 1u) {
      __hip_atomic_store(&sp.bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sp.bar[3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// (the row-width instantiation: two players, rows of exactly up to 3 actions — leduc_poker — with one value per history)
static const void* split_kernel_w3() { return reinterpret_cast<const void*>(&k_cfr_split<3, false, 512, 3>); }
template <int kBound>
static const void* split_kernel_bound(int P, bool br) {
  if (br) return P == 2 ? reinterpret_cast<const void*>(&k_cfr_split<3, true, kBound>)
                        : (P == 3 ? reinterpret_cast<const void*>(&k_cfr_split<4, true, kBound>)
                                  : reinterpret_cast<const void*>(&k_cfr_split<kMaxPlayers + 1, true, kBound>));
  return P == 2 ? reinterpret_cast<const void*>(&k_cfr_split<3, false, kBound>)
                : (P == 3 ? reinterpret_cast<const void*>(&k_cfr_split<4, false, kBound>)
                          : reinterpret_cast<const void*>(&k_cfr_split<kMaxPlayers + 1, false, kBound>));
}
// The instantiation for a launch of `threads` threads per workgroup (see kBound above).
static const void* split_kernel(int P, bool br, int threads, int A = 0, bool one_value = false) {
#ifdef OSG_AB_R4_REGS   // measurement only (tools/build_variant.sh): round 4's instantiation, bound 1024 for every launch
  return split_kernel_bound<1024>(P, br);
#else
  // (the CFR-BR pass set keeps the loop form: its row-width instantiation parks five scalar registers in vector lanes)
  if (threads <= 512 && P == 2 && A == 3 && one_value && !br && !std::getenv("OSG_CFR_SPLIT_W0")) return split_kernel_w3();
  return threads <= 512 ? split_kernel_bound<512>(P, br) : split_kernel_bound<1024>(P, br);
#endif
}

// ---------------------------------------------------------------------------
// Large trees (3-player leduc: 1.8 M histories): the same three phases as k_cfr_small, but
// every phase is a full-grid launch — one kernel per tree level for the values, one for the
// per-history terms, one for the per-infostate fold — so the whole chip works on one tree and
// the stream order provides the barriers.  Same additions in the same order: tables are
// bit-identical with the single-workgroup kernels.
// ---------------------------------------------------------------------------
struct GridCfr {
  Tree t;
  const int32_t* path_off;
  const int32_t* path;
  const int32_t* meta;         // [H] kind | nchild << 2 | (actor + 1) << 10
  const int32_t* info_player;  // [I]
  double* value;               // [H, P]
  double* dreg;                // [M, A]
  double* dpol;                // [M, A]
  int32_t* skip;               // [M]
  Tables tb;
  int M;
  const double* pol = nullptr; // [I, A] the policy a pass plays: tb.cur, or CFR-BR's effective policy (k_gcfr_effpol)
};

__global__ void __launch_bounds__(256) k_gcfr_init_values(GridCfr g) {
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h >= g.t.H) return;
  for (int q = 0; q < g.t.P; ++q) g.value[h * g.t.P + q] = g.t.term_ret[h * g.t.P + q];
}

__global__ void __launch_bounds__(256) k_gcfr_level(GridCfr g, int begin, int end, int q0, int q1) {
  const int h = begin + blockIdx.x * 256 + threadIdx.x;
  if (h >= end) return;
  const int mt = g.meta[h];
  const int k = mt & 3;
  if (k == kTerminalNode) return;
  const int P = g.t.P, A = g.t.A;
  const int fc = g.t.first_child[h], nc = (mt >> 2) & 0xFF;
  const int row = k == kDecisionNode ? g.t.info[h] * A : 0;
  for (int q = q0; q < q1; ++q) {
    double v = 0.0;
    for (int a = 0; a < nc; ++a) {
      const double pr = k == kChanceNode ? g.t.edge_prob[fc + a] : g.pol[row + a];
      v += pr * g.value[(fc + a) * P + q];
    }
    g.value[h * P + q] = v;
  }
}

__global__ void __launch_bounds__(256) k_gcfr_members(GridCfr g, int upd, int iteration, osg_cfr_cfg cfg) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= g.M) return;
  const int P = g.t.P, A = g.t.A;
  const int h = g.t.mem[m];
  const int pl = ((g.meta[h] >> 10) & 15) - 1;
  if (upd >= 0 && pl != upd) { g.skip[m] = 1; return; }
  double reach[kMaxPlayers + 1];
#pragma unroll
  for (int q = 0; q <= kMaxPlayers; ++q) reach[q] = 1.0;
  for (int e = g.path_off[m]; e < g.path_off[m + 1]; ++e) {
    const int code = g.path[e];
    const int slot = (code >> 24) & 0xF, idx = code & 0x7FFFFF;
    const double pr = ((code >> 23) & 1) ? g.t.edge_prob[idx] : g.pol[idx];
#pragma unroll
    for (int q = 0; q <= kMaxPlayers; ++q) reach[q] = (q == slot) ? reach[q] * pr : reach[q];
  }
  bool pruned = true;
  double self_reach = 0.0, cf_reach = 1.0;
#pragma unroll
  for (int q = 0; q <= kMaxPlayers; ++q) {
    if (q < P) pruned &= (reach[q] == 0.0);
    if (q == pl) self_reach = reach[q];
    else if (q <= P) cf_reach *= reach[q];
  }
  g.skip[m] = pruned ? 1 : 0;
  if (pruned) return;
  const int i = g.t.info[h], n = g.t.nact[i], fc = g.t.first_child[h];
  const double vh = g.value[h * P + pl];
  for (int a = 0; a < n; ++a) {
    g.dreg[m * A + a] = cf_reach * (g.value[(fc + a) * P + pl] - vh);
    const double pol = g.pol[i * A + a];   // (the member's own row: the current policy also under CFR-BR's overrides)
    g.dpol[m * A + a] = cfg.linear_averaging ? iteration * self_reach * pol : self_reach * pol;
  }
}
// CFR-BR on large trees (cfr_br.cc:70-81, policy_overrides cfr.cc:365-372): the policy pass `upd` plays — the updating
// player's rows of the current policy, the others' best-response actions (best[i], left by the evaluation) as one-hot rows.
__global__ void __launch_bounds__(256) k_gcfr_effpol(GridCfr g, int upd, const int32_t* __restrict__ best, double* __restrict__ eff) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.t.I) return;
  const int A = g.t.A;
  const bool own = g.info_player[i] == upd;
  for (int a = 0; a < A; ++a) eff[i * A + a] = own ? g.tb.cur[i * A + a] : (a == best[i] ? 1.0 : 0.0);
}

// One WAVEFRONT per infostate (round 5): the lanes fetch the members' skip flags and terms together and the sums are added
// in member order from the lanes' registers (readlane with a uniform index, pruned members stepped over through the
// ballot of the live ones) — a thread per infostate had walked its ~40 members one dependent load after the other.  The
// additions are cfr.cc:379-405's in its order: bit-identical with every other CFR kernel here.
OSG_D double readlane_f64(double v, int lane);   // (defined with k_cfr_sub's helpers)
__global__ void __launch_bounds__(256) k_gcfr_fold(GridCfr g, int upd, osg_cfr_cfg cfg) {
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (i >= g.t.I) return;                                   // (wave-uniform)
  if (upd >= 0 && g.info_player[i] != upd) return;
  const int A = g.t.A, n = g.t.nact[i];
  const int m0 = g.t.mem_off[i], cnt = g.t.mem_off[i + 1] - m0;
  for (int a = 0; a < n; ++a) {
    double reg = g.tb.regrets[i * A + a], cum = g.tb.cum[i * A + a];
    for (int c0 = 0; c0 < cnt; c0 += 64) {
      const int m = m0 + c0 + lane;
      const bool live = c0 + lane < cnt && g.skip[m] == 0;
      double dr = 0.0, dp = 0.0;
      if (live) { dr = g.dreg[m * A + a]; dp = g.dpol[m * A + a]; }
      for (unsigned long long todo = __ballot(live); todo != 0ull; todo &= todo - 1ull) {
        const int j = __builtin_ctzll(todo);
        reg += readlane_f64(dr, j);
        cum += readlane_f64(dp, j);
      }
    }
    if (lane == 0) { g.tb.regrets[i * A + a] = reg; g.tb.cum[i * A + a] = cum; }
  }
  if (lane != 0) return;
  if (cfg.regret_matching_plus)
    for (int a = 0; a < n; ++a)
      if (g.tb.regrets[i * A + a] < 0) g.tb.regrets[i * A + a] = 0;
  regret_match_row(g.tb.regrets + i * A, g.tb.cur + i * A, n);
}

// ---------------------------------------------------------------------------
// Large trees as ONE persistent, cooperative launch (k_cfr_sub): the multi-launch form above spends an iteration of
// 3-player leduc_poker (1.83 M histories) on ~58 launch boundaries of 4-6 us each and on a fold in which 25 856
// threads walk their members serially (540 us per iteration, ~7 % of the bytes' roofline).  Here the tree is cut
// below its leading chance levels like k_cfr_split's and the pieces are dealt to the workgroups as BINS (a subtree
// each, or — with more subtrees than compute units — whole subtrees / the pieces one level deeper packed to one bin
// per workgroup: SubTree's second half).  A workgroup of 1024 threads sweeps its bin bottom-up with the updating
// player's values, the bin's policy rows and chance probabilities in LDS and workgroup barriers only, then writes its
// members' regret / average-policy terms (root-path products as in k_gcfr_members) as 64-byte records in 16-byte
// written-through pieces.  Two grid barriers per player pass: terms -> fold -> next pass; the barrier is two-level
// (group counters, a release word) and every thread spends its wait on fetches of data no workgroup writes (the
// fold's first schedule; the coming pass's terminal values and row indices).  The fold: a workgroup takes a run of the
// updating player's infostates balanced by member count, all its threads fetch the members' records into LDS
// (consecutive threads, consecutive records), ONE THREAD per infostate adds them in member (DFS) order — the same
// additions in the same order as every other CFR kernel here, so the tables stay bit-identical — clamps, matches and
// writes the row through.  Everything that crosses workgroups (records, root values, the three tables) moves with
// written-through stores and cache-bypassing loads: no cache-wide fences.  Alternating updates only (one value per
// history in LDS); the launch is cooperative, so the grid IS co-resident.  DESIGN.md section 6 has what each device was
// worth (3 260 -> 8 260 iterations/s on 3-player leduc in round 5).
// Reference: cfr.cc:331-408 (ComputeCounterFactualRegret), 443-469, 596-615.
// ---------------------------------------------------------------------------
struct SubTree {
  int G, L, NL;                  // subtrees, cut level, padded histories per subtree
  const int32_t* nloc;           // [G] histories of the subtree
  const int32_t* desc;           // [G, NL] kind | nchild << 2 | level << 10 | (actor + 1) << 16
  const int32_t* fc;             // [G, NL] LOCAL index of the first child
  const int32_t* aux;            // [G, NL] decision: its index d among the subtree's decision histories; chance /
                                 //          terminal: the history's global index
  int ND;                        // padded decision histories per subtree
  const int32_t* ndec;           // [G]
  const int32_t* dec_row;        // [G, ND] info * A of decision history d
  const int32_t* mem_off;        // [G * P + 1] the subtree's members of player q: sub_mem[mem_off[g * P + q] ...)
  const int32_t* sub_rec;        // [., 8 + PL / 2 rounded up to 4] (the codes are 16-bit halves, 0xFFFF padded) per member: m (position in Tree::mem), its history's local index, its decision
                                 //   index | actions << 24, its first child's local index; the product of the chance
                                 //   probabilities on its root path (a double, path order), two unused words; then the decision
                                 //   entries of the path GROUPED BY PLAYER, PL / P codes per player in path order, -1 padded:
                                 //   (the ancestor's decision index in this subtree) * A + action index, i.e. an index into the
                                 //   policy rows the sweep has staged in LDS
  int PL;                        // codes per member: P groups of a multiple of 4, at most 4 kSubCodeChunks
  const int32_t* info_off;       // [P + 1] infostates of player q: info_list[info_off[q] ...)
  const int32_t* info_list;
  double* recbuf;                // [M, 8] the members' 64-byte records (kSubRecDoubles)
  int tree_barrier;              // 1: two-level arrival + release word; 0: round 4's flat counter
  unsigned int* bar;             // [0] arrival counter, [1] error flag, [2] release word, [16 + 16 g] group counters (zeroed per launch)
  unsigned int* host_err;        // pinned host word raised on a timeout: the next call reads it without a copy
  unsigned long long timeout_ticks;
  unsigned long long* stamps;    // null, or [P][5] wall-clock stamps of workgroup 0 in the launch's last iteration
  int stamp_wg = 0;              // the workgroup that writes the stamps (OSG_CFR_SUB_STAMPS = its index + 1)
  // Forest form (round 5): when there are more deal subtrees than resident workgroups (3-player leduc: 336 on 256), the
  // tree is cut ONE LEVEL DEEPER into pieces (the children of the deal roots) and the pieces are packed into one bin
  // per workgroup, balanced by size: "subtree g" above is then a forest of pieces in level order, every workgroup
  // sweeps ONE forest per pass and none takes two while the others wait.  The deal roots ("upper" histories) belong to
  // no forest: their policy rows ride in the forests' LDS rows (path codes), the pieces' root values leave through
  // root_value, and an upper member's terms are formed by the fold from those values (skip word = 2 + its index).
  const int32_t* nroot = nullptr;     // [G] piece roots of the forest; null: the bins are whole subtrees
  const int32_t* root_loc = nullptr;  // [G, NR] local index of piece root r
  const int32_t* root_idx = nullptr;  // [G, NR] its slot in root_value
  int NR = 0;
  double* root_value = nullptr;       // [histories of the pieces' level] the updating player's value of every piece root
  const int32_t* upper_rec = nullptr; // [U, 8] first child's slot in root_value, info * A, actions, 0, chance product (lo, hi), 0, 0
  // Round 5: what a pass does not have to fetch again.  The decision rows of a bin are ordered by acting player
  // (dec_off), so with one bin per workgroup (keep_rows) the policy rows STAY in LDS between passes and a pass fetches
  // only the rows the previous pass's fold rewrote (the previous updating player's) and the upper parents' rows; the
  // outcome probabilities of the bin's chance histories sit in LDS too (chance_prob: no trip to memory inside the level
  // loop); the fold takes its infostates from a packed descriptor (fold_info) in shares balanced by members (fold_off).
  const int32_t* dec_off = nullptr;   // [G, P + 2] rows of player q: [dec_off[q], dec_off[q + 1]); upper parents' rows from dec_off[P] to dec_off[P + 1]
  const double* chance_prob = nullptr;// [G, NCP] outcome probabilities of the bin's chance histories (aux = offset of the first)
  int NCP = 0;                        // (even)
  int keep_rows = 0;
  int lds_doubles = 0;                // dynamic LDS of the launch, in doubles: [policy rows ND * A | chance NCP | values NL | spare]
  const int32_t* fold_info = nullptr; // [infostates in info_list order, 4] infostate, actions, first member, members
  const double* term_val = nullptr;   // [G, P, NL] player q's return at every terminal history of the bin, local order (0 elsewhere)
  int prefetch = 1;                   // 0: nothing is fetched in the barriers' windows (measurement)
  const int32_t* fold_off = nullptr;  // [P, grid + 1] the share of workgroup w in pass q: entries [fold_off[q][w], fold_off[q][w + 1])
};
OSG_D double readlane_f64(double v, int lane) {   // lane is wave-uniform: two v_readlane_b32, no LDS permute
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane(static_cast<int>(b), lane), hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), lane);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
OSG_D void store_through_i32(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
OSG_D int32_t load_through_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte written-through stores / bypassing loads (buffer instructions with the sc1 bit: what the 8-byte agent-scope
// atomics above compile to, four words at a time; the compiler keeps the wait counters)
typedef unsigned int osg_u4 __attribute__((ext_vector_type(4)));
typedef double osg_d2 __attribute__((ext_vector_type(2)));
constexpr int kCachePolicySc1 = 16;
OSG_D __amdgpu_buffer_rsrc_t through_buffer(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
OSG_D void store_through16(__amdgpu_buffer_rsrc_t r, unsigned int byte_off, osg_d2 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(osg_u4, v), r, static_cast<int>(byte_off), 0, kCachePolicySc1);
}
OSG_D osg_u4 load_through16(__amdgpu_buffer_rsrc_t r, unsigned int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(byte_off), 0, kCachePolicySc1);
}
// The member records of k_cfr_sub (round 5): 64 bytes per member — regret terms [4], average-policy terms [4] — written
// by the member's thread as whole 16-byte pieces and fetched by the fold the same way (7 eight-byte stores and loads per
// member before).  A record that carries no terms says so in its first word: a quiet NaN whose low word is 1 (the member
// was pruned) or 2 + u (upper member u of the forest form: written once by the host, its terms are formed in the fold).
constexpr unsigned int kSubFlagHi = 0x7FF80000u;
constexpr int kSubRecDoubles = 8;
constexpr int kSubBarWords = 16 + 16 * 64;   // grid barrier words: a cooperative grid of up to 1 024 workgroups

constexpr int kSubThreads = 1024;
constexpr int kSubKD = 4;            // decision histories per thread: ND <= 4096
constexpr int kSubFoldInfos = 64;    // infostates a workgroup folds per round (one wavefront adds them up)
constexpr int kSubFoldX = 2;         // member records a thread fetches per round: <= 2048 per round
constexpr int kSubCodeChunks = 8;    // int4 chunks of path codes a member record holds at most (requested together)
template <int kK>   // histories per thread: NL <= kK * 1024
__global__ void __launch_bounds__(kSubThreads)
k_cfr_sub(Tree t, SmallTree st, SubTree sp, Tables tb, int iters, int iteration0, osg_cfr_cfg cfg) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double* s_pol = s_dyn;                                             // [ND, A] the current policy of the bin's rows (ND * A even)
  double* s_cp = s_pol + sp.ND * t.A;                                // [NCP] outcome probabilities of the bin's chance histories
  double* s_value = s_cp + sp.NCP;                                   // [NL] the updating player's values
  __shared__ int s_ok;
  __shared__ int s_lvl[2 * kK];
  __shared__ int s_fi[kSubFoldInfos], s_fn[kSubFoldInfos], s_fm0[kSubFoldInfos], s_fbase[kSubFoldInfos + 1], s_fne;
  const int P = t.P, A = t.A, tid = threadIdx.x;
  unsigned int epoch = 0;
  // one polling lane per workgroup; false = a workgroup never arrived (cannot happen in a cooperative launch short of
  // a hung device: the bound only keeps a broken device from spinning for ever)
  // Two-level arrival (round 5): a workgroup adds to its group's counter (16 workgroups per group, a cache line each),
  // the last of a group adds to the top counter, the last of all writes the epoch into the release word, and everybody
  // polls that word — which is written once per barrier instead of taking 256 same-address adds under 256 pollers
  // (sp.tree_barrier == 0: the flat counter of round 4).  bar: [0] top / flat counter, [1] error, [2] release word,
  // [16 + 16 g] group g.
  // `window`: work on data no other workgroup writes (the tree, the host's schedules), run by every thread between
  // this workgroup's arrival and its wait — the trips to memory the next phase would start with happen while the
  // slower workgroups are still on their way.
  auto grid_barrier = [&](auto&& window) -> bool {
    ++epoch;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (sp.tree_barrier) {
        const unsigned int grp = blockIdx.x >> 4, ngrp = (gridDim.x + 15u) >> 4;
        const unsigned int gsize = gridDim.x - (grp << 4) < 16u ? gridDim.x - (grp << 4) : 16u;
        if (__hip_atomic_fetch_add(&sp.bar[16 + 16 * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == epoch * gsize) {
          if (__hip_atomic_fetch_add(&sp.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == epoch * ngrp)
            __hip_atomic_store(&sp.bar[2], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        __hip_atomic_fetch_add(&sp.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    window();
    if (tid == 0) {
      const unsigned long long t0 = wall_clock64();
      int ok = 1;
      if (sp.tree_barrier) {
        unsigned int seen;
        while ((seen = __hip_atomic_load(&sp.bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < epoch) {
          if (wall_clock64() - t0 > sp.timeout_ticks) { ok = 0; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (seen == 0xFFFFFFFFu) ok = 0;   // another workgroup gave up
        if (!ok) __hip_atomic_store(&sp.bar[2], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const unsigned int want = epoch * gridDim.x;
        while (__hip_atomic_load(&sp.bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          if (__hip_atomic_load(&sp.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
              wall_clock64() - t0 > sp.timeout_ticks) { ok = 0; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (!ok) {
        __hip_atomic_store(&sp.bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sp.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
  };
  const __amdgpu_buffer_rsrc_t rec_buf = through_buffer(sp.recbuf);
  bool prefetched = false;   // the coming pass's terminal values are in LDS and its rows' indices in rows_pref
  int rows_pref[2] = {-1, -1};
  for (int it = 0; it < iters; ++it) {
    const int iteration = iteration0 + it + 1;
    for (int upd = 0; upd < P; ++upd) {
      const bool stamp = sp.stamps && it == iters - 1 && static_cast<int>(blockIdx.x) == sp.stamp_wg && tid == 0;
      if (stamp) sp.stamps[upd * 5 + 0] = wall_clock64();
      for (int g = blockIdx.x; g < sp.G; g += gridDim.x) {
        // ---- the thread's histories of this subtree: descriptors in registers for the sweep ----
        const int nloc = sp.nloc[g];
        int o_d[kK], o_fc[kK], o_aux[kK];
#pragma unroll
        for (int k = 0; k < kK; ++k) {
          const int j = tid + k * kSubThreads;
          o_d[k] = kTerminalNode | (63 << 10);   // padding: a terminal of a level never swept
          o_fc[k] = 0; o_aux[k] = 0;
          if (j < nloc) {
            o_d[k] = sp.desc[static_cast<size_t>(g) * sp.NL + j];
            o_fc[k] = sp.fc[static_cast<size_t>(g) * sp.NL + j];
            o_aux[k] = sp.aux[static_cast<size_t>(g) * sp.NL + j];
          }
        }
        if (stamp && g == static_cast<int>(blockIdx.x)) sp.stamps[P * 5 + upd * 2 + 1] = wall_clock64();
        // the level range of every slot (its first and its last valid history), for the sweep's (slot, level) walk
#pragma unroll
        for (int k = 0; k < kK; ++k) {
          const int j = tid + k * kSubThreads;
          if (tid == 0 && j < nloc) s_lvl[2 * k] = (o_d[k] >> 10) & 0x3F;
          if (j < nloc && (tid == kSubThreads - 1 || j == nloc - 1)) s_lvl[2 * k + 1] = (o_d[k] >> 10) & 0x3F;
        }
        // ---- A: everything the sweep reads from memory is requested at once — the terminal values and the policy
        //      rows of the subtree's decision histories, into LDS: the levels then cost an LDS round trip and a
        //      workgroup barrier each, not a trip to the L2 (1.6 us per level before: 23 us per sweep) ----
        if (!prefetched) {   // (else: fetched in the window of the previous pass's last barrier)
          const double* tv = sp.term_val + (static_cast<size_t>(g) * P + upd) * sp.NL;
#pragma unroll
          for (int k = 0; k < kK; ++k) {
            const int j = tid + k * kSubThreads;
            if (j < nloc) s_value[j] = tv[j];   // (the terminals' returns; a history that is swept gets its value then)
          }
        }
        const int ndec = sp.ndec[g];
        const bool all_rows = !sp.keep_rows || (it == 0 && upd == 0);
        if (all_rows) {
          int rows[kSubKD];   // the thread's decision histories: all their rows are requested before the first arrives
#pragma unroll
          for (int k = 0; k < kSubKD; ++k) {
            const int d = tid + k * kSubThreads;
            rows[k] = d < ndec ? sp.dec_row[static_cast<size_t>(g) * sp.ND + d] : -1;
          }
#pragma unroll
          for (int k = 0; k < kSubKD; ++k) {
#pragma unroll
            for (int a = 0; a < kSplitMaxA; ++a)
              if (rows[k] >= 0 && a < A) s_pol[(tid + k * kSubThreads) * A + a] = load_through(tb.cur + rows[k] + a);
          }
          for (int c = tid; c < sp.NCP; c += kSubThreads) s_cp[c] = sp.chance_prob[static_cast<size_t>(g) * sp.NCP + c];
        } else {
          // the rows are still in LDS: only the previous pass's fold changed any — the rows of the player it updated —
          // and (forest form) the upper parents' rows ride behind
          const int32_t* doff = sp.dec_off + static_cast<size_t>(g) * (P + 2);
          const int prev = (upd + P - 1) % P;
          const int b0 = doff[prev], n0 = doff[prev + 1] - b0, b1 = doff[P], n1 = doff[P + 1] - b1;
          int rows[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {   // (a third of the bin's rows: at most 2 048 here, the rest in the loop below)
            const int x = tid + k * kSubThreads;
            const int d = x < n0 ? b0 + x : (x - n0 < n1 ? b1 + (x - n0) : -1);
            rows[k] = prefetched ? rows_pref[k] : (d >= 0 ? sp.dec_row[static_cast<size_t>(g) * sp.ND + d] : -1);
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int x = tid + k * kSubThreads;
            const int d = x < n0 ? b0 + x : b1 + (x - n0);
#pragma unroll
            for (int a = 0; a < kSplitMaxA; ++a)
              if (rows[k] >= 0 && a < A) s_pol[d * A + a] = load_through(tb.cur + rows[k] + a);
          }
          for (int x = tid + 2 * kSubThreads; x < n0 + n1; x += kSubThreads) {
            const int d = x < n0 ? b0 + x : b1 + (x - n0);
            const int row = sp.dec_row[static_cast<size_t>(g) * sp.ND + d];
            for (int a = 0; a < A; ++a) s_pol[d * A + a] = load_through(tb.cur + row + a);
          }
        }
        __syncthreads();
        if (stamp && g == static_cast<int>(blockIdx.x)) sp.stamps[P * 5 + upd * 2] = wall_clock64();
        // bottom-up (cfr.cc:443-469).  Slot k of the threads covers the local indices [1024 k, 1024 k + 1023], a
        // contiguous run in level order, i.e. a workgroup-uniform range of levels: the sweep walks (slot, level) pairs
        // from the deepest, ONE slot's body per step (testing all slots at every level cost more instructions than
        // the values themselves; re-measured in round 5 with everything in LDS: one barrier per level — 20 steps instead
        // of 27 — with every slot tested inside a step ran 8.4 us per sweep against 7.5: profiles/r05m_*).  A history's children have larger indices: an earlier step has produced them.
#pragma unroll
        for (int k = kK - 1; k >= 0; --k) {
          if (k * kSubThreads >= nloc) continue;                                  // (workgroup-uniform)
          const int l_lo = s_lvl[2 * k], l_hi = s_lvl[2 * k + 1] < t.D - 2 ? s_lvl[2 * k + 1] : t.D - 2;
          const int kind = o_d[k] & 3, mine = (o_d[k] >> 10) & 0x3F, nc = (o_d[k] >> 2) & 0xFF;
          for (int l = l_hi; l >= l_lo; --l) {
            if (mine == l && kind != kTerminalNode) {
              double v = 0.0;
              if (kind == kChanceNode) {
                for (int a = 0; a < nc; ++a) v += s_cp[o_aux[k] + a] * s_value[o_fc[k] + a];
              } else {
                for (int a = 0; a < nc; ++a) v += s_pol[o_aux[k] * A + a] * s_value[o_fc[k] + a];
              }
              s_value[tid + k * kSubThreads] = v;
            }
            __syncthreads();
          }
        }
        if (stamp && g == static_cast<int>(blockIdx.x)) sp.stamps[upd * 5 + 1] = wall_clock64();
        if (sp.nroot) {   // forest form: the values the upper members' terms are formed from (in the fold)
          const int nr = sp.nroot[g];
          for (int r = tid; r < nr; r += kSubThreads)
            store_through(sp.root_value + sp.root_idx[static_cast<size_t>(g) * sp.NR + r],
                          s_value[sp.root_loc[static_cast<size_t>(g) * sp.NR + r]]);
        }
        // ---- B: the updating player's members of this subtree (k_gcfr_members) ----
        // A member's record is one contiguous run of ints (SubTree::sub_rec) in the order the subtree visits its members:
        // ONE round trip brings all of it.  The probabilities on the root path are the chance product (constant: formed
        // once on the host, in path order) and policy entries of the member's ancestors — decision histories of THIS
        // subtree, whose rows the sweep has staged in LDS: no second trip to memory.  The codes come grouped by player, so
        // a player's reach is one running product in path order (what keeps the tables bit-identical) and the
        // counterfactual reach multiplies the players' products in player order, the chance product last (cfr.cc:309-318).
        const int m_begin = sp.mem_off[g * P + upd], m_end = sp.mem_off[g * P + upd + 1];
        const int n_chunks = sp.PL / 4, per_player = n_chunks / P;
        // two members per thread and round, both records requested before the first is used: a bin holds ~1 050 members
        // of a player (3-player leduc), and a second round for the few beyond 1 024 cost a whole round's latency
        for (int mm0 = m_begin + tid; mm0 < m_end; mm0 += 2 * kSubThreads) {
          int4 head[2], second[2], codes[2][kSubCodeChunks / 2];   // (16-bit codes: two chunks of four per int4)
          bool live[2];
          const int n_words4 = (n_chunks + 1) / 2, rec_ints = 8 + 4 * n_words4;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int mm = mm0 + u * kSubThreads;
            live[u] = mm < m_end;
            const int4* rec = reinterpret_cast<const int4*>(sp.sub_rec + static_cast<size_t>(live[u] ? mm : mm0) * rec_ints);
            head[u] = rec[0]; second[u] = rec[1];
#pragma unroll
            for (int c = 0; c < kSubCodeChunks / 2; ++c) codes[u][c] = rec[2 + (c < n_words4 ? c : n_words4 - 1)];
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (!live[u]) continue;
            const int m = head[u].x, hl = head[u].y, d = head[u].z & 0xFFFFFF, n = (head[u].z >> 24) & 0xFF, lfc = head[u].w;
            const double chance = __longlong_as_double((static_cast<long long>(second[u].y) << 32) | static_cast<unsigned int>(second[u].x));
            bool pruned = true;
            double self_reach = 0.0, cf_reach = 1.0, r = 1.0;
            int q = 0;
#pragma unroll
            for (int c = 0; c < kSubCodeChunks; ++c) {
              if (c < n_chunks) {   // (workgroup-uniform)
                const unsigned int w0 = static_cast<unsigned int>((c & 1) ? codes[u][c >> 1].z : codes[u][c >> 1].x),
                                   w1 = static_cast<unsigned int>((c & 1) ? codes[u][c >> 1].w : codes[u][c >> 1].y);
                const unsigned int cx = w0 & 0xFFFFu, cy = w0 >> 16, cz = w1 & 0xFFFFu, cw = w1 >> 16;   // 0xFFFF: padding
                const double px = s_pol[cx == 0xFFFFu ? 0u : cx], py = s_pol[cy == 0xFFFFu ? 0u : cy],
                             pz = s_pol[cz == 0xFFFFu ? 0u : cz], pw = s_pol[cw == 0xFFFFu ? 0u : cw];
                r = r * (cx == 0xFFFFu ? 1.0 : px);   // (x * 1.0 == x: the padding leaves the product as it is)
                r = r * (cy == 0xFFFFu ? 1.0 : py);
                r = r * (cz == 0xFFFFu ? 1.0 : pz);
                r = r * (cw == 0xFFFFu ? 1.0 : pw);
                if ((c + 1) % per_player == 0) {   // the last chunk of player q's group
                  pruned &= (r == 0.0);
                  if (q == upd) self_reach = r; else cf_reach *= r;
                  ++q;
                  r = 1.0;
                }
              }
            }
            cf_reach *= chance;
            const unsigned int at = static_cast<unsigned int>(m) * (kSubRecDoubles * 8);
            if (pruned) {
              osg_d2 flag;
              flag.x = __longlong_as_double((static_cast<long long>(kSubFlagHi) << 32) | 1ll);
              flag.y = 0.0;
              store_through16(rec_buf, at, flag);
              continue;
            }
            const double vh = s_value[hl];
            double dr[kSplitMaxA], dp[kSplitMaxA];
#pragma unroll
            for (int a = 0; a < kSplitMaxA; ++a) {
              dr[a] = 0.0; dp[a] = 0.0;
              if (a < n) {
                dr[a] = cf_reach * (s_value[lfc + a] - vh);
                const double pol = s_pol[d * A + a];
                dp[a] = cfg.linear_averaging ? iteration * self_reach * pol : self_reach * pol;
              }
            }
            static_assert(kSplitMaxA == 4 && kSubRecDoubles == 8, "the record is two pieces of regret terms, two of policy terms");
            store_through16(rec_buf, at, osg_d2{dr[0], dr[1]});
            store_through16(rec_buf, at + 32, osg_d2{dp[0], dp[1]});
            if (A > 2) {   // (workgroup-uniform)
              store_through16(rec_buf, at + 16, osg_d2{dr[2], dr[3]});
              store_through16(rec_buf, at + 48, osg_d2{dp[2], dp[3]});
            }
          }
        }
        __syncthreads();   // (the next subtree of this workgroup reuses s_value)
      }
      if (stamp) sp.stamps[upd * 5 + 2] = wall_clock64();
      int e0 = 0, e_last = 0;
      // the records are staged behind what stays in LDS (keep_rows: the policy rows and chance probabilities)
      double* s_rec = sp.keep_rows ? s_value : s_dyn;
      const int cap_lds = (sp.lds_doubles - static_cast<int>(s_rec - s_dyn)) / kSubRecDoubles;   // 64-byte records
      const int cap = cap_lds < kSubFoldX * kSubThreads ? cap_lds : kSubFoldX * kSubThreads;
      // this round's infostates of the fold: as many as fit the stage, by a prefix sum over wavefront 1 (wavefront 0
      // holds the barrier's polling lane); the schedule is the host's, so round 0's is formed in the barrier's window
      auto fold_schedule = [&]() {
        static_assert(kSubFoldInfos == 64, "one wavefront schedules a round");
        if (tid >= 64 && tid < 128) {
          const int lane = tid - 64, e = e0 + lane;
          int4 fi = make_int4(0, 0, 0, 0);
          if (e < e_last) fi = reinterpret_cast<const int4*>(sp.fold_info)[e];
          const int cnt = e < e_last ? fi.w : 0;
          int inc = cnt;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(inc, off);
            if (lane >= off) inc += v;
          }
          const bool in = e < e_last && (lane == 0 || inc <= cap);
          const int ne = __popcll(__ballot(in));   // (`in` holds on a prefix of the lanes: inc does not decrease)
          s_fi[lane] = fi.x; s_fn[lane] = fi.y; s_fm0[lane] = fi.z;
          s_fbase[lane] = inc - cnt;
          if (lane == ne - 1) s_fbase[ne] = inc;
          if (lane == 0) { s_fne = ne; if (ne == 0) s_fbase[0] = 0; }
        }
      };
      if (!grid_barrier([&]() {
            e0 = sp.fold_off[upd * (static_cast<int>(gridDim.x) + 1) + static_cast<int>(blockIdx.x)];
            e_last = sp.fold_off[upd * (static_cast<int>(gridDim.x) + 1) + static_cast<int>(blockIdx.x) + 1];
            fold_schedule();
          })) return;
      if (stamp) sp.stamps[upd * 5 + 3] = wall_clock64();
      // ---- C: fold (k_gcfr_fold's additions, in its order).  A workgroup takes a contiguous share of the updating
      //      player's infostates; all its threads fetch the members' records together into LDS (the values / policy
      //      region is free now), then ONE thread per infostate adds its members in member (DFS) order — a serial chain
      //      of ~40 additions fed from LDS — clamps (RM+), regret-matches and writes the row through.  (One wavefront
      //      per infostate with the sums formed by lane broadcasts was 19-23 us: ~13 broadcasts per member.) ----
      {
        bool first_round = true;
        while (e0 < e_last) {
          if (!first_round) {
            fold_schedule();
            __syncthreads();
          }
          first_round = false;
          const int ne = s_fne, total = s_fbase[ne];
          // the infostate's own row, requested now, needed after the barrier
          double reg[kSplitMaxA], cum[kSplitMaxA];
#pragma unroll
          for (int a = 0; a < kSplitMaxA; ++a) { reg[a] = 0.0; cum[a] = 0.0; }
          if (tid < ne) {
#pragma unroll
            for (int a = 0; a < kSplitMaxA; ++a)
              if (a < s_fn[tid]) {
                reg[a] = load_through(tb.regrets + static_cast<size_t>(s_fi[tid]) * A + a);
                cum[a] = load_through(tb.cum + static_cast<size_t>(s_fi[tid]) * A + a);
              }
          }
          {
            int xm[kSubFoldX], xn[kSubFoldX];
            osg_u4 pc[kSubFoldX][4];
#pragma unroll
            for (int u = 0; u < kSubFoldX; ++u) {
              const int x = tid + u * kSubThreads;
              xm[u] = -1; xn[u] = 0;
              if (x < total) {
                int lo = 0, hi = ne;                       // the infostate of record x: s_fbase[lo] <= x < s_fbase[lo + 1]
                while (hi - lo > 1) {
                  const int mid = (lo + hi) >> 1;
                  if (s_fbase[mid] <= x) lo = mid; else hi = mid;
                }
                xm[u] = s_fm0[lo] + (x - s_fbase[lo]);
                xn[u] = s_fn[lo];
              }
            }
            // consecutive threads fetch consecutive 64-byte records (an infostate's members are consecutive): whole
            // lines, four (two for two-action games) 16-byte bypassing loads per record; a pruned member's stale terms
            // are fetched all the same and never added
#pragma unroll
            for (int u = 0; u < kSubFoldX; ++u) {
#pragma unroll
              for (int k = 0; k < 4; ++k) pc[u][k] = osg_u4{0u, 0u, 0u, 0u};
              if (xm[u] < 0) continue;
              const unsigned int at = static_cast<unsigned int>(xm[u]) * (kSubRecDoubles * 8);
              pc[u][0] = load_through16(rec_buf, at);
              pc[u][2] = load_through16(rec_buf, at + 32);
              if (A > 2) {   // (workgroup-uniform)
                pc[u][1] = load_through16(rec_buf, at + 16);
                pc[u][3] = load_through16(rec_buf, at + 48);
              }
            }
#pragma unroll
            for (int u = 0; u < kSubFoldX; ++u) {
              if (xm[u] < 0) continue;
              osg_u4* r4 = reinterpret_cast<osg_u4*>(s_rec + static_cast<size_t>(tid + u * kSubThreads) * kSubRecDoubles);
#pragma unroll
              for (int k = 0; k < 4; ++k) r4[k] = pc[u][k];
            }
            // An upper member (forest form: a deal root; its record carries 2 + its index, written once by the host): its
            // terms are formed here, by k_gcfr_members' expressions — the value of the history is the policy-weighted sum
            // of its children's values in action order (the sweep's), every player's reach on its root path is the
            // empty product 1.0, so the counterfactual reach is 1.0 * ... * chance = chance and the own reach 1.0.
#pragma unroll
            for (int u = 0; u < kSubFoldX; ++u) {
              if (xm[u] < 0 || pc[u][0].y != kSubFlagHi || pc[u][0].x < 2u) continue;
              const int32_t* ur = sp.upper_rec + static_cast<size_t>(pc[u][0].x - 2u) * 8;
              const int slot0 = ur[0], row = ur[1];
              const double chance = __longlong_as_double((static_cast<long long>(ur[5]) << 32) | static_cast<unsigned int>(ur[4]));
              double va[kSplitMaxA], pa[kSplitMaxA];
#pragma unroll
              for (int a = 0; a < kSplitMaxA; ++a) {
                va[a] = 0.0; pa[a] = 0.0;
                if (a < xn[u]) { va[a] = load_through(sp.root_value + slot0 + a); pa[a] = load_through(tb.cur + row + a); }
              }
              double vh = 0.0;
#pragma unroll
              for (int a = 0; a < kSplitMaxA; ++a)
                if (a < xn[u]) vh += pa[a] * va[a];
              const double self_reach = 1.0;
              double cf_reach = 1.0;
              cf_reach *= chance;
              double* r = s_rec + static_cast<size_t>(tid + u * kSubThreads) * kSubRecDoubles;
#pragma unroll
              for (int a = 0; a < kSplitMaxA; ++a) {
                r[a] = 0.0; r[kSplitMaxA + a] = 0.0;
                if (a < xn[u]) {
                  r[a] = cf_reach * (va[a] - vh);
                  r[kSplitMaxA + a] = cfg.linear_averaging ? iteration * self_reach * pa[a] : self_reach * pa[a];
                }
              }
            }
          }
          __syncthreads();
          if (tid < ne) {
            const int n = s_fn[tid];
            // member order; a record is four 16-byte LDS reads, two records in flight (entries beyond the row's actions are
            // zeros in every record and their sums are never written back, so no per-action test)
            const int x_end = s_fbase[tid + 1];
            for (int x = s_fbase[tid]; x < x_end; x += 2) {
              osg_u4 q[2][4];
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const osg_u4* r4 = reinterpret_cast<const osg_u4*>(s_rec + static_cast<size_t>(x + u < x_end ? x + u : x) * kSubRecDoubles);
#pragma unroll
                for (int k = 0; k < 4; ++k) q[u][k] = r4[k];
              }
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                if (x + u >= x_end) continue;
                if (q[u][0].y == kSubFlagHi && q[u][0].x == 1u) continue;   // pruned
                const osg_d2 r01 = __builtin_bit_cast(osg_d2, q[u][0]), r23 = __builtin_bit_cast(osg_d2, q[u][1]);
                const osg_d2 c01 = __builtin_bit_cast(osg_d2, q[u][2]), c23 = __builtin_bit_cast(osg_d2, q[u][3]);
                reg[0] += r01.x; reg[1] += r01.y; reg[2] += r23.x; reg[3] += r23.y;
                cum[0] += c01.x; cum[1] += c01.y; cum[2] += c23.x; cum[3] += c23.y;
              }
            }
            double sum_pos = 0.0;
#pragma unroll
            for (int a = 0; a < kSplitMaxA; ++a) {
              if (cfg.regret_matching_plus && reg[a] < 0) reg[a] = 0;
              if (a < n && reg[a] > 0) sum_pos += reg[a];
            }
#pragma unroll
            for (int a = 0; a < kSplitMaxA; ++a) {
              if (a < n) {
                const double pol = sum_pos > 0 ? (reg[a] > 0 ? reg[a] / sum_pos : 0.0) : 1.0 / n;
                store_through(tb.regrets + static_cast<size_t>(s_fi[tid]) * A + a, reg[a]);
                store_through(tb.cum + static_cast<size_t>(s_fi[tid]) * A + a, cum[a]);
                store_through(tb.cur + static_cast<size_t>(s_fi[tid]) * A + a, pol);
              }
            }
          }
          __syncthreads();   // (the records' LDS is reused by the next round / the next pass's sweep)
          e0 += ne;
        }
      }
      if (stamp) sp.stamps[upd * 5 + 4] = wall_clock64();
      // the coming pass of this workgroup's bin (one bin per workgroup): its terminal values into LDS (the fold's stage
      // is done with) and the indices of the rows it will re-fetch — this pass's updating player's — while waiting
      const bool more = sp.keep_rows && sp.prefetch && !(it == iters - 1 && upd == P - 1);
      if (!grid_barrier([&]() {
            if (!more) return;
            const int g = blockIdx.x, nloc = sp.nloc[g], nxt = upd + 1 < P ? upd + 1 : 0;
            const int32_t* doff = sp.dec_off + static_cast<size_t>(g) * (P + 2);
            const int b0 = doff[upd], n0 = doff[upd + 1] - b0, b1 = doff[P], n1 = doff[P + 1] - b1;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int x = tid + k * kSubThreads;
              const int d = x < n0 ? b0 + x : (x - n0 < n1 ? b1 + (x - n0) : -1);
              rows_pref[k] = d >= 0 ? sp.dec_row[static_cast<size_t>(g) * sp.ND + d] : -1;
            }
            const double* tv = sp.term_val + (static_cast<size_t>(g) * P + nxt) * sp.NL;
#pragma unroll
            for (int k = 0; k < kK; ++k) {
              const int j = tid + k * kSubThreads;
              if (j < nloc) s_value[j] = tv[j];
            }
          })) return;
      prefetched = more;
    }
  }
}

// ---------------------------------------------------------------------------
// Policy evaluation on the flattened tree: ExpectedReturns
// (expected_returns.cc:34-130), TabularBestResponse (best_response.cc:194-227)
// for every player, from which the host derives NashConv / Exploitability
// (tabular_exploitability.cc:30-89).  One workgroup, level-synchronous:
//   values    bottom-up with the evaluated policy -> ev[P]
//   per responder r:
//     cf[m]   counterfactual reach of every decision history of r: product of the
//             chance / opponent-policy probabilities on its root path (root-to-leaf)
//     levels bottom-up; at a level first every infostate of r whose members sit on
//     that level picks argmax_a sum_m cf[m] * brv[child(m, a)] (members in DFS
//     order, strict >: ties go to the lowest action), then the level's nodes get
//     their value (responder nodes: the chosen child's value).
// ---------------------------------------------------------------------------
struct EvalArrays {
  const int32_t* path_off;   // [M+1]
  const int32_t* path;
  const int32_t* info_level; // [I] tree level of the infostate's member histories
  const int32_t* mem_index;  // [H] member position m of a decision history, else -1
  int M;
  double* value;             // [H, P] scratch
  double* brv;               // [H] scratch
  double* cf;                // [M] scratch
  int32_t* best;             // [I] scratch: chosen action index
  double* out;               // [2P]: ev[P] then br[P]
  double* keep = nullptr;    // [H] or null: the best-response value of EVERY history for responder keep_r
  int keep_r = -1;           //          (TabularBestResponse::Value(history), best_response.h:127-128)
};

__global__ void __launch_bounds__(1024)
k_policy_eval(Tree t, EvalArrays ea, const double* pol, int from_cum = 0, double* pol_buf = nullptr) {
  const int P = t.P, A = t.A;
  const int tid = threadIdx.x, nt = blockDim.x;
  if (from_cum) {  // `pol` is the cumulative table: evaluate its normalisation (CFRAveragePolicy, cfr.cc:104-125)
    for (int i = tid; i < t.I; i += nt) {
      const int n = t.nact[i];
      double sum = 0.0;
      for (int a = 0; a < n; ++a) sum += pol[i * A + a];
      for (int a = 0; a < A; ++a) pol_buf[i * A + a] = a >= n ? 0.0 : (sum == 0.0 ? 1. / n : pol[i * A + a] / sum);
    }
    __syncthreads();
    pol = pol_buf;
  }
  // ---- expected returns ----
  for (int l = t.D - 1; l >= 0; --l) {
    for (int h = t.level_off[l] + tid; h < t.level_off[l + 1]; h += nt) {
      const int k = t.kind[h];
      if (k == kTerminalNode) {
        for (int q = 0; q < P; ++q) ea.value[h * P + q] = t.term_ret[h * P + q];
        continue;
      }
      const int fc = t.first_child[h], nc = t.nchild[h];
      const int row = k == kDecisionNode ? t.info[h] * A : 0;
      for (int q = 0; q < P; ++q) {
        double v = 0.0;
        for (int a = 0; a < nc; ++a) {
          const double pr = k == kChanceNode ? t.edge_prob[fc + a] : pol[row + a];
          if (pr > 0.0) v += pr * ea.value[(fc + a) * P + q];
        }
        ea.value[h * P + q] = v;
      }
    }
    __syncthreads();
  }
  if (tid < P) ea.out[tid] = ea.value[tid];
  // ---- best response of every player ----
  for (int r = 0; r < P; ++r) {
    for (int m = tid; m < ea.M; m += nt) {
      const int h = t.mem[m];
      if (t.actor[h] != r) continue;
      double cf = 1.0;
      for (int e = ea.path_off[m]; e < ea.path_off[m + 1]; ++e) {
        const int code = ea.path[e];
        const int slot = (code >> 24) & 0xF, idx = code & 0x7FFFFF;
        const double pr = ((code >> 23) & 1) ? t.edge_prob[idx] : (slot == r ? 1.0 : pol[idx]);
        cf = cf * pr;
      }
      ea.cf[m] = cf;
    }
    __syncthreads();
    for (int l = t.D - 1; l >= 0; --l) {
      for (int i = tid; i < t.I; i += nt) {
        if (t.info_player[i] != r || ea.info_level[i] != l) continue;
        const int n = t.nact[i];
        int best = -1;
        double best_v = -1.7976931348623157e308;  // numeric_limits<double>::lowest()
        for (int a = 0; a < n; ++a) {
          double v = 0.0;
          for (int m = t.mem_off[i]; m < t.mem_off[i + 1]; ++m)
            v += ea.cf[m] * ea.brv[t.first_child[t.mem[m]] + a];
          if (v > best_v) { best_v = v; best = a; }
        }
        ea.best[i] = best < 0 ? 0 : best;
      }
      __syncthreads();
      for (int h = t.level_off[l] + tid; h < t.level_off[l + 1]; h += nt) {
        const int k = t.kind[h];
        double v = 0.0;
        if (k == kTerminalNode) {
          v = t.term_ret[h * P + r];
        } else {
          const int fc = t.first_child[h], nc = t.nchild[h];
          if (k == kDecisionNode && t.actor[h] == r) {
            v += 1.0 * ea.brv[fc + ea.best[t.info[h]]];
          } else {
            const int row = k == kDecisionNode ? t.info[h] * A : 0;
            for (int a = 0; a < nc; ++a) {
              const double pr = k == kChanceNode ? t.edge_prob[fc + a] : pol[row + a];
              v += pr * ea.brv[fc + a];
            }
          }
        }
        ea.brv[h] = v;
      }
      __syncthreads();
    }
    if (tid == 0) ea.out[P + r] = ea.brv[0];
    if (ea.keep && r == ea.keep_r)
      for (int h = tid; h < t.H; h += nt) ea.keep[h] = ea.brv[h];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// The same evaluation for LARGE trees (3-player leduc: 1.83 M histories — one workgroup walks them in 33 ms): one
// full-grid launch per tree level and phase, the stream order is the barrier (the form of k_gcfr_*).  The sums are
// k_policy_eval's, node by node and infostate by infostate, in the same order: bit-identical results.
//   k_geval_policy   the evaluated policy from the cumulative table (mode 0)
//   k_geval_cf       counterfactual reaches of every player's decision histories (each against the others' policy)
//   k_geval_best     the argmax of the infostates whose members sit on level l
//   k_geval_brv      every responder's values of one level; the root's values into out[P ...]
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_geval_policy(Tree t, const double* __restrict__ cum, double* __restrict__ pol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.I) return;
  const int n = t.nact[i], A = t.A;
  double sum = 0.0;
  for (int a = 0; a < n; ++a) sum += cum[i * A + a];
  for (int a = 0; a < A; ++a) pol[i * A + a] = a >= n ? 0.0 : (sum == 0.0 ? 1. / n : cum[i * A + a] / sum);
}
OSG_D double readlane_f64(double v, int lane);   // (defined with k_cfr_sub's helpers)
// (round 5) the best responses of ALL players in one set of launches: a member belongs to one player and an infostate to
// one player, so cf [M] and best [I] hold every responder's entries at once, and the responder values are one [H, P] array
// (the expected-value array, free once its sweep has left the root's values in out[0 .. P)) — a third of the launches of a
// loop over responders, a level without infostates has no argmax launch at all, and the expected returns ride in the
// same sweep (3-player leduc: 172 -> 39 launches).
// Every (history, responder) and every infostate takes the same sums in the same order as before.
__global__ void __launch_bounds__(256) k_geval_cf(Tree t, EvalArrays ea, const double* __restrict__ pol) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= ea.M) return;
  const int h = t.mem[m];
  const int r = t.actor[h];
  double cf = 1.0;
  for (int e = ea.path_off[m]; e < ea.path_off[m + 1]; ++e) {
    const int code = ea.path[e];
    const int slot = (code >> 24) & 0xF, idx = code & 0x7FFFFF;
    const double pr = ((code >> 23) & 1) ? t.edge_prob[idx] : (slot == r ? 1.0 : pol[idx]);
    cf = cf * pr;
  }
  ea.cf[m] = cf;
}
// One WAVEFRONT per infostate of the level (the host's per-level list): the lanes fetch the members' counterfactual
// reaches and child values together (a thread per infostate walked its ~40 members x actions one dependent miss after the
// other: ~60 us per launch, most of an evaluation), form the products, and the sums are added IN MEMBER ORDER from the
// lanes' registers (readlane with a uniform index) — the additions of best_response.cc:194-227 in its order, bit for bit.
__global__ void __launch_bounds__(256) k_geval_best(Tree t, EvalArrays ea, const int32_t* __restrict__ infos, int n_infos) {
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (w >= n_infos) return;   // (wave-uniform)
  const int i = infos[w];
  const int P = t.P, r = t.info_player[i], n = t.nact[i];
  const int m0 = t.mem_off[i], cnt = t.mem_off[i + 1] - m0;
  int best = -1;
  double best_v = -1.7976931348623157e308;  // numeric_limits<double>::lowest()
  for (int a = 0; a < n; ++a) {
    double v = 0.0;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
      const int here = cnt - c0 < 64 ? cnt - c0 : 64;
      double prod = 0.0;
      if (lane < here) {
        const int m = m0 + c0 + lane;
        prod = ea.cf[m] * ea.value[static_cast<size_t>(t.first_child[t.mem[m]] + a) * P + r];
      }
      for (int j = 0; j < here; ++j) v += readlane_f64(prod, j);
    }
    if (v > best_v) { best_v = v; best = a; }
  }
  if (lane == 0) ea.best[i] = best < 0 ? 0 : best;
}
__global__ void __launch_bounds__(256) k_geval_brv(Tree t, EvalArrays ea, const double* __restrict__ pol, int l, double* __restrict__ ev) {
  const int P = t.P, A = t.A;
  const int h = t.level_off[l] + blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= t.level_off[l + 1]) return;
  const int k = t.kind[h];
  const int fc = k == kTerminalNode ? 0 : t.first_child[h], nc = k == kTerminalNode ? 0 : t.nchild[h];
  const int actor = k == kDecisionNode ? t.actor[h] : -1;
  const int row = k == kDecisionNode ? t.info[h] * A : 0;
  if (ev) {   // the expected returns of the same level in the same launch (k_policy_eval's sums; their own [H, P] array)
    for (int q = 0; q < P; ++q) {
      double v = 0.0;
      if (k == kTerminalNode) {
        v = t.term_ret[h * P + q];
      } else {
        for (int a = 0; a < nc; ++a) {
          const double pr = k == kChanceNode ? t.edge_prob[fc + a] : pol[row + a];
          if (pr > 0.0) v += pr * ev[static_cast<size_t>(fc + a) * P + q];
        }
      }
      ev[static_cast<size_t>(h) * P + q] = v;
      if (h == 0) ea.out[q] = v;
    }
  }
  for (int r = 0; r < P; ++r) {
    double v = 0.0;
    if (k == kTerminalNode) {
      v = t.term_ret[h * P + r];
    } else if (actor == r) {
      v += 1.0 * ea.value[static_cast<size_t>(fc + ea.best[t.info[h]]) * P + r];
    } else {
      for (int a = 0; a < nc; ++a) {
        const double pr = k == kChanceNode ? t.edge_prob[fc + a] : pol[row + a];
        v += pr * ea.value[static_cast<size_t>(fc + a) * P + r];
      }
    }
    ea.value[static_cast<size_t>(h) * P + r] = v;
    if (ea.keep && r == ea.keep_r) ea.keep[h] = v;
    if (h == 0) ea.out[P + r] = v;
  }
}

// ---------------------------------------------------------------------------
// The same evaluation for trees that start with their chance deals (leduc_poker: two deal levels, then 30 subtrees
// of 314 histories), spread over the device instead of one workgroup walking 9 457 histories level by level through
// L2.  The quantities are independent below the cut once the work is grouped the right way:
//   * expected returns: every deal subtree on its own (one job per subtree);
//   * the best response of player r: an infostate of r ties together the subtrees that hold its member histories
//     (the deals r cannot tell apart), so the subtrees are grouped into the connected components of that relation
//     (leduc: the 5 deals that share r's private card; 6 components per responder) and one job takes a whole component:
//     the argmax of best_response.cc:194-227 then needs nothing from outside the job.
// A job = one workgroup with its histories, values, counterfactual reaches and the evaluated policy in LDS: a tree
// level costs an LDS round trip.  Every job leaves the values of its subtree roots in memory (written through) and
// takes a ticket; the workgroup that draws the last ticket adds up the chance levels above the cut for all 2 P
// quantities in the recursion's order.  No workgroup waits for another, so the launch is an ordinary one.
// The sums are the ones k_policy_eval forms, in the same order: the results are bit-identical.
// mode 0: `src` is the cumulative-policy table and the evaluated policy is its normalisation (CFRAveragePolicy,
// cfr.cc:104-125); mode 1: `src` is the policy itself.  only_br: the expected-returns jobs do nothing (CFR-BR).
// ---------------------------------------------------------------------------
struct EvalJobs {
  int J, L, G, NT;            // jobs, cut level, subtrees (= histories on level L), histories on levels 0..L
  const int32_t* job;         // [J, 8] kind (0 expected returns, 1 + r best response of r), node_off, nodes, info_off,
                              //        infos, mem_off, members, -
  const int32_t* level_off;   // [J, D + 1] the job's histories of a level: a range of job-local indices
  const int32_t* node_desc;   // per job history: kind | nchild << 2 | (actor + 1) << 10
  const int32_t* node_fc;     //   job-local index of its first child
  const int32_t* node_row;    //   info * A of a decision node
  const int32_t* node_glob;   //   its index in the whole tree
  const int32_t* info_ent;    // per job infostate [4]: id, level, offset of its first member in the job's member list, members
  const int32_t* mem_ent;     // per job member [2]: member index m (position in Tree::mem), job-local history
  double* deal;               // expected returns [G, P], then best-response values [P, G]
  unsigned int* ticket;       // zero between launches
};

__global__ void __launch_bounds__(1024)
k_eval_jobs(Tree t, EvalArrays ea, EvalJobs ej, const double* __restrict__ src, int mode, int only_br) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int s_last;
  const int P = t.P, A = t.A, IA = t.I * t.A, D = t.D;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int32_t* jd = ej.job + blockIdx.x * 8;
  const int kind = jd[0], n0 = jd[1], nn = jd[2], i0 = jd[3], ni = jd[4], m0 = jd[5], nm = jd[6];
  const int r = kind - 1, VP = kind == 0 ? P : 1;
  const int32_t* lvl = ej.level_off + blockIdx.x * (D + 1);
  if (!(only_br && kind == 0)) {
    double* pol = smem;                          // [I, A] the evaluated policy
    double* val = pol + IA;                      // [nodes, P] expected returns, or [nodes] the responder's value
    double* l_edge = val + static_cast<size_t>(nn) * VP;  // [nodes] chance probability of the incoming edge
    double* cf = l_edge + nn;                    // [members] counterfactual reach
    int32_t* l_desc = reinterpret_cast<int32_t*>(cf + nm);
    int32_t* l_fc = l_desc + nn;
    int32_t* l_row = l_fc + nn;
    int32_t* l_mfc = l_row + nn;                 // [members] first child of the member history
    int32_t* l_best = l_mfc + nm;                // [I] chosen action index of the job's infostates
    for (int i = tid; i < t.I; i += nt) {
      const int n = t.nact[i];
      if (mode == 0) {  // CFRAveragePolicy::GetStatePolicyFromInformationStateValues (cfr.cc:104-125)
        double sum = 0.0;
        for (int a = 0; a < n; ++a) sum += src[i * A + a];
        for (int a = 0; a < A; ++a) pol[i * A + a] = a >= n ? 0.0 : (sum == 0.0 ? 1. / n : src[i * A + a] / sum);
      } else {
        for (int a = 0; a < A; ++a) pol[i * A + a] = src[i * A + a];
      }
    }
    for (int x = tid; x < nn; x += nt) {
      const int d = ej.node_desc[n0 + x], hg = ej.node_glob[n0 + x];
      l_desc[x] = d;
      l_fc[x] = ej.node_fc[n0 + x];
      l_row[x] = ej.node_row[n0 + x];
      l_edge[x] = t.edge_prob[hg];
      if ((d & 3) == kTerminalNode) {
        if (kind == 0) for (int q = 0; q < P; ++q) val[x * P + q] = t.term_ret[hg * P + q];
        else val[x] = t.term_ret[hg * P + r];
      }
    }
    if (kind != 0) {
      for (int k = tid; k < nm; k += nt) {
        l_mfc[k] = ej.node_fc[n0 + ej.mem_ent[(m0 + k) * 2 + 1]];
      }
    }
    __syncthreads();
    if (kind == 0) {
      // ---- expected returns (expected_returns.cc:34-130) ----
      for (int l = D - 2; l >= ej.L; --l) {
        for (int x = lvl[l] + tid; x < lvl[l + 1]; x += nt) {
          const int d = l_desc[x], k = d & 3;
          if (k == kTerminalNode) continue;
          const int fc = l_fc[x], nc = (d >> 2) & 0xFF, row = l_row[x];
          for (int q = 0; q < P; ++q) {
            double v = 0.0;
            for (int a = 0; a < nc; ++a) {
              const double pr = k == kChanceNode ? l_edge[fc + a] : pol[row + a];
              if (pr > 0.0) v += pr * val[(fc + a) * P + q];
            }
            val[x * P + q] = v;
          }
        }
        __syncthreads();
      }
      for (int x = lvl[ej.L] + tid; x < lvl[ej.L + 1]; x += nt) {
        const int gidx = ej.node_glob[n0 + x] - t.level_off[ej.L];
        for (int q = 0; q < P; ++q) store_through(ej.deal + static_cast<size_t>(gidx) * P + q, val[x * P + q]);
      }
    } else {
      // ---- the best response of player r (best_response.cc:194-262) ----
      for (int k = tid; k < nm; k += nt) {
        const int m = ej.mem_ent[(m0 + k) * 2];
        double c = 1.0;
        for (int e = ea.path_off[m]; e < ea.path_off[m + 1]; ++e) {
          const int code = ea.path[e];
          const int slot = (code >> 24) & 0xF, idx = code & 0x7FFFFF;
          const double pr = ((code >> 23) & 1) ? t.edge_prob[idx] : (slot == r ? 1.0 : pol[idx]);
          c = c * pr;
        }
        cf[k] = c;
      }
      __syncthreads();
      for (int l = D - 1; l >= ej.L; --l) {
        for (int e = tid; e < ni; e += nt) {
          const int32_t* ie = ej.info_ent + (i0 + e) * 4;
          if (ie[1] != l) continue;
          const int i = ie[0], moff = ie[2], cnt = ie[3], n = t.nact[i];
          int best = -1;
          double best_v = -1.7976931348623157e308;  // numeric_limits<double>::lowest()
          for (int a = 0; a < n; ++a) {
            double v = 0.0;
            for (int k = 0; k < cnt; ++k) v += cf[moff + k] * val[l_mfc[moff + k] + a];
            if (v > best_v) { best_v = v; best = a; }
          }
          best = best < 0 ? 0 : best;
          l_best[i] = best;
          ea.best[i] = best;
        }
        __syncthreads();
        for (int x = lvl[l] + tid; x < lvl[l + 1]; x += nt) {
          const int d = l_desc[x], k = d & 3;
          if (k == kTerminalNode) continue;
          const int fc = l_fc[x], nc = (d >> 2) & 0xFF, row = l_row[x];
          double v = 0.0;
          if (k == kDecisionNode && ((d >> 10) & 15) - 1 == r) {
            v += 1.0 * val[fc + l_best[row / A]];
          } else {
            for (int a = 0; a < nc; ++a) {
              const double pr = k == kChanceNode ? l_edge[fc + a] : pol[row + a];
              v += pr * val[fc + a];
            }
          }
          val[x] = v;
        }
        __syncthreads();
      }
      for (int x = lvl[ej.L] + tid; x < lvl[ej.L + 1]; x += nt) {
        const int gidx = ej.node_glob[n0 + x] - t.level_off[ej.L];
        store_through(ej.deal + static_cast<size_t>(ej.G) * P + static_cast<size_t>(r) * ej.G + gidx, val[x]);
      }
      if (ea.keep && r == ea.keep_r)
        for (int x = tid; x < nn; x += nt) ea.keep[ej.node_glob[n0 + x]] = val[x];
    }
  }
  // ---- the ticket: the last job to finish adds up the chance levels above the cut ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const unsigned int mine = __hip_atomic_fetch_add(ej.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = mine == static_cast<unsigned int>(ej.J) - 1u;
  }
  __syncthreads();
  if (!s_last) return;
  double* top = smem;  // [2 P, NT]: quantity-major, histories of levels 0..L by their index in the whole tree
  const int NT = ej.NT, top0 = t.level_off[ej.L];
  for (int k = tid; k < 2 * P * ej.G; k += nt) {
    int kq, gidx;
    double v;
    if (k < P * ej.G) { gidx = k / P; kq = k % P; v = only_br ? 0.0 : load_through(ej.deal + k); }
    else { kq = P + (k - P * ej.G) / ej.G; gidx = (k - P * ej.G) % ej.G; v = load_through(ej.deal + k); }
    top[kq * NT + top0 + gidx] = v;
  }
  __syncthreads();
  for (int l = ej.L - 1; l >= 0; --l) {
    const int w = t.level_off[l + 1] - t.level_off[l];
    for (int k = tid; k < 2 * P * w; k += nt) {
      const int kq = k / w, h = t.level_off[l] + k % w;
      const int fc = t.first_child[h], nc = t.nchild[h];
      double v = 0.0;
      for (int a = 0; a < nc; ++a) {
        const double pr = t.edge_prob[fc + a];
        if (kq < P) { if (pr > 0.0) v += pr * top[kq * NT + fc + a]; }
        else v += pr * top[kq * NT + fc + a];
      }
      top[kq * NT + h] = v;
    }
    __syncthreads();
  }
  if (tid < 2 * P) ea.out[tid] = top[tid * NT];
  if (ea.keep)
    for (int h = tid; h < top0; h += nt) ea.keep[h] = top[(P + ea.keep_r) * NT + h];
  if (tid == 0) __hip_atomic_store(ej.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// ExternalSamplingMCCFRSolver::UpdateRegrets (external_sampling_mccfr.cc:122-186),
// AverageType::kSimple, one traversal per thread, tables frozen for the launch.
// ---------------------------------------------------------------------------
OSG_D void add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }  // hardware fp64 atomic (LDS and L2)
// The counter stream of an external-sampling trajectory by where the walk is (level = traverser nodes above, 0 .. 2).
OSG_HD uint64_t es_stream(int level, int b1, int b2) {
  return level == 0 ? 0u : (level == 1 ? 1u + static_cast<uint64_t>(b1) : 16u + 8u * static_cast<uint64_t>(b1) + static_cast<uint64_t>(b2));
}

// Streams of trajectory g: (seed, g, 0) in visiting order down to the FIRST node at which the traverser acts; inside
// that node's child b1, down to the traverser's NEXT node on the path, sub-stream 1 + b1; inside that node's child b2
// sub-stream 16 + 8 b1 + b2 — a sub-stream is the same generator after a jump of its counter (Rng::jump_to).  The subtrees below a traverser node are independent but for the order of
// the draws: with a stream each they can be walked by different lanes (k_mccfr_resident<., kSplit>); the oracle's replay
// follows the same rule (osgo_mccfr_minibatch).
// kExtU: the uniforms come from a caller-supplied sequence (ext_u[0], ext_u[1], ... in visiting order) instead
// of the counter streams: with the sequence the reference's std::mt19937 + uniform_real_distribution would
// produce, one trajectory IS one UpdateRegrets call of the reference, draw for draw
// (ExternalSamplingMCCFRSolver::RunIteration(std::mt19937*), external_sampling_mccfr.h:63-100).
template <bool kLdsDelta, bool kExtU = false>
__global__ void __launch_bounds__(256)
k_mccfr(Tree t, const double* __restrict__ regrets, double* g_dreg, double* g_dpol, uint64_t seed,
        int64_t first, int64_t count, const double* __restrict__ ext_u = nullptr, int ext_n = 0,
        int32_t* ext_used = nullptr) {
  extern __shared__ double smem[];
  const int A = t.A, P = t.P, IA = t.I * t.A;
  double* dreg = kLdsDelta ? smem : g_dreg;
  double* dpol = kLdsDelta ? smem + IA : g_dpol;
  if (kLdsDelta) {
    for (int k = threadIdx.x; k < 2 * IA; k += blockDim.x) smem[k] = 0.0;
    __syncthreads();
  }
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < count; j += stride) {
    const int64_t g = first + j;
    const int trav = static_cast<int>(g % P);
    Rng rng(seed, static_cast<uint64_t>(g), 0);
    const uint64_t s0 = rng.s;
    int uk = 0;
    auto next_u = [&]() -> double {
      if (kExtU) { const double u = uk < ext_n ? ext_u[uk] : 0.0; ++uk; return u; }
      return rng.unit();
    };
    int f_node[kMaxFrames];
    int f_a[kMaxFrames];
    double f_value[kMaxFrames];
    double f_cv[kMaxFrames][kMaxA];
    int sp = 0;
    int node = 0;
    for (;;) {
      // ---- descend to a terminal, pushing a frame at every node of the traverser ----
      double ret;
      for (;;) {
        const int k = t.kind[node];
        if (k == kTerminalNode) { ret = t.term_ret[node * P + trav]; break; }
        const int fc = t.first_child[node], nc = t.nchild[node];
        if (k == kChanceNode) {  // SampleAction(ChanceOutcomes(), z) (spiel.cc:372-409)
          const double z = next_u();
          int pick = nc - 1;
          double acc = 0.0;
          for (int c = 0; c < nc; ++c) {
            const double pr = t.edge_prob[fc + c];
            if (acc <= z && z < acc + pr) { pick = c; break; }
            acc += pr;
          }
          node = fc + pick;
          continue;
        }
        const int i = t.info[node];
        if (t.actor[node] != trav) {  // opponent: sample one action from regret matching (:151-154)
          double pol[kMaxA];
          regret_match_row(regrets + i * A, pol, nc);
          const double z = next_u();
          int pick = nc - 1;  // SampleActionIndex(0.0, z) (cfr.cc:617-628)
          double acc = 0.0;
          for (int a = 0; a < nc; ++a) {
            const double pr = 0.0 * 1.0 / nc + (1.0 - 0.0) * pol[a];
            if (z >= acc && z < acc + pr) { pick = a; break; }
            acc += pr;
          }
          if (t.actor[node] == (trav + 1) % P)  // kSimple averaging at player+1's nodes (:177-183)
            for (int a = 0; a < nc; ++a) add_f64(&dpol[i * A + a], pol[a]);
          node = fc + pick;
          continue;
        }
        f_node[sp] = node;  // traverser: walk every action (:155-162)
        f_a[sp] = 0;
        f_value[sp] = 0.0;
        ++sp;
        node = fc;
        if (!kExtU && sp <= 2) rng.jump_to(s0, es_stream(sp, sp == 2 ? f_a[0] : 0, 0));   // child 0 of the traverser's first / second node: its own stream
      }
      // ---- ascend: hand `ret` to the innermost open frame ----
      bool done = false;
      for (;;) {
        if (sp == 0) { done = true; break; }
        const int fn = f_node[sp - 1];
        const int i = t.info[fn];
        const int nc = t.nchild[fn];
        double pol[kMaxA];
        regret_match_row(regrets + i * A, pol, nc);
        const int a = f_a[sp - 1];
        f_cv[sp - 1][a] = ret;
        f_value[sp - 1] += pol[a] * ret;
        if (a + 1 < nc) {
          f_a[sp - 1] = a + 1;
          node = t.first_child[fn] + a + 1;
          if (!kExtU && sp <= 2) rng.jump_to(s0, sp == 1 ? es_stream(1, a + 1, 0) : es_stream(2, f_a[0], a + 1));
          break;
        }
        const double v = f_value[sp - 1];
        for (int b = 0; b < nc; ++b) add_f64(&dreg[i * A + b], f_cv[sp - 1][b] - v);  // (:167-172)
        ret = v;
        --sp;
      }
      if (done) break;
    }
    if (kExtU && ext_used) *ext_used = uk;
  }
  if (kLdsDelta) {
    __syncthreads();
    for (int k = threadIdx.x; k < IA; k += blockDim.x) {
      const double r = smem[k], q = smem[IA + k];
      if (r != 0.0) add_f64(&g_dreg[k], r);
      if (q != 0.0) add_f64(&g_dpol[k], q);
    }
  }
}

// ---------------------------------------------------------------------------
// The same traversal with everything on its dependent chain in LDS: the tree as one
// 8-byte record per history, the launch's regret-matched policy (the table is frozen,
// so ApplyRegretMatching runs once per infostate and workgroup, not once per visit), the
// distinct terminal return vectors / chance probabilities, and the two delta tables.
// One workgroup per CU; the frame on top of the traverser's stack lives in registers,
// deeper frames are spilled to a per-lane backing store on push and reloaded on pop.
//
//   rec.x  kind [0:2) | nchild [2:8) | actor + 1 [8:12) | infostate id [12:32)
//   rec.y  first child (terminal nodes: index of the return vector) [0:24) |
//          index of the incoming edge's chance probability [24:32)
// ---------------------------------------------------------------------------
#ifndef OSG_MCCFR_TREE_GLOBAL_DEFAULT
#define OSG_MCCFR_TREE_GLOBAL_DEFAULT 1   // round 6: 3.04e9 -> 4.23e9 trajectories/s at 16 x 2^20 (profiles/r06c_mccfr_tree_in_l2_ab.txt)
#endif
struct ResidentTree {
  const uint2* rec;      // [H]
  const double* uret;    // [K, P] distinct Returns() vectors
  const double* uprob;   // [nprob] distinct chance probabilities
  int K, nprob;
  int tree_global;       // 1: the traversals read the records from `rec` itself (read-only, L2-resident) and LDS holds
                         // the tables only, so that two workgroups fit a CU (leduc: 67 KB instead of 143 KB); 0: staged in LDS
};

// Fills one workgroup's LDS for the resident traversals: zeroed delta tables, the regret-matched policy
// of every infostate (CFRInfoStateValues::ApplyRegretMatching, cfr.cc:596-615; rows padded with 0),
// the distinct return vectors / chance probabilities and the packed tree.  Caller synchronises.
template <int kA>
OSG_D void resident_load(double* smem, int H, int I, int P, const ResidentTree& rt, const int32_t* __restrict__ nact,
                         const double* __restrict__ regrets, double** o_dreg, double** o_dpol, double** o_pol,
                         double** o_uret, double** o_uprob, const uint2** o_nodes) {
  const int IA = I * kA;
  double* pol = smem + 2 * IA;
  double* uret = smem + 3 * IA;
  double* uprob = uret + rt.K * P;
  uint2* nodes = reinterpret_cast<uint2*>(uprob + rt.nprob);
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int k = tid; k < 2 * IA; k += nt) smem[k] = 0.0;
  // The staging loops keep several independent global loads in flight per thread: written as "load, then
  // store" per element they were a chain of dependent round trips (a 256-thread workgroup staged leduc's 9 457
  // records in 37 of them — 59 us of fixed cost per launch, most of a small mini-batch's time).
  constexpr int kU = 4;
  for (int i0 = tid; i0 < I; i0 += kU * nt) {
    double row[kU][kA];
    int n[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = i0 + u * nt;
      n[u] = i < I ? nact[i] : 0;
#pragma unroll
      for (int a = 0; a < kA; ++a) row[u][a] = i < I ? regrets[i * kA + a] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = i0 + u * nt;
      if (i >= I) continue;
      double sum_pos = 0.0;
#pragma unroll
      for (int a = 0; a < kA; ++a)
        if (a < n[u] && row[u][a] > 0) sum_pos += row[u][a];
#pragma unroll
      for (int a = 0; a < kA; ++a) {
        double o;
        if (a >= n[u]) o = 0.0;
        else if (sum_pos > 0) o = row[u][a] > 0 ? row[u][a] / sum_pos : 0.0;
        else o = 1.0 / n[u];
        pol[i * kA + a] = o;
      }
    }
  }
  for (int k = tid; k < rt.K * P; k += nt) uret[k] = rt.uret[k];
  for (int k = tid; k < rt.nprob; k += nt) uprob[k] = rt.uprob[k];
  if (rt.tree_global) {
    *o_dreg = smem; *o_dpol = smem + IA; *o_pol = pol; *o_uret = uret; *o_uprob = uprob; *o_nodes = rt.rec;
    return;
  }
  {  // the packed tree, two records (16 bytes) per load, kU loads in flight per thread
    const uint4* __restrict__ src4 = reinterpret_cast<const uint4*>(rt.rec);
    const int n4 = H / 2;
    for (int k0 = tid; k0 < n4; k0 += kU * nt) {
      uint4 v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int k = k0 + u * nt;
        v[u] = k < n4 ? src4[k] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int k = k0 + u * nt;
        if (k < n4) {
          nodes[2 * k] = make_uint2(v[u].x, v[u].y);
          nodes[2 * k + 1] = make_uint2(v[u].z, v[u].w);
        }
      }
    }
    if ((H & 1) && tid == 0) nodes[H - 1] = rt.rec[H - 1];
  }
  *o_dreg = smem; *o_dpol = smem + IA; *o_pol = pol; *o_uret = uret; *o_uprob = uprob; *o_nodes = nodes;
}

OSG_D void resident_flush(const double* dreg, const double* dpol, double* g_dreg, double* g_dpol, int IA) {
  __syncthreads();
  for (int k = threadIdx.x; k < IA; k += blockDim.x) {
    const double r = dreg[k], q = dpol[k];
    if (r != 0.0) add_f64(&g_dreg[k], r);
    if (q != 0.0) add_f64(&g_dpol[k], q);
  }
}

// What a lane needs of the staged problem.
template <int kA>
struct EsView {
  const uint2* nodes;
  const double *pol, *uret, *uprob;
  double *dreg, *dpol;
  int P, trav, next;
};
// One step at a node where the traverser does not act: the sampled child (chance: SampleAction(ChanceOutcomes(), z),
// spiel.cc:372-409; opponent: SampleActionIndex(0.0, z) on the regret-matched row, cfr.cc:617-628), with the kSimple
// average-policy update at player + 1's nodes (external_sampling_mccfr.cc:177-183) when `averaging`.
template <int kA>
OSG_D int es_sampled_child(const EsView<kA>& c, uint2 rec, Rng& rng, bool averaging) {
  const int kind = rec.x & 3u, nc = (rec.x >> 2) & 63u, fc = rec.y & 0xFFFFFFu;
  const int i = rec.x >> 12;
  const int actor = static_cast<int>((rec.x >> 8) & 15u) - 1;  // -1 at chance nodes
  const double z = rng.unit();
  int pick = nc - 1;
  if (kind == kChanceNode) {
    double acc = 0.0;
    bool found = false;
    if (i != 0) {  // all outcomes equally likely: the same scan, the probability read once
      const double pr = c.uprob[i - 1];
      for (int k = 0; k < nc; ++k) {
        if (!found && acc <= z && z < acc + pr) { pick = k; found = true; }
        acc += pr;
      }
    } else {
      for (int k = 0; k < nc; ++k) {
        const double pr = c.uprob[c.nodes[fc + k].y >> 24];
        if (!found && acc <= z && z < acc + pr) { pick = k; found = true; }
        acc += pr;
      }
    }
  } else {  // opponent: sample one action from regret matching (:151-154)
    double p[kA];
#pragma unroll
    for (int a = 0; a < kA; ++a) p[a] = c.pol[i * kA + a];
    double acc = 0.0;
    bool found = false;
#pragma unroll
    for (int a = 0; a < kA; ++a) {
      if (!found && a < nc && z >= acc && z < acc + p[a]) { pick = a; found = true; }
      acc += p[a];
    }
    if (averaging && actor == c.next) {
#pragma unroll
      for (int a = 0; a < kA; ++a)
        if (a < nc) add_f64(&c.dpol[i * kA + a], p[a]);
    }
  }
  return fc + pick;
}
// UpdateRegrets from `node` down (external_sampling_mccfr.cc:122-186): the value of `node` for the traverser, the
// regret and average-policy terms of everything below added to the LDS delta tables.  The frame on top of the
// traverser's stack lives in registers, deeper frames in a per-lane backing store touched on push / pop only.
// kBase traverser nodes lie above `node` on the trajectory (0: `node` is the root): the children of the traverser's
// first and second node draw from their own streams (es_stream); b1: the child of the first node the walk is in (kBase >= 1).
template <int kA, int kBase>
OSG_D double es_walk(const EsView<kA>& c, int node, Rng& rng, uint64_t s0, int b1) {   // s0 = the state (seed, g, 0) starts from
  // (the loop is written like k_mccfr_resident_flat's, sampling code in line: through es_sampled_child it ran a quarter slower)
  const uint2* __restrict__ nodes = c.nodes;
  const double* __restrict__ pol = c.pol;
  const int trav = c.trav, next = c.next, P = c.P;
  uint32_t s_x[kMaxFrames], s_fa[kMaxFrames];
  double s_v[kMaxFrames], s_cv[kMaxFrames][kA];
  uint32_t top_x = 0, top_fc = 0;
  int top_a = 0;
  double top_v = 0.0, top_cv[kA];
#pragma unroll
  for (int b = 0; b < kA; ++b) top_cv[b] = 0.0;
  int sp = 0;
  double ret = 0.0;
  for (;;) {
    const uint2 rec = nodes[node];
    const int kind = rec.x & 3u;
    if (kind != kTerminalNode) {
      const int nc = (rec.x >> 2) & 63u, fc = rec.y & 0xFFFFFFu;
      const int i = rec.x >> 12;
      const int actor = static_cast<int>((rec.x >> 8) & 15u) - 1;  // -1 at chance nodes
      if (actor != trav) {
        const double z = rng.unit();
        int pick = nc - 1;
        if (kind == kChanceNode) {  // SampleAction(ChanceOutcomes(), z) (spiel.cc:372-409)
          double acc = 0.0;
          bool found = false;
          if (i != 0) {  // all outcomes equally likely: the same scan, the probability read once
            const double pr = c.uprob[i - 1];
            for (int k = 0; k < nc; ++k) {
              if (!found && acc <= z && z < acc + pr) { pick = k; found = true; }
              acc += pr;
            }
          } else {
            for (int k = 0; k < nc; ++k) {
              const double pr = c.uprob[nodes[fc + k].y >> 24];
              if (!found && acc <= z && z < acc + pr) { pick = k; found = true; }
              acc += pr;
            }
          }
        } else {  // opponent: sample one action from regret matching (:151-154)
          double p[kA];
#pragma unroll
          for (int a = 0; a < kA; ++a) p[a] = pol[i * kA + a];
          double acc = 0.0;  // SampleActionIndex(0.0, z) (cfr.cc:617-628)
          bool found = false;
#pragma unroll
          for (int a = 0; a < kA; ++a) {
            if (!found && a < nc && z >= acc && z < acc + p[a]) { pick = a; found = true; }
            acc += p[a];
          }
          if (actor == next) {  // kSimple averaging at player+1's nodes (:177-183)
#pragma unroll
            for (int a = 0; a < kA; ++a)
              if (a < nc) add_f64(&c.dpol[i * kA + a], p[a]);
          }
        }
        node = fc + pick;
        continue;
      }
      // traverser: walk every action (:155-162)
      if (sp > 0) {
        if (kBase == 0 && sp == 1) b1 = top_a;   // the walk is about to enter the second traverser node inside child top_a
        s_x[sp - 1] = top_x;
        s_fa[sp - 1] = top_fc | (static_cast<uint32_t>(top_a) << 24);
        s_v[sp - 1] = top_v;
#pragma unroll
        for (int b = 0; b < kA; ++b) s_cv[sp - 1][b] = top_cv[b];
      }
      top_x = rec.x; top_fc = fc; top_a = 0; top_v = 0.0;
      ++sp;
      node = fc;
      if (kBase + sp <= 2) rng.jump_to(s0, es_stream(kBase + sp, kBase + sp == 1 ? 0 : b1, 0));
      continue;
    }
    ret = c.uret[(rec.y & 0xFFFFFFu) * P + trav];
    bool done = false;
    for (;;) {  // hand `ret` to the innermost open frame
      if (sp == 0) { done = true; break; }
      const int i = top_x >> 12, nc = (top_x >> 2) & 63u;
      const double pa = pol[i * kA + top_a];
#pragma unroll
      for (int b = 0; b < kA; ++b)
        if (b == top_a) top_cv[b] = ret;
      top_v += pa * ret;
      if (top_a + 1 < nc) {
        ++top_a;
        node = top_fc + top_a;
        if (kBase + sp <= 2) rng.jump_to(s0, kBase + sp == 1 ? es_stream(1, top_a, 0) : es_stream(2, b1, top_a));
        break;
      }
#pragma unroll
      for (int b = 0; b < kA; ++b)
        if (b < nc) add_f64(&c.dreg[i * kA + b], top_cv[b] - top_v);  // (:167-172)
      ret = top_v;
      --sp;
      if (sp > 0) {
        top_x = s_x[sp - 1];
        top_fc = s_fa[sp - 1] & 0xFFFFFFu;
        top_a = s_fa[sp - 1] >> 24;
        top_v = s_v[sp - 1];
#pragma unroll
        for (int b = 0; b < kA; ++b) top_cv[b] = s_cv[sp - 1][b];
      }
    }
    if (done) break;
  }
  return ret;
}

// One trajectory per lane, for mini-batches that fill the chip: the traversal as ONE flat loop (the form the split
// kernels below share their pieces with was measured 27 % slower here: 403 vs 309 us per 2^20 trajectories).
template <int kA>
__global__ void __launch_bounds__(1024)
k_mccfr_resident_flat(int H, int I, int P, ResidentTree rt, const int32_t* __restrict__ nact,
                 const double* __restrict__ regrets, double* g_dreg, double* g_dpol, uint64_t seed, int64_t first,
                 int64_t count) {
  extern __shared__ double smem[];
  const int IA = I * kA;
  double *dreg, *dpol, *pol, *uret, *uprob;
  const uint2* nodes;
  resident_load<kA>(smem, H, I, P, rt, nact, regrets, &dreg, &dpol, &pol, &uret, &uprob, &nodes);
  __syncthreads();

  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t j0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j0 < count; j0 += stride) {
    // Which trajectory a lane takes: within every full group of 64 P consecutive ones, wavefront w of the group takes
    // those with the same traverser (index = lane * P + w), so that the 64 lanes of a wavefront agree at every node
    // on whether they walk all actions or sample one — half the divergence of the natural order, same set of
    // trajectories.  (The last, partial group keeps the natural order.)
    int64_t j = j0;
    {
      const int64_t span = 64 * static_cast<int64_t>(P), group = j0 / span;
      if ((group + 1) * span <= count) {
        const int r = static_cast<int>(j0 - group * span);
        j = group * span + static_cast<int64_t>(r & 63) * P + (r >> 6);
      }
    }
    const int64_t g = first + j;
    // (a 64-bit modulo by a run-time divisor is ~100 instructions: two players take the parity)
    const int trav = P == 2 ? static_cast<int>(g & 1) : static_cast<int>(g % P);
    const int next = trav + 1 == P ? 0 : trav + 1;
    Rng rng(seed, static_cast<uint64_t>(g), 0);
    const uint64_t s0 = rng.s;   // the sub-streams of the traverser's first two levels are jumps of this counter (es_stream)
    int b1 = 0;
    // backing store of the frames below the top one
    uint32_t s_x[kMaxFrames], s_fa[kMaxFrames];
    double s_v[kMaxFrames], s_cv[kMaxFrames][kA];
    uint32_t top_x = 0, top_fc = 0;
    int top_a = 0;
    double top_v = 0.0, top_cv[kA];
#pragma unroll
    for (int b = 0; b < kA; ++b) top_cv[b] = 0.0;
#if OSG_MCCFR_FRAMES2
    // the frame below the top one, in registers too: a pop then takes its frame from registers and only REQUESTS the one
    // that becomes second — nobody waits for the backing store on the traversal's chain, and a push writes to it only
    // from the third level on (leduc_poker: the traverser acts at most four times on a path)
    uint32_t sec_x = 0, sec_fc = 0;
    int sec_a = 0;
    double sec_v = 0.0, sec_cv[kA];
#pragma unroll
    for (int b = 0; b < kA; ++b) sec_cv[b] = 0.0;
#endif
    int sp = 0;
    int node = 0;
    for (;;) {
      const uint2 rec = nodes[node];
#if OSG_MCCFR_PEEK
      // The next uniform of the stream, formed WHILE the node's record is on its way from LDS: the generator is a
      // counter and a mixer, so the draw does not depend on the node — only whether it is consumed does (a node of the
      // traverser or a terminal leaves the counter where it was).  Inside the branch the ~35 instructions of the mixer
      // sat on the traversal's dependent chain behind the record's decode (profiles/r05a_pmc_solvers.json: the waves of
      // this kernel are parked two thirds of their cycles); the empty asm keeps the compiler from sinking them back.
      const uint64_t s_peek = rng.s + 0x9E3779B97F4A7C15ULL;
      double z_peek = static_cast<double>(mix64(s_peek) >> 11) * (1.0 / 9007199254740992.0);
      asm volatile("" : "+v"(z_peek));
#endif
      const int kind = rec.x & 3u;
      if (kind != kTerminalNode) {
        const int nc = (rec.x >> 2) & 63u, fc = rec.y & 0xFFFFFFu;
        const int i = rec.x >> 12;
        const int actor = static_cast<int>((rec.x >> 8) & 15u) - 1;  // -1 at chance nodes
        if (actor != trav) {
#if OSG_MCCFR_PEEK
          const double z = z_peek;
          rng.s = s_peek;
#else
          const double z = rng.unit();
#endif
          int pick = nc - 1;
          if (kind == kChanceNode) {  // SampleAction(ChanceOutcomes(), z) (spiel.cc:372-409)
            double acc = 0.0;
            bool found = false;
            if (i != 0) {  // all outcomes equally likely: the same scan, the probability read once
              const double pr = uprob[i - 1];
              for (int c = 0; c < nc; ++c) {
                if (!found && acc <= z && z < acc + pr) { pick = c; found = true; }
                acc += pr;
              }
            } else {
              for (int c = 0; c < nc; ++c) {
                const double pr = uprob[nodes[fc + c].y >> 24];
                if (!found && acc <= z && z < acc + pr) { pick = c; found = true; }
                acc += pr;
              }
            }
          } else {  // opponent: sample one action from regret matching (:151-154)
            double p[kA];
#pragma unroll
            for (int a = 0; a < kA; ++a) p[a] = pol[i * kA + a];
            double acc = 0.0;  // SampleActionIndex(0.0, z) (cfr.cc:617-628)
            bool found = false;
#pragma unroll
            for (int a = 0; a < kA; ++a) {
              if (!found && a < nc && z >= acc && z < acc + p[a]) { pick = a; found = true; }
              acc += p[a];
            }
            if (actor == next) {  // kSimple averaging at player+1's nodes (:177-183)
#pragma unroll
              for (int a = 0; a < kA; ++a)
                if (a < nc) add_f64(&dpol[i * kA + a], p[a]);
            }
          }
          node = fc + pick;
          continue;
        }
        // traverser: walk every action (:155-162)
        if (sp > 0) {
          if (sp == 1) b1 = top_a;   // entering the traverser's second node inside child top_a of the first
#if OSG_MCCFR_FRAMES2
          if (sp > 1) {              // the frame below the top one leaves for the backing store (slot k = frame k)
            s_x[sp - 2] = sec_x;
            s_fa[sp - 2] = sec_fc | (static_cast<uint32_t>(sec_a) << 24);
            s_v[sp - 2] = sec_v;
#pragma unroll
            for (int b = 0; b < kA; ++b) s_cv[sp - 2][b] = sec_cv[b];
          }
          sec_x = top_x; sec_fc = top_fc; sec_a = top_a; sec_v = top_v;
#pragma unroll
          for (int b = 0; b < kA; ++b) sec_cv[b] = top_cv[b];
#else
          s_x[sp - 1] = top_x;
          s_fa[sp - 1] = top_fc | (static_cast<uint32_t>(top_a) << 24);
          s_v[sp - 1] = top_v;
#pragma unroll
          for (int b = 0; b < kA; ++b) s_cv[sp - 1][b] = top_cv[b];
#endif
        }
        top_x = rec.x; top_fc = fc; top_a = 0; top_v = 0.0;
        ++sp;
        node = fc;
        if (sp <= 2) rng.jump_to(s0, es_stream(sp, sp == 1 ? 0 : b1, 0));
        continue;
      }
      double ret = uret[(rec.y & 0xFFFFFFu) * P + trav];
      bool done = false;
      for (;;) {  // hand `ret` to the innermost open frame
        if (sp == 0) { done = true; break; }
        const int i = top_x >> 12, nc = (top_x >> 2) & 63u;
        const double pa = pol[i * kA + top_a];
#pragma unroll
        for (int b = 0; b < kA; ++b)
          if (b == top_a) top_cv[b] = ret;
        top_v += pa * ret;
        if (top_a + 1 < nc) {
          ++top_a;
          node = top_fc + top_a;
          if (sp <= 2) rng.jump_to(s0, sp == 1 ? es_stream(1, top_a, 0) : es_stream(2, b1, top_a));
          break;
        }
#pragma unroll
        for (int b = 0; b < kA; ++b)
          if (b < nc) add_f64(&dreg[i * kA + b], top_cv[b] - top_v);  // (:167-172)
        ret = top_v;
        --sp;
        if (sp > 0) {
#if OSG_MCCFR_FRAMES2
          top_x = sec_x; top_fc = sec_fc; top_a = sec_a; top_v = sec_v;
#pragma unroll
          for (int b = 0; b < kA; ++b) top_cv[b] = sec_cv[b];
          if (sp > 1) {   // requested now, needed at the NEXT pop (or push): its trip to memory is off the chain
            sec_x = s_x[sp - 2];
            sec_fc = s_fa[sp - 2] & 0xFFFFFFu;
            sec_a = s_fa[sp - 2] >> 24;
            sec_v = s_v[sp - 2];
#pragma unroll
            for (int b = 0; b < kA; ++b) sec_cv[b] = s_cv[sp - 2][b];
          }
#else
          top_x = s_x[sp - 1];
          top_fc = s_fa[sp - 1] & 0xFFFFFFu;
          top_a = s_fa[sp - 1] >> 24;
          top_v = s_v[sp - 1];
#pragma unroll
          for (int b = 0; b < kA; ++b) top_cv[b] = s_cv[sp - 1][b];
#endif
        }
      }
      if (done) break;
    }
  }
  resident_flush(dreg, dpol, g_dreg, g_dpol, IA);
}

// kSplit = 1 / 2: kQ = 2 (kA <= 2) or 4 lanes per traverser level, kQ or kQ^2 lanes per trajectory.  A traversal is
// one dependent chain (leduc: ~100 node visits, 42-45 us on one lane whatever the batch — profiles/r04_mccfr_shard.log):
// all lanes of a group walk the sampled path down to the traverser's first node (the same draws: the same path; lane 0
// does the averaging), the lanes of child b1 walk on from there on its stream — with kSplit = 2 down to the traverser's
// next node, whose child b2 lane (b1, b2) then walks —, the values come back by lane shuffles and a node's own terms are
// added in action order: the sums of the one-lane form.  For mini-batches that leave lanes idle anyway.
template <int kA, int kSplit = 0>
__global__ void __launch_bounds__(1024)
k_mccfr_resident(int H, int I, int P, ResidentTree rt, const int32_t* __restrict__ nact,
                 const double* __restrict__ regrets, double* g_dreg, double* g_dpol, uint64_t seed, int64_t first,
                 int64_t count, unsigned long long* stamps = nullptr) {
  extern __shared__ double smem[];
  const int IA = I * kA;
  double *dreg, *dpol, *pol, *uret, *uprob;
  const uint2* nodes;
  const bool stamp = stamps && blockIdx.x == 0 && threadIdx.x == 0;   // OSG_MCCFR_STAMPS: where a launch's time goes
  if (stamp) stamps[0] = wall_clock64();
  resident_load<kA>(smem, H, I, P, rt, nact, regrets, &dreg, &dpol, &pol, &uret, &uprob, &nodes);
  __syncthreads();
  if (stamp) stamps[1] = wall_clock64();
  constexpr int kQ = kA <= 2 ? 2 : 4;                                      // lanes per traverser level
  constexpr int kLanes = kSplit == 0 ? 1 : (kSplit == 1 ? kQ : kQ * kQ);   // lanes per trajectory
  constexpr int kPerWave = 64 / kLanes;                                    // trajectories per wavefront

  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x / kLanes;
  const int64_t lane_slot = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) / kLanes;
  const int in_group = static_cast<int>(threadIdx.x) & (kLanes - 1);
  const int b1 = kSplit == 2 ? in_group / kQ : in_group, b2 = kSplit == 2 ? in_group % kQ : 0;
  const int group_base = static_cast<int>(threadIdx.x) & 63 & ~(kLanes - 1);   // the group's first lane in its wavefront
  const int64_t rounds = (count + stride - 1) / stride;   // (every lane runs every round: the shuffles need the whole wavefront)
  for (int64_t rd = 0; rd < rounds; ++rd) {
    const int64_t j0 = lane_slot + rd * stride;
    const bool live = j0 < count;
    // Which trajectory a lane takes: within every full group of kPerWave P consecutive ones, wavefront w of the group
    // takes those with the same traverser (index = slot * P + w), so that the lanes of a wavefront agree at every node
    // on whether they walk all actions or sample one — half the divergence of the natural order, same set of
    // trajectories.  (The last, partial group keeps the natural order.)
    int64_t j = live ? j0 : 0;
    {
      const int64_t span = kPerWave * static_cast<int64_t>(P), group = j / span;
      if ((group + 1) * span <= count) {
        const int r = static_cast<int>(j - group * span);
        j = group * span + static_cast<int64_t>(r % kPerWave) * P + (r / kPerWave);
      }
    }
    const uint64_t g = static_cast<uint64_t>(first + j);
    // (a 64-bit modulo by a run-time divisor is ~100 instructions: two players take the parity)
    const int trav = P == 2 ? static_cast<int>(g & 1) : static_cast<int>(g % P);
    const EsView<kA> view{nodes, pol, uret, uprob, dreg, dpol, P, trav, trav + 1 == P ? 0 : trav + 1};
    Rng rng(seed, g, 0);
    const uint64_t s0 = rng.s;
    if (kSplit == 0) {
      if (live) (void)es_walk<kA, 0>(view, 0, rng, s0, 0);
      continue;
    }
    // ---- the shared path down to the traverser's first node ----
    int node = 0;
    uint2 rec = nodes[0];
    bool at1 = false;
    if (live) {
      for (;;) {
        rec = nodes[node];
        if ((rec.x & 3u) == kTerminalNode) break;
        if (static_cast<int>((rec.x >> 8) & 15u) - 1 == trav) { at1 = true; break; }
        node = es_sampled_child<kA>(view, rec, rng, in_group == 0);
      }
    }
    const uint2 rec1 = rec;
    const int nc1 = at1 ? static_cast<int>((rec1.x >> 2) & 63u) : 0, fc1 = static_cast<int>(rec1.y & 0xFFFFFFu);
    double value1 = 0.0;   // the value of child b1 of the first node
    if (b1 < nc1) {
      Rng sub = rng;
      sub.jump_to(s0, es_stream(1, b1, 0));
      if (kSplit == 1) {
        value1 = es_walk<kA, 1>(view, fc1 + b1, sub, s0, b1);
      } else {
        // ---- the path inside child b1 down to the traverser's next node (all b2 lanes: the same draws) ----
        node = fc1 + b1;
        bool at2 = false;
        for (;;) {
          rec = nodes[node];
          if ((rec.x & 3u) == kTerminalNode) break;
          if (static_cast<int>((rec.x >> 8) & 15u) - 1 == trav) { at2 = true; break; }
          node = es_sampled_child<kA>(view, rec, sub, b2 == 0);
        }
        if (!at2) {
          value1 = uret[(rec.y & 0xFFFFFFu) * P + trav];
        } else {
          const int nc2 = static_cast<int>((rec.x >> 2) & 63u), fc2 = static_cast<int>(rec.y & 0xFFFFFFu), i2 = rec.x >> 12;
          double mine = 0.0;
          if (b2 < nc2) {
            Rng sub2 = sub;
            sub2.jump_to(s0, es_stream(2, b1, b2));
            mine = es_walk<kA, 2>(view, fc2 + b2, sub2, s0, b1);
          }
          // (the b2 lanes of this b1 are all here: the shuffle reads them; lanes of other b1 groups shuffle below)
          double cv2[kA];
#pragma unroll
          for (int a = 0; a < kA; ++a) cv2[a] = __shfl(mine, group_base + b1 * kQ + (a < kQ ? a : 0), 64);
          double v2 = 0.0;
#pragma unroll
          for (int a = 0; a < kA; ++a)
            if (a < nc2) v2 += pol[i2 * kA + a] * cv2[a];
#pragma unroll
          for (int a = 0; a < kA; ++a)
            if (a == b2 && a < nc2) add_f64(&dreg[i2 * kA + a], cv2[a] - v2);
          value1 = v2;
        }
      }
    }
    // ---- the first node's own terms: values from the group's lanes, added in action order (:155-172) ----
    double cv[kA];
#pragma unroll
    for (int a = 0; a < kA; ++a) cv[a] = __shfl(value1, group_base + (a < kQ ? a : 0) * (kSplit == 2 ? kQ : 1), 64);
    if (at1) {
      const int i = rec1.x >> 12;
      double v = 0.0;
#pragma unroll
      for (int a = 0; a < kA; ++a)
        if (a < nc1) v += pol[i * kA + a] * cv[a];
#pragma unroll
      for (int a = 0; a < kA; ++a)
        if (a == b1 && b2 == 0 && a < nc1) add_f64(&dreg[i * kA + a], cv[a] - v);
    }
  }
  if (stamp) stamps[2] = wall_clock64();   // (lane 0's own trajectories; the flush below waits for the workgroup's last)
  resident_flush(dreg, dpol, g_dreg, g_dpol, IA);
  if (stamp) { stamps[3] = wall_clock64(); }
}

// ---------------------------------------------------------------------------
// OutcomeSamplingMCCFRSolver::SampleEpisode (outcome_sampling_mccfr.cc:141-241),
// Baseline() == 0: ONE sampled path per thread.  The walk down records, per decision
// node, the regret-matched policy, the sampled action and the three reaches; the walk
// back up turns the terminal return into value estimates and adds the update player's
// regret / average-policy terms (importance weights 1 / sample_reach).  Tables frozen
// for the launch, deltas in LDS like k_mccfr.
// ---------------------------------------------------------------------------
constexpr int kMaxOsDepth = 32;

template <bool kLdsDelta>
__global__ void __launch_bounds__(256)
k_os_mccfr(Tree t, const double* __restrict__ regrets, double* g_dreg, double* g_dpol, uint64_t seed, int64_t first,
           int64_t count, double epsilon) {
  extern __shared__ double smem[];
  const int A = t.A, P = t.P, IA = t.I * t.A;
  double* dreg = kLdsDelta ? smem : g_dreg;
  double* dpol = kLdsDelta ? smem + IA : g_dpol;
  if (kLdsDelta) {
    for (int k = threadIdx.x; k < 2 * IA; k += blockDim.x) smem[k] = 0.0;
    __syncthreads();
  }
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < count; j += stride) {
    const int64_t g = first + j;
    const int upd = static_cast<int>(g % P);
    Rng rng(seed, static_cast<uint64_t>(g), 0);
    int f_info[kMaxOsDepth], f_aidx[kMaxOsDepth];
    double f_my[kMaxOsDepth], f_opp[kMaxOsDepth], f_samp[kMaxOsDepth], f_sp[kMaxOsDepth];
    bool f_upd[kMaxOsDepth];
    int depth = 0;
    int node = 0;
    double my = 1.0, opp = 1.0, samp = 1.0;
    while (t.kind[node] != kTerminalNode && depth < kMaxOsDepth) {
      const int fc = t.first_child[node], nc = t.nchild[node];
      if (t.kind[node] == kChanceNode) {  // SampleAction(ChanceOutcomes(), z) (:146-153)
        const double z = rng.unit();
        int pick = nc - 1;
        double acc = 0.0;
        for (int c = 0; c < nc; ++c) {
          const double pr = t.edge_prob[fc + c];
          if (acc <= z && z < acc + pr) { pick = c; break; }
          acc += pr;
        }
        const double pr = t.edge_prob[fc + pick];
        opp = pr * opp;
        samp = pr * samp;
        node = fc + pick;
        continue;
      }
      const int i = t.info[node];
      const bool is_upd = t.actor[node] == upd;
      double pol[kMaxA];
      regret_match_row(regrets + i * A, pol, nc);
      const double z = rng.unit();
      int pick = nc - 1;
      double acc = 0.0, sp_pick = 0.0;
      for (int a = 0; a < nc; ++a) {
        const double sp = is_upd ? epsilon * 1.0 / nc + (1 - epsilon) * pol[a] : pol[a];  // SamplePolicy (:111-118)
        if (a == nc - 1) sp_pick = sp;
        if (z >= acc && z < acc + sp) { pick = a; sp_pick = sp; break; }
        acc += sp;
      }
      f_info[depth] = i; f_aidx[depth] = pick; f_my[depth] = my; f_opp[depth] = opp; f_samp[depth] = samp;
      f_sp[depth] = sp_pick; f_upd[depth] = is_upd;
      ++depth;
      if (is_upd) my = my * pol[pick]; else opp = opp * pol[pick];
      samp = samp * sp_pick;
      node = fc + pick;
    }
    double v = t.term_ret[node * P + upd];
    for (int d = depth - 1; d >= 0; --d) {
      const int i = f_info[d], n = t.nact[i], sampled = f_aidx[d];
      double pol[kMaxA];
      regret_match_row(regrets + i * A, pol, n);
      // child_values[a] = a == sampled ? 0 + (child_value - 0) / sample_policy[a] : 0 (:126-139)
      const double cv_sampled = 0.0 + (v - 0.0) / f_sp[d];
      double value_estimate = 0.0;
      for (int a = 0; a < n; ++a) value_estimate += pol[a] * (a == sampled ? cv_sampled : 0.0);
      if (f_upd[d]) {
        const double cf_value = value_estimate * f_opp[d] / f_samp[d];
        for (int a = 0; a < n; ++a) {
          const double cf_action_value = (a == sampled ? cv_sampled : 0.0) * f_opp[d] / f_samp[d];
          add_f64(&dreg[i * A + a], cf_action_value - cf_value);
          add_f64(&dpol[i * A + a], f_my[d] * pol[a] / f_samp[d]);
        }
      }
      v = value_estimate;
    }
  }
  if (kLdsDelta) {
    __syncthreads();
    for (int k = threadIdx.x; k < IA; k += blockDim.x) {
      const double r = smem[k], q = smem[IA + k];
      if (r != 0.0) add_f64(&g_dreg[k], r);
      if (q != 0.0) add_f64(&g_dpol[k], q);
    }
  }
}

// k_os_mccfr with the tree, the launch's regret-matched policy and the delta tables in LDS
// (same records as k_mccfr_resident); the per-depth frames stay in the per-lane backing store.
template <int kA>
__global__ void __launch_bounds__(1024)
k_os_mccfr_resident(int H, int I, int P, ResidentTree rt, const int32_t* __restrict__ nact,
                    const double* __restrict__ regrets, double* g_dreg, double* g_dpol, uint64_t seed, int64_t first,
                    int64_t count, double epsilon) {
  extern __shared__ double smem[];
  const int IA = I * kA;
  double *dreg, *dpol, *pol, *uret, *uprob;
  const uint2* nodes;
  resident_load<kA>(smem, H, I, P, rt, nact, regrets, &dreg, &dpol, &pol, &uret, &uprob, &nodes);
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < count; j += stride) {
    const int64_t g = first + j;
    const int upd = static_cast<int>(g % P);
    Rng rng(seed, static_cast<uint64_t>(g), 0);
    uint32_t f_x[kMaxOsDepth];  // infostate id [12:32) | nchild [2:8) | sampled action [0:2) | update player's node [8]
    double f_my[kMaxOsDepth], f_opp[kMaxOsDepth], f_samp[kMaxOsDepth], f_sp[kMaxOsDepth];
    int depth = 0;
    int node = 0;
    double my = 1.0, opp = 1.0, samp = 1.0;
    uint2 rec = nodes[0];
    while ((rec.x & 3u) != kTerminalNode && depth < kMaxOsDepth) {
      const int nc = (rec.x >> 2) & 63u, fc = rec.y & 0xFFFFFFu;
      const double z = rng.unit();
      int pick = nc - 1;
      if ((rec.x & 3u) == kChanceNode) {  // SampleAction(ChanceOutcomes(), z) (:146-153)
        double acc = 0.0, pr_pick = 0.0;
        bool found = false;
        for (int c = 0; c < nc; ++c) {
          const double pr = uprob[nodes[fc + c].y >> 24];
          if (c == nc - 1 && !found) pr_pick = pr;
          if (!found && acc <= z && z < acc + pr) { pick = c; pr_pick = pr; found = true; }
          acc += pr;
        }
        opp = pr_pick * opp;
        samp = pr_pick * samp;
      } else {
        const int i = rec.x >> 12;
        const bool is_upd = static_cast<int>((rec.x >> 8) & 15u) - 1 == upd;
        double p[kA];
#pragma unroll
        for (int a = 0; a < kA; ++a) p[a] = pol[i * kA + a];
        double acc = 0.0, sp_pick = 0.0, p_pick = 0.0;
        bool found = false;
#pragma unroll
        for (int a = 0; a < kA; ++a) {
          if (a < nc) {
            const double sp = is_upd ? epsilon * 1.0 / nc + (1 - epsilon) * p[a] : p[a];  // SamplePolicy (:111-118)
            if (a == nc - 1 && !found) { sp_pick = sp; p_pick = p[a]; }
            if (!found && z >= acc && z < acc + sp) { pick = a; sp_pick = sp; p_pick = p[a]; found = true; }
            acc += sp;
          }
        }
        f_x[depth] = (rec.x & ~0xF03u) | static_cast<uint32_t>(pick) | (is_upd ? 0x100u : 0u);
        f_my[depth] = my; f_opp[depth] = opp; f_samp[depth] = samp; f_sp[depth] = sp_pick;
        ++depth;
        if (is_upd) my = my * p_pick; else opp = opp * p_pick;
        samp = samp * sp_pick;
      }
      node = fc + pick;
      rec = nodes[node];
    }
    double v = uret[(rec.y & 0xFFFFFFu) * P + upd];
    for (int d = depth - 1; d >= 0; --d) {
      const uint32_t x = f_x[d];
      const int i = x >> 12, n = (x >> 2) & 63u, sampled = x & 3u;
      double p[kA];
#pragma unroll
      for (int a = 0; a < kA; ++a) p[a] = pol[i * kA + a];
      // child_values[a] = a == sampled ? 0 + (child_value - 0) / sample_policy[a] : 0 (:126-139)
      const double cv_sampled = 0.0 + (v - 0.0) / f_sp[d];
      double value_estimate = 0.0;
#pragma unroll
      for (int a = 0; a < kA; ++a)
        if (a < n) value_estimate += p[a] * (a == sampled ? cv_sampled : 0.0);
      if (x & 0x100u) {
        const double cf_value = value_estimate * f_opp[d] / f_samp[d];
#pragma unroll
        for (int a = 0; a < kA; ++a) {
          if (a < n) {
            const double cf_action_value = (a == sampled ? cv_sampled : 0.0) * f_opp[d] / f_samp[d];
            add_f64(&dreg[i * kA + a], cf_action_value - cf_value);
            add_f64(&dpol[i * kA + a], f_my[d] * p[a] / f_samp[d]);
          }
        }
      }
      v = value_estimate;
    }
  }
  resident_flush(dreg, dpol, g_dreg, g_dpol, IA);
}

// ---------------------------------------------------------------------------
// ExternalSamplingMCCFRSolver::FullUpdateAverage (external_sampling_mccfr.cc:188-231), AverageType::kFull:
// a full-tree pass that adds reach_probs[cur_player] * sigma(I)[a] to the cumulative policy of every
// decision history (sigma = regret matching of the regrets as they are now), skipping histories every
// player reaches with probability 0.  One workgroup: reach probabilities top-down, one level per step
// (the products round like the reference's recursion), then one thread per infostate adds its members'
// terms in DFS order (= the order the recursion reaches them).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_mccfr_full_average(Tree t, const double* __restrict__ regrets, double* cum, double* reach, double weight) {
  const int P = t.P, A = t.A;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int l = 0; l < t.D; ++l) {
    for (int h = t.level_off[l] + tid; h < t.level_off[l + 1]; h += nt) {
      if (l == 0) {
        for (int q = 0; q < P; ++q) reach[h * P + q] = 1.0;
        continue;
      }
      const int par = t.parent[h];
      double pr = 1.0;
      int slot = -1;
      if (t.kind[par] == kDecisionNode) {
        const int i = t.info[par];
        double sigma[kMaxPolicyRow];
        regret_match_row(regrets + i * A, sigma, t.nact[i]);
        pr = sigma[t.aidx[h]];
        slot = t.actor[par];
      }
      for (int q = 0; q < P; ++q) {
        const double r = reach[par * P + q];
        reach[h * P + q] = (q == slot) ? r * pr : r;
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < t.I; i += nt) {
    const int pl = t.info_player[i], n = t.nact[i];
    double sigma[kMaxPolicyRow];
    regret_match_row(regrets + i * A, sigma, n);
    for (int m = t.mem_off[i]; m < t.mem_off[i + 1]; ++m) {
      const int h = t.mem[m];
      double sum = 0.0;
      for (int q = 0; q < P; ++q) sum += reach[h * P + q];
      if (sum == 0.0) continue;  // external_sampling_mccfr.cc:203-205
      const double own = reach[h * P + pl];
      for (int a = 0; a < n; ++a) {
        const double term = own * sigma[a];
        cum[i * A + a] += weight == 1.0 ? term : weight * term;
      }
    }
  }
}

// Adds a mini-batch's deltas to the tables and leaves the delta tables ZERO: the next sample into them needs no
// memset (a fill launch is ~6 us of a 43 us mini-batch step).  use_policy = 0: AverageType::kFull, the traversals'
// sampled average-policy terms are dropped (external_sampling_mccfr.cc:177).
__global__ void k_fold_deltas(double* regrets, double* cum, double* dreg, double* dpol, int n, int use_policy) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  regrets[k] += dreg[k];
  if (use_policy) cum[k] += dpol[k];
  dreg[k] = 0.0;
  dpol[k] = 0.0;
}

__global__ void k_fill(double* p, double v, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) p[k] = v;
}

// ---------------------------------------------------------------------------
// InformationStateString of the acting player, from the packed state words
// (kuhn_poker.cc:109-166, leduc_poker.cc:198-239).
// ---------------------------------------------------------------------------
std::string kuhn_key(const Kuhn::Params& p, uint64_t word, int player) {
  Kuhn::State s{word};
  std::string r = Kuhn::len(s) > player ? std::to_string(Kuhn::card(s, player)) : std::string();  // not dealt yet: ""
  const int n = Kuhn::nact(p, s);
  for (int j = 0; j < n; ++j) r.push_back(((Kuhn::bets(s) >> j) & 1u) ? 'b' : 'p');
  return r;
}

// The part LeducObserver::StringFrom writes for both recall types (leduc_poker.cc:198-226).  money_ is
// kStartingMoney - ante_ while the hand runs; at the end the pot has been paid out (pot_ = 0,
// money = 100 + returns).
template <class L>
std::string leduc_observer_prefix(const LeducParams& p, const typename L::State& s, int player) {
  const bool term = L::terminal(p, s);
  double ret[kMaxPlayers] = {0};
  if (term) L::returns(p, s, ret);
  const int hole = L::priv(s, player);  // kInvalidCard = -10000 before the deal (leduc_poker.h:63)
  std::string r = "[Observer: " + std::to_string(player) + "][Private: " +
                  std::to_string(hole == L::kNone ? -10000 : hole) + "]";
  r += "[Round " + std::to_string(s.round) + "][Player: " + std::to_string(s.cur) + "][Pot: " +
       std::to_string(term ? 0 : s.pot) + "][Money: ";
  for (int q = 0; q < p.players; ++q) {
    char num[32];
    snprintf(num, sizeof num, "%g", term ? 100.0 + ret[q] : 100.0 - L::ante(s, q));
    if (q) r += " ";
    r += num;
  }
  r += "]";
  if (s.pub != L::kNone) r += "[Public: " + std::to_string(s.pub) + "]";
  return r;
}
template <class L>
std::string leduc_key_of(const LeducParams& p, const typename L::State& s, int player) {
  std::string r = leduc_observer_prefix<L>(p, s, player);
  for (int round = 0; round < 2; ++round) {
    r += round == 0 ? "[Round1: " : "][Round2: ";
    for (int k = 0; k < L::seqlen(s, round); ++k) {
      if (k) r += " ";
      r += std::to_string(static_cast<unsigned>((L::seq(s, round) >> (2 * k)) & 3u));
    }
  }
  r += "]";
  return r;
}
std::string leduc_key(const LeducParams& p, uint64_t w0, uint64_t w1, int player) {
  return leduc_key_of<Leduc>(p, Leduc::unpack(w0, w1), player);
}
// the state of either record from its plane words (2 or 5)
std::string leduc_key_words(const GameSpec& spec, const uint64_t* w, int player) {
  if (spec.leduc_big) return leduc_key_of<LeducBig>(spec.leduc, LeducBig::unpack5(w[0], w[1], w[2], w[3], w[4]), player);
  return leduc_key(spec.leduc, w[0], w[1], player);
}

template <class T> struct TypeTag { using type = T; };

template <class T>
int upload(const std::vector<T>& v, T** d, hipStream_t stream) {
  const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  OSG_HIP(hipMalloc(reinterpret_cast<void**>(d), bytes));
  if (!v.empty()) {
    OSG_HIP(hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    // The callers pass vectors that die when they return, and a large copy from pageable memory may still be reading
    // the host buffer after the call (seen once as a GPU fault at a host address with an 87 MB vector): wait.
    if (v.size() * sizeof(T) > (64u << 10)) OSG_HIP(hipStreamSynchronize(stream));
  }
  return OSG_OK;
}

}  // namespace

struct osg_cfr {
  osg_ctx* ctx = nullptr;
  GameSpec spec;
  osg_cfr_cfg cfg{};
  int P = 0, H = 0, I = 0, A = 0, D = 0;
  int64_t n_chance = 0, n_decision = 0, n_terminal = 0;
  int max_level_width = 0;
  int average_type = 0;  // ES-MCCFR AverageType: 0 kSimple, 1 kFull (external_sampling_mccfr.h:48)
  int iteration = 0;
  const char* last_kernel = "";  // the kernel family the last iterate / sample call launched (osg_cfr_last_kernel)
  // host tree
  std::vector<int32_t> level_off, parent, first_child, info, mem_off, mem, nact, legal;
  std::vector<uint8_t> kind, nchild, aidx;
  std::vector<int32_t> edge_action;  // [H] the action (or chance outcome) on the edge from the parent, -1 at the root
  std::vector<int8_t> actor, info_player;
  std::vector<double> edge_prob, term_ret;
  std::vector<std::string> keys;
  // device tree
  int32_t *d_level_off = nullptr, *d_parent = nullptr, *d_first_child = nullptr, *d_info = nullptr,
          *d_mem_off = nullptr, *d_mem = nullptr, *d_nact = nullptr;
  uint8_t *d_kind = nullptr, *d_nchild = nullptr, *d_aidx = nullptr;
  int8_t *d_actor = nullptr, *d_info_player = nullptr;
  double *d_edge_prob = nullptr, *d_term_ret = nullptr;
  // device tables and work arrays
  double* d_tables = nullptr;  // regrets | cum | cur | dreg | dpol, each [I, A]
  double* d_reach = nullptr;   // [H, P+1]
  double* d_value = nullptr;   // [H, P]
  bool lds_resident = false;
  size_t lds_bytes = 0;
  // small-tree kernel: root paths of the decision histories (member order)
  std::vector<int32_t> path_off, path;
  int32_t *d_path_off = nullptr, *d_path = nullptr;
  bool small_tree = false;   // the all-in-LDS variant fits
  bool path_kernel = false;  // the path-based kernel (k_cfr_small) is usable at all
  int max_path_decisions = 0;  // the most decision entries any member's root path holds
  int first_decision_level = 0;  // the first level with a decision history (SmallTree::L0)
  size_t small_lds_bytes = 0;
  std::vector<int32_t> meta32, info_player32;
  int32_t *d_meta32 = nullptr, *d_info_player32 = nullptr, *d_skip = nullptr;
  double* d_node_delta = nullptr;  // dreg [M, A] | dpol [M, A]
  double* d_spare_delta[2] = {nullptr, nullptr};  // osg_mccfr_spare_delta_buffer: [2, I, A] each, allocated on request
  bool delta_clean[3] = {false, false, false};    // the internal / spare delta buffers are all zero (the last fold left them so)
  // one workgroup per deal subtree (k_cfr_split)
  bool split_ok = false, split_br_ok = false;
  int split_G = 0, split_L = 0, split_NL = 0, split_NM = 0, split_NI = 0, split_threads = 0;
  size_t split_lds_bytes = 0;
  int32_t *d_split_nloc = nullptr, *d_split_desc = nullptr, *d_split_fc = nullptr, *d_split_row = nullptr,
          *d_split_glob = nullptr, *d_split_mem_m = nullptr, *d_split_mem_hloc = nullptr, *d_split_info = nullptr;
  double* d_split_terms = nullptr;
  unsigned int* d_split_bar = nullptr;
  // one cooperative launch, a workgroup per deal subtree of any size (k_cfr_sub)
  bool sub_ok = false;
  int sub_G = 0, sub_L = 0, sub_NL = 0, sub_K = 0, sub_grid = 0;
  size_t sub_lds_bytes = 0;
  int sub_ND = 0, sub_PL = 0;
  int32_t *d_sub_ndec = nullptr, *d_sub_dec_row = nullptr, *d_sub_rec = nullptr;
  int32_t *d_sub_nloc = nullptr, *d_sub_desc = nullptr, *d_sub_fc = nullptr, *d_sub_aux = nullptr, *d_sub_mem_off = nullptr,
          *d_sub_info_off = nullptr, *d_sub_info_list = nullptr;
  unsigned int* d_sub_bar = nullptr;
  // forest form of k_cfr_sub (SubTree's comment): the kernel's own skip words, piece roots, upper members
  bool sub_forest = false;
  int sub_NR = 0, sub_G0 = 0;   // G0: the deal subtrees; sub_G: the bins they (or their pieces) were packed into
  unsigned long long *d_sub_stamps = nullptr, *d_mccfr_stamps = nullptr;   // profiling stamps (per solver: never shared across contexts / devices)
  double *d_sub_recbuf = nullptr, *d_sub_chance_prob = nullptr, *d_sub_term_val = nullptr;
  int32_t *d_sub_dec_off = nullptr, *d_sub_fold_info = nullptr, *d_sub_fold_off = nullptr;
  int sub_NCP = 0;
  bool sub_keep_rows = false;
  int32_t *d_sub_nroot = nullptr, *d_sub_root_loc = nullptr, *d_sub_root_idx = nullptr, *d_sub_upper_rec = nullptr;
  double* d_sub_root_value = nullptr;
  unsigned int* h_sub_err = nullptr;   // pinned: raised by the kernel when a grid barrier times out
  // policy evaluation (k_policy_eval)
  std::vector<int32_t> info_level, mem_index;
  bool eval_ok = true;  // every infostate's members sit on one tree level
  int32_t *d_info_level = nullptr, *d_mem_index = nullptr, *d_best = nullptr;
  double *d_eval = nullptr;  // value [H,P] | brv [H] | cf [M] | out [2P] | policy [I,A]
  double* d_eval_ev = nullptr;   // [H, P]: the expected returns of the large-tree evaluation (allocated on first use)
  std::vector<int32_t> eval_level_off;   // [D + 1] the infostates of level l: d_eval_level_info[eval_level_off[l] ...)
  int32_t* d_eval_level_info = nullptr;
  // the evaluation as independent jobs over the device (k_eval_jobs)
  bool jobs_ok = false;
  int jobs_J = 0, jobs_L = 0, jobs_G = 0, jobs_NT = 0, jobs_threads = 0;
  size_t jobs_lds_bytes = 0;
  int32_t *d_jobs_job = nullptr, *d_jobs_level = nullptr, *d_jobs_desc = nullptr, *d_jobs_fc = nullptr, *d_jobs_row = nullptr,
          *d_jobs_glob = nullptr, *d_jobs_info = nullptr, *d_jobs_mem = nullptr;
  double* d_jobs_deal = nullptr;
  unsigned int* d_jobs_ticket = nullptr;
  double* h_eval_out = nullptr;  // pinned, mapped: the evaluation kernels write their [2 P] results here
  // LDS-resident MCCFR traversal (k_mccfr_resident)
  bool resident_ok = false;
  size_t resident_lds_bytes = 0;
  int n_uret = 0, n_uprob = 0, num_cus = 0;
  uint64_t* d_rec = nullptr;
  double *d_uret = nullptr, *d_uprob = nullptr;

  Tree tree() const {
    Tree t;
    t.H = H; t.I = I; t.A = A; t.P = P; t.D = D;
    t.level_off = d_level_off; t.parent = d_parent; t.first_child = d_first_child; t.kind = d_kind;
    t.nchild = d_nchild; t.aidx = d_aidx; t.actor = d_actor; t.info = d_info; t.edge_prob = d_edge_prob;
    t.term_ret = d_term_ret; t.mem_off = d_mem_off; t.mem = d_mem; t.nact = d_nact; t.info_player = d_info_player;
    return t;
  }
  // Replicas: B independent solvers of the same tree (tables [B][5][I, A]); `selected` is the one the
  // table accessors / evaluation look at.
  int B = 1, selected = 0;
  size_t replica_stride() const { return 5 * static_cast<size_t>(I) * A; }
  double* replica_base(int r) const { return d_tables + static_cast<size_t>(r) * replica_stride(); }
  double* regrets() const { return replica_base(selected); }
  double* cum() const { return replica_base(selected) + static_cast<size_t>(I) * A; }
  double* cur() const { return replica_base(selected) + 2 * static_cast<size_t>(I) * A; }
  double* dreg() const { return replica_base(selected) + 3 * static_cast<size_t>(I) * A; }
  double* dpol() const { return replica_base(selected) + 4 * static_cast<size_t>(I) * A; }
};

static EvalJobs eval_jobs_of(const osg_cfr* s) {
  EvalJobs ej;
  ej.J = s->jobs_J; ej.L = s->jobs_L; ej.G = s->jobs_G; ej.NT = s->jobs_NT;
  ej.job = s->d_jobs_job; ej.level_off = s->d_jobs_level; ej.node_desc = s->d_jobs_desc; ej.node_fc = s->d_jobs_fc;
  ej.node_row = s->d_jobs_row; ej.node_glob = s->d_jobs_glob; ej.info_ent = s->d_jobs_info; ej.mem_ent = s->d_jobs_mem;
  ej.deal = s->d_jobs_deal; ej.ticket = s->d_jobs_ticket;
  return ej;
}

// Trees beyond one workgroup's reach (and too large for the jobs) are evaluated with a launch per level and phase
// (k_geval_*); OSG_EVAL_GRID=1 forces that form, 0 forbids it (the tests compare the three).
static bool eval_takes_the_grid(const osg_cfr* s) {
  const char* e = getenv("OSG_EVAL_GRID");
  if (e && e[0] == '0') return false;
  if (e && e[0] == '1') return true;
  return s->H > 65536;
}
static int launch_grid_eval(const osg_cfr* s, const EvalArrays& ea, const double* src, bool from_cum, double* d_pol, bool only_br) {
  hipStream_t st = s->ctx->stream;
  const Tree t = s->tree();
  const double* pol = src;
  if (from_cum) {
    k_geval_policy<<<dim3((s->I + 255) / 256), dim3(256), 0, st>>>(t, src, d_pol);
    pol = d_pol;
  }
  auto width = [&](int l) { return static_cast<unsigned>((s->level_off[l + 1] - s->level_off[l] + 255) / 256); };
  // (the expected returns ride in the best responses' sweep, in an [H, P] array of their own: allocated on first use)
  double* d_ev = nullptr;
  if (!only_br) {
    osg_cfr* mut = const_cast<osg_cfr*>(s);
    if (!mut->d_eval_ev)
      OSG_HIP(hipMalloc(reinterpret_cast<void**>(&mut->d_eval_ev), sizeof(double) * static_cast<size_t>(s->H) * s->P));
    d_ev = mut->d_eval_ev;
  }
  const unsigned mblocks = static_cast<unsigned>((s->mem.size() + 255) / 256), iblocks = static_cast<unsigned>((s->I + 255) / 256);
  // every player's best response in one bottom-up sweep: the responder values take the expected-value array over (its
  // sweep is done: the root's values are in out); the argmax launch only where the level holds infostates
  osg_cfr* ms = const_cast<osg_cfr*>(s);
  if (ms->eval_level_off.empty()) {   // the infostates of every level, once per solver
    ms->eval_level_off.assign(static_cast<size_t>(s->D) + 1, 0);
    for (int i = 0; i < s->I; ++i)
      if (s->info_level[i] >= 0 && s->info_level[i] < s->D) ++ms->eval_level_off[s->info_level[i] + 1];
    for (int l = 0; l < s->D; ++l) ms->eval_level_off[l + 1] += ms->eval_level_off[l];
    std::vector<int32_t> list(static_cast<size_t>(std::max(ms->eval_level_off[s->D], 1)), 0), at(ms->eval_level_off.begin(), ms->eval_level_off.end() - 1);
    for (int i = 0; i < s->I; ++i)
      if (s->info_level[i] >= 0 && s->info_level[i] < s->D) list[at[s->info_level[i]]++] = i;
    if (int rc = upload(list, &ms->d_eval_level_info, st)) return rc;
  }
  (void)iblocks;
  k_geval_cf<<<dim3(std::max(1u, mblocks)), dim3(256), 0, st>>>(t, ea, pol);
  for (int l = s->D - 1; l >= 0; --l) {
    const int n_infos = ms->eval_level_off[l + 1] - ms->eval_level_off[l];
    if (n_infos > 0)
      k_geval_best<<<dim3(static_cast<unsigned>((n_infos + 3) / 4)), dim3(256), 0, st>>>(t, ea, ms->d_eval_level_info + ms->eval_level_off[l], n_infos);
    k_geval_brv<<<dim3(width(l)), dim3(256), 0, st>>>(t, ea, pol, l, d_ev);
  }
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}
// OSG_EVAL_JOBS=0 keeps the one-workgroup evaluation (k_policy_eval) for trees that could take the jobs: the tests
// compare the two.
static bool OSG_EVAL_JOBS_ENABLED() {
  const char* e = getenv("OSG_EVAL_JOBS");
  return !(e && e[0] == '0');
}
namespace {

// Expands the game tree breadth-first with the batched State kernels.
int build_tree(osg_cfr* s, const char* game_string) {
  osg_ctx* ctx = s->ctx;
  const osg_game_desc& d = s->spec.desc;
  const int P = d.num_players, W = d.mask_words, C = std::max(d.max_chance_outcomes, 1);
  const int words = d.state_words;
  const int64_t kMaxHistories = 1 << 24;
  s->P = P;
  s->A = d.num_distinct_actions;
  if (s->A > 255) return set_error(OSG_ERR_UNSUPPORTED, "osg_cfr_create: more than 255 distinct actions");

  std::unordered_map<std::string, int> key_to_id;
  osg_batch* level = nullptr;
  int rc = osg_batch_create(ctx, game_string, 1, &level);
  if (rc) return rc;
  s->level_off.push_back(0);
  // per-node data pending for the level being expanded: parent + edge data were written when the
  // parent was expanded; this loop fills in what depends on the state itself.
  s->parent.push_back(-1);
  s->aidx.push_back(0);
  s->edge_action.push_back(-1);
  s->edge_prob.push_back(0.0);
  int64_t level_begin = 0;
  while (level) {
    const int64_t n = osg_batch_size(level);
    std::vector<uint32_t> mask(static_cast<size_t>(n) * W);
    std::vector<int8_t> cur(n);
    std::vector<uint8_t> term(n);
    std::vector<double> rets(static_cast<size_t>(n) * P), probs(static_cast<size_t>(n) * C);
    std::vector<uint64_t> raw(static_cast<size_t>(n) * words);
    if ((rc = osg_legal_mask(level, mask.data(), 1)) || (rc = osg_status_query(level, cur.data(), term.data(), rets.data(), 1)) ||
        (d.max_chance_outcomes > 0 && (rc = osg_chance_probs(level, probs.data(), 1))) ||
        (rc = osg_batch_download(level, raw.data()))) {
      osg_batch_destroy(level);
      return rc;
    }
    std::vector<int64_t> gather;
    std::vector<int32_t> actions;
    const int64_t next_begin = level_begin + n;
    for (int64_t i = 0; i < n; ++i) {
      const int64_t h = level_begin + i;
      (void)h;
      s->actor.push_back(cur[i] >= 0 ? cur[i] : -1);
      if (term[i]) {
        s->kind.push_back(kTerminalNode);
        s->nchild.push_back(0);
        s->first_child.push_back(0);
        s->info.push_back(-1);
        for (int q = 0; q < P; ++q) s->term_ret.push_back(rets[i * P + q]);
        ++s->n_terminal;
        continue;
      }
      for (int q = 0; q < P; ++q) s->term_ret.push_back(0.0);
      const bool chance = cur[i] == kChancePlayer;
      s->kind.push_back(chance ? kChanceNode : kDecisionNode);
      s->first_child.push_back(static_cast<int32_t>(next_begin + static_cast<int64_t>(gather.size())));
      std::vector<int32_t> acts;
      for (int a = 0; a < 32 * W; ++a)
        if ((mask[i * W + (a >> 5)] >> (a & 31)) & 1u) acts.push_back(a);
      s->nchild.push_back(static_cast<uint8_t>(acts.size()));
      for (size_t k = 0; k < acts.size(); ++k) {
        gather.push_back(i);
        actions.push_back(acts[k]);
        s->parent.push_back(static_cast<int32_t>(h));
        s->aidx.push_back(static_cast<uint8_t>(k));
        s->edge_action.push_back(acts[k]);
        s->edge_prob.push_back(chance ? probs[i * C + acts[k]] : 0.0);
      }
      if (chance) {
        s->info.push_back(-1);
        ++s->n_chance;
        continue;
      }
      ++s->n_decision;
      std::string key;
      if (d.game_kind == kKuhn) key = kuhn_key(s->spec.kuhn, raw[i], cur[i]);
      else key = leduc_key(s->spec.leduc, raw[i], raw[n + i], cur[i]);
      auto it = key_to_id.find(key);
      int id;
      if (it == key_to_id.end()) {  // InitializeInfostateNodes (cfr.cc:234-261)
        id = static_cast<int>(s->keys.size());
        key_to_id.emplace(key, id);
        s->keys.push_back(key);
        s->nact.push_back(static_cast<int32_t>(acts.size()));
        s->info_player.push_back(cur[i]);
        for (int a = 0; a < s->A; ++a) s->legal.push_back(a < static_cast<int>(acts.size()) ? acts[a] : -1);
      } else {
        id = it->second;
        if (s->nact[id] != static_cast<int32_t>(acts.size())) {
          osg_batch_destroy(level);
          return set_error(OSG_ERR_INVALID, "infostate with inconsistent legal actions: " + key);
        }
      }
      s->info.push_back(id);
    }
    s->max_level_width = std::max<int>(s->max_level_width, static_cast<int>(n));
    s->level_off.push_back(static_cast<int32_t>(next_begin));
    level_begin = next_begin;
    osg_batch* next = nullptr;
    if (!gather.empty()) {
      if (next_begin + static_cast<int64_t>(gather.size()) > kMaxHistories) {
        osg_batch_destroy(level);
        return set_error(OSG_ERR_UNSUPPORTED, "osg_cfr_create: game tree exceeds 2^24 histories");
      }
      if ((rc = osg_batch_create(ctx, game_string, static_cast<int64_t>(gather.size()), &next)) ||
          (rc = osg_batch_gather(next, level, gather.data(), 1)) ||
          (rc = osg_apply(next, actions.data(), 1, nullptr))) {
        osg_batch_destroy(level);
        if (next) osg_batch_destroy(next);
        return rc;
      }
    }
    osg_batch_destroy(level);
    level = next;
  }
  s->H = static_cast<int>(level_begin);
  s->D = static_cast<int>(s->level_off.size()) - 1;
  s->I = static_cast<int>(s->keys.size());
  // Narrow the table width to the widest decision node.
  int amax = 1;
  for (int v : s->nact) amax = std::max(amax, v);
  if (amax != s->A) {
    std::vector<int32_t> legal(static_cast<size_t>(s->I) * amax);
    for (int i = 0; i < s->I; ++i)
      for (int a = 0; a < amax; ++a) legal[i * amax + a] = s->legal[i * s->A + a];
    s->legal.swap(legal);
    s->A = amax;
  }
  // Member histories of every infostate in the reference's DFS visiting order.
  std::vector<int32_t> order;
  order.reserve(s->H);
  std::vector<int32_t> stack{0};
  while (!stack.empty()) {
    const int h = stack.back();
    stack.pop_back();
    order.push_back(h);
    for (int a = s->nchild[h] - 1; a >= 0; --a) stack.push_back(s->first_child[h] + a);
  }
  std::vector<std::vector<int32_t>> members(s->I);
  for (int h : order)
    if (s->kind[h] == kDecisionNode) members[s->info[h]].push_back(h);
  s->mem_off.push_back(0);
  for (int i = 0; i < s->I; ++i) {
    s->mem.insert(s->mem.end(), members[i].begin(), members[i].end());
    s->mem_off.push_back(static_cast<int32_t>(s->mem.size()));
  }
  {  // tree level of every history; infostates must not span levels for the level-synchronous best response
    std::vector<int32_t> level_of(s->H, 0);
    for (int l = 0; l < s->D; ++l)
      for (int h = s->level_off[l]; h < s->level_off[l + 1]; ++h) level_of[h] = l;
    s->info_level.assign(s->I, -1);
    s->mem_index.assign(s->H, -1);
    for (int i = 0; i < s->I; ++i)
      for (int m = s->mem_off[i]; m < s->mem_off[i + 1]; ++m) {
        const int h = s->mem[m];
        s->mem_index[h] = m;
        if (s->info_level[i] < 0) s->info_level[i] = level_of[h];
        else if (s->info_level[i] != level_of[h]) s->eval_ok = false;
      }
  }
  s->first_decision_level = 0;
  for (int l = 0; l < s->D; ++l) {   // the first level that holds a decision history
    bool any = false;
    for (int h = s->level_off[l]; h < s->level_off[l + 1]; ++h) any |= s->kind[h] == kDecisionNode;
    if (any) { s->first_decision_level = l; break; }
  }
  // Root path of every decision history, root-to-leaf: one entry per ancestor edge =
  // (reach slot of the ancestor's actor, where to read the edge probability).
  s->path_off.push_back(0);
  for (int32_t h : s->mem) {
    std::vector<int32_t> rev;
    for (int32_t v = h; s->parent[v] >= 0; v = s->parent[v]) {
      const int32_t par = s->parent[v];
      const bool chance = s->kind[par] == kChanceNode;
      const int slot = chance ? s->P : s->actor[par];
      const int32_t idx = chance ? v : s->info[par] * s->A + s->aidx[v];
      rev.push_back((slot << 24) | ((chance ? 1 : 0) << 23) | idx);
    }
    int decisions = 0;
    for (int32_t code : rev) decisions += ((code >> 23) & 1) ? 0 : 1;
    s->max_path_decisions = std::max(s->max_path_decisions, decisions);
    s->path.insert(s->path.end(), rev.rbegin(), rev.rend());
    s->path_off.push_back(static_cast<int32_t>(s->path.size()));
  }
  return OSG_OK;
}

int init_tables(osg_cfr* s) {
  const int IA = s->I * s->A;
  hipStream_t st = s->ctx->stream;
  const size_t stride = s->replica_stride();
  std::vector<double> host(static_cast<size_t>(s->B) * stride, 0.0);
  for (int r = 0; r < s->B; ++r) {
    double* regrets = host.data() + r * stride;
    double* cum = regrets + IA;
    double* cur = cum + IA;
    for (int i = 0; i < s->I; ++i) {
      const int n = s->nact[i];
      double sum_pos = 0.0;
      for (int a = 0; a < n; ++a) {
        double init = s->cfg.solver >= 1 ? kMccfrInit : 0.0;
        if (s->cfg.random_initial_regrets) {
          // CFRInfoStateValues(la, rng, kRandomInitialRegretsMagnitude = 0.001) (cfr.h:52-61, cfr.cc:31,249-252):
          // regret = magnitude * U[0, 1); the reference draws from mt19937 + absl::Uniform (stream
          // unpinned), here from the counter stream (seed, replica, infostate * Amax + action).
          Rng rng(s->cfg.seed, static_cast<uint64_t>(s->cfg.replica_offset + r), static_cast<uint64_t>(i) * s->A + a);
          init = 0.001 * rng.unit();
        }
        regrets[i * s->A + a] = init;
        cum[i * s->A + a] = s->cfg.solver >= 1 ? kMccfrInit : 0.0;
        if (init > 0) sum_pos += init;
      }
      for (int a = 0; a < n; ++a) {  // ApplyRegretMatching on the fresh row (uniform when all regrets are 0)
        const double rg = regrets[i * s->A + a];
        cur[i * s->A + a] = (s->cfg.random_initial_regrets && sum_pos > 0) ? (rg > 0 ? rg / sum_pos : 0.0) : 1.0 / n;
      }
    }
  }
  OSG_HIP(hipMemcpyAsync(s->d_tables, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice, st));
  OSG_HIP(hipStreamSynchronize(st));
  s->iteration = 0;
  return OSG_OK;
}

// Packs the tree for k_mccfr_resident (8 bytes per history + the distinct return vectors and chance
// probabilities) and decides whether it fits one workgroup's LDS next to three [I, A] tables.
int build_resident_tree(osg_cfr* s) {
  s->resident_ok = false;
  if ((s->cfg.solver != 1 && s->cfg.solver != 2) || s->A < 1 || s->A > kMaxA) return OSG_OK;
  if (s->H >= (1 << 24) || s->I >= (1 << 20)) return OSG_OK;
  std::vector<uint64_t> rec(s->H);
  std::vector<double> uret, uprob;
  std::unordered_map<std::string, uint32_t> ret_id;
  std::unordered_map<uint64_t, uint32_t> prob_id;
  const int P = s->P;
  // traverser frames one path can hold: decision nodes of one player from the root down
  std::vector<uint8_t> own(static_cast<size_t>(s->H) * P, 0);
  int max_frames = 0;
  for (int h = 0; h < s->H; ++h) {
    uint32_t x = s->kind[h] | (static_cast<uint32_t>(s->nchild[h]) << 2) |
                 (static_cast<uint32_t>(s->actor[h] + 1) << 8);
    uint32_t y = 0;
    if (s->nchild[h] > 63 || s->actor[h] + 1 > 15) return OSG_OK;
    if (s->kind[h] == kDecisionNode) x |= static_cast<uint32_t>(s->info[h]) << 12;
    if (s->kind[h] == kTerminalNode) {
      std::string key(reinterpret_cast<const char*>(&s->term_ret[static_cast<size_t>(h) * P]), sizeof(double) * P);
      auto it = ret_id.find(key);
      if (it == ret_id.end()) {
        it = ret_id.emplace(key, static_cast<uint32_t>(ret_id.size())).first;
        for (int p = 0; p < P; ++p) uret.push_back(s->term_ret[static_cast<size_t>(h) * P + p]);
      }
      y = it->second;
    } else {
      y = static_cast<uint32_t>(s->first_child[h]);
    }
    if (h > 0 && s->kind[s->parent[h]] == kChanceNode) {
      uint64_t bits;
      memcpy(&bits, &s->edge_prob[h], sizeof bits);
      auto it = prob_id.find(bits);
      if (it == prob_id.end()) {
        if (prob_id.size() >= 256) return OSG_OK;
        it = prob_id.emplace(bits, static_cast<uint32_t>(prob_id.size())).first;
        uprob.push_back(s->edge_prob[h]);
      }
      y |= it->second << 24;
    }
    rec[h] = static_cast<uint64_t>(x) | (static_cast<uint64_t>(y) << 32);
    if (h > 0) {
      const int par = s->parent[h];
      for (int p = 0; p < P; ++p) {
        int d = own[static_cast<size_t>(par) * P + p] + (s->kind[par] == kDecisionNode && s->actor[par] == p ? 1 : 0);
        if (d > 255) return OSG_OK;
        own[static_cast<size_t>(h) * P + p] = static_cast<uint8_t>(d);
        max_frames = std::max(max_frames, d);
      }
    }
  }
  if (max_frames > kMaxFrames) return OSG_OK;
  // Chance nodes whose outcomes all have the same probability (every chance node of kuhn and leduc: 1 / cards left)
  // say so in the field decision nodes use for their infostate id: index of that probability + 1, else 0.  The
  // traversal then samples without reading the children's records.
  for (int h = 0; h < s->H; ++h) {
    if (s->kind[h] != kChanceNode || s->nchild[h] == 0) continue;
    const uint32_t id0 = static_cast<uint32_t>(rec[s->first_child[h]] >> 56);
    bool same = true;
    for (int c = 1; c < s->nchild[h]; ++c) same &= static_cast<uint32_t>(rec[s->first_child[h] + c] >> 56) == id0;
    if (same) rec[h] |= static_cast<uint64_t>(id0 + 1) << 12;
  }
  if (uprob.empty()) uprob.push_back(1.0);
  s->n_uret = static_cast<int>(uret.size() / std::max(P, 1));
  s->n_uprob = static_cast<int>(uprob.size());
  const size_t IA = static_cast<size_t>(s->I) * s->A;
  s->resident_lds_bytes = sizeof(double) * (3 * IA + uret.size() + uprob.size()) + sizeof(uint64_t) * s->H;
  hipDeviceProp_t prop;
  OSG_HIP(hipGetDeviceProperties(&prop, s->ctx->device));
  s->num_cus = prop.multiProcessorCount;
  if (s->resident_lds_bytes > static_cast<size_t>(prop.sharedMemPerBlockOptin ? prop.sharedMemPerBlockOptin
                                                                              : prop.sharedMemPerBlock))
    return OSG_OK;
  hipStream_t st = s->ctx->stream;
  int rc;
  if ((rc = upload(rec, &s->d_rec, st)) || (rc = upload(uret, &s->d_uret, st)) || (rc = upload(uprob, &s->d_uprob, st)))
    return rc;
  const void* variants[] = {reinterpret_cast<const void*>(&k_mccfr_resident_flat<1>),
                            reinterpret_cast<const void*>(&k_mccfr_resident_flat<2>),
                            reinterpret_cast<const void*>(&k_mccfr_resident_flat<3>),
                            reinterpret_cast<const void*>(&k_mccfr_resident_flat<4>),
                            reinterpret_cast<const void*>(&k_os_mccfr_resident<1>),
                            reinterpret_cast<const void*>(&k_os_mccfr_resident<2>),
                            reinterpret_cast<const void*>(&k_os_mccfr_resident<3>),
                            reinterpret_cast<const void*>(&k_os_mccfr_resident<4>)};
  if (raise_lds_cap(variants[(s->cfg.solver == 2 ? 4 : 0) + s->A - 1], static_cast<int>(s->resident_lds_bytes)) != hipSuccess) {
    (void)hipGetLastError();
    return OSG_OK;
  }
  const void* split_variants[] = {nullptr, nullptr, reinterpret_cast<const void*>(&k_mccfr_resident<2, 1>),
                                  reinterpret_cast<const void*>(&k_mccfr_resident<2, 2>),
                                  reinterpret_cast<const void*>(&k_mccfr_resident<3, 1>),
                                  reinterpret_cast<const void*>(&k_mccfr_resident<3, 2>),
                                  reinterpret_cast<const void*>(&k_mccfr_resident<4, 1>),
                                  reinterpret_cast<const void*>(&k_mccfr_resident<4, 2>)};
  for (int level = 0; level < 2; ++level)
    if (s->cfg.solver != 2 && s->A >= 2 &&
        raise_lds_cap(split_variants[2 * (s->A - 1) + level], static_cast<int>(s->resident_lds_bytes)) != hipSuccess) {
      (void)hipGetLastError();
      return OSG_OK;
    }
  s->resident_ok = true;
  return OSG_OK;
}

// Cuts the tree below its leading chance levels into subtrees for k_cfr_split: one workgroup each, at most one
// per CU, every subtree small enough for one thread per history.
int build_split(osg_cfr* s) {
  s->split_ok = false;
  if (s->cfg.solver != 0 || s->B != 1 || !s->path_kernel || s->A > kSplitMaxA || s->P + 1 > kMaxPlayers + 1) return OSG_OK;
  if (s->H < 2000 || s->D >= 64) return OSG_OK;
  // the cut: the first level that holds a node which is not a chance node
  int L = 0;
  for (; L < s->D; ++L) {
    bool all_chance = true;
    for (int h = s->level_off[L]; h < s->level_off[L + 1]; ++h) all_chance &= s->kind[h] == kChanceNode;
    if (!all_chance) break;
  }
  if (L < 1 || L >= s->D - 1) return OSG_OK;
  const int G = s->level_off[L + 1] - s->level_off[L];
  hipDeviceProp_t prop;
  OSG_HIP(hipGetDeviceProperties(&prop, s->ctx->device));
  if (G < 8 || G > prop.multiProcessorCount) return OSG_OK;
  // a subtree's histories, level by level: the descendants of a level-L node are a contiguous range on every level
  std::vector<std::vector<int32_t>> hist(G);
  std::vector<int32_t> sub_of(s->H, -1), loc_of(s->H, -1), level_of(s->H, 0);
  for (int l = 0; l < s->D; ++l)
    for (int h = s->level_off[l]; h < s->level_off[l + 1]; ++h) level_of[h] = l;
  for (int g = 0; g < G; ++g) sub_of[s->level_off[L] + g] = g;
  for (int h = s->level_off[L]; h < s->H; ++h) {
    if (h >= s->level_off[L + 1]) sub_of[h] = sub_of[s->parent[h]];
    const int g = sub_of[h];
    loc_of[h] = static_cast<int32_t>(hist[g].size());
    hist[g].push_back(h);
  }
  int NL = 0;
  for (int g = 0; g < G; ++g) NL = std::max<int>(NL, static_cast<int>(hist[g].size()));
  if (NL > 1024) return OSG_OK;
  const int threads = std::max(64, (NL + 63) / 64 * 64);
  std::vector<std::vector<int32_t>> mem_m(G), infos(G);
  std::vector<int32_t> seen(s->I, -1);
  for (int i = 0; i < s->I; ++i)
    for (int m = s->mem_off[i]; m < s->mem_off[i + 1]; ++m) {
      const int h = s->mem[m];
      if (sub_of[h] < 0) return OSG_OK;  // a decision node above the cut
      int decisions = 0;
      for (int e = s->path_off[m]; e < s->path_off[m + 1]; ++e) decisions += ((s->path[e] >> 23) & 1) ? 0 : 1;
      if (decisions > kSplitOwnerPath) return OSG_OK;
      mem_m[sub_of[h]].push_back(m);
      if (seen[i] != sub_of[h]) {  // members of one infostate inside one subtree are adjacent in DFS order or not: check all
        bool have = false;
        for (int32_t x : infos[sub_of[h]]) have |= x == i;
        if (!have) infos[sub_of[h]].push_back(i);
        seen[i] = sub_of[h];
      }
    }
  int NM = 1, NI = 1;
  for (int g = 0; g < G; ++g) {
    NM = std::max<int>(NM, static_cast<int>(mem_m[g].size()));
    NI = std::max<int>(NI, static_cast<int>(infos[g].size()));
  }
  if (NM > threads || NI > threads) return OSG_OK;
  const size_t IA = static_cast<size_t>(s->I) * s->A;
  const size_t lds = sizeof(double) * (static_cast<size_t>(NL) * s->P + NL + 3 * IA) + 16;
  if (lds > 150 * 1024) return OSG_OK;
  std::vector<int32_t> nloc(G), desc(static_cast<size_t>(G) * NL, kTerminalNode), fc(static_cast<size_t>(G) * NL, 0),
      row(static_cast<size_t>(G) * NL, 0), glob(static_cast<size_t>(G) * NL, 0), mm(static_cast<size_t>(G) * NM, -1),
      mh(static_cast<size_t>(G) * NM, 0), il(static_cast<size_t>(G) * NI, -1);
  for (int g = 0; g < G; ++g) {
    nloc[g] = static_cast<int32_t>(hist[g].size());
    for (size_t j = 0; j < hist[g].size(); ++j) {
      const int h = hist[g][j];
      const size_t at = static_cast<size_t>(g) * NL + j;
      desc[at] = s->kind[h] | (s->nchild[h] << 2) | (level_of[h] << 10) | ((s->actor[h] + 1) << 16);
      fc[at] = s->kind[h] == kTerminalNode ? 0 : loc_of[s->first_child[h]];
      row[at] = s->kind[h] == kDecisionNode ? s->info[h] * s->A : 0;
      glob[at] = h;
    }
    for (size_t k = 0; k < mem_m[g].size(); ++k) {
      mm[static_cast<size_t>(g) * NM + k] = mem_m[g][k];
      mh[static_cast<size_t>(g) * NM + k] = loc_of[s->mem[mem_m[g][k]]];
    }
    for (size_t k = 0; k < infos[g].size(); ++k) il[static_cast<size_t>(g) * NI + k] = infos[g][k];
  }
  hipStream_t st = s->ctx->stream;
  int rc;
  if ((rc = upload(nloc, &s->d_split_nloc, st)) || (rc = upload(desc, &s->d_split_desc, st)) ||
      (rc = upload(fc, &s->d_split_fc, st)) || (rc = upload(row, &s->d_split_row, st)) ||
      (rc = upload(glob, &s->d_split_glob, st)) || (rc = upload(mm, &s->d_split_mem_m, st)) ||
      (rc = upload(mh, &s->d_split_mem_hloc, st)) || (rc = upload(il, &s->d_split_info, st)))
    return rc;
  const size_t M = s->mem.size();
  OSG_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_split_terms), sizeof(double) * 2 * kSplitRec * std::max<size_t>(M, 1)));
  OSG_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_split_bar), sizeof(unsigned int) * 4));
  OSG_HIP(hipMemsetAsync(s->d_split_bar, 0, sizeof(unsigned int) * 4, st));
  OSG_HIP(hipMemsetAsync(s->d_split_terms, 0, sizeof(double) * 2 * kSplitRec * std::max<size_t>(M, 1), st));
  if (raise_lds_cap(split_kernel(s->P, false, threads), static_cast<int>(lds)) != hipSuccess ||
      raise_lds_cap(split_kernel(s->P, false, threads, s->A, true), static_cast<int>(lds)) != hipSuccess) {
    (void)hipGetLastError();
    return OSG_OK;
  }
  // the CFR-BR pass set keeps one more [I, A] array (the effective policy)
  s->split_br_ok = lds + sizeof(double) * IA <= 158 * 1024;
  if (s->split_br_ok && (raise_lds_cap(split_kernel(s->P, true, threads), static_cast<int>(lds + sizeof(double) * IA)) != hipSuccess ||
                         raise_lds_cap(split_kernel(s->P, true, threads, s->A, true), static_cast<int>(lds + sizeof(double) * IA)) != hipSuccess)) {
    (void)hipGetLastError();
    s->split_br_ok = false;
  }
  s->split_G = G; s->split_L = L; s->split_NL = NL; s->split_NM = NM; s->split_NI = NI; s->split_threads = threads;
  s->split_lds_bytes = lds;
  s->split_ok = true;
  return OSG_OK;
}


// The jobs of k_eval_jobs: the cut of build_split (the first level with a node that is not a chance node); one
// expected-returns job per subtree; for every responder r the subtrees grouped into the connected components of "holds a
// member history of the same infostate of r", one best-response job per component.  Trees of another shape, or with a
// component that does not fit a workgroup's LDS, keep the one-workgroup kernel.
int build_eval_jobs(osg_cfr* s) {
  s->jobs_ok = false;
  if (!s->eval_ok || s->H < 2000 || s->D >= 64 || s->P > 15) return OSG_OK;
  const int P = s->P, A = s->A, D = s->D;
  int L = 0;
  for (; L < D; ++L) {
    bool all_chance = true;
    for (int h = s->level_off[L]; h < s->level_off[L + 1]; ++h) all_chance &= s->kind[h] == kChanceNode;
    if (!all_chance) break;
  }
  if (L < 1 || L >= D - 1) return OSG_OK;
  const int G = s->level_off[L + 1] - s->level_off[L], NT = s->level_off[L + 1];
  if (G < 4 || G > 65536) return OSG_OK;
  std::vector<int32_t> sub_of(s->H, -1), level_of(s->H, 0);
  for (int l = 0; l < D; ++l)
    for (int h = s->level_off[l]; h < s->level_off[l + 1]; ++h) level_of[h] = l;
  for (int h = s->level_off[L]; h < s->H; ++h)
    sub_of[h] = h < s->level_off[L + 1] ? h - s->level_off[L] : sub_of[s->parent[h]];
  std::vector<std::vector<int32_t>> hist(G);
  for (int h = s->level_off[L]; h < s->H; ++h) hist[sub_of[h]].push_back(h);  // ascending h = level-major
  const int M = static_cast<int>(s->mem.size());
  for (int m = 0; m < M; ++m)
    if (sub_of[s->mem[m]] < 0) return OSG_OK;  // a decision node above the cut
  // the groups of subtrees, kind by kind
  std::vector<std::vector<int32_t>> groups;  // subtree lists
  std::vector<int32_t> group_kind;
  for (int g = 0; g < G; ++g) { groups.push_back({g}); group_kind.push_back(0); }
  for (int r = 0; r < P; ++r) {
    std::vector<int32_t> uf(G);
    for (int g = 0; g < G; ++g) uf[g] = g;
    auto find = [&](int x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
    for (int i = 0; i < s->I; ++i) {
      if (s->info_player[i] != r) continue;
      for (int m = s->mem_off[i] + 1; m < s->mem_off[i + 1]; ++m) {
        const int a = find(sub_of[s->mem[s->mem_off[i]]]), b = find(sub_of[s->mem[m]]);
        if (a != b) uf[std::max(a, b)] = std::min(a, b);
      }
    }
    std::vector<int32_t> slot(G, -1);
    for (int g = 0; g < G; ++g) {
      const int root = find(g);
      if (slot[root] < 0) { slot[root] = static_cast<int32_t>(groups.size()); groups.push_back({}); group_kind.push_back(1 + r); }
      groups[slot[root]].push_back(g);
    }
  }
  const int J = static_cast<int>(groups.size());
  std::vector<int32_t> job(static_cast<size_t>(J) * 8, 0), jlevel(static_cast<size_t>(J) * (D + 1), 0), desc, fc, row, glob, ient, ment;
  std::vector<int32_t> loc(s->H, -1), job_of_sub(G, -1);
  size_t lds = sizeof(double) * 2 * P * NT;
  int max_nodes = 0;
  const size_t IA = static_cast<size_t>(s->I) * A;
  for (int j = 0; j < J; ++j) {
    const int kind = group_kind[j], r = kind - 1;
    std::vector<int32_t> nodes;
    for (int g : groups[j]) nodes.insert(nodes.end(), hist[g].begin(), hist[g].end());
    std::sort(nodes.begin(), nodes.end());
    const int nn = static_cast<int>(nodes.size());
    for (int x = 0; x < nn; ++x) loc[nodes[x]] = x;
    const int n0 = static_cast<int>(desc.size());
    int at = 0;
    for (int l = 0; l <= D; ++l) {
      while (l < D && at < nn && level_of[nodes[at]] < l) ++at;
      jlevel[static_cast<size_t>(j) * (D + 1) + l] = l == D ? nn : at;
    }
    for (int x = 0; x < nn; ++x) {
      const int h = nodes[x];
      desc.push_back(s->kind[h] | (s->nchild[h] << 2) | ((s->actor[h] + 1) << 10));
      fc.push_back(s->kind[h] == kTerminalNode ? 0 : loc[s->first_child[h]]);
      row.push_back(s->kind[h] == kDecisionNode ? s->info[h] * A : 0);
      glob.push_back(h);
    }
    const int i0 = static_cast<int>(ient.size() / 4), m0 = static_cast<int>(ment.size() / 2);
    int ni = 0, nm = 0;
    if (kind != 0) {
      for (int g : groups[j]) job_of_sub[g] = j;
      for (int i = 0; i < s->I; ++i) {
        if (s->info_player[i] != r || s->mem_off[i + 1] == s->mem_off[i]) continue;
        if (job_of_sub[sub_of[s->mem[s->mem_off[i]]]] != j) continue;
        ient.insert(ient.end(), {i, s->info_level[i], nm, s->mem_off[i + 1] - s->mem_off[i]});
        ++ni;
        for (int m = s->mem_off[i]; m < s->mem_off[i + 1]; ++m) {
          ment.insert(ment.end(), {m, loc[s->mem[m]]});
          ++nm;
        }
      }
      for (int g : groups[j]) job_of_sub[g] = -1;
    }
    int32_t* jd = &job[static_cast<size_t>(j) * 8];
    jd[0] = kind; jd[1] = n0; jd[2] = nn; jd[3] = i0; jd[4] = ni; jd[5] = m0; jd[6] = nm;
    const size_t bytes = sizeof(double) * (IA + static_cast<size_t>(nn) * (kind == 0 ? P : 1) + nn + nm) +
                         sizeof(int32_t) * (3 * static_cast<size_t>(nn) + nm + s->I);
    lds = std::max(lds, bytes);
    max_nodes = std::max(max_nodes, nn);
  }
  if (lds > 150 * 1024) return OSG_OK;
  hipStream_t st = s->ctx->stream;
  int rc;
  if ((rc = upload(job, &s->d_jobs_job, st)) || (rc = upload(jlevel, &s->d_jobs_level, st)) ||
      (rc = upload(desc, &s->d_jobs_desc, st)) || (rc = upload(fc, &s->d_jobs_fc, st)) || (rc = upload(row, &s->d_jobs_row, st)) ||
      (rc = upload(glob, &s->d_jobs_glob, st)) || (rc = upload(ient, &s->d_jobs_info, st)) || (rc = upload(ment, &s->d_jobs_mem, st)))
    return rc;
  OSG_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_jobs_deal), sizeof(double) * 2 * P * G));
  OSG_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_jobs_ticket), sizeof(unsigned int) * 4));
  OSG_HIP(hipMemsetAsync(s->d_jobs_deal, 0, sizeof(double) * 2 * P * G, st));
  OSG_HIP(hipMemsetAsync(s->d_jobs_ticket, 0, sizeof(unsigned int) * 4, st));
  if (raise_lds_cap(reinterpret_cast<const void*>(&k_eval_jobs), static_cast<int>(lds)) != hipSuccess) {
    (void)hipGetLastError();
    return OSG_OK;
  }
  s->jobs_J = J; s->jobs_L = L; s->jobs_G = G; s->jobs_NT = NT; s->jobs_lds_bytes = lds;
  s->jobs_threads = std::max(64, std::min(1024, (max_nodes + 63) / 64 * 64));
  s->jobs_ok = true;
  return OSG_OK;
}

// The subtrees of k_cfr_sub: the same cut as build_split (the first level with a node that is not a chance node),
// any number of subtrees (a workgroup takes several in turn when the cooperative grid is smaller), up to 8 x 1024
// histories each.  Round 5, forest form: with more subtrees than compute units the bins of the workgroups are packed —
// whole subtrees if they fit, else the pieces one level below the cut (SubTree's comment) — so that every workgroup
// sweeps one bin per pass.  OSG_CFR_SUB_PACK=0 keeps a subtree per bin.
template <int kK> const void* cfr_sub_kernel() { return reinterpret_cast<const void*>(&k_cfr_sub<kK>); }
int build_sub(osg_cfr* s) {
  s->sub_ok = false;
  if (s->cfg.solver != 0 || s->B != 1 || !s->path_kernel || s->A > kSplitMaxA || !s->cfg.alternating_updates) return OSG_OK;
  if (s->H < 4096 || s->D >= 63 || s->H >= (1 << 23)) return OSG_OK;
  int L = 0;
  for (; L < s->D; ++L) {
    bool all_chance = true;
    for (int h = s->level_off[L]; h < s->level_off[L + 1]; ++h) all_chance &= s->kind[h] == kChanceNode;
    if (!all_chance) break;
  }
  if (L < 1 || L >= s->D - 1) return OSG_OK;
  if (s->level_off[L + 1] - s->level_off[L] < 8) return OSG_OK;
  std::vector<int32_t> sub_of(s->H, -1), loc_of(s->H, -1), level_of(s->H, 0);
  for (int l = 0; l < s->D; ++l)
    for (int h = s->level_off[l]; h < s->level_off[l + 1]; ++h) level_of[h] = l;
  // ---- the bins: which histories a workgroup sweeps together ----
  // sizes of every history's subtree (children have larger indices than their parent)
  std::vector<int32_t> sz(s->H, 1), szd(s->H, 0);
  for (int h = s->H - 1; h >= 1; --h) {
    szd[h] += s->kind[h] == kDecisionNode ? 1 : 0;
    sz[s->parent[h]] += sz[h];
    szd[s->parent[h]] += szd[h];
  }
  int cus = s->num_cus;
  if (cus <= 0) {
    hipDeviceProp_t dp;
    if (hipGetDeviceProperties(&dp, s->ctx->device) != hipSuccess) { (void)hipGetLastError(); return OSG_OK; }
    cus = std::max(1, dp.multiProcessorCount);
  }
  // longest-processing-time packing of the histories of level `lp` into at most `cus` bins, balanced by subtree size
  auto pack = [&](int lp, std::vector<int32_t>* bin_of, int* bins) -> bool {
    const int n = s->level_off[lp + 1] - s->level_off[lp], base = s->level_off[lp];
    const int nb = std::min(n, cus);
    std::vector<int32_t> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sz[base + a] > sz[base + b]; });
    std::vector<int64_t> load(nb, 0), loadd(nb, 0), cnt(nb, 0);
    bin_of->assign(n, 0);
    for (int i : order) {
      int best = 0;
      for (int b = 1; b < nb; ++b)
        if (load[b] < load[best]) best = b;
      (*bin_of)[i] = best;
      load[best] += sz[base + i]; loadd[best] += szd[base + i]; ++cnt[best];
    }
    for (int b = 0; b < nb; ++b) {
      const int64_t nl = load[b], nd = loadd[b] + cnt[b];   // (+ the rows of the pieces' upper parents)
      if (nl > 8 * kSubThreads || nd > kSubKD * kSubThreads || sizeof(double) * (nl + nd * s->A) > 150 * 1024) return false;
    }
    *bins = nb;
    return true;
  };
  const int G0 = s->level_off[L + 1] - s->level_off[L];
  int G = G0, piece_level = L;
  bool upper = false;          // the histories of level L belong to no bin
  std::vector<int32_t> bin_of;
  const char* pk = std::getenv("OSG_CFR_SUB_PACK");
  if (G0 > cus && !(pk && pk[0] == '0')) {
    int nb = 0;
    if (pack(L, &bin_of, &nb)) {
      G = nb;
    } else if (L + 1 < s->D - 1 && pack(L + 1, &bin_of, &nb)) {
      G = nb; piece_level = L + 1; upper = true;
    } else {
      bin_of.clear();
    }
  }
  if (bin_of.empty()) {
    bin_of.resize(G0);
    for (int g = 0; g < G0; ++g) bin_of[g] = g;
  }
  std::vector<std::vector<int32_t>> hist(G);
  for (int h = s->level_off[piece_level]; h < s->H; ++h) {
    sub_of[h] = h < s->level_off[piece_level + 1] ? bin_of[h - s->level_off[piece_level]] : sub_of[s->parent[h]];
    const int g = sub_of[h];
    loc_of[h] = static_cast<int32_t>(hist[g].size());
    hist[g].push_back(h);
  }
  int NL = 0;
  for (int g = 0; g < G; ++g) NL = std::max<int>(NL, static_cast<int>(hist[g].size()));
  const int K = NL <= 2 * kSubThreads ? 2 : (NL <= 4 * kSubThreads ? 4 : 8);
  if (NL > 8 * kSubThreads) return OSG_OK;
  const size_t M = s->mem.size();
  std::vector<std::vector<int32_t>> members(static_cast<size_t>(G) * s->P);
  std::vector<int32_t> upper_members;            // forest form: the members of level L, in member order
  std::vector<int32_t> upper_of(std::max<size_t>(M, 1), -1);
  for (size_t m = 0; m < M; ++m) {
    const int h = s->mem[m];
    if (sub_of[h] < 0) {
      if (!upper || level_of[h] != L) return OSG_OK;  // a decision node above the cut
      upper_of[m] = static_cast<int32_t>(upper_members.size());
      upper_members.push_back(static_cast<int32_t>(m));
      continue;
    }
    members[static_cast<size_t>(sub_of[h]) * s->P + s->actor[h]].push_back(static_cast<int32_t>(m));
  }
  if (std::getenv("OSG_CFR_SUB_STATS")) {   // how even the bins are: histories, levels and members per player
    for (int q = -1; q < s->P; ++q) {
      int64_t lo = INT64_MAX, hi = 0, sum = 0;
      for (int g = 0; g < G; ++g) {
        const int64_t v = q < 0 ? static_cast<int64_t>(hist[g].size()) : static_cast<int64_t>(members[static_cast<size_t>(g) * s->P + q].size());
        lo = std::min(lo, v); hi = std::max(hi, v); sum += v;
      }
      fprintf(stderr, "k_cfr_sub bins: %s min %lld mean %.1f max %lld over %d bins\n", q < 0 ? "histories" : (q == 0 ? "members p0" : (q == 1 ? "members p1" : "members p2+")),
              static_cast<long long>(lo), static_cast<double>(sum) / G, static_cast<long long>(hi), G);
    }
    for (int g = 0; g < G; g += std::max(1, G / 12)) {
      fprintf(stderr, "  bin %d: histories %zu members", g, hist[g].size());
      for (int q = 0; q < s->P; ++q) fprintf(stderr, " %zu", members[static_cast<size_t>(g) * s->P + q].size());
      fprintf(stderr, "\n");
    }
  }
  std::vector<int32_t> nloc(G), desc(static_cast<size_t>(G) * NL, kTerminalNode | (63 << 10)), fc(static_cast<size_t>(G) * NL, 0),
      aux(static_cast<size_t>(G) * NL, 0), mem_off(static_cast<size_t>(G) * s->P + 1, 0), sub_rec,
      info_off(s->P + 1, 0), info_list;
  // the decisions of one player on a root path: the codes of a member record are P groups of `per_player` int4 chunks
  int most = 0;
  {
    std::vector<int> cnt(s->P);
    for (size_t m = 0; m < M; ++m) {
      std::fill(cnt.begin(), cnt.end(), 0);
      for (int e = s->path_off[m]; e < s->path_off[m + 1]; ++e)
        if (!((s->path[e] >> 23) & 1)) most = std::max(most, ++cnt[(s->path[e] >> 24) & 0xF]);
    }
  }
  const int per_player = std::max(1, (most + 3) / 4);
  if (per_player * s->P > kSubCodeChunks) return OSG_OK;   // more decisions on a path than the packed record keeps
  const int PL = 4 * per_player * s->P;
  std::vector<std::vector<int32_t>> dec_rows(G);
  std::vector<std::map<int32_t, int32_t>> extra_row(G);
  std::vector<int32_t> anc, filled(s->P);
  int32_t n_members = 0;
  std::vector<int32_t> dec_off(static_cast<size_t>(G) * (s->P + 2), 0);
  std::vector<std::vector<double>> chance_probs(G);
  for (int g = 0; g < G; ++g) {
    nloc[g] = static_cast<int32_t>(hist[g].size());
    // the bin's decision rows ordered by acting player (a pass re-fetches one player's rows only: SubTree's comment)
    std::vector<int32_t> next(s->P + 1, 0);
    for (int h : hist[g])
      if (s->kind[h] == kDecisionNode) ++next[s->actor[h] + 1];
    for (int q = 0; q < s->P; ++q) next[q + 1] += next[q];
    for (int q = 0; q <= s->P; ++q) dec_off[static_cast<size_t>(g) * (s->P + 2) + q] = next[q];
    dec_rows[g].assign(next[s->P], 0);
    for (size_t j = 0; j < hist[g].size(); ++j) {
      const int h = hist[g][j];
      const size_t at = static_cast<size_t>(g) * NL + j;
      desc[at] = s->kind[h] | (s->nchild[h] << 2) | (level_of[h] << 10) | ((s->actor[h] + 1) << 16);
      fc[at] = s->kind[h] == kTerminalNode ? 0 : loc_of[s->first_child[h]];
      aux[at] = h;
      if (s->kind[h] == kDecisionNode) {
        aux[at] = next[s->actor[h]]++;
        dec_rows[g][aux[at]] = s->info[h] * s->A;
      } else if (s->kind[h] == kChanceNode) {
        aux[at] = static_cast<int32_t>(chance_probs[g].size());   // its outcome probabilities, staged in LDS
        for (int c = 0; c < s->nchild[h]; ++c) chance_probs[g].push_back(s->edge_prob[s->first_child[h] + c]);
      }
    }
    for (int q = 0; q < s->P; ++q) {
      for (int32_t m : members[static_cast<size_t>(g) * s->P + q]) {
        const int h = s->mem[m];
        const size_t at = static_cast<size_t>(g) * NL + loc_of[h];
        sub_rec.push_back(m);
        sub_rec.push_back(loc_of[h]);
        sub_rec.push_back(aux[at] | (static_cast<int32_t>(s->nact[s->info[h]]) << 24));
        sub_rec.push_back(fc[at]);
        // the root path, leaf to root: entry e of SmallTree::path is the edge out of the ancestor at depth e
        const int len = s->path_off[m + 1] - s->path_off[m];
        anc.resize(len);
        for (int e = len - 1, x = s->parent[h]; e >= 0; --e, x = s->parent[x]) anc[e] = x;
        double chance = 1.0;
        std::vector<int32_t> codes(PL, -1);
        std::fill(filled.begin(), filled.end(), 0);
        for (int e = 0; e < len; ++e) {
          const int code = s->path[s->path_off[m] + e];
          if ((code >> 23) & 1) {
            chance *= s->edge_prob[code & 0x7FFFFF];
          } else {
            const int pl = (code >> 24) & 0xF, a_idx = (code & 0x7FFFFF) - s->info[anc[e]] * s->A;
            if (a_idx < 0 || a_idx >= s->A) return OSG_OK;   // (cannot happen)
            int d_anc;
            if (sub_of[anc[e]] == g) {
              d_anc = aux[static_cast<size_t>(g) * NL + loc_of[anc[e]]];
            } else if (upper && sub_of[anc[e]] < 0 && level_of[anc[e]] == L) {
              // a deal root above the forest: its policy row rides behind the forest's own rows
              auto it = extra_row[g].find(anc[e]);
              if (it == extra_row[g].end()) {
                it = extra_row[g].emplace(anc[e], static_cast<int32_t>(dec_rows[g].size())).first;
                dec_rows[g].push_back(s->info[anc[e]] * s->A);
              }
              d_anc = it->second;
            } else {
              return OSG_OK;   // (cannot happen: decisions sit below the cut)
            }
            codes[static_cast<size_t>(pl) * 4 * per_player + filled[pl]++] = d_anc * s->A + a_idx;
          }
        }
        int64_t bits;
        memcpy(&bits, &chance, sizeof bits);
        sub_rec.push_back(static_cast<int32_t>(bits & 0xFFFFFFFF));
        sub_rec.push_back(static_cast<int32_t>(bits >> 32));
        sub_rec.push_back(0);
        sub_rec.push_back(0);
        // the codes as 16-bit halves (a code indexes the bin's ND * A <= 16 384 staged policy entries; 0xFFFF pads), the
        // record padded to whole 16-byte pieces
        for (int c = 0; c < PL; c += 2)
          sub_rec.push_back(static_cast<int32_t>((static_cast<uint32_t>(codes[c]) & 0xFFFFu) |
                                                 ((static_cast<uint32_t>(codes[c + 1]) & 0xFFFFu) << 16)));
        for (int c = PL / 2; c % 4 != 0; ++c) sub_rec.push_back(-1);
        ++n_members;
      }
      mem_off[static_cast<size_t>(g) * s->P + q + 1] = n_members;
    }
  }
  for (int q = 0; q < s->P; ++q) {
    for (int i = 0; i < s->I; ++i)
      if (s->info_player[i] == q) info_list.push_back(i);
    info_off[q + 1] = static_cast<int32_t>(info_list.size());
  }
  int ND = 2, NCP = 2;
  for (int g = 0; g < G; ++g) {
    ND = std::max<int>(ND, static_cast<int>(dec_rows[g].size()));
    NCP = std::max<int>(NCP, static_cast<int>(chance_probs[g].size()));
  }
  ND += ND & 1; NCP += NCP & 1;   // (even: the values and the fold's stage behind them stay 16-byte aligned)
  std::vector<int32_t> ndec(G), dec_row(static_cast<size_t>(G) * ND, 0);
  std::vector<double> chance_prob(static_cast<size_t>(G) * NCP, 0.0);
  for (int g = 0; g < G; ++g) {
    ndec[g] = static_cast<int32_t>(dec_rows[g].size());
    dec_off[static_cast<size_t>(g) * (s->P + 2) + s->P + 1] = ndec[g];   // (the upper parents' rows sit behind the players')
    std::copy(dec_rows[g].begin(), dec_rows[g].end(), dec_row.begin() + static_cast<size_t>(g) * ND);
    std::copy(chance_probs[g].begin(), chance_probs[g].end(), chance_prob.begin() + static_cast<size_t>(g) * NCP);
  }
  // dynamic LDS: [policy rows ND * A | chance probabilities NCP | values NL | spare]; the fold stages 64-byte member
  // records from the values on (one bin per workgroup: the rows stay) — the spare takes it to 2 048 records where there is room
  const size_t base_doubles = static_cast<size_t>(ND) * s->A + NCP + NL;
  if (sizeof(double) * base_doubles > 150 * 1024 || ND > kSubKD * kSubThreads) return OSG_OK;
  const size_t lds_doubles = std::min<size_t>(static_cast<size_t>(ND) * s->A + NCP + static_cast<size_t>(kSubFoldX) * kSubThreads * kSubRecDoubles,
                                              (158 * 1024) / sizeof(double));
  const size_t lds = sizeof(double) * std::max(base_doubles, lds_doubles);
  const int fold_cap = static_cast<int>(std::min<size_t>((lds / sizeof(double) - static_cast<size_t>(ND) * s->A - NCP) / kSubRecDoubles,
                                                         static_cast<size_t>(kSubFoldX) * kSubThreads));
  // forest form: the pieces' roots (their values leave through root_value) and the upper members' records
  std::vector<int32_t> nroot, root_loc, root_idx, upper_rec;
  int NR = 0;
  const int root_base = s->level_off[piece_level];
  if (upper) {
    std::vector<std::vector<int32_t>> roots(G);
    for (int h = root_base; h < s->level_off[piece_level + 1]; ++h) roots[sub_of[h]].push_back(h);
    for (int g = 0; g < G; ++g) NR = std::max<int>(NR, static_cast<int>(roots[g].size()));
    nroot.resize(G);
    root_loc.assign(static_cast<size_t>(G) * NR, 0);
    root_idx.assign(static_cast<size_t>(G) * NR, 0);
    for (int g = 0; g < G; ++g) {
      nroot[g] = static_cast<int32_t>(roots[g].size());
      for (size_t r = 0; r < roots[g].size(); ++r) {
        root_loc[static_cast<size_t>(g) * NR + r] = loc_of[roots[g][r]];
        root_idx[static_cast<size_t>(g) * NR + r] = roots[g][r] - root_base;
      }
    }
    for (int32_t m : upper_members) {
      const int h = s->mem[m];
      double chance = 1.0;   // the root path of a deal root holds chance edges only, multiplied in path order
      for (int e = s->path_off[m]; e < s->path_off[m + 1]; ++e) {
        const int code = s->path[e];
        if (!((code >> 23) & 1)) return OSG_OK;   // (cannot happen: every level above the cut is a chance level)
        chance *= s->edge_prob[code & 0x7FFFFF];
      }
      int64_t bits;
      memcpy(&bits, &chance, sizeof bits);
      upper_rec.push_back(s->first_child[h] - root_base);
      upper_rec.push_back(s->info[h] * s->A);
      upper_rec.push_back(s->nact[s->info[h]]);
      upper_rec.push_back(0);
      upper_rec.push_back(static_cast<int32_t>(bits & 0xFFFFFFFF));
      upper_rec.push_back(static_cast<int32_t>(bits >> 32));
      upper_rec.push_back(0);
      upper_rec.push_back(0);
    }
  }
  if (static_cast<unsigned long long>(M) * kSubRecDoubles * 8 >= (1ull << 31)) return OSG_OK;   // 32-bit record offsets
  int widest = 0;   // the fold stages an infostate's member records in LDS: all of one infostate must fit a round
  for (int i = 0; i < s->I; ++i) widest = std::max(widest, s->mem_off[i + 1] - s->mem_off[i]);
  if (widest > fold_cap) return OSG_OK;
  const void* kern = K == 2 ? cfr_sub_kernel<2>() : (K == 4 ? cfr_sub_kernel<4>() : cfr_sub_kernel<8>());
  if (raise_lds_cap(kern, static_cast<int>(lds)) != hipSuccess) {
    (void)hipGetLastError();
    return OSG_OK;
  }
  int per_cu = 0;
  hipError_t e;
  if (K == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cfr_sub<2>, kSubThreads, lds);
  else if (K == 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cfr_sub<4>, kSubThreads, lds);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cfr_sub<8>, kSubThreads, lds);
  hipDeviceProp_t prop;
  if (e != hipSuccess || per_cu < 1 || hipGetDeviceProperties(&prop, s->ctx->device) != hipSuccess || !prop.cooperativeLaunch) {
    (void)hipGetLastError();
    return OSG_OK;
  }
  const int grid = std::min(G, per_cu * prop.multiProcessorCount);
  // the fold's shares: a workgroup's run of the updating player's infostates (info_list order), balanced by members
  std::vector<int32_t> fold_info(info_list.size() * 4), fold_off(static_cast<size_t>(s->P) * (grid + 1), 0);
  for (size_t e = 0; e < info_list.size(); ++e) {
    const int i = info_list[e];
    fold_info[4 * e] = i; fold_info[4 * e + 1] = s->nact[i]; fold_info[4 * e + 2] = s->mem_off[i];
    fold_info[4 * e + 3] = s->mem_off[i + 1] - s->mem_off[i];
  }
  for (int q = 0; q < s->P; ++q) {
    int64_t total = 0;
    for (int e = info_off[q]; e < info_off[q + 1]; ++e) total += fold_info[4 * static_cast<size_t>(e) + 3] + 8;   // (+ the row's own cost)
    int64_t run = 0;
    int e = info_off[q];
    for (int w = 0; w < grid; ++w) {
      fold_off[static_cast<size_t>(q) * (grid + 1) + w] = e;
      const int64_t upto = total * (w + 1) / grid;
      while (e < info_off[q + 1] && run + (fold_info[4 * static_cast<size_t>(e) + 3] + 8) / 2 < upto) {
        run += fold_info[4 * static_cast<size_t>(e) + 3] + 8;
        ++e;
      }
    }
    fold_off[static_cast<size_t>(q) * (grid + 1) + grid] = info_off[q + 1];
    for (int w = grid - 1; w >= 0; --w)   // (everything is handed out: the last share takes what rounding left)
      if (fold_off[static_cast<size_t>(q) * (grid + 1) + w] > fold_off[static_cast<size_t>(q) * (grid + 1) + w + 1])
        fold_off[static_cast<size_t>(q) * (grid + 1) + w] = fold_off[static_cast<size_t>(q) * (grid + 1) + w + 1];
  }
  hipStream_t st = s->ctx->stream;
  int rc;
  if ((rc = upload(nloc, &s->d_sub_nloc, st)) || (rc = upload(desc, &s->d_sub_desc, st)) || (rc = upload(fc, &s->d_sub_fc, st)) ||
      (rc = upload(aux, &s->d_sub_aux, st)) || (rc = upload(mem_off, &s->d_sub_mem_off, st)) ||
      (rc = upload(sub_rec, &s->d_sub_rec, st)) ||
      (rc = upload(info_off, &s->d_sub_info_off, st)) || (rc = upload(info_list, &s->d_sub_info_list, st)) ||
      (rc = upload(ndec, &s->d_sub_ndec, st)) || (rc = upload(dec_row, &s->d_sub_dec_row, st)) ||
      (rc = upload(dec_off, &s->d_sub_dec_off, st)) || (rc = upload(chance_prob, &s->d_sub_chance_prob, st)) ||
      (rc = upload(fold_info, &s->d_sub_fold_info, st)) || (rc = upload(fold_off, &s->d_sub_fold_off, st)))
    return rc;
  {
    // the terminal returns of every bin by player, in the bin's local order: a pass starts with one coalesced copy into LDS
    const size_t n = static_cast<size_t>(G) * s->P * NL;
    if (n * sizeof(double) > (size_t{1} << 31)) return OSG_OK;
    std::vector<double> term_val(n, 0.0);
    for (int g = 0; g < G; ++g)
      for (size_t j = 0; j < hist[g].size(); ++j) {
        const int h = hist[g][j];
        if (s->kind[h] != kTerminalNode) continue;
        for (int q = 0; q < s->P; ++q)
          term_val[(static_cast<size_t>(g) * s->P + q) * NL + j] = s->term_ret[static_cast<size_t>(h) * s->P + q];
      }
    if ((rc = upload(term_val, &s->d_sub_term_val, st))) return rc;
  }
  s->sub_NCP = NCP;
  s->sub_keep_rows = grid >= G;
  {
    // the members' 64-byte records: zero, but an upper member's says which one it is (kSubFlagHi | 2 + u)
    std::vector<double> recbuf(std::max<size_t>(M, 1) * kSubRecDoubles, 0.0);
    for (size_t m = 0; m < M; ++m)
      if (upper_of[m] >= 0) {
        const int64_t bits = (static_cast<int64_t>(kSubFlagHi) << 32) | static_cast<int64_t>(2 + upper_of[m]);
        memcpy(&recbuf[m * kSubRecDoubles], &bits, sizeof bits);
      }
    if ((rc = upload(recbuf, &s->d_sub_recbuf, st))) return rc;
  }
  s->sub_forest = upper;
  if (upper) {
    if ((rc = upload(nroot, &s->d_sub_nroot, st)) || (rc = upload(root_loc, &s->d_sub_root_loc, st)) ||
        (rc = upload(root_idx, &s->d_sub_root_idx, st)) || (rc = upload(upper_rec, &s->d_sub_upper_rec, st)))
      return rc;
    const size_t n_roots = static_cast<size_t>(s->level_off[piece_level + 1] - root_base);
    OSG_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_sub_root_value), sizeof(double) * std::max<size_t>(n_roots, 1)));
    OSG_HIP(hipMemsetAsync(s->d_sub_root_value, 0, sizeof(double) * std::max<size_t>(n_roots, 1), st));
    s->sub_NR = NR;
  }
  s->sub_ND = ND;
  s->sub_PL = PL;
  OSG_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_sub_bar), sizeof(unsigned int) * kSubBarWords));
  OSG_HIP(hipMemsetAsync(s->d_sub_bar, 0, sizeof(unsigned int) * kSubBarWords, st));
  s->sub_G0 = G0;
  s->sub_G = G; s->sub_L = L; s->sub_NL = NL; s->sub_K = K; s->sub_grid = grid; s->sub_lds_bytes = lds;
  s->sub_ok = true;
  return OSG_OK;
}


// A grid barrier of k_cfr_split / k_cfr_sub that timed out (a hung device: the launches are cooperative) leaves the
// tables mixed: the solver refuses further work.  The kernels raise a pinned host word, read here without a copy or a
// wait — by every entry point that advances, reads or hands out the tables.
int cfr_sub_error(const osg_cfr* s) {
  if (s->h_sub_err && __atomic_load_n(s->h_sub_err, __ATOMIC_RELAXED) != 0)
    return set_error(OSG_ERR_HIP, "a grid barrier of the subtree CFR kernel timed out in an earlier launch (the launch is "
                                  "cooperative: a hung device, not contention); the tables are not usable — "
                                  "osg_cfr_cfg.kernel = 3 runs one workgroup");
  return OSG_OK;
}

}  // namespace

extern "C" {

int osg_cfr_create(osg_ctx* ctx, const char* game_string, const osg_cfr_cfg* cfg, osg_cfr** out) {
  if (!ctx || !game_string || !cfg || !out) return set_error(OSG_ERR_INVALID, "osg_cfr_create: null argument");
  if (ctx->closed) return set_error(OSG_ERR_INVALID, "osg_cfr_create: the context was destroyed");
  osg_cfr* s = new osg_cfr;
  s->ctx = ctx;
  s->cfg = *cfg;
  int rc = parse_game(game_string, &s->spec);
  if (rc) { delete s; return rc; }
  if (s->spec.desc.game_kind != kKuhn && s->spec.desc.game_kind != kLeduc) {
    delete s;
    return set_error(OSG_ERR_UNSUPPORTED, "tabular CFR needs information-state strings: kuhn_poker and leduc_poker only");
  }
  if (s->spec.leduc_big) {   // (4-player leduc_poker has ~3e8 histories, 10 players beyond any table)
    delete s;
    return set_error(OSG_ERR_UNSUPPORTED, "tabular solvers: leduc_poker with up to 3 players (the trees of 4+ players do not fit a device)");
  }
  s->B = s->cfg.replicas > 0 ? s->cfg.replicas : 1;
  if (s->B > 1 && s->cfg.solver != 0) { delete s; return set_error(OSG_ERR_UNSUPPORTED, "replicas > 1: CFR family only"); }
  if (s->B > (1 << 16)) { delete s; return set_error(OSG_ERR_INVALID, "osg_cfr_cfg.replicas must be <= 65536"); }
  if (s->cfg.solver < 0 || s->cfg.solver > 2) { delete s; return set_error(OSG_ERR_INVALID, "osg_cfr_cfg.solver must be 0, 1 or 2"); }
  if (s->cfg.solver == 2 && !(s->cfg.epsilon > 0.0 && s->cfg.epsilon <= 1.0)) {
    delete s;
    return set_error(OSG_ERR_INVALID, "outcome sampling needs 0 < epsilon <= 1 (kDefaultEpsilon = 0.6)");
  }
  rc = build_tree(s, game_string);
  if (rc) { delete s; return rc; }
  if (s->cfg.solver == 2 && s->D > kMaxOsDepth) {
    delete s;
    return set_error(OSG_ERR_UNSUPPORTED, "outcome sampling: game tree deeper than 32 levels");
  }
  osg::ctx_retain(ctx);  // from here on the object dies through osg_cfr_destroy, which releases
  hipStream_t st = ctx->stream;
  if ((rc = upload(s->level_off, &s->d_level_off, st)) || (rc = upload(s->parent, &s->d_parent, st)) ||
      (rc = upload(s->first_child, &s->d_first_child, st)) || (rc = upload(s->info, &s->d_info, st)) ||
      (rc = upload(s->mem_off, &s->d_mem_off, st)) || (rc = upload(s->mem, &s->d_mem, st)) ||
      (rc = upload(s->nact, &s->d_nact, st)) || (rc = upload(s->kind, &s->d_kind, st)) ||
      (rc = upload(s->nchild, &s->d_nchild, st)) || (rc = upload(s->aidx, &s->d_aidx, st)) ||
      (rc = upload(s->actor, &s->d_actor, st)) || (rc = upload(s->info_player, &s->d_info_player, st)) ||
      (rc = upload(s->edge_prob, &s->d_edge_prob, st)) || (rc = upload(s->term_ret, &s->d_term_ret, st))) {
    osg_cfr_destroy(s);
    return rc;
  }
  const size_t IA = static_cast<size_t>(s->I) * s->A;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_tables), sizeof(double) * s->B * 5 * std::max<size_t>(IA, 1));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_reach), sizeof(double) * s->H * (s->P + 1));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_value), sizeof(double) * s->H * s->P);
  if (e != hipSuccess) { osg_cfr_destroy(s); return set_error(OSG_ERR_NOMEM, hipGetErrorString(e)); }
  // Whole solver state in LDS when it fits (gfx950: 160 KiB per workgroup; leave headroom).
  s->lds_bytes = sizeof(double) * (static_cast<size_t>(s->H) * (2 * s->P + 1) + 3 * IA);
  s->lds_resident = s->lds_bytes <= 96 * 1024;
  if (s->lds_resident) {
    e = raise_lds_cap(reinterpret_cast<const void*>(&k_cfr<true>), static_cast<int>(s->lds_bytes));
    if (e != hipSuccess) { (void)hipGetLastError(); s->lds_resident = false; }
  }
  {  // the all-in-LDS kernel for small trees
    const size_t M = s->mem.size();
    const size_t doubles = static_cast<size_t>(s->H) * s->P + s->H + 3 * IA + 2 * M * s->A;
    const size_t ints = 3 * static_cast<size_t>(s->H) + M + (s->I + 1) + (M + 1) + s->path.size() + 2 * s->I + M + (s->D + 1);
    s->small_lds_bytes = doubles * 8 + ints * 4;
    const bool index_fits = static_cast<size_t>(s->H) < (1u << 23) && IA < (1u << 23);
    s->small_tree = index_fits && s->small_lds_bytes <= 64 * 1024 && s->P <= kMaxPlayers;
    s->path_kernel = index_fits && s->P <= kMaxPlayers;
    s->meta32.resize(s->H);
    for (int h = 0; h < s->H; ++h) s->meta32[h] = s->kind[h] | (s->nchild[h] << 2) | ((s->actor[h] + 1) << 10);
    s->info_player32.assign(s->info_player.begin(), s->info_player.end());
    if ((rc = upload(s->path_off, &s->d_path_off, st)) || (rc = upload(s->path, &s->d_path, st)) ||
        (rc = upload(s->info_level, &s->d_info_level, st)) || (rc = upload(s->mem_index, &s->d_mem_index, st)) ||
        (rc = upload(s->meta32, &s->d_meta32, st)) || (rc = upload(s->info_player32, &s->d_info_player32, st))) {
      osg_cfr_destroy(s);
      return rc;
    }
    const size_t eval_doubles = static_cast<size_t>(s->H) * (s->P + 1) + M + 2 * s->P + IA;
    e = hipMalloc(reinterpret_cast<void**>(&s->d_eval), sizeof(double) * eval_doubles);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_best), sizeof(int32_t) * std::max(s->I, 1));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_skip), sizeof(int32_t) * std::max<size_t>(M, 1));
    if (e == hipSuccess)
      e = hipMalloc(reinterpret_cast<void**>(&s->d_node_delta), sizeof(double) * 2 * std::max<size_t>(M * s->A, 1));
    if (e != hipSuccess) { osg_cfr_destroy(s); return set_error(OSG_ERR_NOMEM, hipGetErrorString(e)); }
    if (!index_fits) s->eval_ok = false;
    if (s->small_tree) {
      const void* variants[] = {reinterpret_cast<const void*>(&k_cfr_small<true, false, 3>),
                                reinterpret_cast<const void*>(&k_cfr_small<true, false, 4>),
                                reinterpret_cast<const void*>(&k_cfr_small<true, false, kMaxPlayers + 1>),
                                reinterpret_cast<const void*>(&k_cfr_small<true, true, 3>),
                                reinterpret_cast<const void*>(&k_cfr_small<true, true, 3, 2>),
                                reinterpret_cast<const void*>(&k_cfr_small<true, true, 3, 2, 2>),
                                reinterpret_cast<const void*>(&k_cfr_small<true, true, 3, 4>),
                                reinterpret_cast<const void*>(&k_cfr_small<true, true, 4>),
                                reinterpret_cast<const void*>(&k_cfr_small<true, true, kMaxPlayers + 1>)};
      e = hipSuccess;
      for (const void* f : variants)
        if (e == hipSuccess)
          e = raise_lds_cap(f, static_cast<int>(s->small_lds_bytes));
      if (e != hipSuccess) { (void)hipGetLastError(); s->small_tree = false; }
    }
  }
  if (s->B > 1 && !s->small_tree) {
    osg_cfr_destroy(s);
    return set_error(OSG_ERR_UNSUPPORTED, "replicas > 1 need the all-in-LDS kernel (tree too large)");
  }
  rc = build_resident_tree(s);
  if (rc) { osg_cfr_destroy(s); return rc; }
  if (hipHostMalloc(reinterpret_cast<void**>(&s->h_sub_err), sizeof(unsigned int), hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    osg_cfr_destroy(s);
    return set_error(OSG_ERR_NOMEM, "osg_cfr_create: pinned error word");
  }
  *s->h_sub_err = 0;
  rc = build_split(s);
  if (rc == OSG_OK) rc = build_sub(s);
  if (rc == OSG_OK) rc = build_eval_jobs(s);
  if (rc == OSG_OK && hipHostMalloc(reinterpret_cast<void**>(&s->h_eval_out), sizeof(double) * (2 * s->P + 1), hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    rc = set_error(OSG_ERR_NOMEM, "osg_cfr_create: pinned result buffer");
  }
  if (rc) { osg_cfr_destroy(s); return rc; }
  rc = init_tables(s);
  if (rc) { osg_cfr_destroy(s); return rc; }
  *out = s;
  return OSG_OK;
}

int osg_cfr_destroy(osg_cfr* s) {
  if (!s) return OSG_OK;
  (void)hipStreamSynchronize(s->ctx->stream);
  void* ptrs[] = {s->d_level_off, s->d_parent, s->d_first_child, s->d_info, s->d_mem_off, s->d_mem, s->d_nact,
                  s->d_kind, s->d_nchild, s->d_aidx, s->d_actor, s->d_info_player, s->d_edge_prob, s->d_term_ret,
                  s->d_tables, s->d_reach, s->d_value, s->d_path_off, s->d_path, s->d_info_level, s->d_mem_index,
                  s->d_best, s->d_eval, s->d_eval_ev, s->d_eval_level_info, s->d_meta32, s->d_info_player32, s->d_skip, s->d_node_delta, s->d_rec,
                  s->d_uret, s->d_uprob, s->d_spare_delta[0], s->d_spare_delta[1], s->d_split_nloc, s->d_split_desc,
                  s->d_split_fc, s->d_split_row, s->d_split_glob, s->d_split_mem_m, s->d_split_mem_hloc, s->d_split_info,
                  s->d_split_terms, s->d_split_bar, s->d_sub_nloc, s->d_sub_desc, s->d_sub_fc, s->d_sub_aux, s->d_sub_mem_off,
                  s->d_sub_info_off, s->d_sub_info_list, s->d_sub_bar, s->d_sub_ndec, s->d_sub_dec_row, s->d_sub_rec,
                  s->d_sub_stamps, s->d_mccfr_stamps, s->d_sub_recbuf, s->d_sub_chance_prob, s->d_sub_term_val, s->d_sub_dec_off, s->d_sub_fold_info, s->d_sub_fold_off, s->d_sub_nroot, s->d_sub_root_loc, s->d_sub_root_idx, s->d_sub_upper_rec, s->d_sub_root_value,
                  s->d_jobs_job, s->d_jobs_level, s->d_jobs_desc, s->d_jobs_fc, s->d_jobs_row, s->d_jobs_glob, s->d_jobs_info,
                  s->d_jobs_mem, s->d_jobs_deal, s->d_jobs_ticket};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (s->h_sub_err) (void)hipHostFree(s->h_sub_err);
  if (s->h_eval_out) (void)hipHostFree(s->h_eval_out);
  osg::ctx_release(s->ctx);
  delete s;
  return OSG_OK;
}

int osg_cfr_sizes(const osg_cfr* s, int64_t* out) {
  if (!s || !out) return set_error(OSG_ERR_INVALID, "osg_cfr_sizes: null argument");
  out[0] = s->H; out[1] = s->n_chance; out[2] = s->n_decision; out[3] = s->n_terminal; out[4] = s->I; out[5] = s->A;
  return OSG_OK;
}

int osg_cfr_reset(osg_cfr* s) { return init_tables(s); }

// One COOPERATIVE launch of k_cfr_split: the kernel spins on a grid barrier, so its workgroups must be resident
// together — with another stream keeping the device busy (a network's forward pass beside the solver) a plain launch
// can start some workgroups while the others queue behind foreign work, and the barrier's bound then turns a slowdown
// into an error.  The cooperative launch waits until the whole grid fits.  br: the CFR-BR pass set (d_best overrides).
static int launch_split(osg_cfr* s, SmallTree stree, SplitTree sp, Tables tb, int iters, int iteration0, osg_cfr_cfg cfg, bool br) {
  hipStream_t st = s->ctx->stream;
  // (the barrier's counters are zero: build_split zeroed them and every launch leaves them so; a launch that timed out
  // does not — and makes the solver unusable, cfr_sub_error)
  const dim3 grid(static_cast<unsigned>(s->split_G)), block(static_cast<unsigned>(s->split_threads));
  Tree tr = s->tree();
  const int32_t* best = br ? s->d_best : nullptr;
  void* args[] = {&tr, &stree, &sp, &tb, &iters, &iteration0, &cfg, &best};
  const void* kern = split_kernel(s->P, br, s->split_threads, s->A, br || cfg.alternating_updates);
  const size_t lds = s->split_lds_bytes + (br ? sizeof(double) * static_cast<size_t>(s->I) * s->A : 0);
  // OSG_CFR_PLAIN_LAUNCH=1: an ordinary launch, for hosts that run the solver alone on the device — the cooperative
  // launch costs 20 us per call (47.6 vs 27.6 us per one-iteration launch, CFR-BR 1.30e4 vs 1.82e4 it/s), which only the
  // calling pattern "one iteration per call" notices; without it the grid is resident together only as long as nothing
  // else holds the CUs (the barrier's 4 s bound then turns a starved launch into an error instead of a wait)
  static const bool plain = std::getenv("OSG_CFR_PLAIN_LAUNCH") && std::getenv("OSG_CFR_PLAIN_LAUNCH")[0] == '1';
  if (plain) OSG_HIP(hipLaunchKernel(kern, grid, block, args, lds, st));
  else OSG_HIP(hipLaunchCooperativeKernel(kern, grid, block, args, static_cast<unsigned>(lds), st));
  return OSG_OK;
}

int osg_cfr_iterate(osg_cfr* s, int iters) {
  if (!s || iters < 0) return set_error(OSG_ERR_INVALID, "osg_cfr_iterate: bad argument");
  if (iters == 0) return OSG_OK;
  if (int rc = cfr_sub_error(s)) return rc;
  int threads = ((s->max_level_width + 63) / 64) * 64;
  threads = std::max(64, std::min(threads, 1024));
  Tables tb{s->regrets(), s->cum(), s->cur()};
  if (s->B > 1) {  // all replicas advance together: workgroup b works on replica b's tables
    double* base0 = s->replica_base(0);
    tb = Tables{base0, base0 + static_cast<size_t>(s->I) * s->A, base0 + 2 * static_cast<size_t>(s->I) * s->A};
  }
  const unsigned grid_b = static_cast<unsigned>(s->B);
  // Trees far beyond one workgroup: full-grid launches per phase (osg_cfr_cfg.kernel == 2 forces it).
  const bool grid_path = s->path_kernel && s->B == 1 && (s->cfg.kernel == 2 || (s->cfg.kernel == 0 && s->H > 65536));
  // Trees too big for one workgroup: the persistent cooperative launch with a workgroup per deal subtree where the
  // tree has that shape (kernel == 5 forces it, 2 forces the per-phase launches), else a launch per phase.
  const bool sub_path = s->sub_ok && s->B == 1 && (s->cfg.kernel == 5 || (s->cfg.kernel == 0 && grid_path));
  if (sub_path) {
    if (int rc = cfr_sub_error(s)) return rc;
    const int M = static_cast<int>(s->mem.size());
    Tree tr = s->tree();
    SmallTree stree{s->d_path_off, s->d_path, M, static_cast<int>(s->path.size())};
    SubTree sp{s->sub_G, s->sub_L, s->sub_NL, s->d_sub_nloc, s->d_sub_desc, s->d_sub_fc, s->d_sub_aux, s->sub_ND, s->d_sub_ndec,
               s->d_sub_dec_row, s->d_sub_mem_off,
               s->d_sub_rec, s->sub_PL, s->d_sub_info_off, s->d_sub_info_list, s->d_sub_recbuf,
               (std::getenv("OSG_CFR_SUB_FLAT_BARRIER") && std::getenv("OSG_CFR_SUB_FLAT_BARRIER")[0] == '1') ? 0 : 1,
               s->d_sub_bar, s->h_sub_err, 400000000ull /* 4 s at 100 MHz */, nullptr};
    sp.dec_off = s->d_sub_dec_off; sp.chance_prob = s->d_sub_chance_prob; sp.NCP = s->sub_NCP;
    sp.keep_rows = (s->sub_keep_rows && !(std::getenv("OSG_CFR_SUB_KEEP_ROWS") && std::getenv("OSG_CFR_SUB_KEEP_ROWS")[0] == '0')) ? 1 : 0;
    sp.lds_doubles = static_cast<int>(s->sub_lds_bytes / sizeof(double));
    sp.fold_info = s->d_sub_fold_info; sp.fold_off = s->d_sub_fold_off;
    sp.term_val = s->d_sub_term_val;
    sp.prefetch = (std::getenv("OSG_CFR_SUB_PREFETCH") && std::getenv("OSG_CFR_SUB_PREFETCH")[0] == '0') ? 0 : 1;
    if (s->sub_forest) {
      sp.nroot = s->d_sub_nroot; sp.root_loc = s->d_sub_root_loc; sp.root_idx = s->d_sub_root_idx; sp.NR = s->sub_NR;
      sp.root_value = s->d_sub_root_value; sp.upper_rec = s->d_sub_upper_rec;
    }
    unsigned long long*& d_stamps = s->d_sub_stamps;   // OSG_CFR_SUB_STAMPS=1: phase stamps of one workgroup (tools/probe_cfr_sub.py); the solver's own buffer
    if (std::getenv("OSG_CFR_SUB_STAMPS")) {
      sp.stamp_wg = std::max(0, std::min(s->sub_grid - 1, atoi(std::getenv("OSG_CFR_SUB_STAMPS")) - 1));
      fprintf(stderr, "k_cfr_sub: G %d grid %d NL %d ND %d PL %d K %d forest %d NR %d\n", s->sub_G, s->sub_grid, s->sub_NL, s->sub_ND, s->sub_PL, s->sub_K, s->sub_forest ? 1 : 0, s->sub_NR);
      if (!d_stamps) OSG_HIP(hipMalloc(reinterpret_cast<void**>(&d_stamps), sizeof(unsigned long long) * 8 * kMaxPlayers));
      sp.stamps = d_stamps;
    }
    hipStream_t st = s->ctx->stream;
    const int per_launch = std::max(1, (1 << 30) / std::max(1, 2 * s->P * s->sub_grid));  // the arrival counter is 32 bits
    for (int done = 0; done < iters; done += per_launch) {
      int now = std::min(per_launch, iters - done), it0 = s->iteration + done;
      OSG_HIP(hipMemsetAsync(s->d_sub_bar, 0, sizeof(unsigned int) * kSubBarWords, st));
      void* args[] = {&tr, &stree, &sp, &tb, &now, &it0, &s->cfg};
      const void* kern = s->sub_K == 2 ? cfr_sub_kernel<2>() : (s->sub_K == 4 ? cfr_sub_kernel<4>() : cfr_sub_kernel<8>());
      // (OSG_CFR_PLAIN_LAUNCH=1 as for k_cfr_split: an ordinary launch, for hosts that own the device — and for runs under
      // rocprofv3 --kernel-trace, where a process that made a cooperative launch crashes in an exit handler)
      static const bool plain = std::getenv("OSG_CFR_PLAIN_LAUNCH") && std::getenv("OSG_CFR_PLAIN_LAUNCH")[0] == '1';
      if (plain) OSG_HIP(hipLaunchKernel(kern, dim3(static_cast<unsigned>(s->sub_grid)), dim3(kSubThreads), args, s->sub_lds_bytes, st));
      else OSG_HIP(hipLaunchCooperativeKernel(kern, dim3(static_cast<unsigned>(s->sub_grid)), dim3(kSubThreads), args,
                                              static_cast<unsigned>(s->sub_lds_bytes), st));
    }
    if (sp.stamps) {
      unsigned long long h[8 * kMaxPlayers];
      OSG_HIP(hipMemcpyAsync(h, sp.stamps, sizeof(unsigned long long) * 7 * s->P, hipMemcpyDeviceToHost, st));
      OSG_HIP(hipStreamSynchronize(st));
      for (int q = 0; q < s->P; ++q)
        fprintf(stderr, "k_cfr_sub pass %d (one workgroup, us): descriptors %.2f  preload %.2f  levels %.2f |  sweep %.2f  members %.2f  barrier %.2f  fold %.2f  (pass %.2f)\n", q,
                (h[s->P * 5 + q * 2 + 1] - h[q * 5]) / 100.0, (h[s->P * 5 + q * 2] - h[s->P * 5 + q * 2 + 1]) / 100.0,
                (h[q * 5 + 1] - h[s->P * 5 + q * 2]) / 100.0,
                (h[q * 5 + 1] - h[q * 5]) / 100.0, (h[q * 5 + 2] - h[q * 5 + 1]) / 100.0, (h[q * 5 + 3] - h[q * 5 + 2]) / 100.0,
                (h[q * 5 + 4] - h[q * 5 + 3]) / 100.0, q + 1 < s->P ? (h[(q + 1) * 5] - h[q * 5]) / 100.0 : 0.0);
    }
    s->iteration += iters;
    s->last_kernel = s->sub_forest ? "k_cfr_sub<forest>" : (s->sub_G < s->sub_G0 ? "k_cfr_sub<packed>" : "k_cfr_sub");
    return OSG_OK;
  }
  if (grid_path) {
    s->last_kernel = "k_gcfr";
    const int M = static_cast<int>(s->mem.size());
    GridCfr g;
    g.t = s->tree(); g.path_off = s->d_path_off; g.path = s->d_path; g.meta = s->d_meta32;
    g.info_player = s->d_info_player32; g.value = s->d_value; g.dreg = s->d_node_delta;
    g.dpol = s->d_node_delta + static_cast<size_t>(M) * s->A; g.skip = s->d_skip; g.tb = tb; g.M = M;
    g.pol = tb.cur;
    hipStream_t st = s->ctx->stream;
    auto blocks = [](int n) { return dim3(static_cast<unsigned>((n + 255) / 256)); };
    k_gcfr_init_values<<<blocks(s->H), dim3(256), 0, st>>>(g);
    const int passes = s->cfg.alternating_updates ? s->P : 1;
    for (int it = 0; it < iters; ++it) {
      for (int pass = 0; pass < passes; ++pass) {
        const int upd = s->cfg.alternating_updates ? pass : -1;
        const int q0 = upd >= 0 ? upd : 0, q1 = upd >= 0 ? upd + 1 : s->P;
        for (int l = s->D - 2; l >= 0; --l) {
          const int begin = s->level_off[l], end = s->level_off[l + 1];
          k_gcfr_level<<<blocks(end - begin), dim3(256), 0, st>>>(g, begin, end, q0, q1);
        }
        k_gcfr_members<<<blocks(M), dim3(256), 0, st>>>(g, upd, s->iteration + it + 1, s->cfg);
        k_gcfr_fold<<<blocks(s->I * 64), dim3(256), 0, st>>>(g, upd, s->cfg);
      }
    }
    OSG_HIP(hipGetLastError());
    s->iteration += iters;
    return OSG_OK;
  }
  if (s->split_ok && (s->cfg.kernel == 0 || s->cfg.kernel == 4)) {
    // one workgroup per deal subtree, one grid barrier per player pass (k_cfr_split)
    const int M = static_cast<int>(s->mem.size());
    SmallTree stree{s->d_path_off, s->d_path, M, static_cast<int>(s->path.size())};
    SplitTree sp{s->split_G, s->split_L, s->split_NL, s->split_NM, s->split_NI, s->d_split_nloc, s->d_split_desc,
                 s->d_split_fc, s->d_split_row, s->d_split_glob, s->d_split_mem_m, s->d_split_mem_hloc, s->d_split_info,
                 s->d_split_terms, s->d_split_bar, s->h_sub_err};
    const int passes = s->cfg.alternating_updates ? s->P : 1;
    const int per_launch = std::max(1, (1 << 30) / std::max(1, passes * s->split_G));  // the arrival counter is 32 bits
    for (int done = 0; done < iters; done += per_launch) {
      const int rc = launch_split(s, stree, sp, tb, std::min(per_launch, iters - done), s->iteration + done, s->cfg, false);
      if (rc) return rc;
    }
    OSG_HIP(hipGetLastError());
    s->iteration += iters;
    s->last_kernel = "k_cfr_split";
    return OSG_OK;
  }
  if (s->path_kernel && s->cfg.kernel != 1) {
    // Path-based kernel: no top-down reach pass; all-in-LDS when the tree is small enough.
    const int M = static_cast<int>(s->mem.size());
    SmallTree st{s->d_path_off, s->d_path, M, static_cast<int>(s->path.size())};
#ifndef OSG_AB_R4_REGS
    st.L0 = s->first_decision_level;
#endif
    SmallGlobal sg{s->d_value, s->d_node_delta, s->d_node_delta + static_cast<size_t>(M) * s->A, s->d_skip,
                   s->d_meta32, s->d_info_player32};
#define OSG_CFR_SMALL(LDS, OWNER, THREADS, SHMEM)                                                                  \
  do {                                                                                                              \
    if (s->P == 2) k_cfr_small<LDS, OWNER, 3><<<dim3(grid_b), dim3(THREADS), SHMEM, s->ctx->stream>>>(s->tree(), st, sg, tb, iters, s->iteration, s->cfg); \
    else if (s->P == 3) k_cfr_small<LDS, OWNER, 4><<<dim3(grid_b), dim3(THREADS), SHMEM, s->ctx->stream>>>(s->tree(), st, sg, tb, iters, s->iteration, s->cfg); \
    else k_cfr_small<LDS, OWNER, kMaxPlayers + 1><<<dim3(grid_b), dim3(THREADS), SHMEM, s->ctx->stream>>>(s->tree(), st, sg, tb, iters, s->iteration, s->cfg); \
  } while (0)
    if (s->small_tree && s->H <= 1024 && s->A <= kMaxA) {  // one thread per history: descriptors live in registers
      const int owner_threads = std::max(64, ((s->H + 63) / 64) * 64);
#ifdef OSG_AB_R4_REGS
      if (false) {}
#else
      if (s->P == 2 && s->max_path_decisions <= 2 && s->A == 2 && s->cfg.alternating_updates)   // kuhn_poker
        k_cfr_small<true, true, 3, 2, 2><<<dim3(grid_b), dim3(owner_threads), s->small_lds_bytes, s->ctx->stream>>>(s->tree(), st, sg, tb, iters, s->iteration, s->cfg);
      else if (s->P == 2 && s->max_path_decisions <= 2)   // two players, short paths: the 2-entry reach block
        k_cfr_small<true, true, 3, 2><<<dim3(grid_b), dim3(owner_threads), s->small_lds_bytes, s->ctx->stream>>>(s->tree(), st, sg, tb, iters, s->iteration, s->cfg);
      else if (s->P == 2 && s->max_path_decisions <= 4)
        k_cfr_small<true, true, 3, 4><<<dim3(grid_b), dim3(owner_threads), s->small_lds_bytes, s->ctx->stream>>>(s->tree(), st, sg, tb, iters, s->iteration, s->cfg);
#endif
      else
        OSG_CFR_SMALL(true, true, owner_threads, s->small_lds_bytes);
      s->last_kernel = "k_cfr_small<lds, owner>";
    } else if (s->small_tree) {
      OSG_CFR_SMALL(true, false, threads, s->small_lds_bytes);
      s->last_kernel = "k_cfr_small<lds>";
    } else {
      OSG_CFR_SMALL(false, false, threads, 0);
      s->last_kernel = "k_cfr_small<global>";
    }
#undef OSG_CFR_SMALL
  } else if (s->B > 1) {
    return set_error(OSG_ERR_UNSUPPORTED, "replicas > 1 are not available with the general kernel");
  } else if (s->lds_resident) {
    k_cfr<true><<<dim3(1), dim3(threads), s->lds_bytes, s->ctx->stream>>>(s->tree(), tb, s->d_reach, s->d_value, iters,
                                                                         s->iteration, s->cfg);
  } else {
    k_cfr<false><<<dim3(1), dim3(threads), 0, s->ctx->stream>>>(s->tree(), tb, s->d_reach, s->d_value, iters,
                                                                s->iteration, s->cfg);
  }
  OSG_HIP(hipGetLastError());
  s->iteration += iters;
  return OSG_OK;
}

// The traversals of one mini-batch into the delta tables dreg | dpol (the solver's own, or a caller's buffer).
// Which of the solver's own delta buffers `dreg` is (0 internal, 1 / 2 the spare ones), -1 for a caller's buffer.
static int delta_slot(const osg_cfr* s, const double* dreg) {
  if (s->B == 1 && dreg == s->dreg()) return 0;
  if (dreg && dreg == s->d_spare_delta[0]) return 1;
  if (dreg && dreg == s->d_spare_delta[1]) return 2;
  return -1;
}

static int mccfr_sample_impl(osg_cfr* s, uint64_t seed, int64_t first_trajectory, int64_t trajectories, double* dreg,
                             double* dpol) {
  if (!s || trajectories < 0) return set_error(OSG_ERR_INVALID, "osg_mccfr_sample: bad argument");
  if (s->A > kMaxA) return set_error(OSG_ERR_UNSUPPORTED, "osg_mccfr_sample: decision nodes wider than 4 actions");
  const int IA = s->I * s->A;
  hipStream_t st = s->ctx->stream;
  {  // the deltas start from zero: a buffer of the solver that the last fold left clean needs no fill launch
    const int slot = delta_slot(s, dreg);
    if (!(slot >= 0 && s->delta_clean[slot] && dpol == dreg + IA))
      OSG_HIP(hipMemsetAsync(dreg, 0, sizeof(double) * 2 * IA, st));  // dpol == dreg + IA
    if (slot >= 0) s->delta_clean[slot] = false;
  }
  if (trajectories == 0) return OSG_OK;
  const size_t lds = sizeof(double) * 2 * IA;
  const bool use_lds = lds <= 64 * 1024;
  int64_t blocks = (trajectories + 255) / 256;
  // Persistent workgroups: each flushes its LDS delta tables once, so fewer, longer-lived
  // groups mean fewer global atomics (256 CUs x 4 groups).
  if (blocks > 1024) blocks = 1024;
  if (s->resident_ok && s->cfg.kernel != 1) {
    // As many workgroups per CU as the LDS footprint allows.  A footprint that only fits once (leduc:
    // 143 KB) gets one group per CU, sized to the batch — every CU busy, up to 1024 lanes each; small
    // footprints (kuhn) get several 256- or 1024-lane groups per CU.
    int fit = static_cast<int>((160 * 1024) / std::max<size_t>(s->resident_lds_bytes, 1));
    // Mini-batches that leave lanes idle (fewer lanes than one round of the chip even with the split) run the
    // split form of the external-sampling kernel: 2 or 4 lanes per trajectory (OSG_MCCFR_SPLIT=0: never).
    // OSG_MCCFR_SPLIT=0 / 1 / 2: at most that many traverser levels are spread over lanes (default 2)
    static const int split_max = std::getenv("OSG_MCCFR_SPLIT") ? std::atoi(std::getenv("OSG_MCCFR_SPLIT")) : 2;
    const int q = s->A <= 2 ? 2 : 4;
    int split = 0;   // two levels while the lanes fit one round of the chip (2^14 trajectories: 31.4 us per step against 38.7
                     // with one level; 2^13: 26.7 against 37.2), else one level under the same condition, else the flat kernel
    if (s->cfg.solver != 2 && s->A >= 2) {
      const int64_t round = static_cast<int64_t>(s->num_cus) * 1024;
      if (split_max >= 2 && trajectories * q * q <= round) split = 2;
      else if (split_max >= 1 && trajectories * q <= round) split = 1;
    }
    const int64_t sampled = trajectories;
    if (split) trajectories *= split == 2 ? q * q : q;   // (the geometry below counts lanes)
    // Where the flat kernel reads the tree from: when the staged problem allows one workgroup per CU only (leduc: 143 KB,
    // 4 wavefronts per SIMD) but the tables alone would allow two (67 KB), the records can stay in global memory
    // (9 457 x 8 B, read-only: L2-resident) and two 1024-lane workgroups share a CU.  OSG_MCCFR_TREE=global | lds.
    size_t shmem_bytes = s->resident_lds_bytes;
    int tree_global = 0;
    {
      static const char* where = std::getenv("OSG_MCCFR_TREE");
      const size_t tables_only = s->resident_lds_bytes - sizeof(uint64_t) * s->H;
      const bool helps = fit <= 1 && (160 * 1024) / std::max<size_t>(tables_only, 1) >= 2 &&
                         trajectories >= static_cast<int64_t>(s->num_cus) * 2048;
      const bool want = where ? std::strcmp(where, "global") == 0 : OSG_MCCFR_TREE_GLOBAL_DEFAULT != 0;
      if (want && helps && split == 0 && s->cfg.solver != 2) {
        tree_global = 1;
        shmem_bytes = tables_only;
        fit = static_cast<int>((160 * 1024) / std::max<size_t>(tables_only, 1));
      }
    }
    int threads, per_cu;
    if (fit <= 1) {
      const int64_t share = (trajectories + s->num_cus - 1) / std::max(s->num_cus, 1);
      threads = static_cast<int>(std::min<int64_t>(1024, std::max<int64_t>(256, (share + 63) / 64 * 64)));
      per_cu = 1;
    } else {
      threads = trajectories >= static_cast<int64_t>(s->num_cus) * 1024 ? 1024 : 256;
      per_cu = std::max(1, std::min(fit, 2048 / threads));
    }
    int64_t groups = std::min<int64_t>((trajectories + threads - 1) / threads, static_cast<int64_t>(s->num_cus) * per_cu);
    ResidentTree rt{reinterpret_cast<const uint2*>(s->d_rec), s->d_uret, s->d_uprob, s->n_uret, s->n_uprob, tree_global};
    unsigned long long*& d_stamps = s->d_mccfr_stamps;   // OSG_MCCFR_STAMPS=1: phase stamps of workgroup 0 (tools/probe_mccfr_shard.py); the solver's own buffer
    if (std::getenv("OSG_MCCFR_STAMPS") && !d_stamps)
      OSG_HIP(hipMalloc(reinterpret_cast<void**>(&d_stamps), sizeof(unsigned long long) * 4));
    const dim3 grid(static_cast<unsigned>(groups)), block(threads);
    const size_t shmem = shmem_bytes;
#define OSG_MCCFR_RES(KA)                                                                                          \
  do {                                                                                                             \
    if (s->cfg.solver == 2)                                                                                        \
      k_os_mccfr_resident<KA><<<grid, block, shmem, st>>>(s->H, s->I, s->P, rt, s->d_nact, s->regrets(), dreg, \
                                                          dpol, seed, first_trajectory, sampled,             \
                                                          s->cfg.epsilon);                                        \
    else if (split == 1 && KA >= 2)                                                                                \
      k_mccfr_resident<(KA >= 2 ? KA : 2), 1><<<grid, block, shmem, st>>>(s->H, s->I, s->P, rt, s->d_nact, s->regrets(), \
                                                       dreg, dpol, seed, first_trajectory, sampled, d_stamps);   \
    else if (split == 2 && KA >= 2)                                                                                \
      k_mccfr_resident<(KA >= 2 ? KA : 2), 2><<<grid, block, shmem, st>>>(s->H, s->I, s->P, rt, s->d_nact, s->regrets(), \
                                                       dreg, dpol, seed, first_trajectory, sampled, d_stamps);   \
    else                                                                                                           \
      k_mccfr_resident_flat<KA><<<grid, block, shmem, st>>>(s->H, s->I, s->P, rt, s->d_nact, s->regrets(), dreg, \
                                                            dpol, seed, first_trajectory, sampled);            \
  } while (0)
    switch (s->A) {
      case 1: OSG_MCCFR_RES(1); break;
      case 2: OSG_MCCFR_RES(2); break;
      case 3: OSG_MCCFR_RES(3); break;
      default: OSG_MCCFR_RES(4); break;
    }
#undef OSG_MCCFR_RES
    OSG_HIP(hipGetLastError());
    s->last_kernel = s->cfg.solver == 2 ? "k_os_mccfr_resident"
                                        : (split == 2 && s->A >= 2 ? "k_mccfr_resident<split 2>"
                                                                   : (split == 1 && s->A >= 2 ? "k_mccfr_resident<split 1>"
                                                                                              : (tree_global ? "k_mccfr_resident_flat<tree in L2>" : "k_mccfr_resident_flat")));
    if (d_stamps && s->cfg.solver != 2 && split != 0) {   // (the flat kernel writes no stamps)
      unsigned long long h[4];
      OSG_HIP(hipMemcpyAsync(h, d_stamps, sizeof h, hipMemcpyDeviceToHost, st));
      OSG_HIP(hipStreamSynchronize(st));
      fprintf(stderr, "k_mccfr_resident (%lld trajectories, %u x %d lanes; workgroup 0, us): staging %.2f  lane 0's trajectory %.2f  "
                      "rest of the workgroup + flush %.2f\n", static_cast<long long>(sampled), grid.x, threads,
              (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0, (h[3] - h[2]) / 100.0);
    }
    return OSG_OK;
  }
  if (s->cfg.solver == 2) {  // OutcomeSamplingMCCFRSolver
    const double eps = s->cfg.epsilon;
    if (use_lds) {
      static bool os_attr_set = false;
      if (!os_attr_set) {
        (void)raise_lds_cap(reinterpret_cast<const void*>(&k_os_mccfr<true>), 64 * 1024);
        os_attr_set = true;
      }
      k_os_mccfr<true><<<dim3(static_cast<unsigned>(blocks)), dim3(256), lds, st>>>(
          s->tree(), s->regrets(), dreg, dpol, seed, first_trajectory, trajectories, eps);
    } else {
      k_os_mccfr<false><<<dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st>>>(
          s->tree(), s->regrets(), dreg, dpol, seed, first_trajectory, trajectories, eps);
    }
    OSG_HIP(hipGetLastError());
    return OSG_OK;
  }
  if (use_lds) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)raise_lds_cap(reinterpret_cast<const void*>(&k_mccfr<true>), 64 * 1024);
      attr_set = true;
    }
    k_mccfr<true><<<dim3(static_cast<unsigned>(blocks)), dim3(256), lds, st>>>(s->tree(), s->regrets(), dreg,
                                                                                dpol, seed, first_trajectory,
                                                                                trajectories);
  } else {
    k_mccfr<false><<<dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st>>>(s->tree(), s->regrets(), dreg,
                                                                               dpol, seed, first_trajectory,
                                                                               trajectories);
  }
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int osg_mccfr_sample(osg_cfr* s, uint64_t seed, int64_t first_trajectory, int64_t trajectories) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_mccfr_sample: bad argument");
  return mccfr_sample_impl(s, seed, first_trajectory, trajectories, s->dreg(), s->dpol());
}

int osg_mccfr_sample_into(osg_cfr* s, uint64_t seed, int64_t first_trajectory, int64_t trajectories, double* d_delta) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_mccfr_sample_into: bad argument");
  if (!d_delta) return mccfr_sample_impl(s, seed, first_trajectory, trajectories, s->dreg(), s->dpol());
  if (reinterpret_cast<uintptr_t>(d_delta) & 7) return set_error(OSG_ERR_INVALID, "osg_mccfr_sample_into: d_delta must be 8-byte aligned");
  return mccfr_sample_impl(s, seed, first_trajectory, trajectories, d_delta, d_delta + static_cast<size_t>(s->I) * s->A);
}

static int mccfr_fold_impl(osg_cfr* s, double* dreg, double* dpol) {
  const int IA = s->I * s->A;
  // AverageType::kFull: the traversals' sampled average-policy terms are not used (external_sampling_mccfr.cc:177);
  // the average policy comes from osg_mccfr_full_average instead
  const int use_policy = (s->average_type == 1 && s->cfg.solver == 1) ? 0 : 1;
  k_fold_deltas<<<dim3((IA + 255) / 256), dim3(256), 0, s->ctx->stream>>>(s->regrets(), s->cum(), dreg, dpol, IA, use_policy);
  OSG_HIP(hipGetLastError());
  const int slot = delta_slot(s, dreg);
  if (slot >= 0 && dpol == dreg + IA) s->delta_clean[slot] = true;   // the fold left them zero
  ++s->iteration;
  return OSG_OK;
}

int osg_mccfr_apply_deltas(osg_cfr* s) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_mccfr_apply_deltas: null argument");
  return mccfr_fold_impl(s, s->dreg(), s->dpol());
}

int osg_mccfr_spare_delta_buffer(osg_cfr* s, int which, double** d_delta) {
  if (!s || !d_delta || (which != 0 && which != 1)) return set_error(OSG_ERR_INVALID, "osg_mccfr_spare_delta_buffer: bad argument");
  if (!s->d_spare_delta[which]) {
    const size_t bytes = sizeof(double) * 2 * static_cast<size_t>(s->I) * s->A;
    OSG_HIP(hipSetDevice(s->ctx->device));
    OSG_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_spare_delta[which]), bytes));
    OSG_HIP(hipMemsetAsync(s->d_spare_delta[which], 0, bytes, s->ctx->stream));
  }
  *d_delta = s->d_spare_delta[which];
  return OSG_OK;
}

int osg_mccfr_apply_deltas_from(osg_cfr* s, double* d_delta) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_mccfr_apply_deltas_from: null argument");
  if (!d_delta) return mccfr_fold_impl(s, s->dreg(), s->dpol());
  return mccfr_fold_impl(s, d_delta, d_delta + static_cast<size_t>(s->I) * s->A);
}

int osg_mccfr_set_average_type(osg_cfr* s, int average_type) {
  if (!s || (average_type != 0 && average_type != 1)) return set_error(OSG_ERR_INVALID, "osg_mccfr_set_average_type: 0 (kSimple) or 1 (kFull)");
  if (s->cfg.solver != 1) return set_error(OSG_ERR_INVALID, "osg_mccfr_set_average_type: external-sampling solvers only");
  if (average_type == 1 && (s->B != 1 || s->A > kMaxPolicyRow))
    return set_error(OSG_ERR_UNSUPPORTED, "osg_mccfr_set_average_type: kFull needs one solver per object and policy rows of at most 8 actions");
  s->average_type = average_type;
  return OSG_OK;
}

int osg_mccfr_sample_uniforms(osg_cfr* s, int player, const double* h_uniforms, int n, int32_t* consumed) {
  if (!s || !h_uniforms || n < 1 || !consumed) return set_error(OSG_ERR_INVALID, "osg_mccfr_sample_uniforms: bad argument");
  if (s->cfg.solver != 1) return set_error(OSG_ERR_INVALID, "osg_mccfr_sample_uniforms: external-sampling solvers only");
  if (player < 0 || player >= s->P) return set_error(OSG_ERR_INVALID, "osg_mccfr_sample_uniforms: no such player");
  if (s->A > kMaxA) return set_error(OSG_ERR_UNSUPPORTED, "osg_mccfr_sample_uniforms: decision nodes wider than 4 actions");
  if (s->B != 1) return set_error(OSG_ERR_UNSUPPORTED, "osg_mccfr_sample_uniforms: one solver per object");
  const int IA = s->I * s->A;
  hipStream_t st = s->ctx->stream;
  void* scratch = nullptr;
  const size_t bytes = sizeof(double) * static_cast<size_t>(n);
  int rc = osg_ctx_scratch(s->ctx, bytes + 256, &scratch);
  if (rc) return rc;
  double* d_u = static_cast<double*>(scratch);
  int32_t* d_used = reinterpret_cast<int32_t*>(static_cast<char*>(scratch) + ((bytes + 15) & ~static_cast<size_t>(15)));
  OSG_HIP(hipMemcpyAsync(d_u, h_uniforms, bytes, hipMemcpyHostToDevice, st));
  OSG_HIP(hipMemsetAsync(d_used, 0, sizeof(int32_t), st));
  OSG_HIP(hipMemsetAsync(s->dreg(), 0, sizeof(double) * 2 * IA, st));
  s->delta_clean[0] = false;
  // trajectory index == player: the traverser is index mod P
  k_mccfr<false, true><<<dim3(1), dim3(64), 0, st>>>(s->tree(), s->regrets(), s->dreg(), s->dpol(), 0, player, 1, d_u, n, d_used);
  OSG_HIP(hipGetLastError());
  OSG_HIP(hipMemcpyAsync(consumed, d_used, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  OSG_HIP(hipStreamSynchronize(st));
  if (*consumed > n) return set_error(OSG_ERR_INVALID, "osg_mccfr_sample_uniforms: the traversal needed more uniforms than were supplied");
  return OSG_OK;
}

int osg_mccfr_full_average(osg_cfr* s, double weight) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_mccfr_full_average: null argument");
  if (s->B != 1) return set_error(OSG_ERR_UNSUPPORTED, "osg_mccfr_full_average: one solver per object");
  if (s->A > kMaxPolicyRow) return set_error(OSG_ERR_UNSUPPORTED, "osg_mccfr_full_average: policy rows wider than 8 actions");
  if (!(weight > 0.0)) return set_error(OSG_ERR_INVALID, "osg_mccfr_full_average: weight must be positive");
  int threads = ((s->max_level_width + 63) / 64) * 64;
  threads = std::max(64, std::min(threads, 1024));
  k_mccfr_full_average<<<dim3(1), dim3(threads), 0, s->ctx->stream>>>(s->tree(), s->regrets(), s->cum(), s->d_reach, weight);
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int osg_cfr_br_iterate(osg_cfr* s, int iters) {
  if (!s || iters < 0) return set_error(OSG_ERR_INVALID, "osg_cfr_br_iterate: bad argument");
  if (s->cfg.solver != 0) return set_error(OSG_ERR_INVALID, "osg_cfr_br_iterate: needs a CFRSolverBase table (solver 0)");
  if (s->cfg.linear_averaging || s->cfg.regret_matching_plus)
    return set_error(OSG_ERR_INVALID, "osg_cfr_br_iterate: CFRBRSolver is plain CFR (cfr_br.cc:23-29)");
  if (s->B != 1) return set_error(OSG_ERR_UNSUPPORTED, "osg_cfr_br_iterate: one solver per object");
  if (!s->eval_ok) return set_error(OSG_ERR_UNSUPPORTED, "an information state spans several tree levels");
  if (int rc = cfr_sub_error(s)) return rc;
  const size_t M = s->mem.size();
  EvalArrays ea;
  ea.path_off = s->d_path_off; ea.path = s->d_path; ea.info_level = s->d_info_level; ea.mem_index = s->d_mem_index;
  ea.M = static_cast<int>(M);
  ea.value = s->d_eval;
  ea.brv = ea.value + static_cast<size_t>(s->H) * s->P;
  ea.cf = ea.brv + s->H;
  ea.out = ea.cf + M;
  ea.best = s->d_best;
  int threads = ((s->max_level_width + 63) / 64) * 64;
  threads = std::max(64, std::min(threads, 1024));
  osg_cfr_cfg cfg = s->cfg;
  cfg.alternating_updates = 0;
  Tables tb{s->regrets(), s->cum(), s->cur()};
  hipStream_t st = s->ctx->stream;
  // every player's best response to the current policy (cfr_br.cc:55-68), then one regret / average-policy
  // pass per player against the others' best responses (cfr_br.cc:70-81) and ApplyRegretMatching
  const bool jobs = s->jobs_ok && OSG_EVAL_JOBS_ENABLED();
  const bool split = s->split_ok && s->split_br_ok && (s->cfg.kernel == 0 || s->cfg.kernel == 4);
  SmallTree stree{s->d_path_off, s->d_path, static_cast<int>(M), static_cast<int>(s->path.size())};
  SplitTree sp{s->split_G, s->split_L, s->split_NL, s->split_NM, s->split_NI, s->d_split_nloc, s->d_split_desc,
               s->d_split_fc, s->d_split_row, s->d_split_glob, s->d_split_mem_m, s->d_split_mem_hloc, s->d_split_info,
               s->d_split_terms, s->d_split_bar, s->h_sub_err};
  if (!split && !jobs && eval_takes_the_grid(s) && s->path_kernel) {
    // large trees (3-player leduc: one workgroup walked a pass set in tens of milliseconds): the evaluation's sweep leaves
    // every infostate's best-response action, then each player's pass runs as the launch-per-phase CFR pass
    // (k_gcfr_*) on the effective policy — the same additions in the same order as k_cfr<., kBr>
    GridCfr g;
    g.t = s->tree(); g.path_off = s->d_path_off; g.path = s->d_path; g.meta = s->d_meta32;
    g.info_player = s->d_info_player32; g.value = s->d_value; g.dreg = s->d_node_delta;
    g.dpol = s->d_node_delta + M * s->A; g.skip = s->d_skip; g.tb = tb; g.M = static_cast<int>(M);
    double* d_eff = ea.out + 2 * s->P;   // (the evaluation's policy scratch: [I, A], free here)
    g.pol = d_eff;
    auto blocks = [](int n) { return dim3(static_cast<unsigned>((n + 255) / 256)); };
    k_gcfr_init_values<<<blocks(s->H), dim3(256), 0, st>>>(g);
    for (int it = 0; it < iters; ++it) {
      if (int rc = launch_grid_eval(s, ea, s->cur(), false, nullptr, true)) return rc;
      for (int upd = 0; upd < s->P; ++upd) {
        k_gcfr_effpol<<<blocks(s->I), dim3(256), 0, st>>>(g, upd, s->d_best, d_eff);
        for (int l = s->D - 2; l >= 0; --l) {
          const int begin = s->level_off[l], end = s->level_off[l + 1];
          k_gcfr_level<<<blocks(end - begin), dim3(256), 0, st>>>(g, begin, end, upd, upd + 1);
        }
        k_gcfr_members<<<blocks(static_cast<int>(M)), dim3(256), 0, st>>>(g, upd, s->iteration + 1, cfg);
        k_gcfr_fold<<<blocks(s->I * 64), dim3(256), 0, st>>>(g, upd, cfg);
      }
      ++s->iteration;
    }
    OSG_HIP(hipGetLastError());
    s->last_kernel = "k_gcfr<br>";
    return OSG_OK;
  }
  for (int it = 0; it < iters; ++it) {
    if (jobs) k_eval_jobs<<<dim3(s->jobs_J), dim3(s->jobs_threads), s->jobs_lds_bytes, st>>>(s->tree(), ea, eval_jobs_of(s), s->cur(), 1, 1);
    else if (eval_takes_the_grid(s)) { if (int rc = launch_grid_eval(s, ea, s->cur(), false, nullptr, true)) return rc; }
    else k_policy_eval<<<dim3(1), dim3(threads), 0, st>>>(s->tree(), ea, s->cur());
    if (split) {
      const int rc = launch_split(s, stree, sp, tb, 1, s->iteration, cfg, true);
      if (rc) return rc;
    } else {
      k_cfr<false, true><<<dim3(1), dim3(threads), 0, st>>>(s->tree(), tb, s->d_reach, s->d_value, 1, s->iteration, cfg,
                                                            s->d_best);
    }
    ++s->iteration;
  }
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int osg_mccfr_iterate(osg_cfr* s, uint64_t seed, int64_t first_trajectory, int64_t trajectories) {
  int rc = osg_mccfr_sample(s, seed, first_trajectory, trajectories);
  if (rc) return rc;
  rc = osg_mccfr_apply_deltas(s);
  if (rc) return rc;
  // AverageType::kFull: T trajectories stand for T / P iterations, each followed by one FullUpdateAverage
  if (s->average_type == 1 && s->cfg.solver == 1 && trajectories >= s->P)
    return osg_mccfr_full_average(s, static_cast<double>(trajectories / s->P));
  return OSG_OK;
}

int osg_cfr_table_ptrs(osg_cfr* s, double** d_regrets, double** d_cum_policy, double** d_cur_policy) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_cfr_table_ptrs: null argument");
  if (int rc = cfr_sub_error(s)) return rc;
  if (d_regrets) *d_regrets = s->regrets();
  if (d_cum_policy) *d_cum_policy = s->cum();
  if (d_cur_policy) *d_cur_policy = s->cur();
  return OSG_OK;
}

int osg_mccfr_delta_ptrs(osg_cfr* s, double** d_regret_delta, double** d_policy_delta) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_mccfr_delta_ptrs: null argument");
  if (d_regret_delta) *d_regret_delta = s->dreg();
  if (d_policy_delta) *d_policy_delta = s->dpol();
  return OSG_OK;
}

int osg_cfr_upload_tables(osg_cfr* s, const double* h_regrets, const double* h_cum_policy, const double* h_cur_policy) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_cfr_upload_tables: null argument");
  const size_t bytes = sizeof(double) * s->I * s->A;
  hipStream_t st = s->ctx->stream;
  if (h_regrets) OSG_HIP(hipMemcpyAsync(s->regrets(), h_regrets, bytes, hipMemcpyHostToDevice, st));
  if (h_cum_policy) OSG_HIP(hipMemcpyAsync(s->cum(), h_cum_policy, bytes, hipMemcpyHostToDevice, st));
  if (h_cur_policy) OSG_HIP(hipMemcpyAsync(s->cur(), h_cur_policy, bytes, hipMemcpyHostToDevice, st));
  OSG_HIP(hipStreamSynchronize(st));
  return OSG_OK;
}

int osg_cfr_tables(const osg_cfr* s, int32_t* nact, int32_t* legal, double* regrets, double* cum_policy,
                   double* cur_policy, double* avg_policy) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_cfr_tables: null argument");
  const size_t IA = static_cast<size_t>(s->I) * s->A, bytes = sizeof(double) * IA;
  hipStream_t st = s->ctx->stream;
  if (nact) memcpy(nact, s->nact.data(), sizeof(int32_t) * s->I);
  if (legal) memcpy(legal, s->legal.data(), sizeof(int32_t) * IA);
  std::vector<double> cum(IA);
  if (regrets) OSG_HIP(hipMemcpyAsync(regrets, s->regrets(), bytes, hipMemcpyDeviceToHost, st));
  if (cur_policy) OSG_HIP(hipMemcpyAsync(cur_policy, s->cur(), bytes, hipMemcpyDeviceToHost, st));
  OSG_HIP(hipMemcpyAsync(cum.data(), s->cum(), bytes, hipMemcpyDeviceToHost, st));
  unsigned int split_error = 0;
  if (s->split_ok) OSG_HIP(hipMemcpyAsync(&split_error, s->d_split_bar + 2, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
  OSG_HIP(hipStreamSynchronize(st));
  if (split_error)
    return set_error(OSG_ERR_HIP, "k_cfr_split: a grid barrier timed out after seconds (the launch is cooperative: this is a hung "
                                  "device, not contention); the tables are not usable — osg_cfr_cfg.kernel = 3 runs one workgroup");
  if (cum_policy) memcpy(cum_policy, cum.data(), bytes);
  if (avg_policy) {  // CFRAveragePolicy::GetStatePolicyFromInformationStateValues (cfr.cc:104-125)
    for (int i = 0; i < s->I; ++i) {
      const int n = s->nact[i];
      double sum = 0.0;
      for (int a = 0; a < n; ++a) sum += cum[i * s->A + a];
      for (int a = 0; a < s->A; ++a) {
        if (a >= n) avg_policy[i * s->A + a] = 0.0;
        else avg_policy[i * s->A + a] = sum == 0.0 ? 1. / n : cum[i * s->A + a] / sum;
      }
    }
  }
  return OSG_OK;
}

int osg_cfr_infostate_key(const osg_cfr* s, int64_t i, char* buf, int cap) {
  if (!s || i < 0 || i >= s->I || !buf || cap <= 0) return set_error(OSG_ERR_INVALID, "osg_cfr_infostate_key: bad argument");
  const std::string& k = s->keys[i];
  if (static_cast<int>(k.size()) + 1 > cap) return set_error(OSG_ERR_INVALID, "osg_cfr_infostate_key: buffer too small");
  memcpy(buf, k.c_str(), k.size() + 1);
  return static_cast<int>(k.size());
}

int osg_cfr_iteration(const osg_cfr* s) { return s ? s->iteration : 0; }
const char* osg_cfr_last_kernel(const osg_cfr* s) { return s ? s->last_kernel : ""; }
int osg_cfr_infostate_player(const osg_cfr* s, int64_t i) { return (s && i >= 0 && i < s->I) ? s->info_player[i] : -1; }
int osg_cfr_replicas(const osg_cfr* s) { return s ? s->B : 0; }
int osg_cfr_select_replica(osg_cfr* s, int replica) {
  if (!s || replica < 0 || replica >= s->B) return set_error(OSG_ERR_INVALID, "osg_cfr_select_replica: bad replica");
  s->selected = replica;
  return OSG_OK;
}
int osg_cfr_set_iteration(osg_cfr* s, int iteration) {
  if (!s || iteration < 0) return set_error(OSG_ERR_INVALID, "osg_cfr_set_iteration: bad argument");
  s->iteration = iteration;
  return OSG_OK;
}

static int evaluate_policy_impl(osg_cfr* s, int which_policy, const double* h_policy, double* expected_returns,
                                double* best_response_values, double* nash_conv, double* exploitability,
                                int keep_responder, double* h_history_values);

int osg_cfr_evaluate_policy(osg_cfr* s, int which_policy, const double* h_policy, double* expected_returns,
                            double* best_response_values, double* nash_conv, double* exploitability) {
  return evaluate_policy_impl(s, which_policy, h_policy, expected_returns, best_response_values, nash_conv, exploitability,
                              -1, nullptr);
}

int osg_cfr_best_response_history_values(osg_cfr* s, int which_policy, const double* h_policy, int responder,
                                         double* h_history_values) {
  if (!s || !h_history_values) return set_error(OSG_ERR_INVALID, "osg_cfr_best_response_history_values: null argument");
  if (responder < 0 || responder >= s->P) return set_error(OSG_ERR_INVALID, "osg_cfr_best_response_history_values: no such player");
  return evaluate_policy_impl(s, which_policy, h_policy, nullptr, nullptr, nullptr, nullptr, responder, h_history_values);
}

int osg_cfr_tree_edges(const osg_cfr* s, int32_t* parent, int32_t* action) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_cfr_tree_edges: null solver");
  if (parent) memcpy(parent, s->parent.data(), sizeof(int32_t) * s->H);
  if (action) memcpy(action, s->edge_action.data(), sizeof(int32_t) * s->H);
  return OSG_OK;
}

static int evaluate_policy_impl(osg_cfr* s, int which_policy, const double* h_policy, double* expected_returns,
                                double* best_response_values, double* nash_conv, double* exploitability,
                                int keep_responder, double* h_history_values) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_cfr_evaluate_policy: null solver");
  if (!s->eval_ok) return set_error(OSG_ERR_UNSUPPORTED, "an information state spans several tree levels");
  const size_t IA = static_cast<size_t>(s->I) * s->A, M = s->mem.size();
  const int P = s->P;
  hipStream_t st = s->ctx->stream;
  if (which_policy < 0 || which_policy > 2) return set_error(OSG_ERR_INVALID, "osg_cfr_evaluate_policy: which_policy must be 0, 1 or 2");
  if (which_policy != 2)
    if (int rc = cfr_sub_error(s)) return rc;
  if (which_policy == 2 && !h_policy) return set_error(OSG_ERR_INVALID, "osg_cfr_evaluate_policy: which_policy == 2 needs h_policy");
  EvalArrays ea;
  ea.path_off = s->d_path_off; ea.path = s->d_path; ea.info_level = s->d_info_level; ea.mem_index = s->d_mem_index;
  ea.M = static_cast<int>(M);
  ea.value = s->d_eval;
  ea.brv = ea.value + static_cast<size_t>(s->H) * P;
  ea.cf = ea.brv + s->H;
  double* d_pol = ea.cf + M + 2 * P;
  // the 2 P results land in pinned host memory straight from the kernel (the device address of h_eval_out): the call is a
  // launch and a wait — no copy-back launches (two of them were ~10 us of a 45 us call)
  OSG_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&ea.out), s->h_eval_out, 0));
  ea.best = s->d_best;
  if (h_history_values) {  // the responder's value of every history: kept in d_reach ([H, P + 1] doubles, free here)
    ea.keep = s->d_reach;
    ea.keep_r = keep_responder;
  }
  if (s->jobs_ok && OSG_EVAL_JOBS_ENABLED()) {
    // the tables never leave the device: the average policy is formed from the cumulative table inside the jobs
    const double* src = which_policy == 0 ? s->cum() : which_policy == 1 ? s->cur() : d_pol;
    if (which_policy == 2) OSG_HIP(hipMemcpyAsync(d_pol, h_policy, sizeof(double) * IA, hipMemcpyHostToDevice, st));
    k_eval_jobs<<<dim3(s->jobs_J), dim3(s->jobs_threads), s->jobs_lds_bytes, st>>>(s->tree(), ea, eval_jobs_of(s), src,
                                                                                   which_policy == 0 ? 0 : 1, 0);
    OSG_HIP(hipGetLastError());
  } else if (eval_takes_the_grid(s)) {
    const double* src = which_policy == 0 ? s->cum() : which_policy == 1 ? s->cur() : d_pol;
    if (which_policy == 2) OSG_HIP(hipMemcpyAsync(d_pol, h_policy, sizeof(double) * IA, hipMemcpyHostToDevice, st));
    if (int rc = launch_grid_eval(s, ea, src, which_policy == 0, d_pol, false)) return rc;
  } else {
    const double* src = which_policy == 0 ? s->cum() : which_policy == 1 ? s->cur() : d_pol;
    if (which_policy == 2) OSG_HIP(hipMemcpyAsync(d_pol, h_policy, sizeof(double) * IA, hipMemcpyHostToDevice, st));
    int threads = ((s->max_level_width + 63) / 64) * 64;
    threads = std::max(64, std::min(threads, 1024));
    k_policy_eval<<<dim3(1), dim3(threads), 0, st>>>(s->tree(), ea, src, which_policy == 0 ? 1 : 0, d_pol);
    OSG_HIP(hipGetLastError());
  }
  if (h_history_values)
    OSG_HIP(hipMemcpyAsync(h_history_values, s->d_reach, sizeof(double) * s->H, hipMemcpyDeviceToHost, st));
  OSG_HIP(hipStreamSynchronize(st));
  if (which_policy != 2)   // the tables are only as good as the launches that wrote them (the kernels raise the pinned word)
    if (int rc = cfr_sub_error(s)) return rc;
  const double* out = s->h_eval_out;
  double nc = 0.0, total_br = 0.0;
  for (int p = 0; p < P; ++p) {
    if (expected_returns) expected_returns[p] = out[p];
    if (best_response_values) best_response_values[p] = out[P + p];
    nc += out[P + p] - out[p];   // NashConv (tabular_exploitability.cc:60-89)
    total_br += out[P + p];
  }
  if (nash_conv) *nash_conv = nc;
  // Exploitability = (sum of best-response values - UtilitySum) / P (tabular_exploitability.cc:30-47);
  // kuhn_poker and leduc_poker are zero-sum: UtilitySum() == 0.
  if (exploitability) *exploitability = total_br / P;
  return OSG_OK;
}

int osg_cfr_best_response(osg_cfr* s, int which_policy, const double* h_policy, int32_t* h_best_index,
                          double* best_response_values) {
  if (!s || !h_best_index) return set_error(OSG_ERR_INVALID, "osg_cfr_best_response: null argument");
  int rc = osg_cfr_evaluate_policy(s, which_policy, h_policy, nullptr, best_response_values, nullptr, nullptr);
  if (rc) return rc;
  OSG_HIP(hipMemcpyAsync(h_best_index, s->d_best, sizeof(int32_t) * s->I, hipMemcpyDeviceToHost, s->ctx->stream));
  OSG_HIP(hipStreamSynchronize(s->ctx->stream));
  return OSG_OK;
}

int osg_information_state_string(const osg_batch* b, int64_t index, int player, char* buf, int cap) {
  if (!b || !buf || cap <= 0 || index < 0 || index >= b->n)
    return set_error(OSG_ERR_INVALID, "osg_information_state_string: bad argument");
  const osg_game_desc& d = b->spec.desc;
  if (d.game_kind != kKuhn && d.game_kind != kLeduc)
    return set_error(OSG_ERR_INVALID, "this game provides no information state string");
  if (player < 0 || player >= d.num_players) return set_error(OSG_ERR_INVALID, "player id out of range");
  uint64_t w[5] = {0, 0, 0, 0, 0};   // (leduc_poker with 4+ players: five plane words)
  const char* base = static_cast<const char*>(b->d_words);
  for (int k = 0; k < d.state_words; ++k)
    OSG_HIP(hipMemcpyAsync(&w[k], base + (static_cast<size_t>(k) * b->n + index) * sizeof(uint64_t), sizeof(uint64_t),
                           hipMemcpyDeviceToHost, b->ctx->stream));
  OSG_HIP(hipStreamSynchronize(b->ctx->stream));
  const std::string key = d.game_kind == kKuhn ? kuhn_key(b->spec.kuhn, w[0], player) : leduc_key_words(b->spec, w, player);
  if (static_cast<int>(key.size()) + 1 > cap) return set_error(OSG_ERR_INVALID, "buffer too small");
  memcpy(buf, key.c_str(), key.size() + 1);
  return static_cast<int>(key.size());
}

int osg_observation_string(const osg_batch* b, int64_t index, int player, char* buf, int cap) {
  if (!b || !buf || cap <= 0 || index < 0 || index >= b->n)
    return set_error(OSG_ERR_INVALID, "osg_observation_string: bad argument");
  const osg_game_desc& d = b->spec.desc;
  if (player < 0 || player >= d.num_players) return set_error(OSG_ERR_INVALID, "player id out of range");
  uint64_t w[4 * 12 + 1] = {0};   // (hex 19 x 19: 4 planes of 12 words and the meta word)
  const char* base = static_cast<const char*>(b->d_words);
  for (int k = 0; k < d.state_words; ++k)
    OSG_HIP(hipMemcpyAsync(&w[k], base + (static_cast<size_t>(k) * b->n + index) * d.state_word_bytes, d.state_word_bytes,
                           hipMemcpyDeviceToHost, b->ctx->stream));
  OSG_HIP(hipStreamSynchronize(b->ctx->stream));
  std::string out;
  switch (d.game_kind) {
    case kTtt: {  // tic_tac_toe.cc:163-175: rows joined by newlines
      const uint32_t x = static_cast<uint32_t>(w[0]) & 0x1FFu, o = (static_cast<uint32_t>(w[0]) >> 16) & 0x1FFu;
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out += (x >> (3 * r + c) & 1u) ? "x" : ((o >> (3 * r + c) & 1u) ? "o" : ".");
        if (r < 2) out += "\n";
      }
      break;
    }
    case kC4: {  // connect_four.cc:212-222: top row first, every row ends with a newline
      const int R = b->spec.c4.rows, Cn = b->spec.c4.cols;
      osg_u128 x = b->spec.c4_std ? (w[0] & ((1ull << 56) - 1ull)) : w[0], o = w[1];
      if (b->spec.c4_wide) {   // two plane words per colour: x.lo, x.hi, o.lo, o.hi
        x = (static_cast<osg_u128>(w[1]) << 64) | w[0];
        o = (static_cast<osg_u128>(w[3]) << 64) | w[2];
      }
      for (int r = R - 1; r >= 0; --r) {
        for (int c = 0; c < Cn; ++c) {
          const int bit = c * (R + 1) + r;
          out += static_cast<uint32_t>(x >> bit & 1) ? "x" : (static_cast<uint32_t>(o >> bit & 1) ? "o" : ".");
        }
        out += "\n";
      }
      break;
    }
    case kHex: {  // hex.cc:341-359: one line per row, indented by the row number, a space after every cell
      const int NW = b->spec.hex_nw;
      int cols = 0, rows_unused = 0, cells = 0;
      hex_dims(b->spec, &rows_unused, &cols, &cells);
      auto bit = [&](int plane, int cell) { return (w[plane * NW + (cell >> 5)] >> (cell & 31)) & 1ull; };
      int line = 0;
      for (int i = 0; i < cells; ++i) {
        if (i && i % cols == 0) {
          out += "\n";
          out += std::string(++line, ' ');
        }
        const bool black = bit(0, i), white = bit(1, i);
        const int mag = 1 + 2 * static_cast<int>(bit(2, i)) + static_cast<int>(bit(3, i));  // plain, B, A, win
        if (!black && !white) out += ".";
        else if (!b->spec.hex_explicit) out += black ? "x" : "o";
        else out += black ? (mag == 1 ? "x" : mag == 2 ? "z" : mag == 3 ? "y" : "X")   // hex.cc:173-227
                          : (mag == 1 ? "o" : mag == 2 ? "q" : mag == 3 ? "p" : "O");
        out += " ";
      }
      break;
    }
    case kKuhn: {  // kuhn_poker.cc:109-166, default observer: own card, then every player's contribution
      const Kuhn::Params& kp = b->spec.kuhn;
      Kuhn::State st{w[0]};
      if (Kuhn::len(st) > player) {
        out += std::to_string(Kuhn::card(st, player));
        for (int q = 0; q < kp.players; ++q) out += std::to_string(Kuhn::did_bet(kp, st, q) ? 2 : 1);
      }
      break;
    }
    case kLeduc: {  // leduc_poker.cc:198-239, imperfect recall: pot contributions instead of the sequences
      const LeducParams& lp = b->spec.leduc;
      auto text = [&](auto tag, const auto& st) {
        using L = typename decltype(tag)::type;
        std::string r = leduc_observer_prefix<L>(lp, st, player) + "[Ante: ";
        for (int q = 0; q < lp.players; ++q) r += (q ? " " : "") + std::to_string(L::ante(st, q));
        return r + "]";
      };
      if (b->spec.leduc_big) out = text(TypeTag<LeducBig>{}, LeducBig::unpack5(w[0], w[1], w[2], w[3], w[4]));
      else out = text(TypeTag<Leduc>{}, Leduc::unpack(w[0], w[1]));
      break;
    }
    default: return set_error(OSG_ERR_INVALID, "bad game kind");
  }
  if (static_cast<int>(out.size()) + 1 > cap) return set_error(OSG_ERR_INVALID, "buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return static_cast<int>(out.size());
}

namespace {
// The packed words of one state, on the host.
int fetch_state_words(const osg_batch* b, int64_t index, uint64_t* w) {
  const osg_game_desc& d = b->spec.desc;
  const char* base = static_cast<const char*>(b->d_words);
  for (int k = 0; k < d.state_words; ++k)
    OSG_HIP(hipMemcpyAsync(&w[k], base + (static_cast<size_t>(k) * b->n + index) * d.state_word_bytes, d.state_word_bytes,
                           hipMemcpyDeviceToHost, b->ctx->stream));
  OSG_HIP(hipStreamSynchronize(b->ctx->stream));
  return OSG_OK;
}
const char* leduc_action_name(int a) { return a == 0 ? "Fold" : (a == 1 ? "Call" : "Raise"); }  // leduc_poker.cc:869-875
void hex_geometry(const GameSpec& spec, int* cols, int* rows, int* cells) { hex_dims(spec, rows, cols, cells); }
int return_string(const std::string& out, char* buf, int cap) {
  if (static_cast<int>(out.size()) + 1 > cap) return set_error(OSG_ERR_INVALID, "buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return static_cast<int>(out.size());
}
}  // namespace

int osg_state_string(const osg_batch* b, int64_t index, char* buf, int cap) {
  if (!b || !buf || cap <= 0 || index < 0 || index >= b->n)
    return set_error(OSG_ERR_INVALID, "osg_state_string: bad argument");
  const osg_game_desc& d = b->spec.desc;
  if (d.game_kind == kTtt || d.game_kind == kC4 || d.game_kind == kHex)
    return osg_observation_string(b, index, 0, buf, cap);  // ObservationString is ToString() in these games
  uint64_t w[4 * 12 + 1] = {0};   // (hex 19 x 19: 4 planes of 12 words and the meta word)
  int rc = fetch_state_words(b, index, w);
  if (rc) return rc;
  std::string out;
  if (d.game_kind == kKuhn) {  // kuhn_poker.cc:253-268: the dealt cards, then p / b
    const Kuhn::Params& kp = b->spec.kuhn;
    Kuhn::State st{w[0]};
    const int h = Kuhn::len(st);
    for (int i = 0; i < h && i < kp.players; ++i) out += (i ? " " : "") + std::to_string(Kuhn::card(st, i));
    if (h > kp.players) out += ' ';
    for (int j = 0; j < Kuhn::nact(kp, st); ++j) out.push_back(((Kuhn::bets(st) >> j) & 1u) ? 'b' : 'p');
  } else {  // leduc_poker.cc:463-496
    const LeducParams& lp = b->spec.leduc;
    auto text = [&](auto tag, const auto& st) {
      using L = typename decltype(tag)::type;
      const bool term = L::terminal(lp, st);
      double ret[kMaxPlayers] = {0};
      if (term) L::returns(lp, st, ret);
      const int P = lp.players;
      auto card = [](int c) { return std::to_string(c == L::kNone ? -10000 : c); };
      std::string r = "Round: " + std::to_string(st.round) + "\nPlayer: " + std::to_string(st.cur) + "\nPot: " +
                      std::to_string(term ? 0 : st.pot) + "\nMoney (player_0 player_1" + (P > 2 ? " [...]):" : "):");
      for (int q = 0; q < P; ++q) {
        char num[32];
        snprintf(num, sizeof num, "%g", term ? 100.0 + ret[q] : 100.0 - L::ante(st, q));
        r += std::string(" ") + num;
      }
      r += std::string("\nCards (public player_0 player_1") + (P > 2 ? " [...]): " : "): ") + card(st.pub) + " ";
      for (int q = 0; q < P; ++q) r += card(L::priv(st, q)) + " ";
      for (int round = 0; round < 2; ++round) {
        r += round == 0 ? "\nRound 1 sequence: " : "\nRound 2 sequence: ";
        for (int k = 0; k < L::seqlen(st, round); ++k)
          r += std::string(k ? ", " : "") + leduc_action_name(static_cast<int>((L::seq(st, round) >> (2 * k)) & 3u));
      }
      return r + "\n";
    };
    if (b->spec.leduc_big) out = text(TypeTag<LeducBig>{}, LeducBig::unpack5(w[0], w[1], w[2], w[3], w[4]));
    else out = text(TypeTag<Leduc>{}, Leduc::unpack(w[0], w[1]));
  }
  return return_string(out, buf, cap);
}

int osg_action_string(const osg_batch* b, int64_t index, int player, int32_t action, char* buf, int cap) {
  if (!b || !buf || cap <= 0 || index < 0 || index >= b->n)
    return set_error(OSG_ERR_INVALID, "osg_action_string: bad argument");
  const osg_game_desc& d = b->spec.desc;
  if (player < -1 || player >= d.num_players) return set_error(OSG_ERR_INVALID, "player id out of range");
  std::string out;
  switch (d.game_kind) {
    case kTtt:  // tic_tac_toe.cc:266-270
      out = std::string(player == 0 ? "x" : "o") + "(" + std::to_string(action / 3) + "," + std::to_string(action % 3) + ")";
      break;
    case kC4:  // connect_four.cc:158-161
      out = std::string(player == 0 ? "x" : "o") + std::to_string(action);
      break;
    case kHex: {  // hex.cc:295-314 (its "row" is the column letter)
      int cols, rows, cells;
      hex_geometry(b->spec, &cols, &rows, &cells);
      if (action == cells) { out = "swap"; break; }
      const int x = action % cols, y = action / cols;
      // Always the standard form, also for string_rep=explicit: hex.cc:301 tests `StringRep() == StringRep::kStandard`,
      // a value-initialised enum (= kStandard), not the state's string_rep(), so the reference never reaches its
      // explicit branch (:307-310).  Checked against the running reference (tests/test_oracle_vs_reference.py);
      // the board string (ToString, hex.cc:341-359) does honour string_rep.
      out = std::string(1, static_cast<char>('a' + x)) + std::to_string(y + 1);
      break;
    }
    case kKuhn:  // kuhn_poker.cc:244-251
      out = player < 0 ? "Deal:" + std::to_string(action) : std::string(action == 0 ? "Pass" : "Bet");
      break;
    case kLeduc:  // leduc_poker.cc:459-461
      out = player < 0 ? "Chance outcome:" + std::to_string(action) : std::string(leduc_action_name(action));
      break;
    default: return set_error(OSG_ERR_INVALID, "bad game kind");
  }
  return return_string(out, buf, cap);
}

}  // extern "C"
