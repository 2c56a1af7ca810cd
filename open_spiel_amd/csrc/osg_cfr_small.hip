// The one-workgroup CFR kernels: k_cfr (any tree, reach + value passes) and k_cfr_small (path-based; the whole
// problem in LDS for small trees: kuhn_poker, BASELINE config 3).  See osg_cfr_internal.h for the family map.
#include "osg_cfr_internal.h"

namespace {

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


}  // namespace

namespace osg_cfr_impl {

// Whole solver state in LDS when it fits (gfx950: 160 KiB per workgroup; leave headroom): the caps of the instantiations
// a launch may pick are raised once, at creation; a refusal falls back to the global-memory forms.
void cfr_small_prepare(osg_cfr* s) {
  if (s->lds_resident) {
    hipError_t e = raise_lds_cap(reinterpret_cast<const void*>(&k_cfr<true>), static_cast<int>(s->lds_bytes));
    if (e != hipSuccess) { (void)hipGetLastError(); s->lds_resident = false; }
  }
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
    hipError_t e = hipSuccess;
      for (const void* f : variants)
        if (e == hipSuccess)
          e = raise_lds_cap(f, static_cast<int>(s->small_lds_bytes));
      if (e != hipSuccess) { (void)hipGetLastError(); s->small_tree = false; }
  }
}

// The one-workgroup kernels (osg_cfr_iterate's last branch): the path-based kernel, all-in-LDS when the tree is small
// enough, else the general kernel.
int cfr_small_iterate(osg_cfr* s, Tables tb, int iters, int threads, unsigned grid_b) {
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
  return OSG_OK;
}

void cfr_general_br_pass(osg_cfr* s, Tables tb, int threads, osg_cfr_cfg cfg) {
  k_cfr<false, true><<<dim3(1), dim3(threads), 0, s->ctx->stream>>>(s->tree(), tb, s->d_reach, s->d_value, 1, s->iteration, cfg,
                                                                    s->d_best);
}

}  // namespace osg_cfr_impl
