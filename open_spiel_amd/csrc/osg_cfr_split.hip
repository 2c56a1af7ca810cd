// k_cfr_split: one workgroup per deal subtree, one grid barrier per player pass (leduc_poker).
#include "osg_cfr_internal.h"

namespace {

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


}  // namespace

namespace osg_cfr_impl {

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

// One COOPERATIVE launch of k_cfr_split: the kernel spins on a grid barrier, so its workgroups must be resident
// together — with another stream keeping the device busy (a network's forward pass beside the solver) a plain launch
// can start some workgroups while the others queue behind foreign work, and the barrier's bound then turns a slowdown
// into an error.  The cooperative launch waits until the whole grid fits.  br: the CFR-BR pass set (d_best overrides).
int launch_split(osg_cfr* s, SmallTree stree, SplitTree sp, Tables tb, int iters, int iteration0, osg_cfr_cfg cfg, bool br) {
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

}  // namespace osg_cfr_impl
