// Large trees: k_cfr_sub (ONE persistent cooperative launch, a bin of subtrees / pieces per workgroup) and k_gcfr_*
// (a full-grid launch per tree level and phase).  3-player leduc_poker: 1.83 M histories.
#include "osg_cfr_internal.h"

namespace {

// ---------------------------------------------------------------------------
// Large trees (3-player leduc: 1.8 M histories): the same three phases as k_cfr_small, but
// every phase is a full-grid launch — one kernel per tree level for the values, one for the
// per-history terms, one for the per-infostate fold — so the whole chip works on one tree and
// the stream order provides the barriers.  Same additions in the same order: tables are
// bit-identical with the single-workgroup kernels.
// ---------------------------------------------------------------------------
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
#ifndef OSG_SUB_KEEP_DESCRIPTORS
#define OSG_SUB_KEEP_DESCRIPTORS 1
#endif
#ifndef OSG_SUB_QUAD_STORE
#define OSG_SUB_QUAD_STORE 1
#endif
OSG_D unsigned int dlo(double v) { return static_cast<unsigned int>(__double_as_longlong(v)); }
OSG_D unsigned int dhi(double v) { return static_cast<unsigned int>(__double_as_longlong(v) >> 32); }
// Lane kSrc of every quad (four consecutive lanes) broadcast to the quad: a DPP move, no LDS.
template <int kSrc>
OSG_D unsigned int quad_bcast(unsigned int v) {
  return static_cast<unsigned int>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), kSrc * 0x55, 0xF, 0xF, true));
}
// The record of the quad's lane kSrc (its four 16-byte pieces own[piece][word]) written by the whole quad: lane rq writes
// piece rq at at + 16 rq — 64 contiguous, 64-byte aligned bytes per quad and instruction.
template <int kSrc>
OSG_D void quad_store(__amdgpu_buffer_rsrc_t rec_buf, const unsigned int (&own)[4][4], unsigned int at, unsigned int wr, int rq) {
  const unsigned int at_s = quad_bcast<kSrc>(at), wr_s = quad_bcast<kSrc>(wr);
  osg_u4 mine;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const unsigned int c0 = quad_bcast<kSrc>(own[0][w]), c1 = quad_bcast<kSrc>(own[1][w]), c2 = quad_bcast<kSrc>(own[2][w]),
                       c3 = quad_bcast<kSrc>(own[3][w]);
    mine[w] = rq == 0 ? c0 : (rq == 1 ? c1 : (rq == 2 ? c2 : c3));
  }
  if (wr_s) __builtin_amdgcn_raw_buffer_store_b128(mine, rec_buf, static_cast<int>(at_s + 16u * static_cast<unsigned int>(rq)), 0, kCachePolicySc1);
}

// kBr (round 6): a CFR-BR pass set (cfr_br.cc:70-81) — the rows pass `upd` plays are the updating player's rows of the current
// policy and, for every other player, the one-hot row of the best-response action the evaluation left in sp.br_best
// (policy_overrides, cfr.cc:365-372): every pass stages all its rows, nothing else changes.  A template argument so that
// the plain CFR kernel's code object is what it was.
template <int kK, bool kBr = false>   // histories per thread: NL <= kK * 1024
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
          if (sp.tree_barrier == 2) {   // (round 6) the top counter IS what the pollers watch: no release word, one hop less after the last arrival
            (void)ngrp;
            __hip_atomic_fetch_add(&sp.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else if (__hip_atomic_fetch_add(&sp.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == epoch * ngrp) {
            __hip_atomic_store(&sp.bar[2], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      } else {
        __hip_atomic_fetch_add(&sp.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    window();
    if (tid == 0) {
      const unsigned long long t0 = wall_clock64();
      int ok = 1;
      if (sp.tree_barrier == 2) {
        // 16 adds per barrier land on the top counter (one per group), so 256 pollers reading it do not starve them as they
        // starved round 4's 256 adds; another workgroup's give-up shows in the error word, looked at every 16th poll
        const unsigned int want = epoch * ((gridDim.x + 15u) >> 4);
        unsigned int polls = 0;
        while (__hip_atomic_load(&sp.bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          if (wall_clock64() - t0 > sp.timeout_ticks) { ok = 0; break; }
          if ((++polls & 15u) == 0u && __hip_atomic_load(&sp.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      } else if (sp.tree_barrier) {
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
  // ---- the thread's histories of a bin: descriptors in registers for the sweep.  With one bin per workgroup (the
  //      packed / forest forms) they are the same in every pass: fetched ONCE per launch (round 6: 0.8 us per pass) ----
  int o_d[kK], o_fc[kK], o_aux[kK];
  auto load_descriptors = [&](int g, int nloc) {
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
  };
  const bool one_bin = OSG_SUB_KEEP_DESCRIPTORS && sp.G <= static_cast<int>(gridDim.x);
  if (one_bin && static_cast<int>(blockIdx.x) < sp.G) load_descriptors(blockIdx.x, sp.nloc[blockIdx.x]);
  for (int it = 0; it < iters; ++it) {
    const int iteration = iteration0 + it + 1;
    for (int upd = 0; upd < P; ++upd) {
      const bool stamp = sp.stamps && it == iters - 1 && static_cast<int>(blockIdx.x) == sp.stamp_wg && tid == 0;
      if (stamp) sp.stamps[upd * 5 + 0] = wall_clock64();
      for (int g = blockIdx.x; g < sp.G; g += gridDim.x) {
        const int nloc = sp.nloc[g];
        if (!one_bin) load_descriptors(g, nloc);
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
        const bool all_rows = kBr || !sp.keep_rows || (it == 0 && upd == 0);
        if (all_rows) {
          int rows[kSubKD];   // the thread's decision histories: all their rows are requested before the first arrives
#pragma unroll
          for (int k = 0; k < kSubKD; ++k) {
            const int d = tid + k * kSubThreads;
            rows[k] = d < ndec ? sp.dec_row[static_cast<size_t>(g) * sp.ND + d] : -1;
          }
          if constexpr (kBr) {
            int own[kSubKD], best[kSubKD];
#pragma unroll
            for (int k = 0; k < kSubKD; ++k) {
              const int i = rows[k] >= 0 ? rows[k] / A : 0;
              own[k] = sp.br_player[i] == upd ? 1 : 0;
              best[k] = sp.br_best[i];
            }
#pragma unroll
            for (int k = 0; k < kSubKD; ++k) {
#pragma unroll
              for (int a = 0; a < kSplitMaxA; ++a)
                if (rows[k] >= 0 && a < A)
                  s_pol[(tid + k * kSubThreads) * A + a] = own[k] ? load_through(tb.cur + rows[k] + a) : (a == best[k] ? 1.0 : 0.0);
            }
          } else {
#pragma unroll
          for (int k = 0; k < kSubKD; ++k) {
#pragma unroll
            for (int a = 0; a < kSplitMaxA; ++a)
              if (rows[k] >= 0 && a < A) s_pol[(tid + k * kSubThreads) * A + a] = load_through(tb.cur + rows[k] + a);
          }
          }
          if (!kBr || !sp.keep_rows || (it == 0 && upd == 0))   // (a CFR-BR pass re-stages the rows only: the probabilities stay)
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
        // One or two members per thread and round (the two-member form keeps both dependent chains in flight).  A record's
        // request is not moved before the sweep: holding 24 - 48 registers across it measured 2 - 9 % slower (profiles/r06j_*).
        const int m_begin = sp.mem_off[g * P + upd], m_end = sp.mem_off[g * P + upd + 1];
        const int n_chunks = sp.PL / 4, per_player = n_chunks / P;
        const int n_words4 = (n_chunks + 1) / 2, rec_ints = 8 + 4 * n_words4;
        auto members_round = [&](int mbase, auto nu_tag) {
          constexpr int NU = decltype(nu_tag)::value;
          int4 head[NU], second[NU], codes[NU][kSubCodeChunks / 2];   // (16-bit codes: two chunks of four per int4)
          bool live[NU];
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            const int mm = mbase + tid + u * kSubThreads;
            live[u] = mm < m_end;
            const int4* rec = reinterpret_cast<const int4*>(sp.sub_rec + static_cast<size_t>(live[u] ? mm : m_begin) * rec_ints);
            head[u] = rec[0]; second[u] = rec[1];
#pragma unroll
            for (int c = 0; c < kSubCodeChunks / 2; ++c) codes[u][c] = rec[2 + (c < n_words4 ? c : n_words4 - 1)];
          }
          bool pruned[NU];
          double self_reach[NU], cf_reach[NU], r[NU];
#pragma unroll
          for (int u = 0; u < NU; ++u) { pruned[u] = true; self_reach[u] = 0.0; cf_reach[u] = 1.0; r[u] = 1.0; }
          int q = 0;
#pragma unroll
          for (int c = 0; c < kSubCodeChunks; ++c) {
            if (c < n_chunks) {   // (workgroup-uniform)
#pragma unroll
              for (int u = 0; u < NU; ++u) {
                const unsigned int w0 = static_cast<unsigned int>((c & 1) ? codes[u][c >> 1].z : codes[u][c >> 1].x),
                                   w1 = static_cast<unsigned int>((c & 1) ? codes[u][c >> 1].w : codes[u][c >> 1].y);
                const unsigned int cx = w0 & 0xFFFFu, cy = w0 >> 16, cz = w1 & 0xFFFFu, cw = w1 >> 16;   // 0xFFFF: padding
                const double px = s_pol[cx == 0xFFFFu ? 0u : cx], py = s_pol[cy == 0xFFFFu ? 0u : cy],
                             pz = s_pol[cz == 0xFFFFu ? 0u : cz], pw = s_pol[cw == 0xFFFFu ? 0u : cw];
                r[u] = r[u] * (cx == 0xFFFFu ? 1.0 : px);   // (x * 1.0 == x: the padding leaves the product as it is)
                r[u] = r[u] * (cy == 0xFFFFu ? 1.0 : py);
                r[u] = r[u] * (cz == 0xFFFFu ? 1.0 : pz);
                r[u] = r[u] * (cw == 0xFFFFu ? 1.0 : pw);
              }
              if ((c + 1) % per_player == 0) {   // the last chunk of player q's group
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                  pruned[u] &= (r[u] == 0.0);
                  if (q == upd) self_reach[u] = r[u]; else cf_reach[u] *= r[u];
                  r[u] = 1.0;
                }
                ++q;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            const int m = head[u].x, hl = head[u].y, d = head[u].z & 0xFFFFFF, n = (head[u].z >> 24) & 0xFF, lfc = head[u].w;
            const double chance = __longlong_as_double((static_cast<long long>(second[u].y) << 32) | static_cast<unsigned int>(second[u].x));
            const double cf = cf_reach[u] * chance;
            const unsigned int at = static_cast<unsigned int>(m) * (kSubRecDoubles * 8);
            const double vh = s_value[live[u] ? hl : 0];
            double dr[kSplitMaxA], dp[kSplitMaxA];
#pragma unroll
            for (int a = 0; a < kSplitMaxA; ++a) {
              dr[a] = 0.0; dp[a] = 0.0;
              if (live[u] && !pruned[u] && a < n) {
                dr[a] = cf * (s_value[lfc + a] - vh);
                const double pol = s_pol[d * A + a];
                dp[a] = cfg.linear_averaging ? iteration * self_reach[u] * pol : self_reach[u] * pol;
              }
            }
            if (pruned[u]) {   // the flag record: a quiet NaN whose low word is 1 in the first word, nothing else is read
              dr[0] = __longlong_as_double((static_cast<long long>(kSubFlagHi) << 32) | 1ll);
              dr[1] = 0.0;
            }
            static_assert(kSplitMaxA == 4 && kSubRecDoubles == 8, "the record is two pieces of regret terms, two of policy terms");
            if (A > 2 && OSG_SUB_QUAD_STORE) {   // (workgroup-uniform)
              // A member's 64-byte record leaves as ONE contiguous piece of memory traffic: the four lanes of a quad write
              // the four 16-byte pieces of ONE member's record together, member by member (round 6).  Written by its own
              // lane piece by piece, every store instruction put 64 scattered 16-byte fragments on the fabric — the records
              // of a bin's members lie ~2.5 KB apart — and the phase spent 7 of its 12 us there (profiles/r06h_*, r06i_*).
              const int rq = tid & 3;
              const unsigned int own[4][4] = {
                  {dlo(dr[0]), dhi(dr[0]), dlo(dr[1]), dhi(dr[1])}, {dlo(dr[2]), dhi(dr[2]), dlo(dr[3]), dhi(dr[3])},
                  {dlo(dp[0]), dhi(dp[0]), dlo(dp[1]), dhi(dp[1])}, {dlo(dp[2]), dhi(dp[2]), dlo(dp[3]), dhi(dp[3])}};
              const unsigned int wr = live[u] ? 1u : 0u;
              quad_store<0>(rec_buf, own, at, wr, rq);
              quad_store<1>(rec_buf, own, at, wr, rq);
              quad_store<2>(rec_buf, own, at, wr, rq);
              quad_store<3>(rec_buf, own, at, wr, rq);
            } else if (live[u]) {
              store_through16(rec_buf, at, osg_d2{dr[0], dr[1]});
              if (!pruned[u]) {
                store_through16(rec_buf, at + 32, osg_d2{dp[0], dp[1]});
                if (A > 2) {   // (workgroup-uniform)
                  store_through16(rec_buf, at + 16, osg_d2{dr[2], dr[3]});
                  store_through16(rec_buf, at + 48, osg_d2{dp[2], dp[3]});
                }
              }
            }
          }
        };
        // A round covers 2 048 members: thread t takes members mbase + t and mbase + 1 024 + t.  Which form a WAVEFRONT runs
        // is its own (wave-uniform, so the quad stores may read their neighbours' registers): two members only where its
        // lanes have a second one — with 1 052 - 1 086 members that is the first wavefront alone; the whole workgroup in the
        // two-member form doubled the phase's instructions and put ~2 us on the 64 workgroups every barrier waits for.
        for (int mbase = m_begin; mbase < m_end; mbase += 2 * kSubThreads) {
          const int wave_first = mbase + (tid & ~63);
          if (wave_first + kSubThreads < m_end) members_round(mbase, std::integral_constant<int, 2>{});
          else if (wave_first < m_end) members_round(mbase, std::integral_constant<int, 1>{});
        }
        __syncthreads();   // (the next subtree of this workgroup reuses s_value)
      }
      if (stamp) sp.stamps[upd * 5 + 2] = wall_clock64();
      // (profiling) when EVERY workgroup reaches the two barriers of the launch's last iteration: who the others wait for
      if (sp.stamps && it == iters - 1 && tid == 0) sp.stamps[8 * kMaxPlayers + (upd * 2 + 0) * gridDim.x + blockIdx.x] = wall_clock64();
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
      if (sp.stamps && it == iters - 1 && tid == 0) sp.stamps[8 * kMaxPlayers + (upd * 2 + 1) * gridDim.x + blockIdx.x] = wall_clock64();
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


}  // namespace

namespace osg_cfr_impl {

// The subtrees of k_cfr_sub: the same cut as build_split (the first level with a node that is not a chance node),
// any number of subtrees (a workgroup takes several in turn when the cooperative grid is smaller), up to 8 x 1024
// histories each.  Round 5, forest form: with more subtrees than compute units the bins of the workgroups are packed —
// whole subtrees if they fit, else the pieces one level below the cut (SubTree's comment) — so that every workgroup
// sweeps one bin per pass.  OSG_CFR_SUB_PACK=0 keeps a subtree per bin.
template <int kK, bool kBr = false> const void* cfr_sub_kernel() { return reinterpret_cast<const void*>(&k_cfr_sub<kK, kBr>); }
const void* cfr_sub_kernel_of(int K, bool br) {
  if (br) return K == 2 ? cfr_sub_kernel<2, true>() : (K == 4 ? cfr_sub_kernel<4, true>() : cfr_sub_kernel<8, true>());
  return K == 2 ? cfr_sub_kernel<2>() : (K == 4 ? cfr_sub_kernel<4>() : cfr_sub_kernel<8>());
}
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
  const void* kern = cfr_sub_kernel_of(K, false);
  if (raise_lds_cap(kern, static_cast<int>(lds)) != hipSuccess) {
    (void)hipGetLastError();
    return OSG_OK;
  }
  // the CFR-BR form of the same kernel: taken when it can be resident on the same grid (else CFR-BR keeps the launches per phase)
  s->sub_br_ok = false;
  if (raise_lds_cap(cfr_sub_kernel_of(K, true), static_cast<int>(lds)) == hipSuccess) {
    int per_cu_br = 0;
    hipError_t eb;
    if (K == 2) eb = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_br, k_cfr_sub<2, true>, kSubThreads, lds);
    else if (K == 4) eb = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_br, k_cfr_sub<4, true>, kSubThreads, lds);
    else eb = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_br, k_cfr_sub<8, true>, kSubThreads, lds);
    if (eb == hipSuccess) s->sub_br_ok = per_cu_br >= 1;
    else (void)hipGetLastError();
  } else {
    (void)hipGetLastError();
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

// The persistent cooperative launch (osg_cfr_iterate's sub_path); br_best != null: ONE CFR-BR pass set on the best responses
// the evaluation left there (cfr_sub_br_iterate below).
static int cfr_sub_run(osg_cfr* s, Tables tb, int iters, const int32_t* br_best, osg_cfr_cfg cfg) {
  if (int rc = cfr_sub_error(s)) return rc;
  const int M = static_cast<int>(s->mem.size());
  Tree tr = s->tree();
  SmallTree stree{s->d_path_off, s->d_path, M, static_cast<int>(s->path.size())};
  SubTree sp{s->sub_G, s->sub_L, s->sub_NL, s->d_sub_nloc, s->d_sub_desc, s->d_sub_fc, s->d_sub_aux, s->sub_ND, s->d_sub_ndec,
             s->d_sub_dec_row, s->d_sub_mem_off,
             s->d_sub_rec, s->sub_PL, s->d_sub_info_off, s->d_sub_info_list, s->d_sub_recbuf,
             // the grid barrier: 2 = two-level arrival, the pollers watch the top counter (round 6); 1 = the same with a release
             // word (round 5; OSG_CFR_SUB_BARRIER=release); 0 = round 4's flat counter (OSG_CFR_SUB_FLAT_BARRIER=1)
             (std::getenv("OSG_CFR_SUB_FLAT_BARRIER") && std::getenv("OSG_CFR_SUB_FLAT_BARRIER")[0] == '1')
                 ? 0 : ((std::getenv("OSG_CFR_SUB_BARRIER") && std::strcmp(std::getenv("OSG_CFR_SUB_BARRIER"), "release") == 0) ? 1 : 2),
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
  sp.br_best = br_best; sp.br_player = s->d_info_player32;
  unsigned long long*& d_stamps = s->d_sub_stamps;   // OSG_CFR_SUB_STAMPS=1: phase stamps of one workgroup (tools/probe_cfr_sub.py); the solver's own buffer
  if (std::getenv("OSG_CFR_SUB_STAMPS")) {
    sp.stamp_wg = std::max(0, std::min(s->sub_grid - 1, atoi(std::getenv("OSG_CFR_SUB_STAMPS")) - 1));
    fprintf(stderr, "k_cfr_sub: G %d grid %d NL %d ND %d PL %d K %d forest %d NR %d\n", s->sub_G, s->sub_grid, s->sub_NL, s->sub_ND, s->sub_PL, s->sub_K, s->sub_forest ? 1 : 0, s->sub_NR);
    if (!d_stamps) OSG_HIP(hipMalloc(reinterpret_cast<void**>(&d_stamps), sizeof(unsigned long long) * (8 * kMaxPlayers + 2 * kMaxPlayers * 1024)));
    sp.stamps = d_stamps;
  }
  hipStream_t st = s->ctx->stream;
  const int per_launch = std::max(1, (1 << 30) / std::max(1, 2 * s->P * s->sub_grid));  // the arrival counter is 32 bits
  for (int done = 0; done < iters; done += per_launch) {
    int now = std::min(per_launch, iters - done), it0 = s->iteration + done;
    OSG_HIP(hipMemsetAsync(s->d_sub_bar, 0, sizeof(unsigned int) * kSubBarWords, st));
    void* args[] = {&tr, &stree, &sp, &tb, &now, &it0, &cfg};
    const void* kern = cfr_sub_kernel_of(s->sub_K, br_best != nullptr);
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
  if (sp.stamps && s->sub_grid <= 1024) {   // arrival of every workgroup at the two barriers of each pass (s_memrealtime: one clock for the chip)
    std::vector<unsigned long long> a(static_cast<size_t>(2) * s->P * s->sub_grid);
    OSG_HIP(hipMemcpyAsync(a.data(), sp.stamps + 8 * kMaxPlayers, sizeof(unsigned long long) * a.size(), hipMemcpyDeviceToHost, st));
    OSG_HIP(hipStreamSynchronize(st));
    for (int q = 0; q < s->P; ++q)
      for (int w = 0; w < 2; ++w) {
        const unsigned long long* v = a.data() + static_cast<size_t>(q * 2 + w) * s->sub_grid;
        unsigned long long lo = v[0], hi = v[0];
        int last = 0;
        for (int g = 0; g < s->sub_grid; ++g) { if (v[g] < lo) lo = v[g]; if (v[g] > hi) { hi = v[g]; last = g; } }
        int late1 = 0, late2 = 0, late_big = 0;
        for (int g = 0; g < s->sub_grid; ++g) { late1 += hi - v[g] <= 100; late2 += hi - v[g] <= 200; late_big += (hi - v[g] <= 200 && g < 64); }
        fprintf(stderr, "k_cfr_sub pass %d barrier %c arrivals (us after the first): last %.2f (workgroup %d); within 1 us of the last: %d, within 2 us: %d (%d of them among workgroups 0-63); wg 0 %.2f, 32 %.2f, 100 %.2f, 200 %.2f, %d %.2f\n",
                q, w ? 'B' : 'A', (hi - lo) / 100.0, last, late1, late2, late_big, (v[0] - lo) / 100.0, (v[std::min(32, s->sub_grid - 1)] - lo) / 100.0,
                (v[std::min(100, s->sub_grid - 1)] - lo) / 100.0, (v[std::min(200, s->sub_grid - 1)] - lo) / 100.0, s->sub_grid - 1, (v[s->sub_grid - 1] - lo) / 100.0);
      }
  }
  s->iteration += iters;
  if (br_best) s->last_kernel = s->sub_forest ? "k_cfr_sub<forest,br>" : (s->sub_G < s->sub_G0 ? "k_cfr_sub<packed,br>" : "k_cfr_sub<br>");
  else s->last_kernel = s->sub_forest ? "k_cfr_sub<forest>" : (s->sub_G < s->sub_G0 ? "k_cfr_sub<packed>" : "k_cfr_sub");
  return OSG_OK;
}
int cfr_sub_iterate(osg_cfr* s, Tables tb, int iters) { return cfr_sub_run(s, tb, iters, nullptr, s->cfg); }

// CFRBRSolver::EvaluateAndUpdatePolicy on large trees through the persistent kernel (round 6): per iteration the
// evaluation's one sweep (k_geval_*: every player's best response to the current policy, cfr_br.cc:55-68) and ONE launch of
// k_cfr_sub<., kBr> for the P passes (cfr_br.cc:70-81) — where cfr_grid_br_iterate below spends ~58 launches on them.
// The same additions in the same order: tables bit-identical with the launch-per-phase form.
int cfr_sub_br_iterate(osg_cfr* s, Tables tb, const EvalArrays& ea, osg_cfr_cfg cfg, int iters) {
  for (int it = 0; it < iters; ++it) {
    if (int rc = launch_grid_eval(s, ea, s->cur(), false, nullptr, true)) return rc;
    if (int rc = cfr_sub_run(s, tb, 1, s->d_best, cfg)) return rc;
  }
  return OSG_OK;
}

// A launch per tree level and phase (osg_cfr_iterate's grid_path).
int cfr_grid_iterate(osg_cfr* s, Tables tb, int iters) {
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

// CFRBRSolver::EvaluateAndUpdatePolicy on large trees (3-player leduc: one workgroup walked a pass set in tens of
// milliseconds): the evaluation's sweep leaves every infostate's best-response action, then each player's pass runs as the
// launch-per-phase CFR pass (k_gcfr_*) on the effective policy — the same additions in the same order as k_cfr<., kBr>.
int cfr_grid_br_iterate(osg_cfr* s, Tables tb, const EvalArrays& ea, osg_cfr_cfg cfg, int iters) {
  const size_t M = s->mem.size();
  hipStream_t st = s->ctx->stream;
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

}  // namespace osg_cfr_impl
