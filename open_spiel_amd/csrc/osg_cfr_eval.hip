// Policy evaluation on the flattened tree: ExpectedReturns, TabularBestResponse, NashConv / Exploitability
// (SURVEY.md 8(f) row 1): k_policy_eval (one workgroup), k_geval_* (a launch per level), k_eval_jobs (independent jobs).
#include "osg_cfr_internal.h"

namespace {

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
// The large-tree evaluation as ONE persistent cooperative launch (round 6): the phases of k_geval_cf / _best / _brv above
// as grid-stride loops of one resident grid, a two-level counter barrier (k_cfr_sub's: group counters, the pollers watch
// the top counter) where the launches had a kernel boundary — 3-player leduc: 39 launches of ~8 us each, most of it the
// boundaries and the ramps of short kernels.  Everything a later phase reads that an earlier phase of the SAME launch
// wrote (cf, the [H, P] responder values, the expected returns, best) moves with written-through stores and
// cache-bypassing loads: the XCDs' L2s are not coherent with each other inside a launch.  The tree, the policy and the
// host's per-level lists are read-only here: ordinary loads.  Same sums in the same order as the launches: bit-identical
// (tests/test_gpu_cfr.py compares the two forms and the one-workgroup walk).
// ---------------------------------------------------------------------------
struct GEvalPlan {
  const int32_t* level_info;      // the infostates of every level, level by level (the host's list)
  const int32_t* level_info_off;  // [D + 1]
  unsigned int* bar;              // [kSubBarWords] zeroed before the launch: [0] top counter, [1] error, [16 + 16 g] group g
  unsigned int* host_err;         // pinned word raised on a timeout
  unsigned long long timeout_ticks;
};
OSG_D bool geval_barrier(const GEvalPlan& gp, unsigned int& epoch, int* s_ok) {
  ++epoch;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int grp = blockIdx.x >> 4, ngrp = (gridDim.x + 15u) >> 4;
    const unsigned int gsize = gridDim.x - (grp << 4) < 16u ? gridDim.x - (grp << 4) : 16u;
    if (__hip_atomic_fetch_add(&gp.bar[16 + 16 * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == epoch * gsize)
      __hip_atomic_fetch_add(&gp.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    const unsigned int want = epoch * ngrp;
    unsigned int polls = 0;
    int ok = 1;
    while (__hip_atomic_load(&gp.bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      if (wall_clock64() - t0 > gp.timeout_ticks) { ok = 0; break; }
      if ((++polls & 15u) == 0u && __hip_atomic_load(&gp.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) {
      __hip_atomic_store(&gp.bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gp.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *s_ok = ok;
  }
  __syncthreads();
  // (an acquire fence here — buffer_inv sc1 per wavefront — with ordinary loads afterwards measured 1.58 ms per evaluation
  // against 0.61 with a bypassing load per value: profiles/r06t_*)
  return *s_ok != 0;
}
constexpr int kGEvalThreads = 1024;
// A value another workgroup wrote is fetched by a buffer load with the sc1 bit (bypassing, and — unlike an agent-scope
// atomic load — an ordinary load to the scheduler: the loads of a history's children are all requested before the first
// is used)
typedef unsigned int osg_u2 __attribute__((ext_vector_type(2)));
OSG_D double geval_load(__amdgpu_buffer_rsrc_t buf, size_t idx) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(buf, static_cast<int>(idx * 8), 0, kCachePolicySc1));
}
OSG_D int32_t geval_load_i32(__amdgpu_buffer_rsrc_t buf, size_t idx) {
  return static_cast<int32_t>(__builtin_amdgcn_raw_buffer_load_b32(buf, static_cast<int>(idx * 4), 0, kCachePolicySc1));
}
__global__ void __launch_bounds__(kGEvalThreads)
k_geval_persist(Tree t, EvalArrays ea, const double* __restrict__ pol, double* ev, GEvalPlan gp) {
  __shared__ int s_ok;
  const int P = t.P, A = t.A;
  const int tid = threadIdx.x;
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kGEvalThreads + tid, gsize = static_cast<int64_t>(gridDim.x) * kGEvalThreads;
  unsigned int epoch = 0;
  const __amdgpu_buffer_rsrc_t vbuf = through_buffer(ea.value), cbuf = through_buffer(ea.cf), bbuf = through_buffer(ea.best),
                               ebuf = through_buffer(ev ? ev : ea.value);
  // ---- counterfactual reaches of every player's decision histories (k_geval_cf) ----
  for (int64_t m = gtid; m < ea.M; m += gsize) {
    const int h = t.mem[m];
    const int r = t.actor[h];
    double cf = 1.0;
    for (int e = ea.path_off[m]; e < ea.path_off[m + 1]; ++e) {
      const int code = ea.path[e];
      const int slot = (code >> 24) & 0xF, idx = code & 0x7FFFFF;
      const double pr = ((code >> 23) & 1) ? t.edge_prob[idx] : (slot == r ? 1.0 : pol[idx]);
      cf = cf * pr;
    }
    store_through(ea.cf + m, cf);
  }
  // (the first argmax needs cf; the deepest level holds terminals only, whose values need nothing: one barrier covers both)
  for (int l = t.D - 1; l >= 0; --l) {
    const int i0 = gp.level_info_off[l], n_infos = gp.level_info_off[l + 1] - i0;
    if (n_infos > 0 || l == t.D - 1) {
      if (!geval_barrier(gp, epoch, &s_ok)) return;
    }
    if (n_infos > 0) {
      // ---- the argmax of the infostates whose members sit on level l: a wavefront each (k_geval_best) ----
      const int lane = tid & 63;
      const int64_t wave = gtid >> 6, waves = gsize >> 6;
      for (int64_t w = wave; w < n_infos; w += waves) {
        const int i = gp.level_info[i0 + w];
        const int r = t.info_player[i], n = t.nact[i];
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
              prod = geval_load(cbuf, m) * geval_load(vbuf, static_cast<size_t>(t.first_child[t.mem[m]] + a) * P + r);
            }
            for (int j = 0; j < here; ++j) v += readlane_f64(prod, j);
          }
          if (v > best_v) { best_v = v; best = a; }
        }
        if (lane == 0) store_through_i32(ea.best + i, best < 0 ? 0 : best);
      }
      if (!geval_barrier(gp, epoch, &s_ok)) return;
    }
    // ---- every responder's values (and the expected returns) of level l (k_geval_brv) ----
    for (int64_t h = t.level_off[l] + gtid; h < t.level_off[l + 1]; h += gsize) {
      const int k = t.kind[h];
      const int fc = k == kTerminalNode ? 0 : t.first_child[h], nc = k == kTerminalNode ? 0 : t.nchild[h];
      const int actor = k == kDecisionNode ? t.actor[h] : -1;
      const int row = k == kDecisionNode ? t.info[h] * A : 0;
      if (ev) {
        for (int q = 0; q < P; ++q) {
          double v = 0.0;
          if (k == kTerminalNode) {
            v = t.term_ret[h * P + q];
          } else {
            for (int a = 0; a < nc; ++a) {
              const double pr = k == kChanceNode ? t.edge_prob[fc + a] : pol[row + a];
              if (pr > 0.0) v += pr * geval_load(ebuf, static_cast<size_t>(fc + a) * P + q);
            }
          }
          store_through(ev + static_cast<size_t>(h) * P + q, v);
          if (h == 0) ea.out[q] = v;
        }
      }
      const int chosen = actor >= 0 ? geval_load_i32(bbuf, t.info[h]) : 0;
      for (int r = 0; r < P; ++r) {
        double v = 0.0;
        if (k == kTerminalNode) {
          v = t.term_ret[h * P + r];
        } else if (actor == r) {
          v += 1.0 * geval_load(vbuf, static_cast<size_t>(fc + chosen) * P + r);
        } else {
          for (int a = 0; a < nc; ++a) {
            const double pr = k == kChanceNode ? t.edge_prob[fc + a] : pol[row + a];
            v += pr * geval_load(vbuf, static_cast<size_t>(fc + a) * P + r);
          }
        }
        store_through(ea.value + static_cast<size_t>(h) * P + r, v);
        if (ea.keep && r == ea.keep_r) ea.keep[h] = v;
        if (h == 0) ea.out[P + r] = v;
      }
    }
    if (l > 0 && gp.level_info_off[l] - gp.level_info_off[l - 1] == 0) {   // (a level with infostates opens with its own barrier)
      if (!geval_barrier(gp, epoch, &s_ok)) return;
    }
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


}  // namespace

namespace osg_cfr_impl {

EvalJobs eval_jobs_of(const osg_cfr* s) {
  EvalJobs ej;
  ej.J = s->jobs_J; ej.L = s->jobs_L; ej.G = s->jobs_G; ej.NT = s->jobs_NT;
  ej.job = s->d_jobs_job; ej.level_off = s->d_jobs_level; ej.node_desc = s->d_jobs_desc; ej.node_fc = s->d_jobs_fc;
  ej.node_row = s->d_jobs_row; ej.node_glob = s->d_jobs_glob; ej.info_ent = s->d_jobs_info; ej.mem_ent = s->d_jobs_mem;
  ej.deal = s->d_jobs_deal; ej.ticket = s->d_jobs_ticket;
  return ej;
}

// Trees beyond one workgroup's reach (and too large for the jobs) are evaluated with a launch per level and phase
// (k_geval_*); OSG_EVAL_GRID=1 forces that form, 0 forbids it (the tests compare the three).
bool eval_takes_the_grid(const osg_cfr* s) {
  const char* e = getenv("OSG_EVAL_GRID");
  if (e && e[0] == '0') return false;
  if (e && e[0] == '1') return true;
  return s->H > 65536;
}
int launch_grid_eval(const osg_cfr* s, const EvalArrays& ea, const double* src, bool from_cum, double* d_pol, bool only_br) {
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
  // OSG_EVAL_PERSIST=1: ONE persistent cooperative launch for the whole sweep (round 6).  Measured SLOWER than the launches
  // below on 3-player leduc (0.61 against 0.40 ms per NashConv, profiles/r06t_*): not the default; kept as the cross-check
  {
    const char* pe = getenv("OSG_EVAL_PERSIST");
    if (pe && pe[0] == '1') {
      if (!ms->d_eval_level_off) {
        if (int rc = upload(ms->eval_level_off, &ms->d_eval_level_off, st)) return rc;
        OSG_HIP(hipMalloc(reinterpret_cast<void**>(&ms->d_geval_bar), sizeof(unsigned int) * kSubBarWords));
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_geval_persist, kGEvalThreads, 0) != hipSuccess || per_cu < 1 ||
            hipGetDeviceProperties(&prop, s->ctx->device) != hipSuccess || !prop.cooperativeLaunch) {
          (void)hipGetLastError();
          ms->geval_grid = -1;   // (no resident grid: the launches below)
        } else {
          const char* w = getenv("OSG_EVAL_PERSIST_PER_CU");
          const int want = w ? std::max(1, atoi(w)) : 1;
          ms->geval_grid = std::min(std::min(per_cu, want) * prop.multiProcessorCount, 1024);
          if (static_cast<unsigned long long>(s->H) * s->P * sizeof(double) >= (1ull << 31)) ms->geval_grid = -1;   // (32-bit buffer offsets)
        }
      }
      if (ms->geval_grid > 0) {
        GEvalPlan gp{ms->d_eval_level_info, ms->d_eval_level_off, ms->d_geval_bar, s->h_sub_err, 400000000ull /* 4 s at 100 MHz */};
        OSG_HIP(hipMemsetAsync(ms->d_geval_bar, 0, sizeof(unsigned int) * kSubBarWords, st));
        Tree tt = t;
        EvalArrays eaa = ea;
        const double* pp = pol;
        void* args[] = {&tt, &eaa, &pp, &d_ev, &gp};
        static const bool plain = std::getenv("OSG_CFR_PLAIN_LAUNCH") && std::getenv("OSG_CFR_PLAIN_LAUNCH")[0] == '1';
        const void* kern = reinterpret_cast<const void*>(&k_geval_persist);
        if (plain) OSG_HIP(hipLaunchKernel(kern, dim3(static_cast<unsigned>(ms->geval_grid)), dim3(kGEvalThreads), args, 0, st));
        else OSG_HIP(hipLaunchCooperativeKernel(kern, dim3(static_cast<unsigned>(ms->geval_grid)), dim3(kGEvalThreads), args, 0, st));
        ms->last_eval_kernel = "k_geval_persist";
        return OSG_OK;
      }
    }
  }
  ms->last_eval_kernel = "k_geval";
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
bool OSG_EVAL_JOBS_ENABLED() {
  const char* e = getenv("OSG_EVAL_JOBS");
  return !(e && e[0] == '0');
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

// Every player's best response to the current policy (cfr_br.cc:55-68): the first half of a CFR-BR iteration.
int cfr_best_responses_to_current(osg_cfr* s, const EvalArrays& ea, int threads, bool jobs) {
  hipStream_t st = s->ctx->stream;
  if (jobs) { k_eval_jobs<<<dim3(s->jobs_J), dim3(s->jobs_threads), s->jobs_lds_bytes, st>>>(s->tree(), ea, eval_jobs_of(s), s->cur(), 1, 1); s->last_eval_kernel = "k_eval_jobs"; }
  else if (eval_takes_the_grid(s)) { if (int rc = launch_grid_eval(s, ea, s->cur(), false, nullptr, true)) return rc; }
  else { k_policy_eval<<<dim3(1), dim3(threads), 0, st>>>(s->tree(), ea, s->cur()); s->last_eval_kernel = "k_policy_eval"; }
  return OSG_OK;
}

}  // namespace osg_cfr_impl

extern "C" {

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
    s->last_eval_kernel = "k_eval_jobs";
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
    s->last_eval_kernel = "k_policy_eval";
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

}  // extern "C"
