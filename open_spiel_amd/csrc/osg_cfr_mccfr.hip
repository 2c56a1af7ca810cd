// External-sampling and outcome-sampling MCCFR (external_sampling_mccfr.{h,cc}, outcome_sampling_mccfr.cc): the
// general kernels, the LDS-resident forms (BASELINE config 5), FullUpdateAverage, the delta fold, and their entry points.
#include "osg_cfr_internal.h"

namespace {

// ---------------------------------------------------------------------------
// ExternalSamplingMCCFRSolver::UpdateRegrets (external_sampling_mccfr.cc:122-186),
// AverageType::kSimple, one traversal per thread, tables frozen for the launch.
// ---------------------------------------------------------------------------

// Streams of trajectory g: (seed, g, 0) in visiting order down to the FIRST node at which the traverser acts; inside
// that node's child b1, down to the traverser's NEXT node on the path, sub-stream 1 + b1; inside that node's child b2
// sub-stream 16 + 8 b1 + b2 — a sub-stream is the same generator after a jump of its counter (Rng::jump_to).  The subtrees below a traverser node are independent but for the order of
// the draws: with a stream each they can be walked by different lanes (k_mccfr_resident<., kSplit>); the oracle's replay
// follows the same rule (osgo_mccfr_minibatch).
// kExtU: the uniforms come from a caller-supplied sequence (ext_u[0], ext_u[1], ... in visiting order) instead
// of the counter streams: with the sequence the reference's std::mt19937 + uniform_real_distribution would
// produce, one trajectory IS one UpdateRegrets call of the reference, draw for draw
// (ExternalSamplingMCCFRSolver::RunIteration(std::mt19937*), external_sampling_mccfr.h:63-100).
#ifdef OSG_MCCFR_DIAG_NOATOMIC   // MEASUREMENT ONLY: what the general kernel would last without its global atomics
OSG_D void k_add_f64(double* p, double v) { *p = v; }
#else
OSG_D void k_add_f64(double* p, double v) { add_f64(p, v); }
#endif
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
            for (int a = 0; a < nc; ++a) k_add_f64(&dpol[i * A + a], pol[a]);
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
        for (int b = 0; b < nc; ++b) k_add_f64(&dreg[i * A + b], f_cv[sp - 1][b] - v);  // (:167-172)
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
      if (r != 0.0) k_add_f64(&g_dreg[k], r);
      if (q != 0.0) k_add_f64(&g_dpol[k], q);
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
#if defined(OSG_MCCFR_DIAG_COHERENT) && OSG_MCCFR_DIAG_COHERENT == 1
    const int64_t jj = j0 & ~static_cast<int64_t>(63);   // MEASUREMENT ONLY: a wavefront's lanes all walk its first lane's trajectory (no divergence at all)
#else
    const int64_t jj = j0;
#endif
    int64_t j = jj;
    {
      const int64_t span = 64 * static_cast<int64_t>(P), group = jj / span;
      if ((group + 1) * span <= count) {
        const int r = static_cast<int>(jj - group * span);
        j = group * span + static_cast<int64_t>(r & 63) * P + (r >> 6);
      }
    }
    const int64_t g = first + j;
    // (a 64-bit modulo by a run-time divisor is ~100 instructions: two players take the parity)
    const int trav = P == 2 ? static_cast<int>(g & 1) : static_cast<int>(g % P);
    const int next = trav + 1 == P ? 0 : trav + 1;
    Rng rng(seed, static_cast<uint64_t>(g), 0);
#if defined(OSG_MCCFR_DIAG_COHERENT) && OSG_MCCFR_DIAG_COHERENT == 2
    // MEASUREMENT ONLY: the first two draws of a trajectory (leduc: the two private cards) are the wavefront's, the rest
    // its own — what sorting the trajectories by their deal would give a wavefront
    Rng rng_w(seed, static_cast<uint64_t>(first + (j0 & ~static_cast<int64_t>(63))), 0);
    int diag_draws = 0;
#endif
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
#elif defined(OSG_MCCFR_DIAG_COHERENT) && OSG_MCCFR_DIAG_COHERENT == 2
          double z = rng.unit();
          if (diag_draws < 2) z = rng_w.unit();
          ++diag_draws;
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


}  // namespace

namespace osg_cfr_impl {

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

}  // namespace osg_cfr_impl

extern "C" {

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

int osg_mccfr_delta_ptrs(osg_cfr* s, double** d_regret_delta, double** d_policy_delta) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_mccfr_delta_ptrs: null argument");
  if (d_regret_delta) *d_regret_delta = s->dreg();
  if (d_policy_delta) *d_policy_delta = s->dpol();
  return OSG_OK;
}

}  // extern "C"
