// algorithms::MCTSBot (open_spiel/algorithms/mcts.{h,cc}), ONE WAVEFRONT PER ROOT.
//
// The 64 lanes of a wavefront cooperate on one search tree:
//   * selection   (mcts.cc:324-341)  children of a node are contiguous; lane l scans
//                 children l, l+64, ... (header + statistics, one branch-free round of
//                 loads), scores them (UCTValue, mcts.cc:90-101: in fp32 first, the fp64
//                 formula only where that cannot name the maximum — select_child), a DPP
//                 reduction finds the maximum and a ballot who holds it; the chosen child's
//                 header is handed down by readlane, so a tree level costs one memory round trip
//   * expansion   (mcts.cc:281-299)  lane l initialises children l, l+64, ...
//   * evaluation  (mcts.cc:43-72)    n_rollouts playouts spread over the lanes; a
//                 hex playout is ONE wave-parallel random fill of the board
//                 (lane = cell): a binary search on the key threshold hands the mover
//                 its half, then a lane-parallel flood fill over uniform cell sets
//                 decides the winner
//   * backup      (mcts.cc:383-395)  lane d updates the d-th node of the path from the
//                 statistics the path carried down (stores only)
//   * MCTS-Solver (mcts.cc:398-434)  lanes scan the children, ballot / shuffle reduce
// The position itself is wave-uniform and kept in scalar registers (hex: HexWT<kS>, kS
// 64-bit cell sets per colour: 1 up to 64 cells, 2 up to 128, 3 / 4 / 6 up to 19 x 19), so the rule code is scalar set algebra plus per-lane neighbour
// tests.  Everything the compiler must see as wave-uniform is made so explicitly
// (readfirstlane / readlane / ballot); `opt -passes=print<uniformity>` on this file is
// the check — a single lane-varying value on the loop-carried path turns the whole
// position into vector registers and the scalar branches into exec-mask branches.
//
// What bounds the hex kernel is the ISSUE RATE OF SCALAR-UNIT INSTRUCTIONS (ALU, branches,
// waits: one per 4 cycles per SIMD), measured with rocprofv3 --pmc SQ_INSTS_SALU /
// SQ_INSTS_BRANCH / SQ_INSTS_VALU per variant (tools/pmc_variants.sh): 652 + 83 scalar +
// branch and 533 vector instructions per simulation gave 7.97e8 simulations/s, 387 + 88
// and 513 give 1.12e9.  Hence the shape of the code below: no lane-dependent control flow
// where a clamped index or a select does, bookkeeping that is the same in every lane done
// by the vector unit when the scalar unit is the busier one (the key-threshold search and
// the flood of the playout, the visit path), and no per-level state that is not needed
// (the hex position carries no edge labels on the way down, see HexW).
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
#include <type_traits>

#include "osg_mcts_internal.h"

using namespace osg;

#ifdef OSG_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[8];
extern "C" int osg_debug_phase_cycles(unsigned long long* out8, int reset) {
  if (out8) (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z); }
  return 0;
}
#define PT_DECL unsigned long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_t = clock64();
#define PT_MARK(k) do { const unsigned long long pt_n = clock64(); pt_acc[k] += pt_n - pt_t; pt_t = pt_n; } while (0)
#define PT_FLUSH do { if (lane_id() == 0) for (int q = 0; q < 8; ++q) atomicAdd(&g_phase_cycles[q], pt_acc[q]); } while (0)
#define PT_ARGS , unsigned long long* pt_acc, unsigned long long& pt_t
#define PT_PASS , pt_acc, pt_t
#else
#define PT_ARGS
#define PT_PASS
#define PT_DECL
#define PT_MARK(k)
#define PT_FLUSH
#endif

namespace {

#ifndef OSG_WAVES_PER_BLOCK
#define OSG_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = OSG_WAVES_PER_BLOCK;
constexpr int kMaxPath = 160;
// Where the hex playout's bookkeeping runs: 1 = on the vector unit (measured on MI355X, config 4: key threshold
// 0 -> 1: 9.48e8 -> 1.003e9 sims/s; flood 0 -> 1: 9.89e8 -> 1.003e9), 0 = the scalar formulation.  Flood mode 2 (round 5:
// the two exits on the vector unit as well, tested once per two steps — ~20 -> ~8 scalar instructions per pair of
// steps) measured 1.105e9 -> 1.090e9: the vector pipe is as full as the scalar one (SQ_ACTIVE_INST_VALU 549 quad-cycles
// per simulation against SQ_WAVE_CYCLES / 7 resident wavefronts = 503, SQ_ACTIVE_INST_SCA 382), so moving work across
// no longer pays (profiles/r05y_hex_flood_exits_ab.txt).
#ifndef OSG_THR_MODE
#define OSG_THR_MODE 3
#endif
#ifndef OSG_FLOOD_MODE
#define OSG_FLOOD_MODE 1
#endif
#ifndef OSG_HASH_VALU
#define OSG_HASH_VALU 1
#endif
// The hex fill kernel's expansion from the legal cells as lane masks (1) or through the 4-word action mask (0, the form up
// to round 5; the boards above 128 cells always take the lane masks).
// The UCT arg-max through an fp32 filter (1) or always in fp64 (0): see select_child.
#ifndef OSG_UCT_FILTER
#define OSG_UCT_FILTER 1
#endif
#ifndef OSG_EXPAND_SETS
#define OSG_EXPAND_SETS 1
#endif

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
OSG_D double uniform_f64(double v) {
  const uint64_t u = uniform64(__double_as_longlong(v));
  return __longlong_as_double(static_cast<long long>(u));
}
OSG_D uint32_t read_lane(uint32_t v, int src) {  // src must be wave-uniform
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), src));
}
OSG_D uint64_t read_lane_u64(uint64_t v, int src) {
  const uint32_t lo = read_lane(static_cast<uint32_t>(v), src), hi = read_lane(static_cast<uint32_t>(v >> 32), src);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
OSG_D double read_lane_f64(double v, int src) {
  const uint64_t u = static_cast<uint64_t>(__double_as_longlong(v));
  const uint32_t lo = read_lane(static_cast<uint32_t>(u), src), hi = read_lane(static_cast<uint32_t>(u >> 32), src);
  return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}
// Orders this wavefront's own memory operations for the compiler.  The lanes of one wavefront issue
// their loads and stores as one in-order instruction stream, so a store followed by a load of the
// same address needs no wait — only that the compiler keeps them in program order.
OSG_D void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
// A zero the compiler must treat as a per-lane value: whatever is combined with it is computed by
// the vector unit, even when every lane holds the same number.
OSG_D uint32_t vector_zero() {
  uint32_t r;
  asm("v_mov_b32 %0, 0" : "=v"(r));
  return r;
}
// base[byte_off / sizeof(T)] with a 32-bit per-lane byte offset: the uniform base stays in scalar registers and the
// load needs no 64-bit address arithmetic on the vector unit (a node pool holds fewer than 2^28 nodes per root).
template <class T>
OSG_D T load_at(const T* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
OSG_D void store_at(T* base, uint32_t byte_off, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
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
// Wave-wide max / min as a DPP reduction (gfx9 row_shr 1, 2, 4, 8, then row_bcast 15 and 31): six
// register-to-register steps, no LDS permutes; lane 63 ends up with the result, which is handed back
// wave-uniform.  All 64 lanes must be active.
template <int kCtrl, int kRowMask>
OSG_D uint32_t dpp_move(uint32_t identity, uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(identity), static_cast<int>(v), kCtrl,
                                                           kRowMask, 0xf, false));
}
template <int kCtrl, int kRowMask>
OSG_D double dpp_max_step(double v) {
  const uint64_t u = static_cast<uint64_t>(__double_as_longlong(v));
  constexpr uint64_t kNegInf = 0xFFF0000000000000ull;
  const uint32_t lo = dpp_move<kCtrl, kRowMask>(static_cast<uint32_t>(kNegInf), static_cast<uint32_t>(u));
  const uint32_t hi = dpp_move<kCtrl, kRowMask>(static_cast<uint32_t>(kNegInf >> 32), static_cast<uint32_t>(u >> 32));
  const double o = __longlong_as_double(static_cast<long long>(static_cast<uint64_t>(hi) << 32 | lo));
  return fmax(o, v);  // one v_max_f64 (the values are never NaN: a compare + two selects would be three)
}
OSG_D double wave_max(double v) {
  v = dpp_max_step<0x111, 0xf>(v);  // row_shr:1
  v = dpp_max_step<0x112, 0xf>(v);  // row_shr:2
  v = dpp_max_step<0x114, 0xf>(v);  // row_shr:4
  v = dpp_max_step<0x118, 0xf>(v);  // row_shr:8
  v = dpp_max_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_max_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return read_lane_f64(v, 63);
}
template <int kCtrl, int kRowMask>
OSG_D uint32_t dpp_min_step(uint32_t v) {
  const uint32_t o = dpp_move<kCtrl, kRowMask>(0xFFFFFFFFu, v);
  return o < v ? o : v;
}
template <int kCtrl, int kRowMask>
OSG_D float dpp_maxf_step(float v) {
  const float o = __uint_as_float(dpp_move<kCtrl, kRowMask>(0xFF800000u, __float_as_uint(v)));   // identity: -infinity
  return fmaxf(o, v);
}
OSG_D float wave_max_f32(float v) {  // never NaN-sensitive here: a NaN input loses every v_max and the caller falls back
  v = dpp_maxf_step<0x111, 0xf>(v);
  v = dpp_maxf_step<0x112, 0xf>(v);
  v = dpp_maxf_step<0x114, 0xf>(v);
  v = dpp_maxf_step<0x118, 0xf>(v);
  v = dpp_maxf_step<0x142, 0xa>(v);
  v = dpp_maxf_step<0x143, 0xc>(v);
  return __uint_as_float(read_lane(__float_as_uint(v), 63));
}
OSG_D uint32_t wave_min_u32(uint32_t v) {
  v = dpp_min_step<0x111, 0xf>(v);
  v = dpp_min_step<0x112, 0xf>(v);
  v = dpp_min_step<0x114, 0xf>(v);
  v = dpp_min_step<0x118, 0xf>(v);
  v = dpp_min_step<0x142, 0xa>(v);
  v = dpp_min_step<0x143, 0xc>(v);
  return read_lane(v, 63);
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

// ---- selection at one node (mcts.cc:324-341) ----------------------------------------------------------
// arg-max of UCTValue / PUCTValue (mcts.cc:90-112) over the node's children, ties to the smallest order key (=
// first in the reference's shuffled order, mcts.cc:294,336).  Lane l scans child l (and l + 64 when kSlots = 2:
// the caller picks the instantiation by the child count, so a node with at most 64 children carries no trace of
// the second slot).  Headers AND statistics come in one memory round trip with no lane-dependent control flow:
// lanes beyond the last child read child 0 again (a clamped index instead of an exec-mask branch) and are kept out
// of every candidate set.  Candidate sets travel as lane masks; the chosen child's header is handed down by
// readlane from the lane that scanned it.  (The kernel is bound by scalar / branch issue, not by memory: a few
// redundant loads are cheaper than the branches that would avoid them.)
struct Chosen {
  int k;
  uint32_t meta, cnt, first;
  double tot;
};
template <int kSlots, bool kBoard, bool kWide = false>
OSG_D Chosen select_child(const uint32_t* __restrict__ META, const uint32_t* __restrict__ COUNT,
                          const uint32_t* __restrict__ FIRST, const double* __restrict__ TOTAL, uint32_t first, int c,
                          uint32_t cnt, const osg_mcts_cfg& cfg, const double* __restrict__ log_table, uint64_t obase,
                          uint64_t ph) {
  const int lane = lane_id();
  constexpr uint32_t kOutcomeBit = 1u << 20;
  uint32_t cm[kSlots], cc[kSlots], cf[kSlots];
  double ct[kSlots];
  bool in[kSlots];
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {  // every load of the level is in flight before the first value is looked at
    const int k = lane + 64 * j;
    in[j] = k < c;
    const uint32_t off = (first + static_cast<uint32_t>(in[j] ? k : 0)) * 4u;
    cm[j] = load_at(META, off);
    cc[j] = load_at(COUNT, off);
    cf[j] = load_at(FIRST, off);
    ct[j] = load_at(TOTAL, off * 2u);
  }
  uint64_t unvisited[kSlots], cand[kSlots];
  uint64_t any_unvisited = 0ull, any_outcome = 0ull;
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {
    // never visited and without a proven outcome, in one compare: count | outcome bit | "not a child"
    unvisited[j] = __ballot((cc[j] | (cm[j] & kOutcomeBit) | (in[j] ? 0u : 1u)) == 0u);
    any_unvisited |= unvisited[j];
    any_outcome |= __ballot((cm[j] & kOutcomeBit) != 0u);
  }
  const bool puct = cfg.child_selection_policy == 1;
  if (!puct && any_unvisited != 0ull) {
    // Some child has never been visited: its value is +infinity (mcts.cc:95), so the maximum is +infinity
    // whatever the others score — no UCT arithmetic at this node.
#pragma unroll
    for (int j = 0; j < kSlots; ++j) cand[j] = unvisited[j];
  } else {
    bool decided = false;
#if OSG_UCT_FILTER
    if constexpr (kBoard) {
      if (!puct && any_outcome == 0ull) {
        // The arg-max WITHOUT the fp64 divisions and square root where single precision already decides it.  Every value
        // is computed in fp32 first (error below 2^-20 of 1 + |c| sqrt(log n): the returns of these games lie in
        // [-1, 1], each of the five operations is good to an ulp or two).  Only children whose fp32 value is within 2^-18 of
        // the largest (four times that bound) can hold the exact maximum.  One such child: it is the arg-max.  Several
        // with IDENTICAL statistics (the usual case deep in the tree: siblings visited once or twice each): their exact
        // values are the same number, so they are exactly the tied maxima, and the order key below picks among them as
        // it would have.  Anything else (values that close from different statistics, overflow, NaN) takes the exact path.
        const float lf = static_cast<float>(log_table[cnt]), cf = static_cast<float>(cfg.uct_c);
        float a[kSlots];
        float am = -INFINITY;
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
          const float rc = __builtin_amdgcn_rcpf(static_cast<float>(cc[j]));
          const float val = static_cast<float>(ct[j]) * rc + cf * __builtin_amdgcn_sqrtf(lf * rc);
          a[j] = in[j] ? val : -INFINITY;
          am = fmaxf(am, a[j]);
        }
        const float top = wave_max_f32(am);
        const float floor_v = top - 0x1p-18f * (1.0f + fabsf(cf) * __builtin_amdgcn_sqrtf(lf));
        uint64_t near[kSlots];
        int n_near = 0;
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
          near[j] = __ballot(a[j] >= floor_v);
          n_near += __builtin_popcountll(near[j]);
        }
        decided = n_near == 1;
        if (n_near > 1) {  // the same (count, total) in every one of them?
          int slot = kSlots - 1;
          uint64_t set = near[kSlots - 1];
#pragma unroll
          for (int j = kSlots - 2; j >= 0; --j) {
            const bool here = near[j] != 0ull;
            slot = here ? j : slot;
            set = here ? near[j] : set;
          }
          const int src = uniform(static_cast<int>(__builtin_ctzll(set)));
          uint32_t sc = cc[0];
          double st = ct[0];
#pragma unroll
          for (int j = 1; j < kSlots; ++j) {
            sc = slot == j ? cc[j] : sc;
            st = slot == j ? ct[j] : st;
          }
          const uint32_t c0 = read_lane(sc, src);
          const uint64_t t0 = read_lane_u64(static_cast<uint64_t>(__double_as_longlong(st)), src);
          uint64_t odd = 0ull;
#pragma unroll
          for (int j = 0; j < kSlots; ++j)
            odd |= near[j] & ~__ballot(cc[j] == c0 && static_cast<uint64_t>(__double_as_longlong(ct[j])) == t0);
          decided = odd == 0ull;
        }
        if (decided) {
#pragma unroll
          for (int j = 0; j < kSlots; ++j) cand[j] = near[j];
        }
      }
    }
#endif
    if (!decided) {
    double v[kSlots];
    if (!puct && any_outcome == 0ull) {
      // the common case, straight-line: every child has been visited, none has a proven outcome
      const double logn = log_table[cnt];
#pragma unroll
      for (int j = 0; j < kSlots; ++j) {
        const double val = ct[j] / cc[j] + cfg.uct_c * sqrt(logn / cc[j]);
        v[j] = in[j] ? val : -INFINITY;
      }
    } else if (puct) {  // uniform branch: the two policies share nothing but the loads
      const double prior = 1.0 / c, sqrt_n = sqrt(static_cast<double>(cnt));
#pragma unroll
      for (int j = 0; j < kSlots; ++j) {
        v[j] = -INFINITY;
        if (in[j]) {
          if (m_has_outcome(cm[j])) v[j] = outcome_value<kBoard>(cm[j], cc[j], ct[j], m_player(cm[j]));
          else v[j] = (cc[j] != 0 ? ct[j] / cc[j] : 0.0) + cfg.uct_c * prior * sqrt_n / (cc[j] + 1);
        }
      }
    } else {
      const double logn = log_table[cnt];
#pragma unroll
      for (int j = 0; j < kSlots; ++j) {
        double val = ct[j] / cc[j] + cfg.uct_c * sqrt(logn / cc[j]);
        if (m_has_outcome(cm[j])) val = outcome_value<kBoard>(cm[j], cc[j], ct[j], m_player(cm[j]));
        v[j] = in[j] ? val : -INFINITY;
      }
    }
    double vm = v[0];
#pragma unroll
    for (int j = 1; j < kSlots; ++j) vm = v[j] > vm ? v[j] : vm;
    const double vmax = wave_max(vm);  // value only: 2 dwords per reduction step
    // (lanes without a child hold -infinity and c >= 1, so they never equal the maximum)
#pragma unroll
    for (int j = 0; j < kSlots; ++j) cand[j] = __ballot(v[j] == vmax);
    }
  }
  int total = 0;
#pragma unroll
  for (int j = 0; j < kSlots; ++j) total += __builtin_popcountll(cand[j]);
  if (total != 1) {  // several maxima: the smallest order key among them
    uint32_t key[kSlots];
    uint32_t km = 0xFFFFFFFFu;  // no candidate's key: the low byte of a key is an action, never 0xFF
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
      const uint32_t kj = order_key(obase, ph, static_cast<int>(mw_action<kWide>(cm[j])));
      key[j] = __builtin_amdgcn_inverse_ballot_w64(cand[j]) ? kj : 0xFFFFFFFFu;
      km = key[j] < km ? key[j] : km;
    }
    const uint32_t kmin = wave_min_u32(km);
#pragma unroll
    for (int j = 0; j < kSlots; ++j) cand[j] = __ballot(key[j] == kmin);
  }
  Chosen r;
  if constexpr (kSlots == 1) {
    const int src = uniform(static_cast<int>(__builtin_ctzll(cand[0])));
    r.k = src;
    r.meta = read_lane(cm[0], src);
    r.cnt = read_lane(cc[0], src);
    r.first = read_lane(cf[0], src);
    r.tot = read_lane_f64(ct[0], src);
  } else if constexpr (kSlots == 2) {
    const bool hi = cand[0] == 0ull;
    const int src = uniform(static_cast<int>(__builtin_ctzll(hi ? cand[1] : cand[0])));
    r.k = src + (hi ? 64 : 0);
    // the slot is picked per lane first so that each field costs one readlane
    r.meta = read_lane(hi ? cm[1] : cm[0], src);
    r.cnt = read_lane(hi ? cc[1] : cc[0], src);
    r.first = read_lane(hi ? cf[1] : cf[0], src);
    r.tot = read_lane_f64(hi ? ct[1] : ct[0], src);
  } else {
    // the first slot that holds a candidate (children are in action order: the smallest action among equal keys)
    int slot = kSlots - 1;
    uint64_t set = cand[kSlots - 1];
#pragma unroll
    for (int j = kSlots - 2; j >= 0; --j) {
      const bool here = cand[j] != 0ull;
      slot = here ? j : slot;
      set = here ? cand[j] : set;
    }
    const int src = uniform(static_cast<int>(__builtin_ctzll(set)));
    r.k = src + 64 * slot;
    uint32_t sm = cm[0], sc = cc[0], sf = cf[0];
    double st = ct[0];
#pragma unroll
    for (int j = 1; j < kSlots; ++j) {
      sm = slot == j ? cm[j] : sm;
      sc = slot == j ? cc[j] : sc;
      sf = slot == j ? cf[j] : sf;
      st = slot == j ? ct[j] : st;
    }
    r.meta = read_lane(sm, src);
    r.cnt = read_lane(sc, src);
    r.first = read_lane(sf, src);
    r.tot = read_lane_f64(st, src);
  }
  return r;
}

// --- hex playout as a wave-parallel random fill --------------------------------------------
// Lane l owns cells l, l + 64, ... l + 64 (kS - 1): kS "slots" (kS = 2 for boards of up to 128 cells, 3 / 4 / 6 for the
// boards with six / eight / twelve plane words: 13 x 13, 15 x 15, 19 x 19).  Per lane and slot: the set of the cell's (up
// to six) neighbours as 64-bit masks over the cell sets it can reach.  Per wavefront (uniform, in SGPRs): which cells
// are on the board / on black's two edges.  Sets of cells travel as kS 64-bit lane masks (cells 64 j ... 64 j + 63), so
// set algebra runs on the scalar unit and the vector unit only does the per-cell tests.
//
// A cell's neighbours are at most `cols` cells away, so with cols <= 64 (the launcher's condition for kS > 2) the
// neighbours of a cell of set j lie in sets j - 1, j, j + 1: a band of kB = min(kS, 3) masks per slot, the window of
// slot j starting at set hex_win<kS>(j).  (kS = 2: both sets for both slots, as before.)
template <int kS>
constexpr int hex_band() { return kS < 3 ? kS : 3; }
template <int kS>
constexpr int hex_win(int j) { return j - 1 < 0 ? 0 : (j - 1 > kS - hex_band<kS>() ? kS - hex_band<kS>() : j - 1); }
// (OSG_HEX_ONE_SET=1: hex boards of up to 64 cells — two plane words — walk with ONE cell set per colour and one child slot
// per lane; 0 keeps two, the form up to round 6's first sessions)
#ifndef OSG_HEX_ONE_SET
#define OSG_HEX_ONE_SET 1
#endif
template <class G> struct hex_plane_words { static constexpr int value = kMaskWords; };
template <int NW> struct hex_plane_words<HexT<NW>> { static constexpr int value = NW; };
template <class G>
constexpr int wave_sets() {
  return G::kMaskW > kMaskWords ? (G::kMaskW + 1) / 2 : ((OSG_HEX_ONE_SET != 0 && is_hex<G>::value && hex_plane_words<G>::value <= 2) ? 1 : 2);
}
// Which boards flood on packed per-lane flags (HexLaneT): those above 128 cells; OSG_PACKED_FLOOD_2=1 builds the form
// for the smaller boards too (an A/B build: hex(9) config 4 1.255e9 -> 1.11e9 simulations/s at 7 and at 6 wavefronts per
// SIMD, 10 / 4 vector registers in scratch — with two cell sets the neighbour bands are the cheaper step;
// profiles/r06zw_*).
#ifndef OSG_PACKED_FLOOD_2
#define OSG_PACKED_FLOOD_2 0
#endif
template <int kS>
constexpr bool hex_packed() { return kS > 2 || OSG_PACKED_FLOOD_2 != 0; }
template <int kS>
struct HexLaneT {
  uint64_t nb[kS][hex_band<kS>()];  // slot j: neighbours among the cells of sets hex_win(j) ... hex_win(j) + kB - 1
  uint32_t edge;      // slot j, bits 4 j ... 4 j + 3: first row 1, last row 2, first column 4, last column 8
  uint64_t board[kS], first_row[kS], last_row[kS];  // wave-uniform cell sets
  // kS > 2 (the boards above 128 cells): the floods run on PACKED per-lane flags instead — bit kFlag0 + j of a lane's
  // word = its cell of slot j — and a step PULLS the frontier flags of the six neighbour directions (+1, -1, +C, -C,
  // +C-1, -C+1: hex.cc:316-329) from the lanes that hold them: six cross-lane reads whatever kS is, against kS neighbour
  // bands of six 32-bit ANDs each.  fl_addr[d]: the byte address (lane * 4) of the lane holding cell + d;  fl_w[d]: bits
  // kFlag0 + j = "the cell of slot j has a neighbour in direction d", bit 0 = the neighbour sits one slot further (d > 0)
  // / one slot back (d < 0) because lane + d left 0 ... 63;  fl_edge[e]: the lane's cells on the first row, last row,
  // first column, last column.
  uint32_t fl_addr[6], fl_w[6], fl_edge[4];
};
constexpr int kFlag0 = 8;
// (+1, -1, +C, -C, +C-1, -C+1 for C columns)
OSG_D int hex_dir(int d, int cols) { return d == 0 ? 1 : d == 1 ? -1 : d == 2 ? cols : d == 3 ? -cols : d == 4 ? cols - 1 : 1 - cols; }
template <class G>
OSG_D HexLaneT<wave_sets<G>()> hex_lane_setup(const typename G::Params& p) {
  constexpr int kS = wave_sets<G>(), kB = hex_band<kS>();
  HexLaneT<kS> hl;
  const int lane = lane_id();
  hl.edge = 0u;
#pragma unroll
  for (int j = 0; j < kS; ++j) {
    const int cell = lane + 64 * j;
    const bool on_board = cell < p.cells;
    const typename G::Bits nb = G::neighbours(p, G::single(on_board ? cell : 0));
    constexpr int kWords = static_cast<int>(sizeof(nb.w) / sizeof(nb.w[0]));
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      const int set = hex_win<kS>(j) + t;
      const uint32_t lo = 2 * set < kWords ? nb.w[2 * set < kWords ? 2 * set : 0] : 0u;
      const uint32_t hi = 2 * set + 1 < kWords ? nb.w[2 * set + 1 < kWords ? 2 * set + 1 : 0] : 0u;
      hl.nb[j][t] = on_board ? (static_cast<uint64_t>(hi) << 32 | lo) : 0ull;
    }
    hl.board[j] = uniform64(__ballot(on_board));
    hl.first_row[j] = uniform64(__ballot(on_board && G::test(p.row_first, cell)));
    hl.last_row[j] = uniform64(__ballot(on_board && G::test(p.row_last, cell)));
    uint32_t e = 0;
    if (on_board) {
      e = (G::test(p.row_first, cell) ? 1u : 0u) | (G::test(p.row_last, cell) ? 2u : 0u) |
          (G::test(p.col_first, cell) ? 4u : 0u) | (G::test(p.col_last, cell) ? 8u : 0u);
    }
    hl.edge |= e << (4 * j);
    if constexpr (hex_packed<kS>()) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (j == 0) hl.fl_edge[q] = 0u;
        hl.fl_edge[q] |= ((e >> q) & 1u) << (kFlag0 + j);
      }
#pragma unroll
      for (int d = 0; d < 6; ++d) {
        const int step = hex_dir(d, p.cols);
        if (j == 0) {
          const int to = lane + step;
          hl.fl_addr[d] = static_cast<uint32_t>(to & 63) << 2;
          hl.fl_w[d] = (to < 0 || to > 63) ? 1u : 0u;
        }
        const int other = cell + step;
        const bool has = on_board && other >= 0 && other < p.cells && G::test(nb, other < 0 ? 0 : (other < p.cells ? other : 0));
        hl.fl_w[d] |= (has ? 1u : 0u) << (kFlag0 + j);
      }
    }
  }
  return hl;
}
// Packed flags (kS > 2): the cells of kS wave-uniform sets as one word per lane, bit kFlag0 + j = the lane's cell of slot j.
template <int kS>
OSG_D uint32_t hex_pack(const uint64_t* sets) {
  uint32_t bits = 0u;
#pragma unroll
  for (int j = 0; j < kS; ++j) bits |= __builtin_amdgcn_inverse_ballot_w64(sets[j]) ? (1u << (kFlag0 + j)) : 0u;
  return bits;
}
// One flood step in the packed form: the cells that have a neighbour in `front`.
template <int kS>
OSG_D uint32_t hex_pull(const HexLaneT<kS>& hl, uint32_t front) {
  uint32_t acc = 0u;
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    uint32_t g = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(hl.fl_addr[d]), static_cast<int>(front)));
    // the neighbour's flag sits one slot off where lane + d wrapped (the shift reads bits [4:0] of fl_w: 0 or 1)
    g = (d == 0 || d == 2 || d == 4) ? g >> (hl.fl_w[d] & 31u) : g << (hl.fl_w[d] & 31u);
    acc |= g & hl.fl_w[d];   // (flags live at bits >= kFlag0, so fl_w's bit 0 never matches one)
  }
  return acc;
}
// Cells 64 j ... 64 j + 63 of a (wave-uniform) bitboard as one 64-bit set.
template <class G>
OSG_D uint64_t hex_cells64(const typename G::Bits& b, int j) {
  constexpr int kWords = static_cast<int>(sizeof(b.w) / sizeof(b.w[0]));
  const uint32_t lo = 2 * j < kWords ? b.w[2 * j < kWords ? 2 * j : 0] : 0u;
  const uint32_t hi = 2 * j + 1 < kWords ? b.w[2 * j + 1 < kWords ? 2 * j + 1 : 0] : 0u;
  return uniform64(static_cast<uint64_t>(hi) << 32 | lo);
}
// The cells of a slot's neighbour band that lie in `front` (kS sets): nonzero iff one of the cell's neighbours is there.
template <int kS>
OSG_D uint32_t hex_touch(const HexLaneT<kS>& hl, int j, const uint64_t* front) {
  constexpr int kB = hex_band<kS>();
  uint64_t x = 0ull;
#pragma unroll
  for (int t = 0; t < kB; ++t) x |= hl.nb[j][t] & front[hex_win<kS>(j) + t];
  return static_cast<uint32_t>(x) | static_cast<uint32_t>(x >> 32);
}
template <int kS>
OSG_D bool sets_meet(const uint64_t* a, const uint64_t* b) {  // (a & b) != 0 over the kS sets
  uint64_t x = 0ull;
#pragma unroll
  for (int j = 0; j < kS; ++j) x |= a[j] & b[j];
  return x != 0ull;
}
template <int kS>
OSG_D bool sets_any(const uint64_t* a) {
  uint64_t x = 0ull;
#pragma unroll
  for (int j = 0; j < kS; ++j) x |= a[j];
  return x != 0ull;
}

// The hex position the search walks with (no swap rule, at least two rows and two columns): the two
// colours' stones as kS 64-bit cell sets each and the player to move, wave-uniform, so that the rules
// run as scalar set algebra plus lane-parallel neighbour tests against HexLane.
//
// The reference keeps an edge-connection label on every stone and relabels a group at every move
// (hex.cc:108-171, 229-278) because IsTerminal() must be cheap at every state.  A tree descent does not
// need that: a node that has children was not terminal when it was expanded, and a node found terminal
// keeps that in its header, so the only state whose IsTerminal() the search ever asks for is the one a
// simulation stops at on its FIRST visit.  There the question is answered from the stones alone: the
// game ended with the last move iff the group of the stone just placed touches both of its colour's
// edges — the labels' invariant (a stone carries an edge label iff its group reaches that edge; the
// `else if` of hex.cc:122-126 only matters on a one-row / one-column board, which this kernel is not
// launched for) makes the two statements the same.
template <int kS>
struct HexWT {
  uint64_t occ[kS], blk[kS];  // all stones, black's stones (white = occ & ~blk: one set to update per move)
  uint32_t meta;              // to move [0], result [1:3) as HexT::State::meta
};
// The legal actions of a hex position without the swap rule: its empty cells, as kS lane masks (hex.cc:280-293).
template <int kS>
struct HexLegalT {
  uint64_t e[kS];
  OSG_D int count() const {
    int c = 0;
#pragma unroll
    for (int j = 0; j < kS; ++j) c += __builtin_popcountll(e[j]);
    return c;
  }
};
template <class G>
OSG_D HexWT<wave_sets<G>()> hexw_from_state(const typename G::State& s) {
  constexpr int kS = wave_sets<G>();
  HexWT<kS> w;
#pragma unroll
  for (int j = 0; j < kS; ++j) {
    w.blk[j] = hex_cells64<G>(s.black, j);
    w.occ[j] = w.blk[j] | hex_cells64<G>(s.white, j);
  }
  w.meta = uniform(s.meta) & 7u;
  return w;
}
template <int kS>
OSG_D bool hexw_terminal(const HexWT<kS>& w) { return ((w.meta >> 1) & 3u) != 0; }
template <int kS>
OSG_D int hexw_current_player(const HexWT<kS>& w) { return hexw_terminal(w) ? kTerminalPlayer : static_cast<int>(w.meta & 1u); }
template <int kS>
OSG_D HexLegalT<kS> hexw_legal(const HexLaneT<kS>& hl, const HexWT<kS>& w) {
  HexLegalT<kS> m;
  const bool over = hexw_terminal(w);
#pragma unroll
  for (int j = 0; j < kS; ++j) m.e[j] = over ? 0ull : hl.board[j] & ~w.occ[j];
  return m;
}
template <int kS>
OSG_D void hexw_returns(const HexWT<kS>& w, double* out) {  // hex.cc:363-365
  const int res = (w.meta >> 1) & 3u;
  const double r = res == 1 ? 1.0 : (res == 2 ? -1.0 : 0.0);
  out[0] = r;
  out[1] = -r + 0.0;
}
// DoApplyAction (hex.cc:229-278) on the way down the tree: the stone and the turn, nothing else.
template <int kS>
OSG_D void hexw_apply(const HexLaneT<kS>&, HexWT<kS>& w, int move) {
  const uint64_t bit = 1ull << (move & 63);
  const uint64_t black = (w.meta & 1u) == 0 ? ~0ull : 0ull;
  if constexpr (kS == 2) {
    const bool hi = move >= 64;
    const uint64_t bit0 = hi ? 0ull : bit, bit1 = hi ? bit : 0ull;
    w.occ[0] |= bit0;
    w.occ[1] |= bit1;
    w.blk[0] |= bit0 & black;
    w.blk[1] |= bit1 & black;
  } else {
    const int set = move >> 6;
#pragma unroll
    for (int j = 0; j < kS; ++j) {
      const uint64_t bj = set == j ? bit : 0ull;
      w.occ[j] |= bj;
      w.blk[j] |= bj & black;
    }
  }
  w.meta ^= 1u;
}
// Did the stone just placed on `move` end the game?  (= would hex.cc:248 have labelled it Win.)  Lane-parallel
// flood of its group: a stone of the same colour joins when one of its neighbours is in the frontier.
template <int kS>
OSG_D bool hexw_last_stone_wins(const HexLaneT<kS>& hl, const HexWT<kS>& w, int move) {
  const bool black = (w.meta & 1u) != 0;  // the owner of the stone is the player who is NOT to move now
  if constexpr (hex_packed<kS>()) {  // packed flags (see HexLaneT)
    uint64_t own_sets[kS];
#pragma unroll
    for (int j = 0; j < kS; ++j) own_sets[j] = black ? w.blk[j] : w.occ[j] & ~w.blk[j];
    const uint32_t own = hex_pack<kS>(own_sets);
    const uint32_t fe = black ? hl.fl_edge[0] : hl.fl_edge[2], le = black ? hl.fl_edge[1] : hl.fl_edge[3];
    if ((__ballot((own & fe) != 0u) == 0ull) | (__ballot((own & le) != 0u) == 0ull)) return false;
    uint32_t front = lane_id() == (move & 63) ? 1u << (kFlag0 + (move >> 6)) : 0u;
    uint32_t group = front, avail = own & ~front;
    for (int it = 0; it < 64 * kS; ++it) {
      const uint32_t joined = hex_pull<kS>(hl, front) & avail;
      if (__ballot(joined != 0u) == 0ull) break;
      avail ^= joined;
      group |= joined;
      front = joined;
    }
    return (__ballot((group & fe) != 0u) != 0ull) & (__ballot((group & le) != 0u) != 0ull);
  }
  uint64_t own[kS], f[kS], l[kS];
  // the colour's two edges as cell sets: from the lanes' edge flags (black: rows = bits 0, 1; white: columns = bits 2, 3)
  const uint32_t fbit = black ? 1u : 4u, lbit = black ? 2u : 8u;
#pragma unroll
  for (int j = 0; j < kS; ++j) {
    own[j] = black ? w.blk[j] : w.occ[j] & ~w.blk[j];
    f[j] = __ballot((hl.edge & (fbit << (4 * j))) != 0u);
    l[j] = __ballot((hl.edge & (lbit << (4 * j))) != 0u);
  }
  // a chain needs a stone on each of the two edges
  if (!sets_meet<kS>(own, f) | !sets_meet<kS>(own, l)) return false;
  const uint64_t bit = 1ull << (move & 63);
  const int set = move >> 6;
  // (bookkeeping on the vector unit, as in the playout's flood: every lane keeps an all-ones word per own-colour cell
  // that has not joined yet; the scalar unit only sees the new frontier of a step)
  uint64_t group[kS], front[kS];
  uint32_t avail[kS];
#pragma unroll
  for (int j = 0; j < kS; ++j) {
    group[j] = set == j ? bit : 0ull;
    front[j] = group[j];
    avail[j] = __builtin_amdgcn_inverse_ballot_w64(own[j] & ~group[j]) ? ~0u : 0u;
  }
  for (int it = 0; it < 64 * kS; ++it) {
    uint32_t jn[kS];
#pragma unroll
    for (int j = 0; j < kS; ++j) jn[j] = hex_touch<kS>(hl, j, front) & avail[j];
#pragma unroll
    for (int j = 0; j < kS; ++j) front[j] = __ballot(jn[j] != 0u);
    if (!sets_any<kS>(front)) break;
#pragma unroll
    for (int j = 0; j < kS; ++j) {
      group[j] |= front[j];
      avail[j] = jn[j] != 0u ? 0u : avail[j];
    }
  }
  return sets_meet<kS>(group, f) & sets_meet<kS>(group, l);
}

// The rules as the search sees them: HexWT for the hex fill kernel, G::State otherwise.
template <class G>
OSG_D bool w_terminal(const typename G::Params& p, const typename G::State& s) { return G::terminal(p, s); }
template <class G, int kS>
OSG_D bool w_terminal(const typename G::Params&, const HexWT<kS>& w) { return hexw_terminal(w); }
template <class G>
OSG_D int w_current_player(const typename G::Params& p, const typename G::State& s) { return G::current_player(p, s); }
template <class G, int kS>
OSG_D int w_current_player(const typename G::Params&, const HexWT<kS>& w) { return hexw_current_player(w); }
template <class G, int kS>
OSG_D Mask w_legal(const typename G::Params& p, const HexLaneT<kS>&, const typename G::State& s) { return G::legal(p, s); }
template <class G, int kS>
OSG_D HexLegalT<kS> w_legal(const typename G::Params&, const HexLaneT<kS>& hl, const HexWT<kS>& w) { return hexw_legal(hl, w); }
template <class G, int kS>
OSG_D void w_apply(const typename G::Params& p, const HexLaneT<kS>&, typename G::State& s, int a) { G::apply(p, s, a); }
template <class G, int kS>
OSG_D void w_apply(const typename G::Params&, const HexLaneT<kS>& hl, HexWT<kS>& w, int a) { hexw_apply(hl, w, a); }
template <class G>
OSG_D void w_returns(const typename G::Params& p, const typename G::State& s, double* out) { G::returns(p, s, out); }
template <class G, int kS>
OSG_D void w_returns(const typename G::Params&, const HexWT<kS>& w, double* out) { hexw_returns(w, out); }

// The key that orders the empty cells of a playout: fill_key (32 mixed bits | the low byte of the cell id); on boards
// above 256 cells two cells 256 apart could tie, and the order is defined as (fill_key, cell) — here as ONE key with
// the cell's ninth bit below everything else, so that the threshold search still sees distinct keys.
template <int kS>
constexpr int fill_bits() { return kS > 4 ? kFillKeyBits + 1 : kFillKeyBits; }
template <int kS>
OSG_D uint64_t wave_fill_key(uint64_t base, int cell) {
  const uint64_t k = fill_key(base, cell);
  if constexpr (kS > 4) return (k << 1) | static_cast<uint64_t>(cell >> 8);
  else return k;
}

template <int kS>
OSG_D int hex_fill_winner(const HexWT<kS>& s, uint64_t base, const HexLaneT<kS>& hl PT_ARGS) {
  const int lane = lane_id();
  constexpr int kBits = fill_bits<kS>();
  uint64_t empty[kS], key[kS];
  int m = 0;
#pragma unroll
  for (int j = 0; j < kS; ++j) {
    empty[j] = hl.board[j] & ~s.occ[j];
    key[j] = wave_fill_key<kS>(base, lane + 64 * j);
    m += __builtin_popcountll(empty[j]);
  }
  const int want = (m + 1) >> 1;  // plies 0, 2, 4, ... belong to the player to move
  // The `want` smallest keys among the empty cells = the keys below a threshold T with exactly `want` keys
  // under it.  T is built most-significant bit first (a binary search on the key space): a bit stays
  // set while no more than `want` keys lie below.  Keys are distinct, so the search ends as soon as the
  // count is exact — about log2(m) + 2 steps of two compares and a handful of scalar instructions.
  uint64_t sel[kS];
#if defined(OSG_DIAG_NOTHR)  // measurement only: an arbitrary subset instead of the exact half
#pragma unroll
  for (int j = 0; j < kS; ++j) sel[j] = __ballot((key[j] >> 20) & 1ull) & empty[j];
#elif OSG_THR_MODE == 0
  uint64_t thr = 0ull;
  if (want > 0) {
    uint64_t step = 1ull << (kBits - 1);
    bool exact;
    do {  // straight-line body: one select, no inner branch
      const uint64_t probe = thr | step;
      int below = 0;
#pragma unroll
      for (int j = 0; j < kS; ++j) below += __builtin_popcountll(__ballot(key[j] < probe) & empty[j]);
      thr = below <= want ? probe : thr;
      exact = below == want;
      step >>= 1;
    } while (!exact && step != 0ull);
  }
#pragma unroll
  for (int j = 0; j < kS; ++j) sel[j] = __ballot(key[j] < thr) & empty[j];
#else
  // The search runs on the VECTOR unit (the kernel is bound by scalar issue): threshold and step live in
  // vector registers holding the same value in every lane, occupied cells carry the key 2^64 - 1 so that
  // the ballots need no masking, and only the counting and the loop branch are left to the scalar unit.
  const uint32_t vz = vector_zero();
#if OSG_THR_MODE == 3
  // First on the keys' 32 mixed bits alone (32-bit compares and selects; the cell-id bits below them only break ties):
  // a threshold with exactly `want` of those words under it selects the same cells as the full keys do.  Where two
  // empty cells share their 32 bits across the threshold (~m^2 / 2^33 of the playouts), or the largest word is all
  // ones, no such threshold exists and the full-key search below runs.
  bool found = want == 0;
  {
    uint32_t h[kS];
#pragma unroll
    for (int j = 0; j < kS; ++j)
      h[j] = __builtin_amdgcn_inverse_ballot_w64(empty[j]) ? static_cast<uint32_t>(fill_key(base, lane + 64 * j) >> 8) : ~0u;
    uint32_t t32 = vz, step32 = 0x80000000u | vz;
    const uint32_t want32 = static_cast<uint32_t>(want) | vz;
    if (want > 0) {
      for (int it = 0; it < 32; ++it) {
        const uint32_t probe = t32 | step32;
        int below = 0;
#pragma unroll
        for (int j = 0; j < kS; ++j) below += __builtin_popcountll(__ballot(h[j] < probe));
        const uint32_t below_v = static_cast<uint32_t>(below) | vz;
        t32 = below_v <= want32 ? probe : t32;
        step32 >>= 1;
        if (below == want) { found = true; break; }
      }
    }
#pragma unroll
    for (int j = 0; j < kS; ++j) sel[j] = __ballot(h[j] < t32);
  }
  if (!found) {
#endif
  uint64_t k[kS];
#pragma unroll
  for (int j = 0; j < kS; ++j) k[j] = __builtin_amdgcn_inverse_ballot_w64(empty[j]) ? key[j] : ~0ull;
  uint64_t thr = vz;
  if (want > 0) {
    uint64_t step = (1ull << (kBits - 1)) | vz;
    const uint32_t want_v = static_cast<uint32_t>(want) | vz;
    for (int it = 0; it < kBits; ++it) {
      const uint64_t probe = thr | step;
      int below = 0;
#pragma unroll
      for (int j = 0; j < kS; ++j) below += __builtin_popcountll(__ballot(k[j] < probe));
      const uint32_t below_v = static_cast<uint32_t>(below) | vz;
      thr = below_v <= want_v ? probe : thr;
      step >>= 1;
      if (below == want) break;
    }
  }
#pragma unroll
  for (int j = 0; j < kS; ++j) sel[j] = __ballot(k[j] < thr);
#if OSG_THR_MODE == 3
  }
#endif
#endif
  PT_MARK(4);
  // The filled board: the mover's new stones are `sel`, the opponent's the other empty cells.
  const bool black_moves = (s.meta & 1u) == 0;
  uint64_t blk[kS];
#pragma unroll
  for (int j = 0; j < kS; ++j) blk[j] = s.blk[j] | (black_moves ? sel[j] : empty[j] & ~sel[j]);
  // Black wins iff its stones join the first row to the last row (hex.cc:108-171 edge labels).
  // Lane-parallel flood: a black cell joins the region when one of its neighbours is in it.
  // Stops as soon as the last row is reached.
#if defined(OSG_DIAG_NOFLOOD)  // measurement only
  return static_cast<int>((blk[0] ^ blk[1] ^ (blk[0] >> 17)) & 1ull);
#elif OSG_FLOOD_MODE == 0
  uint64_t reach[kS];
#pragma unroll
  for (int j = 0; j < kS; ++j) reach[j] = blk[j] & hl.first_row[j];
#pragma unroll 4  // measured: 1 -> 7.80e8, compiler's choice (2) -> 7.93e8, 4 -> 8.01e8 sims/s
  for (int it = 0; it < 64 * kS; ++it) {
    if (sets_meet<kS>(reach, hl.last_row)) return 0;  // black
    uint64_t g[kS];
#pragma unroll
    for (int j = 0; j < kS; ++j) g[j] = __ballot(hex_touch<kS>(hl, j, reach) != 0u) & blk[j] & ~reach[j];
    if (!sets_any<kS>(g)) break;
#pragma unroll
    for (int j = 0; j < kS; ++j) reach[j] |= g[j];
  }
#elif OSG_FLOOD_MODE == 2
  static_assert(kS == 2, "flood mode 2 (a measurement variant) is written for boards of up to 128 cells");
  // As mode 1, with the two exits decided on the VECTOR unit too and only once per TWO steps: every lane remembers
  // whether one of its cells that joined lies on the last row (the "black arrived" exit) and whether its cells joined
  // in the second step (the "nothing new" exit); each exit is one compare into vcc and one branch.  Running one step
  // past either condition is harmless: an empty frontier stays empty, and a last-row cell that joined stays remembered.
  uint64_t front[2] = {blk[0] & hl.first_row[0], blk[1] & hl.first_row[1]};
  uint32_t avail0 = __builtin_amdgcn_inverse_ballot_w64(blk[0] & ~front[0]) ? ~0u : 0u;
  uint32_t avail1 = __builtin_amdgcn_inverse_ballot_w64(blk[1] & ~front[1]) ? ~0u : 0u;
  if (sets_meet<kS>(front, hl.last_row)) return 0;  // a one-row chain
  // all ones where the lane's cell is on the last row (edge bit 1 of cell l, bit 5 of cell l + 64)
  const uint32_t last0 = static_cast<uint32_t>(static_cast<int32_t>(hl.edge << 30) >> 31);
  const uint32_t last1 = static_cast<uint32_t>(static_cast<int32_t>(hl.edge << 26) >> 31);
  uint32_t hit;  // (lane-local) one of the lane's cells that joined in this pair of steps lies on the last row
  for (;;) {
    const uint32_t ja0 = hex_touch<kS>(hl, 0, front) & avail0;
    const uint32_t ja1 = hex_touch<kS>(hl, 1, front) & avail1;
    const uint64_t mid[2] = {__ballot(ja0 != 0u), __ballot(ja1 != 0u)};
    avail0 = ja0 != 0u ? 0u : avail0;
    avail1 = ja1 != 0u ? 0u : avail1;
    const uint32_t jb0 = hex_touch<kS>(hl, 0, mid) & avail0;
    const uint32_t jb1 = hex_touch<kS>(hl, 1, mid) & avail1;
    front[0] = __ballot(jb0 != 0u);
    front[1] = __ballot(jb1 != 0u);
    avail0 = jb0 != 0u ? 0u : avail0;
    avail1 = jb1 != 0u ? 0u : avail1;
    // (both exits lead to the same place and the answer is read off `hit` there: the loop stays two compares into
    // vcc and two branches, no exit-code bookkeeping on the scalar unit)
    hit = ((ja0 | jb0) & last0) | ((ja1 | jb1) & last1);
    if (__ballot(hit != 0u) != 0ull) break;            // black reached its last row
    if (__ballot((jb0 | jb1) != 0u) == 0ull) break;    // nothing new: black's region is closed
  }
  return __ballot(hit != 0u) != 0ull ? 0 : 1;
#else
  if constexpr (hex_packed<kS>()) {  // packed flags (see HexLaneT): a step costs six cross-lane reads whatever kS is
    const uint32_t black = hex_pack<kS>(blk);
    uint32_t front = black & hl.fl_edge[0];
    uint32_t avail = black & ~front;
    if (__ballot((front & hl.fl_edge[1]) != 0u) != 0ull) return 0;  // a one-row chain
    for (int it = 0; it < 64 * kS; ++it) {
      const uint32_t joined = hex_pull<kS>(hl, front) & avail;
      if (__ballot((joined & hl.fl_edge[1]) != 0u) != 0ull) return 0;  // black reached its last row
      if (__ballot(joined != 0u) == 0ull) break;
      avail ^= joined;
      front = joined;
    }
    return 1;
  }
  // The bookkeeping of the flood on the vector unit: every lane keeps, for each of its cells, an all-ones word
  // while the cell is black and not reached yet ("available") and clears it when the cell joins; the scalar
  // unit only sees the kS ballots of a step (the new frontier) and decides the two exits.
  uint64_t front[kS];
  uint32_t avail[kS];
#pragma unroll
  for (int j = 0; j < kS; ++j) {
    front[j] = blk[j] & hl.first_row[j];
    avail[j] = __builtin_amdgcn_inverse_ballot_w64(blk[j] & ~front[j]) ? ~0u : 0u;
  }
  if (sets_meet<kS>(front, hl.last_row)) return 0;  // a one-row chain
#ifndef OSG_FLOOD_UNROLL
#define OSG_FLOOD_UNROLL 2
#endif
#pragma unroll OSG_FLOOD_UNROLL
  for (int it = 0; it < 64 * kS; ++it) {
    uint32_t jn[kS];
#pragma unroll
    for (int j = 0; j < kS; ++j) jn[j] = hex_touch<kS>(hl, j, front) & avail[j];
#pragma unroll
    for (int j = 0; j < kS; ++j) front[j] = __ballot(jn[j] != 0u);
    if (sets_meet<kS>(front, hl.last_row)) return 0;  // black reached its last row
    if (!sets_any<kS>(front)) break;
#pragma unroll
    for (int j = 0; j < kS; ++j) avail[j] = jn[j] != 0u ? 0u : avail[j];
  }
#endif
  return 1;  // white: on a filled board exactly one side connects
}

// The visit path of the running simulation: the d-th entry is the node id [0:28) | META's player field [28:32)
// and the node's visit count / total reward as they were when the path went through it, so that the backup is
// stores only (no read-modify-write round trip to the pool).  LANE d KEEPS ENTRY d IN REGISTERS (three selects per
// tree level, no memory and nothing for the scalar unit; the backup has lane d own path node d anyway); entries
// 64 ... kMaxPath - 1 — reachable only in games longer than 64 plies — live in LDS.  Entry 0 is the root, whose
// statistics persist in lane 0 from one simulation to the next.
#ifndef OSG_PATH_REGS
#define OSG_PATH_REGS 64  // (a test build with 2 sends every deeper entry through the LDS part)
#endif
constexpr int kPathRegs = OSG_PATH_REGS;
struct VisitPath {
  uint32_t node, cnt;
  double tot;
  uint32_t* l_node;
  uint32_t* l_cnt;
  double* l_tot;
  OSG_D void set(int d, uint32_t e, uint32_t c, double t) {  // d is wave-uniform
    if (d < kPathRegs) {
      const bool me = lane_id() == d;
      node = me ? e : node;
      cnt = me ? c : cnt;
      tot = me ? t : tot;
    } else if (lane_id() == 0) {
      l_node[d - kPathRegs] = e;
      l_cnt[d - kPathRegs] = c;
      l_tot[d - kPathRegs] = t;
    }
  }
  OSG_D uint32_t node_at(int d) const {  // d is wave-uniform
    return d < kPathRegs ? read_lane(node, d) : uniform(l_node[d - kPathRegs]);
  }
};

#ifndef OSG_HEX_WPE
#define OSG_HEX_WPE 7
#endif
// Wavefronts per SIMD the hex fill kernel is compiled for, by the number of cell sets of the position (the boards above
// 128 cells hold kS sets per colour in scalar registers and kS child slots per lane: fewer, fatter wavefronts).
#ifndef OSG_HEX_WPE_3
#define OSG_HEX_WPE_3 5
#endif
#ifndef OSG_HEX_WPE_4
#define OSG_HEX_WPE_4 5
#endif
#ifndef OSG_HEX_WPE_6
#define OSG_HEX_WPE_6 4
#endif
template <class G, bool kHexFill>
constexpr int wave_wpe() {
  if (!kHexFill) return 4;
  return wave_sets<G>() <= 2 ? OSG_HEX_WPE : (wave_sets<G>() == 3 ? OSG_HEX_WPE_3 : (wave_sets<G>() == 4 ? OSG_HEX_WPE_4 : OSG_HEX_WPE_6));
}

// The hex fill kernel at 7 waves per SIMD.  What the code object says (tools/kernel_resources.py ->
// profiles/r05_kernel_resources.txt): 72 vector + 94 scalar registers, 44 scalar registers parked in the lanes of one
// vector register (v_writelane in the prologue only; 7 v_readlane at the head of a simulation — the root position —
// and ~16 more on the expansion / first-visit paths, against ~520 vector instructions per simulation, on the vector
// pipe, which is not the one that bounds this kernel) and 2 vector registers (12 B per lane) in scratch, touched on the
// first-visit path.  Round 5 tried to take the parked registers out — the wave-uniform cell sets (board, first / last
// row) dropped in favour of "off-board cells are occupied" and two ballots per playout: 44 -> 39 parked scalar and
// 2 -> 6 spilled vector registers, 1.108e9 -> 1.078e9 simulations/s on the same box (profiles/r05_ab_register_work.txt)
// — so the round-4 form stays; at 6 waves per SIMD the scalar budget is 100 instead of 88 (28 parked) and the search
// runs at 1.06e9.  Measured on config 4 with the final kernel: 6 waves 1.06e9, 7 waves 1.12e9, 8 waves (64 registers,
// more spilled) 1.08e9 simulations/s; the 2^13-root shard of an 8-GPU run 8.7e8 / 8.9e8 / 9.0e8.  One wavefront per workgroup instead of four: the same
// at 2^16 roots, 5 % less at 2^13.  The generic instantiations carry more per-lane state (their playouts run one per
// lane): 4 waves with a little scratch measured faster than 2-3 without.
// kGc: the instantiation that can garbage-collect (mcts.cc:441-482): it also records every node's parent.
// Kept out of the default instantiation so that the hex kernel's register budget is untouched.
// ONE search: root r by the calling wavefront.
template <class G, bool kBoard, bool kHexFill, bool kGc>
OSG_D void wave_search(const typename G::Params& p, const typename G::word_t* base, int64_t n, int num_players,
                       int num_actions, const osg_mcts_cfg& cfg, double max_utility, const double* __restrict__ log_table,
                       const Pool& pool, const MctsOut& out, const int64_t r, uint32_t* my_path, uint32_t* my_pcnt,
                       double* my_ptot) {
  const int lane = lane_id();
  // without chance nodes (kBoard) a path may use the LDS entries; the chance-skipping backup below looks entries
  // up across lanes and stays within the register entries (the games with chance nodes are far shorter anyway)
  constexpr int kPathLimit = kBoard ? kMaxPath : kPathRegs;
  const uint64_t gr = static_cast<uint64_t>(cfg.index_offset + r);
  const int cap = pool.cap;
  uint32_t* META = pool.meta + r * cap;
  uint32_t* FIRST = pool.first + r * cap;
  uint32_t* COUNT = pool.count + r * cap;
  double* TOTAL = pool.total + r * cap;
  uint32_t* PARENT = pool.parent + r * cap;  // kGc only
  uint32_t* REMAP = kGc ? pool.remap + r * cap : nullptr;
  int gc_limit = kMinGcLimit;
  const uint64_t obase = order_base(cfg.seed, gr);
  constexpr int kS = wave_sets<G>();   // 64-cell sets per position (hex fill) = child slots per lane
  constexpr bool kWide = kS > 2;       // nine-bit action / child-count fields in a node's header (osg_mcts_internal.h)
  static_assert(kHexFill || !kWide, "the boards above 128 actions are searched as hex without the swap rule only");
  HexLaneT<kS> hl{};
  if constexpr (kHexFill) hl = hex_lane_setup<G>(p);

  using WState = std::conditional_t<kHexFill, HexWT<kS>, typename G::State>;
  const typename G::State loaded_root = G::load(p, base, n, r);
  WState root_state;
  if constexpr (kHexFill) root_state = hexw_from_state<G>(loaded_root);
  else root_state = loaded_root;
  const int root_player = w_current_player<G>(p, root_state);
  // The root's header stays in registers (its count / total in path slot 0); every other node's header
  // comes out of its parent's child scan by readlane, so a tree level costs ONE memory round trip.
  uint32_t root_meta = mw_make<kWide>(kWide ? 0x1FF : 0xFF, root_player, 0);  // mcts.cc:356-357
  uint32_t root_first = 0;
  if (lane == 0) {
    META[0] = root_meta;
    FIRST[0] = 0;
    COUNT[0] = 0;
    TOTAL[0] = 0.0;
  }
  VisitPath vp{(root_meta >> 8 & 15u) << 28, 0u, 0.0, my_path, my_pcnt, my_ptot};
  wave_fence();
  uint32_t used = 1;
  int sims_done = 0;
  // Hash chains whose value is the same in every lane (the path hash behind the sibling order, the playout's
  // fill base) run on the vector unit: the kernel is bound by scalar issue, and their consumers are per-lane anyway.
#if OSG_HASH_VALU
  const uint32_t vz = vector_zero();
#else
  const uint32_t vz = 0u;
#endif
  const uint32_t fill_word = kHexFill ? fill_root(cfg.seed, gr) : 0u;
  PT_DECL

  for (int sim = 0; sim < cfg.max_simulations; ++sim) {
    Rng trng(cfg.seed ^ kTreeSalt, gr, static_cast<uint64_t>(sim));
    // ---- ApplyTreePolicy (mcts.cc:273-351) ----
    WState s = root_state;
    uint32_t node = 0;
    int depth = 0;
    uint32_t ph = static_cast<uint32_t>(path_hash_root()) | vz;  // the same in every lane, kept off the scalar unit
    uint32_t meta = root_meta, first = root_first;
    uint32_t cnt = read_lane(vp.cnt, 0);
    bool term;
    PT_MARK(7);
    for (;;) {
      if constexpr (kHexFill) {
        // The walk goes on only through nodes that were visited before and are not known to be terminal (one
        // exit test; what kind of stop it was is sorted out after the loop).
        term = false;
        if (m_terminal(meta) | (cnt == 0) | (depth + 1 >= kPathLimit)) break;
      } else {
        term = w_terminal<G>(p, s);
        if (term || cnt == 0 || depth + 1 >= kPathLimit) break;
      }
      const int cur = w_current_player<G>(p, s);
      const auto legal = w_legal<G>(p, hl, s);
      PT_MARK(0);
      if (mw_nchild<kWide>(meta) == 0) {  // expand: one child per Prior() entry, in action order
        const int c = legal.count();
        // slots exhausted (unreachable unless the caller's HBM could not hold max_nodes + slack), or a record that is
        // not terminal and has no legal action (only an uploaded inconsistent one): leaf evaluation
        if (c == 0 || used + static_cast<uint32_t>(c) > static_cast<uint32_t>(cap)) break;
        first = used;
        used += c;
        // Children in action order.  Lane l looks at actions l, l + 64, ...: a legal action's slot is its
        // rank among the legal ones (popcount of the mask below it) — the cheap direction of the
        // k <-> action mapping — so the writes are still one compacted, coalesced span.
        if constexpr (kHexFill && (kWide || OSG_EXPAND_SETS)) {  // the legal set as lane masks: a rank is a running count + the lanes below
          int before = 0;
#pragma unroll
          for (int j = 0; j < kS; ++j) {
            const uint64_t e = legal.e[j];
            if (__builtin_amdgcn_inverse_ballot_w64(e)) {
              const int rank = before + static_cast<int>(__builtin_amdgcn_mbcnt_hi(
                                            static_cast<uint32_t>(e >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(e), 0u)));
              META[first + rank] = mw_make<kWide>(lane + 64 * j, cur, 0);
              FIRST[first + rank] = 0;
              COUNT[first + rank] = 0;
              TOTAL[first + rank] = 0.0;
              if constexpr (kGc) PARENT[first + rank] = node;
            }
            before += __builtin_popcountll(e);
          }
        } else {
          Mask lm;
          if constexpr (kHexFill) {
#pragma unroll
            for (int j = 0; j < kS; ++j) {
              lm.w[2 * j] = static_cast<uint32_t>(legal.e[j]);
              lm.w[2 * j + 1] = static_cast<uint32_t>(legal.e[j] >> 32);
            }
          } else {
            lm = legal;
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int a = lane + 64 * j;
            if (lm.test(a)) {
              int rank = 0;
#pragma unroll
              for (int w = 0; w < kMaskWords; ++w) {
                const int lo = 32 * w;
                if (a >= lo + 32) rank += __builtin_popcount(lm.w[w]);
                else if (a > lo) rank += __builtin_popcount(lm.w[w] & ((1u << (a - lo)) - 1u));
              }
              META[first + rank] = make_meta(a, cur, 0);
              FIRST[first + rank] = 0;
              COUNT[first + rank] = 0;
              TOTAL[first + rank] = 0.0;
              if constexpr (kGc) PARENT[first + rank] = node;
            }
          }
        }
        meta = mw_make<kWide>(static_cast<int>(mw_action<kWide>(meta)), m_player(meta), c) | (meta & kMetaOutcomeBits);
        if (lane == 0) {
          META[node] = meta;
          FIRST[node] = first;
        }
        if (node == 0) {
          root_meta = meta;
          root_first = first;
        }
        wave_fence();
      }
      PT_MARK(1);
      const int c = mw_nchild<kWide>(meta);
      int chosen_k, action;
      uint32_t n_meta, n_cnt, n_first;
      double n_tot;
      bool at_chance = false;
      if constexpr (!kHexFill) at_chance = cur == kChancePlayer;
      if (at_chance) {  // mcts.cc:311-322; children are in outcome order
        action = 0;
        chosen_k = 0;
        if constexpr (!kHexFill) {
          action = sample_action_chance<G>(p, s, legal, trng);
          int below = 0;
#pragma unroll
          for (int w = 0; w < kMaskWords; ++w) {
            const int lo = 32 * w;
            if (action >= lo + 32) below += __builtin_popcount(legal.w[w]);
            else if (action > lo) below += __builtin_popcount(legal.w[w] & ((1u << (action - lo)) - 1u));
          }
          chosen_k = below;
        }
        n_meta = uniform(META[first + chosen_k]);
        n_cnt = uniform(COUNT[first + chosen_k]);
        n_first = uniform(FIRST[first + chosen_k]);
        n_tot = uniform_f64(TOTAL[first + chosen_k]);
      } else {  // arg-max of UCTValue (mcts.cc:90-101), ties to the smallest order key
        Chosen ch;
        if constexpr (kWide) {  // the instantiation by the child count: a late position of a big board scans one slot
          if (c > 256 && kS > 4) ch = select_child<kS, kBoard, true>(META, COUNT, FIRST, TOTAL, first, c, cnt, cfg, log_table, obase, ph);
          else if (c > 192 && kS > 3) ch = select_child<4, kBoard, true>(META, COUNT, FIRST, TOTAL, first, c, cnt, cfg, log_table, obase, ph);
          else if (c > 128) ch = select_child<3, kBoard, true>(META, COUNT, FIRST, TOTAL, first, c, cnt, cfg, log_table, obase, ph);
          else if (c > 64) ch = select_child<2, kBoard, true>(META, COUNT, FIRST, TOTAL, first, c, cnt, cfg, log_table, obase, ph);
          else ch = select_child<1, kBoard, true>(META, COUNT, FIRST, TOTAL, first, c, cnt, cfg, log_table, obase, ph);
        } else {
          if (c > 64) ch = select_child<2, kBoard>(META, COUNT, FIRST, TOTAL, first, c, cnt, cfg, log_table, obase, ph);
          else ch = select_child<1, kBoard>(META, COUNT, FIRST, TOTAL, first, c, cnt, cfg, log_table, obase, ph);
        }
        chosen_k = ch.k;
        n_meta = ch.meta;
        n_cnt = ch.cnt;
        n_first = ch.first;
        n_tot = ch.tot;
        action = static_cast<int>(mw_action<kWide>(n_meta));
      }
      PT_MARK(2);
      node = first + static_cast<uint32_t>(chosen_k);
      ph = static_cast<uint32_t>(path_hash_child(ph, action));
      ++depth;
      // (kept ahead of apply on purpose: were a lane-dependent block the last thing in the loop body, its join
      // would be the loop latch, and the compiler would then treat every loop-carried value as lane-varying)
      vp.set(depth, node | ((n_meta >> 8 & 15u) << 28), n_cnt, n_tot);
      w_apply<G>(p, hl, s, action);
      PT_MARK(3);
      meta = n_meta;
      cnt = n_cnt;
      first = n_first;
    }
    if constexpr (kHexFill) {
      // IsTerminal() only where the search needs it (see HexW): a header that says so, or a first visit.
      if (m_terminal(meta)) {
        s.meta = (s.meta & 1u) | ((m_code(meta) == 2 ? 1u : 2u) << 1);
        term = true;
      } else if (cnt == 0) {
        if (depth == 0) {
          term = hexw_terminal(s);
        } else if (hexw_last_stone_wins(hl, s, static_cast<int>(mw_action<kWide>(meta)))) {
          s.meta |= ((s.meta & 1u) ? 1u : 2u) << 1;  // the player who moved last won
          term = true;
        }
      }
    }
    PT_MARK(0);
    // ---- evaluate (mcts.cc:372-381) ----
    double returns[kMaxPlayers];
    bool solved = false;
    if (term) {
      w_returns<G>(p, s, returns);
      meta |= (1u << 20) | (1u << 23);
      if (kBoard) meta = (meta & ~(3u << 21)) | (static_cast<uint32_t>(static_cast<int>(returns[0]) + 1) << 21);
      meta = uniform(meta);  // `returns` is a per-lane array to the compiler; the header is wave-uniform
      if (lane == 0) META[node] = meta;
      if (node == 0) root_meta = meta;
      solved = cfg.solve != 0;
    } else if constexpr (kHexFill) {
      double r0 = 0.0;
      for (int ro = 0; ro < cfg.n_rollouts; ++ro) {
        const uint64_t fb = fill_base_of(fill_word | vz, static_cast<uint64_t>(sim) * cfg.n_rollouts + ro);
        r0 += hex_fill_winner(s, fb, hl PT_PASS) == 0 ? 1.0 : -1.0;
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
          playout_returns<G>(p, s, rng, rr);
        }
        // Returns() of these games are multiples of 0.5 with small magnitude: sums are exact in
        // any order, so the butterfly equals the reference's sequential accumulation.
        for (int q = 0; q < num_players; ++q) returns[q] += wave_sum(rr[q]);
      }
      for (int q = 0; q < num_players; ++q) returns[q] /= cfg.n_rollouts;
    }
    PT_MARK(5);
    // ---- backup (mcts.cc:383-395): lane d owns the d-th node of the visit path ----
    {
      const uint32_t e = vp.node;
      int pl = static_cast<int>(e >> 28) - 1;
      if constexpr (!kBoard) {
        // chance-player entries (poker trees) take the player of the nearest entry above them that is not the
        // chance player (player 0 when there is none); all lanes walk up together, one entry per step
        for (int hop = 1; __ballot(pl == kChancePlayer && lane <= depth) != 0ull; ++hop) {
          const int up = lane - hop;
          const uint32_t ue = __shfl(e, up < 0 ? 0 : up);
          if (pl == kChancePlayer) pl = up < 0 ? 0 : static_cast<int>(ue >> 28) - 1;
        }
      }
      if (lane < kPathRegs && lane <= depth) {
        const uint32_t v = e & 0x0FFFFFFFu;
        double rv = returns[0];
        if constexpr (kBoard) {  // two players, no chance nodes
          rv = pl == 1 ? returns[1] : rv;
        } else {
          for (int q = 1; q < num_players; ++q) rv = (pl == q) ? returns[q] : rv;
        }
        const double nt = vp.tot + rv;
        const uint32_t nc = vp.cnt + 1;
        store_at(TOTAL, v * 8u, nt);
        store_at(COUNT, v * 4u, nc);
        vp.tot = nt;  // (only lane 0's — the root's — is ever read again before it is set anew)
        vp.cnt = nc;
      }
    }
    if constexpr (kBoard) {
      for (int d = kPathRegs + lane; d <= depth; d += 64) {  // entries in LDS: two players, no chance nodes
        const uint32_t e = vp.l_node[d - kPathRegs];
        const uint32_t v = e & 0x0FFFFFFFu;
        const int pl = static_cast<int>(e >> 28) - 1;
        const double rv = pl == 1 ? returns[1] : returns[0];
        TOTAL[v] = vp.l_tot[d - kPathRegs] + rv;
        COUNT[v] = vp.l_cnt[d - kPathRegs] + 1;
      }
    }
    wave_fence();
    // ---- MCTS-Solver (mcts.cc:398-434), leaf to root ----
    if (kBoard && solved) {
      for (int d = depth; d >= 0 && solved; --d) {
        const uint32_t v = vp.node_at(d) & 0x0FFFFFFFu;
        const uint32_t meta = uniform(META[v]);
        const int c = mw_nchild<kWide>(meta);
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
        // every lane holds the same `best` after the butterfly; say so, or the compiler treats the
        // branch — and with it the root header and the whole position — as lane-varying
        const bool mark = uniform(static_cast<int>(best.v > -INFINITY && (all_solved || best.v == max_utility))) != 0;
        if (mark) {
          const uint32_t solved_meta = (meta & ~(3u << 21)) | (1u << 20) | (static_cast<uint32_t>(uniform(best.k)) << 21);
          if (lane == 0) META[v] = solved_meta;
          if (v == 0) root_meta = solved_meta;
        } else {
          solved = false;
        }
      }
      wave_fence();
    }
    PT_MARK(6);
    ++sims_done;
    if (m_has_outcome(root_meta) || mw_nchild<kWide>(root_meta) == 1) break;  // mcts.cc:437-440 (a terminal root has an outcome too)
    // ---- GarbageCollect (mcts.cc:441-482): when nodes_ >= max_nodes_, every node with explore_count <
    // gc_limit_ loses its children.  Visit counts never grow from parent to child, so a node survives exactly
    // when its parent's count reaches the limit: 64 nodes per step, a ballot prefix gives the survivors their
    // new (order-preserving) index, a second sweep moves them down and rewrites the links.
    if constexpr (kGc) {
      if (used >= static_cast<uint32_t>(pool.gc_nodes)) {
        const uint32_t limit = static_cast<uint32_t>(gc_limit);
        uint32_t w = 0;
        for (uint32_t i0 = 0; i0 < used; i0 += 64) {
          const uint32_t i = i0 + lane;
          bool alive = false;
          if (i < used) alive = i == 0 || COUNT[PARENT[i]] >= limit;
          const uint64_t b = __ballot(alive);
          const uint32_t idx = w + static_cast<uint32_t>(__builtin_popcountll(b & ((1ull << lane) - 1ull)));
          if (i < used) REMAP[i] = alive ? idx : kNoNode;
          w += static_cast<uint32_t>(__builtin_popcountll(b));
        }
        wave_fence();
        for (uint32_t i0 = 0; i0 < used; i0 += 64) {
          const uint32_t i = i0 + lane;
          uint32_t to = kNoNode, nm = 0, nf = 0, nc = 0, np = kNoNode;
          double nt = 0.0;
          if (i < used) to = REMAP[i];
          if (to != kNoNode) {
            nm = META[i]; nf = FIRST[i]; nc = COUNT[i]; nt = TOTAL[i];
            if (i != 0) np = REMAP[PARENT[i]];
            if (mw_nchild<kWide>(nm) > 0) {
              if (nc < limit) { nm = mw_clear_children<kWide>(nm); nf = 0; }  // children.clear(); the outcome stays
              else nf = REMAP[nf];
            }
          }
          wave_fence();  // every load of this step before its stores (targets are <= the sources of this step)
          if (to != kNoNode) {
            META[to] = nm; FIRST[to] = nf; COUNT[to] = nc; TOTAL[to] = nt; PARENT[to] = np;
          }
          wave_fence();
        }
        used = uniform(w);
        root_meta = uniform(META[0]);
        root_first = uniform(FIRST[0]);
        gc_limit = next_gc_limit(gc_limit, used, pool.gc_nodes);
      }
    }
  }

  PT_FLUSH;
  // ---- results: BestChild (mcts.cc:114-143) + per-action statistics ----
  const uint32_t rm = root_meta;
  const int c = mw_nchild<kWide>(rm);
  const uint32_t first = root_first;
  for (int a = lane; a < num_actions; a += 64) {
    if (out.child_visits) out.child_visits[r * num_actions + a] = 0;
    if (out.child_reward) out.child_reward[r * num_actions + a] = 0.0;
    if (out.child_outcome) out.child_outcome[r * num_actions + a] = 3;
  }
  wave_fence();
  Final best{-INFINITY, 0u, 0.0, ~0ull, -1};
  const uint64_t root_ph = path_hash_root();
  for (int k = lane; k < c; k += 64) {
    const uint32_t cm = META[first + k];
    const uint32_t cc = COUNT[first + k];
    const double ct = TOTAL[first + k];
    const int a = static_cast<int>(mw_action<kWide>(cm));
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

// One wavefront per root, statically: wave w of workgroup b searches root 4 b + w.
template <class G, bool kBoard, bool kHexFill, bool kGc>
__global__ void __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(wave_wpe<G, kHexFill>(), 8)))
k_mcts_wave(typename G::Params p, const typename G::word_t* base, int64_t n, int num_players, int num_actions,
            osg_mcts_cfg cfg, double max_utility, const double* __restrict__ log_table, Pool pool, MctsOut out) {
  __shared__ uint32_t s_path[kWavesPerBlock][kMaxPath - kPathRegs];
  __shared__ uint32_t s_pcnt[kWavesPerBlock][kMaxPath - kPathRegs];
  __shared__ double s_ptot[kWavesPerBlock][kMaxPath - kPathRegs];
  const int wave_in_block = static_cast<int>(threadIdx.x >> 6);
  const int64_t r = uniform(static_cast<int>(blockIdx.x * kWavesPerBlock + wave_in_block));
  if (r >= n) return;
  wave_search<G, kBoard, kHexFill, kGc>(p, base, n, num_players, num_actions, cfg, max_utility, log_table, pool, out, r,
                                        s_path[wave_in_block], s_pcnt[wave_in_block], s_ptot[wave_in_block]);
}

// The same searches as a work queue: a persistent grid (a few wavefronts per SIMD) whose wavefronts take the next
// ticket from a counter until the roots are used up.  Searches differ in length (a root with 70 empty cells costs
// several times one with 45), so with one wavefront per root the launch ends with a few SIMDs still holding their
// longest searches; with tickets handed out in order of decreasing expected cost (queue.order: roots sorted by their
// number of legal actions, most first) the long searches start first and the short ones fill the gaps.  Results are
// written under the root's own index: the outputs do not depend on the order or on which wavefront ran what.
template <class G, bool kBoard, bool kHexFill, bool kGc>
__global__ void __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(wave_wpe<G, kHexFill>(), 8)))
k_mcts_wave_queue(typename G::Params p, const typename G::word_t* base, int64_t n, int num_players, int num_actions,
                  osg_mcts_cfg cfg, double max_utility, const double* __restrict__ log_table, Pool pool, MctsOut out,
                  WaveQueue queue) {
  __shared__ uint32_t s_path[kWavesPerBlock][kMaxPath - kPathRegs];
  __shared__ uint32_t s_pcnt[kWavesPerBlock][kMaxPath - kPathRegs];
  __shared__ double s_ptot[kWavesPerBlock][kMaxPath - kPathRegs];
  const int wave_in_block = static_cast<int>(threadIdx.x >> 6);
  for (;;) {
    int ticket = 0;
    if (lane_id() == 0) ticket = atomicAdd(queue.ticket, 1);
    ticket = uniform(ticket);
    if (ticket >= n) return;
    const int64_t r = queue.order ? uniform(queue.order[ticket]) : ticket;
    wave_search<G, kBoard, kHexFill, kGc>(p, base, n, num_players, num_actions, cfg, max_utility, log_table, pool, out,
                                          r, s_path[wave_in_block], s_pcnt[wave_in_block], s_ptot[wave_in_block]);
    wave_fence();
  }
}

// Cost key of a root for the queue's order: its number of legal actions (hex: the empty cells — the playout length
// and the width of the tree).  Counting sort, most expensive first: histogram, bucket offsets, scatter.
template <class G>
__global__ void __launch_bounds__(256) k_queue_histogram(typename G::Params p, const typename G::word_t* base, int64_t n,
                                                         int32_t* hist) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const typename G::State s = G::load(p, base, n, i);
  const int key = G::terminal(p, s) ? 0 : G::legal(p, s).count();
  atomicAdd(&hist[key < kQueueBuckets ? key : kQueueBuckets - 1], 1);
}
__global__ void __launch_bounds__(64) k_queue_offsets(int32_t* hist) {  // hist -> first position of every bucket
  if (threadIdx.x != 0) return;
  int32_t at = 0;
  for (int key = kQueueBuckets - 1; key >= 0; --key) {
    const int32_t c = hist[key];
    hist[key] = at;
    at += c;
  }
}
template <class G>
__global__ void __launch_bounds__(256) k_queue_scatter(typename G::Params p, const typename G::word_t* base, int64_t n,
                                                       int32_t* offsets, int32_t* order) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const typename G::State s = G::load(p, base, n, i);
  const int key = G::terminal(p, s) ? 0 : G::legal(p, s).count();
  order[atomicAdd(&offsets[key < kQueueBuckets ? key : kQueueBuckets - 1], 1)] = static_cast<int32_t>(i);
}

// Schedule of the wave-per-root search.  Measured on hex(9), 1024 simulations per root (MI355X,
// profiles/r03_mcts_schedule.log): with 2^16 roots — nine rounds of the wave slots — one wavefront per root is the
// fastest (1.106e9 simulations/s; the queue in index order 1.098e9, most-expensive-first 1.078e9: all the widest
// trees in flight at once); with the 2^13 roots one of 8 GPUs gets (8 192 roots, 7 168 wave slots: the launch is as
// long as its longest searches) most-expensive-first gains 6 % (8.80e8 -> 9.35e8; index order 8.84e8; 6 wavefronts
// per SIMD the same, 5 / 4 less).  Hence: the queue, most expensive roots first, for batches of up to four rounds
// of the wave slots, one wavefront per root beyond.  OSG_MCTS_SCHEDULE (read at every launch; a tuning knob, results
// do not depend on it) overrides: "static" | "queue" (index order) | "lpt", optionally ":<wavefronts per SIMD>".
struct Schedule { int mode; int waves_per_simd; bool forced; };
inline Schedule schedule_from_env(int compiled_waves_per_simd) {
  Schedule sc{2, compiled_waves_per_simd, false};
  const char* e = std::getenv("OSG_MCTS_SCHEDULE");
  if (!e || !*e) return sc;
  sc.forced = true;
  const std::string v(e);
  const size_t colon = v.find(':');
  const std::string mode = v.substr(0, colon);
  if (mode == "static") sc.mode = 0; else if (mode == "queue") sc.mode = 1; else if (mode == "lpt") sc.mode = 2;
  if (colon != std::string::npos) {
    const int w = std::atoi(v.c_str() + colon + 1);
    if (w >= 1 && w <= 16) sc.waves_per_simd = w;
  }
  return sc;
}

template <class G, bool kBoard, bool kHexFill>
int launch(const typename G::Params& P, const osg_batch* roots, const osg_mcts_cfg& cfg, const double* d_logs,
           const Pool& pool, const MctsOut& out) {
  const osg_game_desc& d = roots->spec.desc;
  osg_ctx* ctx = roots->ctx;
  hipStream_t st = ctx->stream;
  const int64_t n = roots->n;
  const auto* words = static_cast<const typename G::word_t*>(roots->d_words);
  const bool gc = pool.gc_nodes > 1 && pool.remap;
  const Schedule sc = schedule_from_env(wave_wpe<G, kHexFill>());
  if (ctx->num_cus == 0) {
    hipDeviceProp_t prop;
    OSG_HIP(hipGetDeviceProperties(&prop, ctx->device));
    ctx->num_cus = prop.multiProcessorCount;
  }
  // a queue pays only when there are more roots than wave slots to hand them to — and, measured, only while the batch is
  // little more than one round of slots (hex(9), 7 168 slots: 8 192 roots 8.84e8 static -> 9.37e8 longest-first, but
  // 16 384 roots 1.00e9 -> 9.6e8, 32 768 roots 1.06e9 -> 1.03e9: with two or more rounds the static order's own
  // mixing balances the SIMDs and the queue's atomics and sort are pure cost)
  const int64_t slots = static_cast<int64_t>(ctx->num_cus) * 4 * sc.waves_per_simd;
  // (the boards above 128 cells always launch statically: the queue's loop around the search costs the fat instantiations
  // 50-180 vector registers in scratch, and its gain was 6 % on one-round batches of hex(9))
  constexpr bool kQueue = wave_sets<G>() <= 2;
  if (!kQueue || sc.mode == 0 || n <= slots || n >= (int64_t{1} << 31) || (!sc.forced && n > slots + slots / 2)) {
    const unsigned grid = static_cast<unsigned>((n + kWavesPerBlock - 1) / kWavesPerBlock);
    if (gc)
      k_mcts_wave<G, kBoard, kHexFill, true><<<dim3(grid), dim3(64 * kWavesPerBlock), 0, st>>>(
          P, words, n, d.num_players, d.num_distinct_actions, cfg, d.max_utility, d_logs, pool, out);
    else
      k_mcts_wave<G, kBoard, kHexFill, false><<<dim3(grid), dim3(64 * kWavesPerBlock), 0, st>>>(
          P, words, n, d.num_players, d.num_distinct_actions, cfg, d.max_utility, d_logs, pool, out);
    return OSG_OK;
  }
  if constexpr (kQueue) {
  if (ctx->mcts_queue_roots < n) {
    if (ctx->d_mcts_queue) OSG_HIP(hipFree(ctx->d_mcts_queue));
    ctx->d_mcts_queue = nullptr;
    ctx->mcts_queue_roots = 0;
    OSG_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_mcts_queue), sizeof(int32_t) * (kQueueHeader + static_cast<size_t>(n))));
    ctx->mcts_queue_roots = n;
  }
  int32_t* ticket = ctx->d_mcts_queue;
  int32_t* hist = ctx->d_mcts_queue + 1;
  int32_t* order = ctx->d_mcts_queue + kQueueHeader;
  OSG_HIP(hipMemsetAsync(ctx->d_mcts_queue, 0, sizeof(int32_t) * kQueueHeader, st));
  WaveQueue queue{ticket, nullptr};
  if (sc.mode == 2) {
    const unsigned g = static_cast<unsigned>((n + 255) / 256);
    k_queue_histogram<G><<<dim3(g), dim3(256), 0, st>>>(P, words, n, hist);
    k_queue_offsets<<<dim3(1), dim3(64), 0, st>>>(hist);
    k_queue_scatter<G><<<dim3(g), dim3(256), 0, st>>>(P, words, n, hist, order);
    queue.order = order;
  }
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>((n + kWavesPerBlock - 1) / kWavesPerBlock,
                                                                slots / kWavesPerBlock));
  if (gc)
    k_mcts_wave_queue<G, kBoard, kHexFill, true><<<dim3(grid), dim3(64 * kWavesPerBlock), 0, st>>>(
        P, words, n, d.num_players, d.num_distinct_actions, cfg, d.max_utility, d_logs, pool, out, queue);
  else
    k_mcts_wave_queue<G, kBoard, kHexFill, false><<<dim3(grid), dim3(64 * kWavesPerBlock), 0, st>>>(
        P, words, n, d.num_players, d.num_distinct_actions, cfg, d.max_utility, d_logs, pool, out, queue);
  }
  return OSG_OK;
}

}  // namespace

namespace osg {

int launch_mcts_wave(const osg_batch* roots, const osg_mcts_cfg& cfg, const double* d_logs, const Pool& pool,
                     const MctsOut& out) {
  const GameSpec& spec = roots->spec;
  int rc = OSG_OK;
  switch (spec.desc.game_kind) {
    case kTtt: rc = launch<Ttt, true, false>(spec.ttt, roots, cfg, d_logs, pool, out); break;
    case kC4:
      if (spec.c4_std) rc = launch<C4Std, true, false>(spec.c4, roots, cfg, d_logs, pool, out);
      else rc = launch<C4, true, false>(spec.c4, roots, cfg, d_logs, pool, out);
      break;
    case kKuhn: rc = launch<Kuhn, false, false>(spec.kuhn, roots, cfg, d_logs, pool, out); break;
    case kLeduc: rc = launch<Leduc, false, false>(spec.leduc, roots, cfg, d_logs, pool, out); break;
    case kHex: {
      // The random-fill playout needs "legal moves == empty cells": not with the swap rule.  The search's own
      // position (HexW) needs two distinct edges per colour.
#define OSG_HEX_CASE(NW, member)                                                             \
  if (spec.member.swap || spec.member.rows < 2 || spec.member.cols < 2)                      \
    rc = launch<HexT<NW>, true, false>(spec.member, roots, cfg, d_logs, pool, out);          \
  else rc = launch<HexT<NW>, true, true>(spec.member, roots, cfg, d_logs, pool, out)
      // The boards above 128 cells (six / eight / twelve plane words) are searched in the fill form only (the caller,
      // osg_mcts_search, has checked: no swap rule, two rows and columns at least, at most 64 columns).
#define OSG_HEX_WIDE_CASE(NW, member)                                                        \
  if (spec.member.swap || spec.member.rows < 2 || spec.member.cols < 2 || spec.member.cols > 64) \
    return set_error(OSG_ERR_UNSUPPORTED, "wave-per-root search of a board above 128 cells: hex without the swap rule, 2 ... 64 columns"); \
  rc = launch<HexT<NW>, true, true>(spec.member, roots, cfg, d_logs, pool, out)
      switch (spec.hex_nw) {   // (a folded record — HexT::folded — differs in the root load only)
        case 1: OSG_HEX_CASE(1, hex1); break;
        case 2: OSG_HEX_CASE(2, hex2); break;
        case 3: OSG_HEX_CASE(3, hex3); break;
        case 4: OSG_HEX_CASE(4, hex4); break;
        case 6: OSG_HEX_WIDE_CASE(6, hex6); break;
        case 8: OSG_HEX_WIDE_CASE(8, hex8); break;
        default: OSG_HEX_WIDE_CASE(12, hex12); break;
      }
#undef OSG_HEX_WIDE_CASE
#undef OSG_HEX_CASE
      break;
    }
    default: return set_error(OSG_ERR_INVALID, "bad game kind");
  }
  if (rc) return rc;
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

}  // namespace osg
