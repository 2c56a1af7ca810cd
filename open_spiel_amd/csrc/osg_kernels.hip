// Batched State kernels for the five games + the C-ABI around them.
//
// Data layout in HBM: struct-of-arrays.  A batch of n states of a game with W
// words per state is ONE allocation of W planes of n elements (u32 or u64);
// lane i of a wavefront touches element i of every plane, so each plane access
// is a fully coalesced 256-512 B transaction per wave.  All kernels are
// HBM-bound byte/integer work: no MFMA, no LDS needed for the pure step path
// (the state lives in VGPRs between load and store).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <type_traits>

#include "osg_internal.h"
#include "osg_c4_step.h"
#include "osg_ttt_step.h"

using namespace osg;

namespace {

constexpr int kBlock = 256;  // 4 wavefronts
inline int grid_for(int64_t n) { return static_cast<int>((n + kBlock - 1) / kBlock); }

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
template <class G>
__global__ void __launch_bounds__(kBlock) k_init(typename G::Params p, typename G::word_t* base, int64_t n) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  G::store(p, base, n, i, G::initial(p));
}

// 16 bytes per lane, 4 KiB per workgroup: the plain-copy ceiling (osg_copy_bytes).
__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
#ifdef OSG_COPY_PLAIN
  if (i < n16) dst[i] = src[i];
#else
  // non-temporal stores, like the kernels it is the ceiling of (a copy with plain stores is slower: §9)
  if (i < n16) {
    const uint4 v = src[i];
    __builtin_nontemporal_store(v.x, &dst[i].x);
    __builtin_nontemporal_store(v.y, &dst[i].y);
    __builtin_nontemporal_store(v.z, &dst[i].z);
    __builtin_nontemporal_store(v.w, &dst[i].w);
  }
#endif
}

template <class G>
__global__ void __launch_bounds__(kBlock)
k_gather(typename G::Params p, typename G::word_t* dst, int64_t nd, const typename G::word_t* src, int64_t ns,
         const int64_t* index, unsigned long long* illegal) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= nd) return;
  int64_t j = index[i];
  if (j < 0 || j >= ns) {  // device-resident indices cannot be checked on the host: initial state + counted
    G::store(p, dst, nd, i, G::initial(p));
    atomicAdd(illegal, 1ull);
    return;
  }
  for (int k = 0; k < p.words; ++k) dst[k * nd + i] = src[k * ns + j];
}

template <class G>
__global__ void __launch_bounds__(kBlock)
k_legal_mask(typename G::Params p, const typename G::word_t* base, int64_t n, uint32_t* mask, int mask_words) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  auto m = G::legal(p, G::load(p, base, n, i));
#pragma unroll
  for (int w = 0; w < G::kMaskW; ++w)  // static indices only: a runtime index would spill the mask to scratch
    if (w < mask_words) mask[i * mask_words + w] = m.w[w];
}

template <class G>
__global__ void __launch_bounds__(kBlock)
k_apply(typename G::Params p, typename G::word_t* base, int64_t n, const int32_t* actions,
        unsigned long long* illegal) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  int a = actions[i];
  if (a == OSG_INVALID_ACTION) return;
  typename G::State s = G::load(p, base, n, i);
  auto m = G::legal(p, s);
  if (a < 0 || a >= 32 * G::kMaskW || !m.test(a)) {
    atomicAdd(illegal, 1ull);  // the compiler folds this into one add per wave
    return;
  }
  G::apply(p, s, a);
  G::store(p, base, n, i, s);
}

template <class G>
__global__ void __launch_bounds__(kBlock)
k_status(typename G::Params p, const typename G::word_t* base, int64_t n, int num_players, int8_t* cur,
         uint8_t* term, double* rets) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  typename G::State s = G::load(p, base, n, i);
  if (cur) cur[i] = static_cast<int8_t>(G::current_player(p, s));
  if (term) term[i] = G::terminal(p, s) ? 1 : 0;
  if (rets) {
    double r[kMaxPlayers];
    G::returns(p, s, r);
    for (int q = 0; q < num_players; ++q) rets[i * num_players + q] = r[q];
  }
}

template <class G>
__global__ void __launch_bounds__(kBlock)
k_chance_probs(typename G::Params p, const typename G::word_t* base, int64_t n, int max_chance, double* probs) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  typename G::State s = G::load(p, base, n, i);
  bool chance = G::current_player(p, s) == kChancePlayer;
  auto m = G::legal(p, s);
  for (int o = 0; o < max_chance; ++o)
    probs[i * max_chance + o] = (chance && m.test(o)) ? G::chance_prob(p, s, o) : 0.0;
}

// The fused headline kernel (connect_four = 35 B per state:
// 16 B state in + 16 B out + 1 B action + 1 B mask + 1 B status).
OSG_D uint8_t encode_status(bool terminal, bool illegal, int cur, int outcome) {
  uint8_t v = illegal ? 0x40 : 0;
  if (terminal) return v | 0x80 | static_cast<uint8_t>(outcome & 7);
  return v | static_cast<uint8_t>((cur + 1) & 15);
}
// (connect_four geometries without a stored result take C4T::fused_step: one pass instead of the generic sequence)
template <class G> struct has_fused_step : std::false_type {};
template <int R, int C, int K, class BB> struct has_fused_step<C4T<R, C, K, BB>> : std::integral_constant<bool, !C4T<R, C, K, BB>::kStored> {};
template <class G, typename MaskT>
__global__ void __launch_bounds__(kBlock)
k_step(typename G::Params p, const typename G::word_t* src, typename G::word_t* dst, int64_t n,
       const uint8_t* actions, MaskT* mask_out, int mask_elems, uint8_t* status) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  typename G::State s = G::load(p, src, n, i);
  int a = actions[i];
  if constexpr (has_fused_step<G>::value) {
    bool illegal_f, term_f;
    int outcome_f;
    const uint32_t open = G::fused_step(p, s, a, illegal_f, term_f, outcome_f);
    G::store(p, dst, n, i, s);
    if (sizeof(MaskT) < 4) {
      mask_out[i] = static_cast<MaskT>(open);
    } else {
#pragma unroll
      for (int w = 0; w < G::kMaskW; ++w)
        if (w < mask_elems) mask_out[i * mask_elems + w] = static_cast<MaskT>(w == 0 ? open : 0u);
    }
    status[i] = encode_status(term_f, illegal_f, term_f ? 0 : (G::plies(s) & 1), term_f ? outcome_f : 0);
    return;
  }
  bool illegal = false;
  if (a != 0xFF) {
    auto before = G::legal(p, s);
    if (a < 32 * G::kMaskW && before.test(a)) G::apply(p, s, a); else illegal = true;
  }
  G::store(p, dst, n, i, s);  // (plain stores: non-temporal ones measured mixed here — hex 22.8 -> 24.6 us at 2^20, 118 -> 104 at 2^22)
  bool term = G::terminal(p, s);
  auto after = G::legal(p, s);
  if (sizeof(MaskT) < 4) {
    mask_out[i] = static_cast<MaskT>(after.w[0]);
  } else {
#pragma unroll
    for (int w = 0; w < G::kMaskW; ++w)  // static indices only (see k_legal_mask)
      if (w < mask_elems) mask_out[i * mask_elems + w] = static_cast<MaskT>(after.w[w]);
  }
  status[i] = encode_status(term, illegal, term ? 0 : G::current_player(p, s), term ? G::outcome_code(p, s) : 0);
}

// The fused step for the games whose state is one or two words (tic_tac_toe 4 B, kuhn_poker 8 B, leduc_poker
// 2 x 8 B): V consecutive states per thread so that every plane access is ONE 16-byte vector load / store per lane
// (1 KiB per wave-instruction) instead of V narrow ones, and the V actions / masks / statuses move as one word.
// The per-state code is the generic one (G::legal / apply / terminal) on a register-resident mini-batch.
template <class G, typename MaskT, int V, int W>  // W = words per state
__global__ void __launch_bounds__(kBlock)
k_step_vec(typename G::Params p, const typename G::word_t* src, typename G::word_t* dst, int64_t n,  // src may BE dst (in-place step)
           const uint8_t* __restrict__ actions, MaskT* __restrict__ mask_out, uint8_t* __restrict__ status) {
  using word_t = typename G::word_t;
  typedef word_t wvec __attribute__((ext_vector_type(V)));
  typedef uint8_t bvec __attribute__((ext_vector_type(V)));
  typedef MaskT mvec __attribute__((ext_vector_type(V)));
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * V;
  if (i >= n) return;
  word_t tmp[W * V];  // plane-major mini-batch: G::load(p, tmp, V, j) reads tmp[w * V + j]
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const wvec v = *reinterpret_cast<const wvec*>(src + w * n + i);
#pragma unroll
    for (int j = 0; j < V; ++j) tmp[w * V + j] = v[j];
  }
  const bvec av = *reinterpret_cast<const bvec*>(actions + i);
  bvec sv;
  mvec mv;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    if constexpr (std::is_same<G, Ttt>::value) {
      // the whole step as straight-line code on the packed word, both players' lines in one pass (osg_ttt_step.h)
      const uint32_t r = ttt_fused_step(tmp[j], av[j]);
      mv[j] = static_cast<MaskT>(r & 0xFFFFu);
      sv[j] = static_cast<uint8_t>(r >> 16);
      continue;
    }
    typename G::State s = G::load(p, tmp, V, j);
    const int a = av[j];
    bool illegal = false;
    if (a != 0xFF) {
      const auto before = G::legal(p, s);
      if (a < 32 * G::kMaskW && before.test(a)) G::apply(p, s, a); else illegal = true;
    }
    G::store(p, tmp, V, j, s);
    const bool term = G::terminal(p, s);
    const auto after = G::legal(p, s);
    mv[j] = static_cast<MaskT>(after.w[0]);
    sv[j] = encode_status(term, illegal, term ? 0 : G::current_player(p, s), term ? G::outcome_code(p, s) : 0);
  }
#pragma unroll
  for (int w = 0; w < W; ++w) {
    wvec v;
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = tmp[w * V + j];
    // non-temporal, as in k_step_c4std (2^24 states: kuhn 52.6 -> 49.4 us, leduc 103.1 -> 92.9 us, tic_tac_toe unchanged)
    __builtin_nontemporal_store(v, reinterpret_cast<wvec*>(dst + w * n + i));
  }
  __builtin_nontemporal_store(mv, reinterpret_cast<mvec*>(mask_out + i));
  __builtin_nontemporal_store(sv, reinterpret_cast<bvec*>(status + i));
}

// hex: the fused step with V consecutive states per thread.  A hex(9) state is 13 planes of 32-bit words: with one
// state per thread every plane access moves 4 bytes per lane (256 B per wave-instruction, 26 such streams per launch);
// with V = 2 it is 8 bytes per lane and plane, and the V legal masks of a thread (NW words each, rows of the [n, NW]
// output) are one contiguous span written as NW vectors.  kNt: the successor records, masks and status bytes leave
// through non-temporal stores (batches beyond the Infinity Cache).  The rules are the generic HexT<NW>::legal /
// apply on a register-resident mini-batch (the flood runs only for a placement that touches an edge-connected
// group or an edge, hex.cc:253-276).  Measured (MI355X, 118 B per step, profiles/r03_hex_step.log; V:nt):
//   2^20 states  1:0 22.2 us   2:0 20.1 us   2:1 23.4   4:0 23.2   4:1 29.0
//   2^22 states  1:0 90.2 us   2:0 97.0      2:1 80.8   4:0 99.5   4:1 96.3
//   2^24 states  1:0 334.7 us  2:0 335.3     2:1 330.1  4:0 337.7  4:1 350.7   (0.74-0.75 of 8 TB/s: DRAM)
// Four states per thread (94 vector registers, four divergent floods in a row) never pays; two do, with ordinary
// stores while the batch fits the Infinity Cache and non-temporal ones beyond.
// kMask = false (round 5; osg_step with d_mask == NULL): the successor's mask row is not written — on a hex board it
// is ~occupied of the successor record, which the caller holds anyway (SURVEY.md 8(d) prices the hex step without
// it: 109 B instead of 118 B moved for hex(9)).
template <int NW, int V, bool kNt, bool kMask = true, bool kFold = false>
__global__ void __launch_bounds__(kBlock)
k_step_hexvec(typename HexT<NW>::Params p, const uint32_t* src, uint32_t* dst, int64_t n,  // src may BE dst
              const uint8_t* __restrict__ actions, uint32_t* __restrict__ mask_out, uint8_t* __restrict__ status) {
  using G = HexT<NW>;
  constexpr int W = kFold ? 4 * NW : 4 * NW + 1;   // (folded: the meta word rides in the planes' spare bits; hex(9): 12)
  typedef uint32_t wvec __attribute__((ext_vector_type(V)));
  typedef uint8_t bvec __attribute__((ext_vector_type(V)));
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * V;
  if (i >= n) return;
  uint32_t tmp[W * V];  // plane-major mini-batch: G::load(p, tmp, V, j) reads tmp[w * V + j]
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const wvec v = *reinterpret_cast<const wvec*>(src + w * n + i);
#pragma unroll
    for (int j = 0; j < V; ++j) tmp[w * V + j] = v[j];
  }
  const bvec av = *reinterpret_cast<const bvec*>(actions + i);
  uint32_t mk[NW * V];  // the thread's V mask rows, in output order
  bvec sv;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    typename G::State s = G::template load_as<kFold>(tmp, V, j);
    const int a = av[j];
    bool illegal = false;
    if (a != 0xFF) {
      const auto before = G::legal(p, s);
      if (a < 32 * G::kMaskW && before.test(a)) G::apply(p, s, a); else illegal = true;
    }
    G::template store_as<kFold>(tmp, V, j, s);
    const bool term = G::terminal(p, s);
    if constexpr (kMask) {
      const auto after = G::legal(p, s);
#pragma unroll
      for (int w = 0; w < NW; ++w) mk[j * NW + w] = after.w[w];
    }
    sv[j] = encode_status(term, illegal, term ? 0 : G::current_player(p, s), term ? G::outcome_code(p, s) : 0);
  }
#pragma unroll
  for (int w = 0; w < W; ++w) {
    wvec v;
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = tmp[w * V + j];
    if constexpr (kNt) __builtin_nontemporal_store(v, reinterpret_cast<wvec*>(dst + w * n + i));
    else *reinterpret_cast<wvec*>(dst + w * n + i) = v;
  }
  if constexpr (kMask) {
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      wvec v;
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = mk[k * V + j];
      if constexpr (kNt) __builtin_nontemporal_store(v, reinterpret_cast<wvec*>(mask_out + i * NW) + k);
      else reinterpret_cast<wvec*>(mask_out + i * NW)[k] = v;
    }
  }
  if constexpr (kNt) __builtin_nontemporal_store(sv, reinterpret_cast<bvec*>(status + i));
  else *reinterpret_cast<bvec*>(status + i) = sv;
}

// connect_four, other geometries than 6 x 7 x 4: TWO consecutive states per thread so that every state access is
// one 16-byte vector load/store per lane per plane; actions / masks / statuses move as u16.
#ifndef OSG_C4STEP_BLOCK
#define OSG_C4STEP_BLOCK 128
#endif
constexpr int kC4StepBlock = OSG_C4STEP_BLOCK;
template <class G>
__global__ void __launch_bounds__(kC4StepBlock)
k_step_c4x2(typename G::Params p, const uint64_t* src, uint64_t* dst, int64_t n,  // src may BE dst (in-place step)
            const uint8_t* __restrict__ actions, uint8_t* __restrict__ mask_out, uint8_t* __restrict__ status) {
  const int64_t pair = static_cast<int64_t>(blockIdx.x) * kC4StepBlock + threadIdx.x;
  const int64_t i = pair * 2;
  if (i >= n) return;
  const ulonglong2 xs = *reinterpret_cast<const ulonglong2*>(src + i);
  const ulonglong2 os = *reinterpret_cast<const ulonglong2*>(src + n + i);
  const uint32_t a2 = *reinterpret_cast<const uint16_t*>(actions + i);
  uint64_t x[2] = {xs.x, xs.y}, o[2] = {os.x, os.y};
  uint32_t m2 = 0, s2 = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    typename G::State s = G::unpack(x[j], o[j]);
    const int a = (a2 >> (8 * j)) & 0xFF;
    bool term = G::terminal(p, s);
    bool illegal = false;
    if (a != 0xFF) {
      if (!term && G::column_has_room(p, s, a)) {
        G::apply(p, s, a);
        term = G::terminal(p, s);
      } else {
        illegal = true;
      }
    }
    const uint32_t open = G::open_columns(p, s);
    const int to_move = G::plies(s) & 1;
    x[j] = G::pack0(s);
    o[j] = s.o;
    m2 |= (term ? 0u : (open & 0xFFu)) << (8 * j);
    s2 |= static_cast<uint32_t>(encode_status(term, illegal, to_move, term ? G::outcome_code(p, s) : 0)) << (8 * j);
  }
  *reinterpret_cast<ulonglong2*>(dst + i) = make_ulonglong2(x[0], x[1]);
  *reinterpret_cast<ulonglong2*>(dst + n + i) = make_ulonglong2(o[0], o[1]);
  *reinterpret_cast<uint16_t*>(mask_out + i) = static_cast<uint16_t>(m2);
  *reinterpret_cast<uint16_t*>(status + i) = static_cast<uint16_t>(s2);
}

// THE HEADLINE KERNEL (with k_step_c4std2 below): the fused step of the standard connect_four board (6 x 7, four in a
// row), one state per thread, workgroups of 128.  The step itself is c4_fused_step (osg_c4_step.h: straight-line selects on the two
// packed planes, ONE line test — the mover's —, the successor's legal mask gathered by two 24-bit multiplies; the
// result of the game lives in plane 0's spare byte).  2^20 states are 16 384 wavefronts, TWO rounds of the chip's
// 8 192 wave slots: the second round's loads overlap the first round's stores, which measured faster than two
// states per thread with 16-byte accesses in one round (6.18-6.30 vs 6.48-6.67 us per launch at 2^20 states in the
// same runs, 97.8-99.7 vs 101.8-103.6 us at 2^24; workgroups of 64 / 256 / 1024: 6.73 / 6.17-6.35 / 6.16-6.19 us at
// 2^20 and 107.7 / 99.6-102.3 / 105.0-105.8 us at 2^24).  Any batch size, no alignment requirement on the side arrays.
__global__ void __launch_bounds__(kC4StepBlock)
k_step_c4std(const uint64_t* src, uint64_t* dst, int64_t n, const uint8_t* __restrict__ actions,  // src may BE dst
             uint8_t* __restrict__ mask_out, uint8_t* __restrict__ status) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kC4StepBlock + threadIdx.x;
  if (i >= n) return;
  uint64_t x = src[i], o = src[n + i];
  const uint32_t r = c4_fused_step(x, o, actions[i]);
  // Non-temporal stores: the successor records are not read again by this launch, and written around the L2 they
  // neither displace the inputs still to be read nor wait for a write-back at the end of the kernel (measured:
  // 6.1-6.3 -> 5.0-5.1 us per launch at 2^20 states, 97-99 -> 89-91 us at 2^24; non-temporal LOADS as well: 6.8 us).
  __builtin_nontemporal_store(x, dst + i);
  __builtin_nontemporal_store(o, dst + n + i);
  __builtin_nontemporal_store(static_cast<uint8_t>(r), mask_out + i);
  __builtin_nontemporal_store(static_cast<uint8_t>(r >> 8), status + i);
}

// The same step with TWO consecutive states per thread (16-byte plane accesses, u16 side arrays) for even batches
// with 2-byte aligned side arrays — the headline configuration.  With ordinary stores one state per thread was the
// faster layout (two rounds of wavefronts: the second round's loads overlap the first round's stores); with
// non-temporal stores the wider accesses win again (same runs, 2^20 states: 5.07-5.08 vs 5.14-5.21 us per launch;
// 2^24 states: 86.7-88.6 vs 90.6-91.7 us).
__global__ void __launch_bounds__(kC4StepBlock)
k_step_c4std2(const uint64_t* src, uint64_t* dst, int64_t n, const uint8_t* __restrict__ actions,  // src may BE dst
              uint8_t* __restrict__ mask_out, uint8_t* __restrict__ status) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kC4StepBlock + threadIdx.x) * 2;
  if (i >= n) return;
  typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
  typedef uint8_t u8x2 __attribute__((ext_vector_type(2)));
  const u64x2 xs = *reinterpret_cast<const u64x2*>(src + i), os = *reinterpret_cast<const u64x2*>(src + n + i);
  const u8x2 av = *reinterpret_cast<const u8x2*>(actions + i);
  uint64_t x0 = xs.x, x1 = xs.y, o0 = os.x, o1 = os.y;
  const uint32_t r0 = c4_fused_step(x0, o0, av.x), r1 = c4_fused_step(x1, o1, av.y);
  u64x2 xo, oo;
  xo.x = x0; xo.y = x1; oo.x = o0; oo.y = o1;
  u8x2 mo, so;
  mo.x = static_cast<uint8_t>(r0); mo.y = static_cast<uint8_t>(r1);
  so.x = static_cast<uint8_t>(r0 >> 8); so.y = static_cast<uint8_t>(r1 >> 8);
  __builtin_nontemporal_store(xo, reinterpret_cast<u64x2*>(dst + i));
  __builtin_nontemporal_store(oo, reinterpret_cast<u64x2*>(dst + n + i));
  __builtin_nontemporal_store(mo, reinterpret_cast<u8x2*>(mask_out + i));
  __builtin_nontemporal_store(so, reinterpret_cast<u8x2*>(status + i));
}

// Observation / information-state tensors: write-bound ([n, size] fp32, zero-filled
// then set like ContiguousAllocator / TensorView do, observer.h:174-185).  A row is cut
// into chunks of four floats; one lane produces one chunk — one state load, one cursor,
// four entries, one 16-byte store (the last chunk of a row may be shorter) — so lanes never
// straddle two states and the kernel has no divergent reloads.  Consecutive lanes write
// consecutive addresses (1 KiB per wave-instruction).
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));  // rows are only 4-byte aligned
// Tensor rows are written once and not read again by the kernel: non-temporal stores (see k_step_c4std) — where they
// measured faster (2^24 states, fraction of 8 TB/s, plain -> non-temporal): leduc [n, 16] 0.76 -> 0.89, [n, 30] 0.74 ->
// 0.83, kuhn [n, 7] 0.60 -> 0.67, [n, 11] 0.69 -> 0.76, hex(9) 0.67 -> 0.71; the tic_tac_toe rows keep plain stores
// (0.77 -> 0.70 with non-temporal ones), the connect_four planes take them from 2^22 states on (see the kernel).
OSG_D void store_row4(float4u* dst, const float4u& v) {  // 4-byte aligned rows
#ifdef OSG_OBS_PLAIN
  *dst = v;
#else
  __builtin_nontemporal_store(v, dst);
#endif
}
template <bool kNt = true>
OSG_D void store_row4(float4* dst, const float4& v) {  // 16-byte aligned spans
  if constexpr (!kNt) { *dst = v; return; }
#ifdef OSG_OBS_PLAIN
  *dst = v;
#else
  __builtin_nontemporal_store(v.x, &dst->x);
  __builtin_nontemporal_store(v.y, &dst->y);
  __builtin_nontemporal_store(v.z, &dst->z);
  __builtin_nontemporal_store(v.w, &dst->w);
#endif
}
template <class G, int F>  // F = floats per lane (a multiple of 4): 4 for short rows, 16 for long ones
__global__ void __launch_bounds__(kBlock)
k_observation(typename G::Params p, const typename G::word_t* base, int64_t n, int size, int seg_len, int chunks_per_seg,
              int player, int which, float* out) {
  // A row of `size` floats is a sequence of segments of `seg_len` floats (hex: one per tensor plane; other
  // games: the whole row); chunks never cross a segment, so a cursor never changes plane mid-chunk.
  // One 64-bit division per workgroup on wave-uniform values; lanes divide a small offset in 32 bits.
  // F == 16: a lane's four float4 pieces are 64 bytes apart from its neighbour's, so storing them directly
  // would make every store instruction hit 64 different cache lines with 16 bytes each.  Instead each
  // wavefront stages its 4 KiB through LDS and writes it back piece-major: instruction j stores pieces
  // 64 j ... 64 j + 63, i.e. whole consecutive chunks -> whole cache lines.
  __shared__ float4 s_tile[F >= 16 ? kBlock * (F / 4) : 1];
  __shared__ float* s_dst[F >= 16 ? kBlock : 1];
  __shared__ int s_count[F >= 16 ? kBlock : 1];
  const int chunks = (size / seg_len) * chunks_per_seg;  // per state
  const int64_t tb = static_cast<int64_t>(blockIdx.x) * kBlock;
  const int64_t ib = tb / chunks;
  const uint32_t local = static_cast<uint32_t>(tb - ib * chunks) + threadIdx.x;
  const uint32_t il = local / static_cast<uint32_t>(chunks);
  const int64_t i = ib + il;
  const bool live = i < n;
  int count = 0;
  float* dst = out;
  float4 q[F / 4];
#pragma unroll
  for (int g = 0; g < F / 4; ++g) q[g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (live) {
    const uint32_t c_in_state = local - il * static_cast<uint32_t>(chunks);
    const uint32_t seg = c_in_state / static_cast<uint32_t>(chunks_per_seg);
    const int off = static_cast<int>(c_in_state - seg * chunks_per_seg) * F;  // offset inside the segment
    const int idx = static_cast<int>(seg) * seg_len + off;
    const typename G::State s = G::load(p, base, n, i);
    int pl = player;
    if (pl < 0) {
      pl = G::current_player(p, s);
      if (pl < 0) pl = 0;
    }
    typename G::ObsCursor cur;
    cur.init(p, s, pl, which, idx);
    count = seg_len - off;  // >= 1; only the last chunk of a segment has fewer than F
    if (count > F) count = F;
    dst = out + i * size + idx;
#pragma unroll
    for (int g = 0; g < F / 4; ++g) {
      if (4 * g >= count) break;
      float v[4];  // indexed by unrolled constants only: stays in registers
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (4 * g + k < count) ? cur.next(p, s, pl, which) : 0.0f;
      q[g] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  if (F == 4) {
    if (!live) return;
    if (count >= 4) {
      float4u w = {q[0].x, q[0].y, q[0].z, q[0].w};
      store_row4(reinterpret_cast<float4u*>(dst), w);
    } else {
      dst[0] = q[0].x;
      if (count > 1) dst[1] = q[0].y;
      if (count > 2) dst[2] = q[0].z;
    }
    return;
  }
  // ---- F >= 16: piece-major write-back through LDS (per wavefront; no workgroup barrier needed, every
  //      wave only reads what it wrote itself) ----
  const int lane = threadIdx.x & 63, wave0 = threadIdx.x & ~63;
#pragma unroll
  for (int g = 0; g < F / 4; ++g) s_tile[(wave0 + lane) * (F / 4) + g] = q[g];
  s_dst[threadIdx.x] = dst;
  s_count[threadIdx.x] = live ? count : 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int j = 0; j < F / 4; ++j) {
    const int piece = j * 64 + lane;        // piece index inside the wave's tile
    const int src_lane = piece / (F / 4), part = piece % (F / 4);
    const float4 w4 = s_tile[(wave0 + src_lane) * (F / 4) + part];
    float* d = s_dst[wave0 + src_lane] + 4 * part;
    const int left = s_count[wave0 + src_lane] - 4 * part;
    if (left >= 4) {
      float4u w = {w4.x, w4.y, w4.z, w4.w};
      store_row4(reinterpret_cast<float4u*>(d), w);
    } else if (left > 0) {
      d[0] = w4.x;
      if (left > 1) d[1] = w4.y;
      if (left > 2) d[2] = w4.z;
    }
  }
}

// Short rows (tic_tac_toe 27 floats, kuhn_poker 7 / 11, leduc_poker 16 / 30, ...): ONE LANE PER STATE.  The lane
// loads its state once and walks the cursor over the whole row into the wavefront's LDS tile (row stride padded
// to an odd number of words: conflict-free); the 64 rows of a wavefront are one contiguous, 16-byte aligned span
// of the output (64 * size floats), which the wavefront then writes as aligned float4 — 1 KiB per store
// instruction instead of 64 scattered 12-28 byte pieces.  Needs a 16-byte aligned output.
constexpr int kRowsBlock = 256;
constexpr int kRowsMaxSize = 63;
// kR: states per lane.  A workgroup of the shortest rows (kuhn_poker: 4 bytes in, 28 out per state) carries 7 KiB; with
// eight of them per CU the bytes in flight (57 KiB per CU) do not cover bandwidth x latency of the memory system, so
// the launch is bound by how long a workgroup LIVES, not by what it moves.  kR consecutive blocks of 64 states per
// wavefront (all kR state loads issued before the first cursor step) put kR times the bytes behind every wavefront.
template <class G, int kR>
__global__ void __launch_bounds__(kRowsBlock)
k_observation_rows(typename G::Params p, const typename G::word_t* base, int64_t n, int size, int player, int which,
                   float* __restrict__ out) {
  extern __shared__ float s_rows[];  // [waves][kR * 64 * pad]
  const int pad = size | 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tile = s_rows + wave * (kR * 64) * pad;
  const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * kRowsBlock + wave * 64) * kR;  // first state of this wavefront
  if (i0 >= n) return;
  typename G::State st[kR];
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    const int64_t i = i0 + r * 64 + lane;
    if (i < n) st[r] = G::load(p, base, n, i);
  }
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    const int64_t i = i0 + r * 64 + lane;
    if (i < n) {
      const typename G::State& s = st[r];
      int pl = player;
      if (pl < 0) {
        pl = G::current_player(p, s);
        if (pl < 0) pl = 0;
      }
      typename G::ObsCursor cur;
      cur.init(p, s, pl, which, 0);
      float* row = tile + (r * 64 + lane) * pad;
      for (int k = 0; k < size; ++k) row[k] = cur.next(p, s, pl, which);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int rows = static_cast<int>(n - i0 < 64 * kR ? n - i0 : 64 * kR);
  const int total = rows * size;                       // floats this wavefront writes
  float* dst = out + i0 * size;                        // 64 * kR * size * 4 bytes per wavefront: 16-byte aligned
  for (int j = 4 * lane; j < total; j += 256) {
    int r = j / size, k = j - r * size;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = (j + e < total) ? tile[r * pad + k] : 0.0f;
      if (++k == size) { k = 0; ++r; }
    }
    if (j + 4 <= total) {
      store_row4<!std::is_same<G, Ttt>::value>(reinterpret_cast<float4*>(dst + j), make_float4(v[0], v[1], v[2], v[3]));
    } else {
      for (int e = 0; e < 4 && j + e < total; ++e) dst[j + e] = v[e];
    }
  }
}

// connect_four 6x7 tensor pack, fallback for an output pointer that is only 4-byte aligned: one lane per
// BOARD ROW of the tensor (3 planes x 6 rows per state, 7 floats each).  The seven cells of a row sit at
// bit stride 7 in the column-major bitboard; one multiply gathers them (same identity as
// C4T::open_columns), then each float is a bit-field extract; stores are 16 + 12 bytes per lane.
typedef float float3u __attribute__((ext_vector_type(3), aligned(4)));
__global__ void __launch_bounds__(kBlock)
k_observation_c4std(C4Params p, const uint64_t* __restrict__ base, int64_t n, int player, float* __restrict__ out) {
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (row >= n * 18) return;
  const int64_t i = row / 18;
  const int rem = static_cast<int>(row - i * 18);
  const int plane = rem / 6, r = rem - plane * 6;
  const C4Std::State s = C4Std::unpack(base[i], base[n + i]);
  uint64_t first = s.x, second = s.o;
  if (p.ego) {  // PlayerRelative (connect_four.cc:299-310)
    int pl = player;
    if (pl < 0) {
      pl = C4Std::current_player(p, s);
      if (pl < 0) pl = 0;
    }
    first = pl == 0 ? s.o : s.x;
    second = pl == 0 ? s.x : s.o;
  }
  const uint64_t bits = plane == 0 ? first : (plane == 1 ? second : ~(s.x | s.o));
  const uint64_t stride7 = 1ull | (1ull << 7) | (1ull << 14) | (1ull << 21) | (1ull << 28) | (1ull << 35) | (1ull << 42);
  const uint64_t M = (1ull << 36) | (1ull << 30) | (1ull << 24) | (1ull << 18) | (1ull << 12) | (1ull << 6) | 1ull;
  const uint32_t g = static_cast<uint32_t>((((bits >> r) & stride7) * M) >> 36) & 0x7Fu;  // bit c = column c
  float v[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) v[c] = static_cast<float>((g >> c) & 1u);
  float* dst = out + row * 7;
  float4u lo = {v[0], v[1], v[2], v[3]};
  float3u hi = {v[4], v[5], v[6]};
  store_row4(reinterpret_cast<float4u*>(dst), lo);
  *reinterpret_cast<float3u*>(dst + 4) = hi;
}

// connect_four 6x7 fast path of the tensor pack: one lane per (state, PLANE), 6 rows x 7 floats = 168 bytes
// per lane, which amortises the index arithmetic and the state load over six times more output than a
// row per lane would (about 1.2 instructions per output byte instead of 5; 117 -> 94 us for [2^20, 126]).  A wavefront owns a contiguous, 16-byte
// aligned span of 64 x 42 floats; it is staged in LDS (8-byte writes at a 168-byte lane stride) and
// written back as aligned float4, one KiB per store instruction.  Needs a 16-byte aligned output.
#ifndef OSG_C4OBS_BLOCK
#define OSG_C4OBS_BLOCK 128
#endif
constexpr int kC4ObsBlock = OSG_C4OBS_BLOCK;
// kNt: non-temporal stores — slower while the tensor is small (2^20 states: 95.7 vs 90.8 us), faster once it is
// gigabytes (2^24 states, 8.5 GB: 1 395 vs 1 485 us); the launcher picks by size.
template <bool kNt>
__global__ void __launch_bounds__(kC4ObsBlock)
k_observation_c4std_planes(C4Params p, const uint64_t* __restrict__ base, int64_t n, int player, float* __restrict__ out) {
  __shared__ float2 s_stage[kC4ObsBlock * 21];
  const int64_t gl = static_cast<int64_t>(blockIdx.x) * kC4ObsBlock + threadIdx.x;
  const int64_t lanes = n * 3;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2* w2 = s_stage + wave * (64 * 21);
  if (gl < lanes) {
    const int64_t i = gl / 3;
    const int plane = static_cast<int>(gl - i * 3);
    const C4Std::State s = C4Std::unpack(base[i], base[n + i]);
    uint64_t first = s.x, second = s.o;
    if (p.ego) {  // PlayerRelative (connect_four.cc:299-310)
      int pl = player;
      if (pl < 0) {
        pl = C4Std::current_player(p, s);
        if (pl < 0) pl = 0;
      }
      first = pl == 0 ? s.o : s.x;
      second = pl == 0 ? s.x : s.o;
    }
    const uint64_t bits = plane == 0 ? first : (plane == 1 ? second : ~(s.x | s.o));
    const uint64_t stride7 = 1ull | (1ull << 7) | (1ull << 14) | (1ull << 21) | (1ull << 28) | (1ull << 35) | (1ull << 42);
    const uint64_t M = (1ull << 36) | (1ull << 30) | (1ull << 24) | (1ull << 18) | (1ull << 12) | (1ull << 6) | 1ull;
    float v[42];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const uint32_t g = static_cast<uint32_t>((((bits >> r) & stride7) * M) >> 36);  // bit c = column c of row r
#pragma unroll
      for (int c = 0; c < 7; ++c) v[r * 7 + c] = static_cast<float>((g >> c) & 1u);
    }
#pragma unroll
    for (int j = 0; j < 21; ++j) w2[lane * 21 + j] = make_float2(v[2 * j], v[2 * j + 1]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int64_t wave_lane0 = static_cast<int64_t>(blockIdx.x) * kC4ObsBlock + wave * 64;
  if (wave_lane0 >= lanes) return;
  const int64_t left = (lanes - wave_lane0) * 42;
  const int valid = left < 64 * 42 ? static_cast<int>(left) : 64 * 42;  // floats this wavefront owns
  float* gdst = out + wave_lane0 * 42;
  const float4* w4 = reinterpret_cast<const float4*>(w2);
  const float* w1 = reinterpret_cast<const float*>(w2);
#pragma unroll
  for (int j = 0; j < 11; ++j) {
    const int piece = lane + 64 * j;  // 672 float4 pieces
    if (piece * 4 + 4 <= valid) {
      store_row4<kNt>(reinterpret_cast<float4*>(gdst) + piece, w4[piece]);
    } else {
      for (int k = piece * 4; k < valid && k < piece * 4 + 4; ++k) gdst[k] = w1[k];
    }
  }
}

// hex 9-plane tensor: one lane per (state, plane).  The plane's membership mask is boolean algebra on the
// bitboards (HexT::plane_mask); the lane turns its `cells` bits into floats, stages them in LDS at a lane
// stride of `cells` words, and the wavefront's span — 64 x cells floats, contiguous and 16-byte aligned —
// goes out as aligned float4, one KiB per store instruction.  One wavefront per workgroup: the stage is
// 256 x cells bytes (20 KiB for 9 x 9), so seven wavefronts share a CU's LDS (157 us with two-wave groups,
// 148 us with one).  Needs a 16-byte aligned output.
constexpr int kHexObsBlock = 64;
template <class G>
__global__ void __launch_bounds__(kHexObsBlock)
k_observation_hex_planes(typename G::Params p, const typename G::word_t* base, int64_t n, int planes, float* __restrict__ out) {
  extern __shared__ float s_hex_stage[];
  const int cells = p.cells;
  const int64_t gl = static_cast<int64_t>(blockIdx.x) * kHexObsBlock + threadIdx.x;
  const int64_t lanes = n * planes;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* w = s_hex_stage + wave * 64 * cells;
  if (gl < lanes) {
    const int64_t i = gl / planes;
    const int plane = static_cast<int>(gl - i * planes);
    const typename G::State s = G::load(p, base, n, i);
    const typename G::Bits m = G::plane_mask(p, s, plane);
    float* mine = w + lane * cells;
    constexpr int kWords = static_cast<int>(sizeof(m.w) / sizeof(m.w[0]));
#pragma unroll
    for (int k = 0; k < kWords; ++k) {
      const int count = cells - 32 * k < 32 ? cells - 32 * k : 32;  // wave-uniform
      uint32_t bits = m.w[k];
#pragma unroll 8
      for (int b = 0; b < count; ++b) {
        mine[32 * k + b] = static_cast<float>(bits & 1u);
        bits >>= 1;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int64_t wave_lane0 = static_cast<int64_t>(blockIdx.x) * kHexObsBlock + wave * 64;
  if (wave_lane0 >= lanes) return;
  const int64_t left = (lanes - wave_lane0) * cells;
  const int span = 64 * cells;
  const int valid = left < span ? static_cast<int>(left) : span;  // floats this wavefront owns
  float* gdst = out + wave_lane0 * cells;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  for (int piece = lane; piece * 4 < valid; piece += 64) {
    if (piece * 4 + 4 <= valid) {
      store_row4(reinterpret_cast<float4*>(gdst) + piece, w4[piece]);
    } else {
      for (int k = piece * 4; k < valid; ++k) gdst[k] = w[k];
    }
  }
}

// ---------------------------------------------------------------------------
// Tensor pack, "one aligned 16-byte piece per thread": thread t computes floats [4t, 4t + 4) of the flat
// [n, size] output and stores them once.  The store pattern is the one tools/fill_probe.hip measured as the
// write-only ceiling (0.86 of 8 TB/s: every wave-instruction covers one aligned KiB, consecutive waves
// consecutive KiB), where any form in which a wavefront owns a multi-KiB span of its own stays at 0.69-0.77
// (profiles/r03_fill_probe.log) — which is where the span-per-wavefront packers above sit.  The price is that a
// thread recomputes its position (state, offset) from the piece index and that the ~size / 4 threads of one state
// all need that state; what it takes to reach the ceiling (profiles/r04_obs_forms.log has every step):
//   * few instructions: at 0.85 a SIMD retires a 64-piece wavefront every ~150 ns, ~145 vector instructions — a
//     game's generic cursor per piece is far too slow (0.29-0.59), each game below has its own bit arithmetic;
//   * several pieces per thread with all loads issued first: a wavefront is load latency, then stores; one
//     piece per thread keeps too few bytes in flight (0.75), four reach 0.84, six / eight fall back (0.79);
//   * non-temporal stores: plain stores halve the rate (0.41-0.48) as soon as loads share the launch;
//   * whole aligned KiB per wave-instruction: spans that start at 16-byte but not 1 KiB boundaries cost 0.77 -> 0.55.
// Needs a 16-byte aligned output of fewer than 2^32 floats (the span-per-wavefront kernels serve the rest).
// ---------------------------------------------------------------------------
constexpr int kPieceBlock = 256;
// connect_four 6 x 7 in the piece form.  A piece is four consecutive cells in row-major order (row r, columns
// c .. c + 3, running on into the next row / plane / state); in the column-major bitboard (bit = 7 col + row) the cells
// of a row sit 7 bits apart, so ONE 64-bit shift per row brings a whole row's bits into a 32-bit window and each
// float is a bit-field extract: window 0 = the first cell's row from column c on, window 1 = the following row
// (of the same plane, the next plane, or plane 0 of the next state).  126 floats per row is even and a piece
// starts at a multiple of four, so floats 0 and 1 of a piece never leave the first state.
// What the form costs is instructions, not bytes: at 0.85 of 8 TB/s a SIMD retires a 64-piece wavefront every
// ~150 ns, i.e. ~145 vector instructions (the first version of this kernel had 127 + a 64-bit scalar division:
// 0.77).  Hence: 32-bit indices throughout (the launcher sends tensors of 2^32 floats or more elsewhere), the
// workgroup's first state by a 32-bit division by a constant, no unpacking of the result byte (no window ever
// reaches bits 49+ unmasked), and the four cells as bit-field extracts of ONE word U built from the two windows
// (float k = bit 7k of U).  kEgo: egocentric_obs_tensor (connect_four.cc:299-310), its own instantiation.
// kPer pieces per thread: piece j of a thread lies T = threads-of-the-launch pieces after piece j - 1 (every store
// instruction of the grid still covers consecutive KiB); all state loads are issued before the first float is formed.
template <bool kNt, bool kEgo, int kPer>
__global__ void __launch_bounds__(kPieceBlock)
k_observation_c4std_pieces(C4Params p, const uint64_t* __restrict__ base, uint32_t n, uint32_t total, int player,
                           float* __restrict__ out) {
  uint64_t A0[kPer], A1[kPer], B0[kPer], B1[kPer];
  uint32_t offs[kPer];
  bool live[kPer];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const uint32_t wg = blockIdx.x + j * gridDim.x;       // < 2^22
    const uint32_t ib = (wg * 512u) / 63u;                // = 1024 wg / 126: first state the workgroup touches
    const uint32_t local = (wg * 1024u - ib * 126u) + 4u * threadIdx.x;   // < 126 + 1024
    const uint32_t il = (local * 1041u) >> 17;            // local / 126 for local < 2^13
    uint32_t i = ib + il;
    live[j] = i < n;
    if (!live[j]) i = n - 1;
    offs[j] = local - il * 126u;                          // even: floats off, off + 1 are in state i
    const uint32_t i1 = i + 1u < n ? i + 1u : i;
    A0[j] = base[i]; A1[j] = base[n + i]; B0[j] = base[i1];
    if (kEgo) B1[j] = base[n + i1];
  }
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    if (!live[j]) continue;
    const uint32_t wg = blockIdx.x + j * gridDim.x;
    uint64_t a0 = A0[j], a1 = A1[j], b0 = B0[j];
    if (kEgo) {  // PlayerRelative (connect_four.cc:299-310)
      const C4Std::State sa = C4Std::unpack(a0, a1), sb = C4Std::unpack(b0, B1[j]);
      int pa = player, pb = player;
      if (player < 0) {
        pa = C4Std::current_player(p, sa); if (pa < 0) pa = 0;
        pb = C4Std::current_player(p, sb); if (pb < 0) pb = 0;
      }
      a0 = pa == 0 ? sa.o : sa.x;
      a1 = pa == 0 ? sa.x : sa.o;
      b0 = pb == 0 ? sb.o : sb.x;
    }
    const uint64_t a2 = ~(a0 | a1);
    const uint32_t off = offs[j];
    const uint32_t plane = (off >= 42u) + (off >= 84u);
    const uint32_t cell = off - __umul24(42u, plane);
    const uint32_t row = __umul24(cell, 37u) >> 8;         // cell / 7 for cell < 42
    const uint32_t col = cell - __umul24(7u, row);
    // window 0: the first cell's row from its column on (bit 7k = column col + k); window 1: the following row from
    // column 0 — of the same plane, of the next plane, or of plane 0 of the next state
    const uint64_t bits0 = plane == 0 ? a0 : (plane == 1 ? a1 : a2);
    const uint32_t w0 = static_cast<uint32_t>(bits0 >> (__umul24(7u, col) + row));
    const bool last_row = row == 5u;
    const uint32_t plane1 = plane + (last_row ? 1u : 0u);
    const uint64_t bits1 = plane1 == 0 ? a0 : (plane1 == 1 ? a1 : (plane1 == 2 ? a2 : b0));
    const uint32_t w1 = static_cast<uint32_t>(bits1 >> (last_row ? 0u : row + 1u));
    const uint32_t t7 = 49u - __umul24(7u, col);           // bits of window 0 that are cells of this row: 7 (7 - col)
    const uint32_t u = t7 >= 28u ? w0 : ((w0 & ((1u << t7) - 1u)) | (w1 << t7));
    const float4 v = make_float4(static_cast<float>(u & 1u), static_cast<float>((u >> 7) & 1u),
                                 static_cast<float>((u >> 14) & 1u), static_cast<float>((u >> 21) & 1u));
    const uint32_t f0 = wg * 1024u + 4u * threadIdx.x;
    float* dst = out + (static_cast<size_t>(wg) * 1024u) + 4u * threadIdx.x;   // scalar base + 32-bit lane offset
    if (f0 + 4u <= total) {
      store_row4<kNt>(reinterpret_cast<float4*>(dst), v);
    } else {  // the last piece of the tensor (126 n is even, not always a multiple of four).  Atomic stores: plain ones
      // are merged with the vector store above into a 12-byte + a 4-byte store on EVERY lane (seen in the ISA: 0.34)
      __hip_atomic_store(dst, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The piece form for the other board tensors, written once.  The mapping is FLAT — thread t of workgroup w forms
// floats [1024 w + 4 t, + 4): every wave-instruction stores one whole, 1 KiB-aligned KiB; a first version that gave a
// workgroup whole rows (972 floats for tic_tac_toe: spans aligned to 16 bytes only) ran at 0.55 where this runs at
// the fill ceiling — so the workgroup's first state is a division of 1024 w by the row length: five scalar
// instructions with a host-made multiplier (FastDiv, libdivide's branch-free form), and the lane's row a
// multiply-shift the host has verified for its range.  Like the connect_four kernel above, every thread serves kPer
// spans, all state loads issued first.  F is the game's piece functor:
//   Words F::load(i)                           the raw state words a piece of state i may need
//   float4 F::piece(Words a, Words b, off)     floats off .. off + 3 of state a's row, running on into state b's
struct FastDiv {   // x / d for any 32-bit x: t = mulhi(x, m); q = (((x - t) >> 1) + t) >> s
  uint32_t m, s, d;
  OSG_HD uint32_t div(uint32_t x) const {
#ifdef __HIP_DEVICE_COMPILE__
    const uint32_t t = __umulhi(x, m);
#else
    const uint32_t t = static_cast<uint32_t>((static_cast<uint64_t>(x) * m) >> 32);
#endif
    return (((x - t) >> 1) + t) >> s;
  }
};
inline FastDiv make_fast_div(uint32_t d) {  // d >= 2
  FastDiv f;
  f.d = d;
  const uint32_t k = 31u - static_cast<uint32_t>(__builtin_clz(d));
  if ((d & (d - 1)) == 0) { f.m = 0; f.s = k - 1; return f; }   // 2^k: t = 0, q = (x >> 1) >> (k - 1)
  const uint64_t two = uint64_t{1} << (32 + k);
  uint64_t m = two / d;
  const uint64_t rem = two - m * d;
  m += m;
  const uint64_t twice = rem + rem;
  if (twice >= d) m += 1;
  f.m = static_cast<uint32_t>(m + 1);
  f.s = k;
  return f;
}
template <class F, bool kNt, int kPer>
__global__ void __launch_bounds__(kPieceBlock)
k_observation_row_pieces(F f, uint32_t n, FastDiv by_size, uint32_t lmagic, uint32_t lshift, uint32_t total,
                         float* __restrict__ out) {
  typename F::Words wa[kPer], wb[kPer];
  uint32_t offs[kPer];
  const uint32_t size = by_size.d;
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const uint32_t fb = (blockIdx.x + j * gridDim.x) * 1024u;        // first float of the span (scalar)
    const uint32_t ib = by_size.div(fb);
    const uint32_t local = (fb - ib * size) + 4u * threadIdx.x;      // < size + 1024
    const uint32_t il = (local * lmagic) >> lshift;                  // local / size
    uint32_t i = ib + il;
    offs[j] = local - il * size;
    if (i >= n) i = n - 1u;                                          // (a piece past the end: not stored)
    wa[j] = f.load(i);
    wb[j] = f.load(i + 1u < n ? i + 1u : i);
  }
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const uint32_t f0 = (blockIdx.x + j * gridDim.x) * 1024u + 4u * threadIdx.x;
    if (f0 >= total) continue;
    const float4 v = f.piece(wa[j], wb[j], offs[j]);
    float* dst = out + static_cast<size_t>(blockIdx.x + j * gridDim.x) * 1024u + 4u * threadIdx.x;
    if (f0 + 4u <= total) {
      store_row4<kNt>(reinterpret_cast<float4*>(dst), v);
    } else {  // the tensor's last piece (atomic stores: see k_observation_c4std_pieces)
      const uint32_t left = total - f0;
      __hip_atomic_store(dst, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (left > 1u) __hip_atomic_store(dst + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (left > 2u) __hip_atomic_store(dst + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// The same pieces with the rows' IMAGES staged in LDS: a span of 1024 floats belongs to 1024 / size + 2 states, and
// in the kernel above every piece rebuilds the images of its two states (leduc_poker's information row: 82 vector
// instructions per piece, the vector unit 84 % busy — profiles/r04_pmc_obs_rows_pieces.txt).  Here the workgroup's
// kSpans spans first get their states' images, one (span, state) per thread, then a piece is two LDS reads, a shift
// and four conversions.
template <class F, bool kNt, int kSpans>
__global__ void __launch_bounds__(kPieceBlock)
k_observation_row_pieces_lds(F f, uint32_t n, FastDiv by_size, uint32_t lmagic, uint32_t lshift, uint32_t cap, uint32_t cmagic,
                             uint32_t cshift, uint32_t total, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(8))) unsigned char s_raw[];
  typename F::Img* s_img = reinterpret_cast<typename F::Img*>(s_raw);   // [kSpans][cap]
  const uint32_t size = by_size.d;
  for (uint32_t flat = threadIdx.x; flat < kSpans * cap; flat += kPieceBlock) {
    const uint32_t j = (flat * cmagic) >> cshift, sl = flat - j * cap;                  // flat / cap
    const uint32_t fb = (blockIdx.x + j * gridDim.x) * 1024u;
    if (fb >= total) continue;
    uint32_t i = by_size.div(fb) + sl;
    if (i >= n) i = n - 1u;
    s_img[flat] = f.image(f.load(i));
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSpans; ++j) {
    const uint32_t fb = (blockIdx.x + j * gridDim.x) * 1024u;
    const uint32_t f0 = fb + 4u * threadIdx.x;
    if (f0 >= total) continue;
    const uint32_t ib = by_size.div(fb);
    const uint32_t local = (fb - ib * size) + 4u * threadIdx.x;      // < size + 1024
    const uint32_t il = (local * lmagic) >> lshift;                  // local / size
    const uint32_t off = local - il * size;
    const float4 v = f.piece_img(s_img[j * cap + il], s_img[j * cap + il + 1u], off);
    float* dst = out + static_cast<size_t>(blockIdx.x + j * gridDim.x) * 1024u + 4u * threadIdx.x;
    if (f0 + 4u <= total) {
      store_row4<kNt>(reinterpret_cast<float4*>(dst), v);
    } else {  // the tensor's last piece (atomic stores: see k_observation_c4std_pieces)
      const uint32_t left = total - f0;
      __hip_atomic_store(dst, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (left > 1u) __hip_atomic_store(dst + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (left > 2u) __hip_atomic_store(dst + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
OSG_D float4 low_four_bits(uint32_t u) {
  return make_float4(static_cast<float>(u & 1u), static_cast<float>((u >> 1) & 1u), static_cast<float>((u >> 2) & 1u),
                     static_cast<float>((u >> 3) & 1u));
}
// tic_tac_toe (tic_tac_toe.cc:241-251): the row is a 27-bit image, plane 0 empty | plane 1 o | plane 2 x.
struct TttPieces {
  const uint32_t* base;
  using Words = uint32_t;
  OSG_D Words load(uint32_t i) const { return base[i]; }
  OSG_D static uint32_t image(uint32_t w) {
    const uint32_t x = w & 0x1FFu, o = (w >> 16) & 0x1FFu;
    return (~(x | o) & 0x1FFu) | (o << 9) | (x << 18);
  }
  OSG_D float4 piece(Words a, Words b, uint32_t off) const {  // off <= 26
    return low_four_bits((image(a) >> off) | (image(b) << (27u - off)));
  }
  using Img = uint32_t;
  OSG_D float4 piece_img(Img a, Img b, uint32_t off) const { return low_four_bits((a >> off) | (b << (27u - off))); }
};
// kuhn_poker, two players (KuhnObserver::WriteTensor, kuhn_poker.cc:72-107), in the piece form.  The round-3 rows
// kernel spends its whole launch issuing vector instructions (profiles/r04_pmc_k_observation_rows_kuhn_2p24.txt:
// SQ_ACTIVE_INST_VALU x 4 cycles / SIMD = the launch's duration): here a row is one small image —
//   observation  [player 2 | private card 3 | pot contribution 2]: seven 4-bit entries (the contributions are 1 .. 3)
//   information  [player 2 | private card 3 | betting 3 x 2]:      eleven 1-bit entries
// — two images per piece (its state and the next), one 64-bit shift, four field extracts.
template <int kWhich>
struct Kuhn2Pieces {
  const uint64_t* base;
  int player;
  using Words = uint64_t;
  static constexpr uint32_t kSize = kWhich == 0 ? 7u : 11u, kBits = kWhich == 0 ? 4u : 1u;
  OSG_D Words load(uint32_t i) const { return base[i]; }
  OSG_D uint64_t image(uint64_t h) const {
    const Kuhn::Params p{1, 2};
    const Kuhn::State s{h};
    int pl = player;
    if (pl < 0) {
      pl = Kuhn::current_player(p, s);
      if (pl < 0) pl = 0;
    }
    const uint32_t len = static_cast<uint32_t>(h & 31ull), bets = static_cast<uint32_t>(h >> 45);
    const uint32_t card = static_cast<uint32_t>(h >> (5 + 4 * pl)) & 15u;
    const bool dealt = len > static_cast<uint32_t>(pl);
    if (kWhich == 0) {
      const uint32_t c0 = 1u + (bets & 1u) + ((bets >> 2) & 1u), c1 = 1u + ((bets >> 1) & 1u);   // contribution(), P = 2
      uint32_t img = (1u << (4 * pl)) | (c0 << 20) | (c1 << 24);
      if (dealt) img |= 1u << (8 + 4 * card);
      return img;
    }
    uint32_t img = 1u << pl;
    if (dealt) img |= 1u << (2 + card);
    const uint32_t nact = len > 2u ? len - 2u : 0u;
#pragma unroll
    for (uint32_t j = 0; j < 3; ++j)
      if (j < nact) img |= 1u << (5u + 2u * j + ((bets >> j) & 1u));
    return img;
  }
  OSG_D float4 piece(Words a, Words b, uint32_t off) const {
    const uint64_t both = (image(a) | (image(b) << (kSize * kBits))) >> (kBits * off);
    const uint32_t u = static_cast<uint32_t>(both), m = (1u << kBits) - 1u;
    return make_float4(static_cast<float>(u & m), static_cast<float>((u >> kBits) & m),
                       static_cast<float>((u >> (2 * kBits)) & m), static_cast<float>((u >> (3 * kBits)) & m));
  }
};
// leduc_poker, two players (LeducObserver::WriteTensor, leduc_poker.cc:103-192), in the piece form, K = the number
// of card ranks the tensor distinguishes (6, or 3 with suit isomorphism):
//   observation  [player 2 | private K | public K | pot contribution 2]: 4-bit entries (contributions <= 13), <= 64 bits
//   information  [player 2 | private K | public K | betting 2 x 4 x 2]: 1-bit entries; a move's pair (call "10",
//                raise "01", fold "00") IS its 2-bit code in the record (1 call, 2 raise, 0 fold), so a round's
//                betting bits are its move sequence masked to its length.
template <int kWhich>
struct Leduc2Pieces {
  const uint64_t* base;
  uint32_t n;
  int player, K;
  Leduc::Params p;
  struct Words { uint64_t a, b; };
  static constexpr uint32_t kBits = kWhich == 0 ? 4u : 1u;
  OSG_D Words load(uint32_t i) const { return {base[i], base[n + i]}; }
  using Img = typename std::conditional<kWhich == 0, uint64_t, uint32_t>::type;   // information rows: 30 bits
  OSG_D Img image(const Words& w) const {
    // only the fields a row shows are taken out of the two packed words (Leduc::unpack's layout)
    const uint32_t ante_pk = static_cast<uint32_t>(w.a >> 42) & 0xFFFu, priv_pk = static_cast<uint32_t>(w.b >> 34) & 0xFFFu;
    const int pub = static_cast<int>((w.a >> 18) & 15ull) - 1;
    int pl = player;
    if (pl < 0) {
      pl = Leduc::current_player(p, Leduc::unpack(w.a, w.b));
      if (pl < 0) pl = 0;
    }
    const int pc = static_cast<int>((priv_pk >> (4 * pl)) & 15u) - 1;
    Img img = Img{1} << (kBits * pl);
    if (pc >= 0) img |= Img{1} << (kBits * (2 + pc));
    if (pub >= 0) img |= Img{1} << (kBits * (2 + K + pub));
    if (kWhich == 0) {
      img |= static_cast<Img>(ante_pk & 0xFFu) << (4 * (2 + 2 * K));   // ante[0] | ante[1] << 4: two entries
    } else {
      const uint32_t lo = static_cast<uint32_t>(w.b);
      const uint32_t len0 = lo & 7u, seq0 = (lo >> 3) & 0xFFu, len1 = (lo >> 17) & 7u, seq1 = (lo >> 20) & 0xFFu;
      const uint32_t r0 = seq0 & ((1u << (2 * len0)) - 1u), r1 = seq1 & ((1u << (2 * len1)) - 1u);
      img |= static_cast<Img>((r0 & 0xFFu) | ((r1 & 0xFFu) << 8)) << (2 + 2 * K);
    }
    return img;
  }
  OSG_D float4 piece(const Words& a, const Words& b, uint32_t off) const {
    const uint32_t size = kWhich == 0 ? 4u + 2u * K : 18u + 2u * K;
    Img both = image(a) >> (kBits * off);
    const uint32_t in_a = size - off;                        // entries of the piece that lie in a's row (>= 1)
    if (in_a < 4u) both |= image(b) << (kBits * in_a);
    const uint32_t u = static_cast<uint32_t>(both), m = (1u << kBits) - 1u;
    return make_float4(static_cast<float>(u & m), static_cast<float>((u >> kBits) & m),
                       static_cast<float>((u >> (2 * kBits)) & m), static_cast<float>((u >> (3 * kBits)) & m));
  }
  OSG_D float4 piece_img(Img ia, Img ib, uint32_t off) const {
    const uint32_t size = kWhich == 0 ? 4u + 2u * K : 18u + 2u * K;
    const uint32_t in_a = size - off;                        // entries of the piece that lie in a's row (>= 1)
    Img both = ia >> (kBits * off);
    both |= in_a < 4u ? ib << (kBits * in_a) : Img{0};
    const uint32_t u = static_cast<uint32_t>(both), m = (1u << kBits) - 1u;
    return make_float4(static_cast<float>(u & m), static_cast<float>((u >> kBits) & m),
                       static_cast<float>((u >> (2 * kBits)) & m), static_cast<float>((u >> (3 * kBits)) & m));
  }
};
// hex, the 9-plane tensor (hex.cc:379-398: plane = label + 4), in the piece form: a piece is four consecutive cells
// of one plane's membership mask (HexT::plane_mask: boolean algebra on the stone / edge-connection planes), running
// on into the next plane of the same state or plane 0 of the next state.  It needs nine words (three planes x {two
// words of the first mask — the four bits may straddle a word —, word 0 of the following mask}), each in another
// plane of the SoA image; as nine global loads per piece that is 0.35 of 8 TB/s (the texture path spends its cycles
// on load INSTRUCTIONS, not bytes).  So a workgroup owns spans of 4096 consecutive floats (16 KiB: four aligned KiB
// per wavefront); the 5-7 states a span belongs to are fetched ONCE into LDS, one word per thread (the only global
// loads), and the pieces read their nine words from there (same-address LDS reads within a wavefront: broadcasts).
constexpr int kHexLdsSpan = 4096;             // floats per workgroup
constexpr int kHexLdsMaxStates = 18;          // 4096 / (9 * 29 cells) + 2
// kSpans spans per workgroup (span j of workgroup w = span w + j * gridDim.x): all their states are fetched before the
// one barrier, so kSpans x 16 KiB of stores stand behind one load round trip.
template <int NW, bool kNt, int kSpans>
__global__ void __launch_bounds__(kPieceBlock)
k_observation_hex_pieces_lds(const uint32_t* __restrict__ base, uint32_t n, uint32_t cells, uint32_t cmagic, uint32_t cshift,
                             FastDiv by_size, uint32_t lmagic, uint32_t lshift, uint32_t total, float* __restrict__ out,
                             uint32_t last_word_mask) {   // (0x07FFFFFF where the planes' last words carry the meta bits)
  __shared__ uint32_t s_words[kSpans][kHexLdsMaxStates * 4 * NW];
  const uint32_t size = by_size.d;   // 9 cells
  uint32_t ias[kSpans];
#pragma unroll
  for (int j = 0; j < kSpans; ++j) {
    const uint32_t fb = (blockIdx.x + j * gridDim.x) * static_cast<uint32_t>(kHexLdsSpan);
    ias[j] = 0;
    if (fb >= total) continue;
    const uint32_t ia = by_size.div(fb);                               // first state of the span (scalar)
    ias[j] = ia;
    uint32_t last = fb + kHexLdsSpan - 1u;
    if (last >= total) last = total - 1u;
    const uint32_t count = by_size.div(last) - ia + 1u;                // states the span touches ...
    const uint32_t fetch = (count + 1u) * 4u * NW;                     // ... and one more (a piece's next mask), clamped
    if (threadIdx.x < fetch) {
      const uint32_t sl = threadIdx.x / (4u * NW), w = threadIdx.x - sl * 4u * NW;
      uint32_t i = ia + sl;
      if (i >= n) i = n - 1u;
      const uint32_t word = base[w * n + i];                           // plane-major SoA: word w of state i
      s_words[j][threadIdx.x] = (w % NW == NW - 1u) ? (word & last_word_mask) : word;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSpans; ++j) {
    const uint32_t span = blockIdx.x + j * gridDim.x;
    const uint32_t fb = span * static_cast<uint32_t>(kHexLdsSpan);
    if (fb >= total) break;
    const uint32_t rem = fb - ias[j] * size;
    const uint32_t* words = s_words[j];
#pragma unroll
    for (int r = 0; r < kHexLdsSpan / (4 * kPieceBlock); ++r) {
      const uint32_t p4 = 4u * (r * kPieceBlock + threadIdx.x);        // float inside the span
      const uint32_t f0 = fb + p4;
      if (f0 >= total) break;
      const uint32_t local = rem + p4;                                 // < size + 4096
      const uint32_t il = (local * lmagic) >> lshift;                  // local / size
      const uint32_t off = local - il * size;
      const uint32_t plane = (off * cmagic) >> cshift;                 // off / cells
      const uint32_t cell0 = off - plane * cells;
      const uint32_t k0 = cell0 >> 5, k1 = k0 + 1u < NW ? k0 + 1u : k0;
      const bool wrap = plane == 8u;                                   // the following mask: plane 0 of the next state
      const int l0 = static_cast<int>(plane) - 4, l1 = wrap ? -4 : l0 + 1;
      const uint32_t sa = il * 4u * NW, sb = wrap ? sa + 4u * NW : sa;
      // planes: 0 black, 1 white, 2 edge A, 3 edge B.  l == 0 (empty): X = black, Y = white.  else X = own, Y = ea, Z = eb
      const uint32_t px0 = (l0 >= 0 ? 0u : 1u) * NW, py0 = (l0 == 0 ? 1u : 2u) * NW;
      const uint32_t px1 = (l1 >= 0 ? 0u : 1u) * NW, py1 = (l1 == 0 ? 1u : 2u) * NW;
      uint32_t X[3], Y[3], Z[3];
      X[0] = words[sa + px0 + k0]; Y[0] = words[sa + py0 + k0]; Z[0] = words[sa + 3u * NW + k0];
      X[1] = words[sa + px0 + k1]; Y[1] = words[sa + py0 + k1]; Z[1] = words[sa + 3u * NW + k1];
      X[2] = words[sb + px1];      Y[2] = words[sb + py1];      Z[2] = words[sb + 3u * NW];
      uint32_t m[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int l = k < 2 ? l0 : l1;
        const int mag = l > 0 ? l : -l;                                 // 1 plain, 2 edge B only, 3 edge A only, 4 both
        const uint32_t fa = mag >= 3 ? 0u : ~0u, fb2 = (mag == 2 || mag == 4) ? 0u : ~0u;
        const uint32_t labelled = X[k] & (Y[k] ^ fa) & (Z[k] ^ fb2);
        m[k] = l == 0 ? ~(X[k] | Y[k]) : labelled;
      }
      const uint32_t second = k0 + 1u < NW ? m[1] : 0u;
      const uint32_t w0 = __funnelshift_r(m[0], second, cell0 & 31u);   // bit k = cell0 + k of mask 0
      const uint32_t t = cells - cell0;                                 // cells left in the plane (>= 1)
      const uint32_t u = t >= 4u ? w0 : ((w0 & ((1u << t) - 1u)) | (m[2] << t));
      const float4 v = low_four_bits(u);
      float* dst = out + static_cast<size_t>(span) * kHexLdsSpan + p4;
      if (f0 + 4u <= total) {
        store_row4<kNt>(reinterpret_cast<float4*>(dst), v);
      } else {
        const uint32_t left = total - f0;
        __hip_atomic_store(dst, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left > 1u) __hip_atomic_store(dst + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left > 2u) __hip_atomic_store(dst + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
// (x * magic) >> shift == x / d for every x < limit?  The host picks the pair with this check, so the kernels' lane
// arithmetic is a multiply and a shift whatever the row length.
inline bool find_div_magic(uint32_t d, uint32_t limit, uint32_t* magic, uint32_t* shift) {
  for (uint32_t s = 8; s <= 24; ++s) {
    const uint64_t m = ((uint64_t{1} << s) + d - 1) / d;
    if (m * (limit - 1) >= (uint64_t{1} << 32)) break;
    bool ok = true;
    for (uint32_t x = 0; x < limit && ok; ++x) ok = ((x * m) >> s) == x / d;
    if (ok) { *magic = static_cast<uint32_t>(m); *shift = s; return true; }
  }
  return false;
}

// `steps` uniformly random env steps per state with auto-reset, the state in registers throughout.
// Persistent grid (grid-stride over the states).  The two counters are reduced per workgroup and then
// added to one of 64 partial slots — 32 768 same-address atomics (one per wavefront) were measured at
// ~12 ns each, 400 us per launch, dwarfing the steps themselves; k_fold_counters sums the slots.
constexpr int kCounterSlots = 64;
template <class G>
__global__ void __launch_bounds__(kBlock)
k_random_steps(typename G::Params p, typename G::word_t* base, int64_t n, uint64_t seed, int64_t index_offset,
               int steps, unsigned long long* partials) {
  __shared__ unsigned long long s_sum[2][kBlock / 64];
  unsigned long long applied = 0, episodes = 0;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    typename G::State s = G::load(p, base, n, i);
    Rng rng(seed, static_cast<uint64_t>(index_offset + i), 0);
    for (int t = 0; t < steps; ++t) {
      if (G::terminal(p, s)) {
        s = G::initial(p);
        ++episodes;
      }
      auto m = G::legal(p, s);
      int a = sample_action<G>(p, s, m, G::current_player(p, s), rng);
      G::apply(p, s, a);
      ++applied;
    }
    G::store(p, base, n, i, s);
  }
  unsigned long long a = applied, e = episodes;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off);
    e += __shfl_xor(e, off);
  }
  if ((threadIdx.x & 63) == 0) {
    s_sum[0][threadIdx.x >> 6] = a;
    s_sum[1][threadIdx.x >> 6] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long ta = 0, te = 0;
    for (int w = 0; w < kBlock / 64; ++w) { ta += s_sum[0][w]; te += s_sum[1][w]; }
    const int slot = blockIdx.x & (kCounterSlots - 1);
    atomicAdd(&partials[2 * slot], ta);
    atomicAdd(&partials[2 * slot + 1], te);
  }
}
// Adds the partial slots into the caller's two counters and clears them for the next launch.
__global__ void __launch_bounds__(64) k_fold_counters(unsigned long long* partials, unsigned long long* counters) {
  const int lane = threadIdx.x;
  unsigned long long a = partials[2 * lane], e = partials[2 * lane + 1];
  partials[2 * lane] = 0;
  partials[2 * lane + 1] = 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off);
    e += __shfl_xor(e, off);
  }
  if (lane == 0) {
    counters[0] += a;
    counters[1] += e;
  }
}

// SURVEY.md 8(d) synthetic inputs on the counter stream, so that the CPU oracle regenerates the very batch a
// benchmark times (oracle/spiel_oracle_capi.cpp osgo_synth_batch restates this loop call for call):
//   rng   = Rng(seed, global index, kSynthSub)
//   depth = rng.below(depth_mod)                                       "d_i = hash(i) mod 36"
//   play `depth` moves from the initial state, chance outcomes by their distribution, player actions
//   uniformly over LegalActions(); a trajectory that ends before `depth` moves is thrown away and
//   re-drawn from the SAME stream ("re-drawn if terminal before d_i"), up to kSynthMaxAttempts times
//   (then the state is the initial state and depth 0: never reached by the configurations served);
//   action = one more draw of the same kind at the accepted, non-terminal state.
// One flat loop per lane, "step, or judge the finished attempt", so lanes on different attempts run the same code.
constexpr uint64_t kSynthSub = 0x53594E5448ULL;  // "SYNTH"
constexpr int kSynthMaxAttempts = 1 << 14;
template <class G>
__global__ void __launch_bounds__(kBlock)
k_synth(typename G::Params p, typename G::word_t* base, int64_t n, uint64_t seed, int64_t index_offset, int depth_mod,
        uint8_t* actions, int32_t* depth_out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  Rng rng(seed, static_cast<uint64_t>(index_offset + i), kSynthSub);
  int depth = static_cast<int>(rng.below(static_cast<uint32_t>(depth_mod)));
  typename G::State s = G::initial(p);
  int t = 0, attempt = 0;
  for (;;) {
    const bool term = G::terminal(p, s);
    if (t == depth || term) {
      if (!term) break;                        // accepted
      if (++attempt >= kSynthMaxAttempts) { s = G::initial(p); depth = 0; break; }
      s = G::initial(p);                       // re-draw the whole trajectory from the same stream
      t = 0;
      continue;
    }
    const auto m = G::legal(p, s);
    G::apply(p, s, sample_action<G>(p, s, m, G::current_player(p, s), rng));
    ++t;
  }
  G::store(p, base, n, i, s);
  const auto m = G::legal(p, s);
  const int a = sample_action<G>(p, s, m, G::current_player(p, s), rng);
  if (actions) actions[i] = static_cast<uint8_t>(a);
  if (depth_out) depth_out[i] = depth;
}

// One fused reinforcement-learning environment step for every state of the batch
// (python/rl_environment.py:379-418 Environment.step + :257-318 get_time_step, and
// python/vector_env.py:51-54 which loops over environments in Python):
//   * an environment whose previous time step was LAST (should_reset) starts a new
//     episode and ignores its action (rl_environment.py:405-406);
//   * otherwise the action is applied (illegal actions are counted, state unchanged);
//   * chance nodes are then resolved by sampling ChanceOutcomes() (_sample_external_events,
//     rl_environment.py:454-461) from the counter stream (seed, global env index, step);
//   * outputs: current player, step type (0 FIRST, 1 MID, 2 LAST), rewards (terminal
//     returns at LAST, zeros otherwise; the reference yields None at FIRST), the legal
//     mask of the new state, and the next should_reset flag.
template <class G>
__global__ void __launch_bounds__(kBlock)
k_env_step(typename G::Params p, typename G::word_t* base, int64_t n, int num_players, const int32_t* actions,
           uint8_t* should_reset, uint64_t seed, int64_t index_offset, int64_t step_index, int8_t* cur_player,
           uint8_t* step_type, double* rewards, uint32_t* mask, int mask_words, unsigned long long* illegal) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  typename G::State s = G::load(p, base, n, i);
  int type = 1;
  if (should_reset[i]) {
    s = G::initial(p);
    type = 0;
  } else {
    const int a = actions[i];
    if (a != OSG_INVALID_ACTION) {  // -1: leave this environment as it is (get_time_step without stepping)
      const auto m = G::legal(p, s);
      if (a < 0 || a >= 32 * G::kMaskW || !m.test(a)) atomicAdd(illegal, 1ull);
      else G::apply(p, s, a);
    }
  }
  Rng rng(seed, static_cast<uint64_t>(index_offset + i), static_cast<uint64_t>(step_index));
  for (int guard = 0; guard < 64 && !G::terminal(p, s) && G::current_player(p, s) == kChancePlayer; ++guard) {
    const auto m = G::legal(p, s);
    G::apply(p, s, sample_action<G>(p, s, m, kChancePlayer, rng));
  }
  G::store(p, base, n, i, s);
  const bool term = G::terminal(p, s);
  if (term && type != 0) type = 2;
  should_reset[i] = type == 2 ? 1 : 0;
  cur_player[i] = static_cast<int8_t>(G::current_player(p, s));
  step_type[i] = static_cast<uint8_t>(type);
  double r[kMaxPlayers];
  G::returns(p, s, r);
  for (int q = 0; q < num_players; ++q) rewards[i * num_players + q] = type == 2 ? r[q] : 0.0;
  const auto after = G::legal(p, s);
#pragma unroll
  for (int w = 0; w < G::kMaskW; ++w)
    if (w < mask_words) mask[i * mask_words + w] = after.w[w];
}

// The same step for games of two 64-bit planes and two players (connect_four up to 64 board bits, leduc_poker with 2
// players), TWO consecutive environments per thread (round 5): every plane access is one 16-byte access per lane, the
// two reward rows are 32 contiguous bytes (two 16-byte stores), actions and mask words 8 bytes, the byte arrays 2 —
// the one-environment form moved 8 + 8 + 4 + 1 bytes in and eleven 1- to 8-byte pieces out per lane (4.7 TB/s
// cache-resident against 7.2 for the step kernel).  Same arithmetic, same outputs: tests/test_gpu_vector_env.py compares
// the two forms (odd batch sizes and unaligned side arrays keep the one-environment form).
template <class G>
__global__ void __launch_bounds__(kBlock)
k_env_step_x2(typename G::Params p, uint64_t* base, int64_t n, const int32_t* __restrict__ actions,
              uint8_t* should_reset, uint64_t seed, int64_t index_offset, int64_t step_index,
              int8_t* __restrict__ cur_player, uint8_t* __restrict__ step_type, double* __restrict__ rewards,
              uint32_t* __restrict__ mask, unsigned long long* illegal) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 2;
  if (i >= n) return;
  const ulonglong2 w0 = *reinterpret_cast<const ulonglong2*>(base + i);
  const ulonglong2 w1 = *reinterpret_cast<const ulonglong2*>(base + n + i);
  const int2 a2 = *reinterpret_cast<const int2*>(actions + i);
  const uchar2 r2 = *reinterpret_cast<const uchar2*>(should_reset + i);
  uint64_t tmp[4] = {w0.x, w0.y, w1.x, w1.y};   // plane-major mini-batch of two: G::load(p, tmp, 2, j) reads tmp[w * 2 + j]
  uint32_t reset_out[2], cur_out[2], type_out[2], mask_out[2];
  double rew[4];
  int bad = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    typename G::State s = G::load(p, tmp, 2, j);
    int type = 1;
    if (j == 0 ? r2.x : r2.y) {
      s = G::initial(p);
      type = 0;
    } else {
      const int a = j == 0 ? a2.x : a2.y;
      if (a != OSG_INVALID_ACTION) {
        const auto m = G::legal(p, s);
        if (a < 0 || a >= 32 * G::kMaskW || !m.test(a)) ++bad;
        else G::apply(p, s, a);
      }
    }
    Rng rng(seed, static_cast<uint64_t>(index_offset + i + j), static_cast<uint64_t>(step_index));
    for (int guard = 0; guard < 64 && !G::terminal(p, s) && G::current_player(p, s) == kChancePlayer; ++guard) {
      const auto m = G::legal(p, s);
      G::apply(p, s, sample_action<G>(p, s, m, kChancePlayer, rng));
    }
    G::store(p, tmp, 2, j, s);
    const bool term = G::terminal(p, s);
    if (term && type != 0) type = 2;
    reset_out[j] = type == 2 ? 1u : 0u;
    cur_out[j] = static_cast<uint32_t>(static_cast<uint8_t>(static_cast<int8_t>(G::current_player(p, s))));
    type_out[j] = static_cast<uint32_t>(type);
    double r[kMaxPlayers];
    G::returns(p, s, r);
    rew[2 * j] = type == 2 ? r[0] : 0.0;
    rew[2 * j + 1] = type == 2 ? r[1] : 0.0;
    mask_out[j] = G::legal(p, s).w[0];
  }
  if (bad) atomicAdd(illegal, static_cast<unsigned long long>(bad));
  *reinterpret_cast<ulonglong2*>(base + i) = make_ulonglong2(tmp[0], tmp[1]);
  *reinterpret_cast<ulonglong2*>(base + n + i) = make_ulonglong2(tmp[2], tmp[3]);
  *reinterpret_cast<uint16_t*>(should_reset + i) = static_cast<uint16_t>(reset_out[0] | (reset_out[1] << 8));
  *reinterpret_cast<uint16_t*>(cur_player + i) = static_cast<uint16_t>(cur_out[0] | (cur_out[1] << 8));
  *reinterpret_cast<uint16_t*>(step_type + i) = static_cast<uint16_t>(type_out[0] | (type_out[1] << 8));
  *reinterpret_cast<double2*>(rewards + 2 * i) = make_double2(rew[0], rew[1]);
  *reinterpret_cast<double2*>(rewards + 2 * i + 2) = make_double2(rew[2], rew[3]);
  *reinterpret_cast<uint2*>(mask + i) = make_uint2(mask_out[0], mask_out[1]);
}

// The environment step with COMPACT side arrays (round 6; osg_env_step_compact): the reference's TimeStep types cost the
// step above 20 of its 60 bytes per connect_four environment (int32 actions, float64 rewards, three flag bytes); here an
// action is one byte (0xFF: leave the environment as it is), the three flag bytes are ONE in/out byte — bits 0-1 the step
// type (LAST on input = "restart": what should_reset carried), bits 2-7 the current player + 4 — and a reward is one
// signed byte holding TWICE the return (every game here pays multiples of 0.5; games whose returns do not fit are
// refused by the entry point).  41 bytes per connect_four environment.  Same rules, same counter streams, same order of
// operations as k_env_step: tests/test_gpu_vector_env.py steps the two forms side by side.
OSG_D uint32_t env_flag_byte(int type, int cur) { return static_cast<uint32_t>(type) | (static_cast<uint32_t>(cur + 4) << 2); }
template <class G>
OSG_D int env_step_one(const typename G::Params& p, typename G::State& s, bool restart, int a /* -1: leave */, uint64_t seed,
                       uint64_t index, uint64_t step_index, int* bad) {
  int type = 1;
  if (restart) {
    s = G::initial(p);
    type = 0;
  } else if (a != OSG_INVALID_ACTION) {
    const auto m = G::legal(p, s);
    if (a < 0 || a >= 32 * G::kMaskW || !m.test(a)) ++*bad;
    else G::apply(p, s, a);
  }
  Rng rng(seed, index, step_index);
  for (int guard = 0; guard < 64 && !G::terminal(p, s) && G::current_player(p, s) == kChancePlayer; ++guard) {
    const auto m = G::legal(p, s);
    G::apply(p, s, sample_action<G>(p, s, m, kChancePlayer, rng));
  }
  if (G::terminal(p, s) && type != 0) type = 2;
  return type;
}
template <class G>
__global__ void __launch_bounds__(kBlock)
k_env_step_compact(typename G::Params p, typename G::word_t* base, int64_t n, int num_players, const uint8_t* __restrict__ actions,
                   uint8_t* flags, uint64_t seed, int64_t index_offset, int64_t step_index, int8_t* __restrict__ rewards_x2,
                   uint32_t* __restrict__ mask, int mask_words, unsigned long long* illegal) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  typename G::State s = G::load(p, base, n, i);
  const int a8 = actions[i];
  int bad = 0;
  const int type = env_step_one<G>(p, s, (flags[i] & 3u) == 2u, a8 == 0xFF ? OSG_INVALID_ACTION : a8, seed,
                                   static_cast<uint64_t>(index_offset + i), static_cast<uint64_t>(step_index), &bad);
  if (bad) atomicAdd(illegal, 1ull);
  G::store(p, base, n, i, s);
  flags[i] = static_cast<uint8_t>(env_flag_byte(type, G::current_player(p, s)));
  double r[kMaxPlayers];
  G::returns(p, s, r);
  for (int q = 0; q < num_players; ++q) rewards_x2[i * num_players + q] = type == 2 ? static_cast<int8_t>(2.0 * r[q]) : static_cast<int8_t>(0);
  const auto after = G::legal(p, s);
#pragma unroll
  for (int w = 0; w < G::kMaskW; ++w)
    if (w < mask_words) mask[i * mask_words + w] = after.w[w];
}
// Two consecutive environments per thread for the two-plane two-player games (as k_env_step_x2): 16-byte plane accesses,
// the two action bytes / flag bytes as one 16-bit access, the two reward rows as one 32-bit store, the two mask words 8 bytes.
template <class G>
__global__ void __launch_bounds__(kBlock)
k_env_step_compact_x2(typename G::Params p, uint64_t* base, int64_t n, const uint8_t* __restrict__ actions, uint8_t* flags,
                      uint64_t seed, int64_t index_offset, int64_t step_index, int8_t* __restrict__ rewards_x2,
                      uint32_t* __restrict__ mask, unsigned long long* illegal) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 2;
  if (i >= n) return;
  const ulonglong2 w0 = *reinterpret_cast<const ulonglong2*>(base + i);
  const ulonglong2 w1 = *reinterpret_cast<const ulonglong2*>(base + n + i);
  const uint32_t a2 = *reinterpret_cast<const uint16_t*>(actions + i);
  const uint32_t f2 = *reinterpret_cast<const uint16_t*>(flags + i);
  uint64_t tmp[4] = {w0.x, w0.y, w1.x, w1.y};   // plane-major mini-batch of two: G::load(p, tmp, 2, j) reads tmp[w * 2 + j]
  uint32_t flag_out = 0, rew_out = 0, mask_out[2];
  int bad = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    typename G::State s = G::load(p, tmp, 2, j);
    const int a8 = static_cast<int>((a2 >> (8 * j)) & 0xFFu);
    const int type = env_step_one<G>(p, s, ((f2 >> (8 * j)) & 3u) == 2u, a8 == 0xFF ? OSG_INVALID_ACTION : a8, seed,
                                     static_cast<uint64_t>(index_offset + i + j), static_cast<uint64_t>(step_index), &bad);
    G::store(p, tmp, 2, j, s);
    flag_out |= env_flag_byte(type, G::current_player(p, s)) << (8 * j);
    double r[kMaxPlayers];
    G::returns(p, s, r);
    const uint32_t r0 = type == 2 ? static_cast<uint32_t>(static_cast<uint8_t>(static_cast<int8_t>(2.0 * r[0]))) : 0u;
    const uint32_t r1 = type == 2 ? static_cast<uint32_t>(static_cast<uint8_t>(static_cast<int8_t>(2.0 * r[1]))) : 0u;
    rew_out |= (r0 | (r1 << 8)) << (16 * j);
    mask_out[j] = G::legal(p, s).w[0];
  }
  if (bad) atomicAdd(illegal, static_cast<unsigned long long>(bad));
  *reinterpret_cast<ulonglong2*>(base + i) = make_ulonglong2(tmp[0], tmp[1]);
  *reinterpret_cast<ulonglong2*>(base + n + i) = make_ulonglong2(tmp[2], tmp[3]);
  *reinterpret_cast<uint16_t*>(flags + i) = static_cast<uint16_t>(flag_out);
  *reinterpret_cast<uint32_t*>(rewards_x2 + 2 * i) = rew_out;
  *reinterpret_cast<uint2*>(mask + i) = make_uint2(mask_out[0], mask_out[1]);
}

// RandomRolloutEvaluator::Evaluate (mcts.cc:43-72), persistent form: every lane owns a strided list of
// work items and runs ONE flat loop whose body is "step the playout, or retire it and start the next", so
// lanes in different phases of different playouts still execute the same instructions.  A work item is
// (root, share j of `group`): the lane plays rollouts j, j + group, j + 2 group, ... of that root back to
// back, adds their returns up in registers and stores the sums into its own slot [root, j]; k_rollout_fold
// then adds the `group` slots of every root in order.  No atomics: the L2 retires only ~2e10 atomics/s
// chip-wide, and one per playout and player was the whole run time of the short games.  Rollout r of
// root i always plays from the counter stream (seed, i, r), whatever the split.
template <class G>
__global__ void __launch_bounds__(kBlock)
k_rollout(typename G::Params p, const typename G::word_t* base, int64_t n, int num_players, uint64_t seed,
          int64_t index_offset, int n_rollouts, int group, double* sum_returns, int32_t* steps_out) {
  const int64_t total = n * group;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  int64_t item = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (item >= total) return;
  int64_t root = item / group;
  int r = static_cast<int>(item - root * group);  // current rollout of this share
  typename G::State s = G::load(p, base, n, root);
  Rng rng(seed, static_cast<uint64_t>(index_offset + root), static_cast<uint64_t>(r));
  double acc[kMaxPlayers];
#pragma unroll
  for (int q = 0; q < kMaxPlayers; ++q) acc[q] = 0.0;
  int plies = 0, ply = 0;  // moves of this share so far / of the running playout
  for (;;) {
    if (G::terminal(p, s) || ply >= kMaxPlayoutPlies) {
      double ret[kMaxPlayers];
      G::returns(p, s, ret);
#pragma unroll
      for (int q = 0; q < kMaxPlayers; ++q)
        if (q < num_players) acc[q] += ret[q];  // small multiples of 0.5: exact in any order
      r += group;
      if (r >= n_rollouts) {  // this share is done: hand in its sums, fetch the next item
#pragma unroll
        for (int q = 0; q < kMaxPlayers; ++q) {
          if (q < num_players) sum_returns[item * num_players + q] = acc[q];  // slot of (root, share)
          acc[q] = 0.0;
        }
        if (steps_out) steps_out[item] = plies;
        plies = 0;
        item += stride;
        if (item >= total) break;
        root = item / group;
        r = static_cast<int>(item - root * group);
      }
      s = G::load(p, base, n, root);
      rng = Rng(seed, static_cast<uint64_t>(index_offset + root), static_cast<uint64_t>(r));
      ply = 0;
      continue;
    }
    auto m = G::legal(p, s);
    int a = sample_action<G>(p, s, m, G::current_player(p, s), rng);
    G::apply(p, s, a);
    ++plies;
    ++ply;
  }
}

// The same work items for hex when nobody asks for the ply counts (round 6): a playout is HexT::fill_playout_winner —
// the stones placed with the same draws until the board is full, the winner read off by one flood — so every playout of
// a root has the same length and the loop needs no retire / refill phase.  Same sums as k_rollout.
#ifndef OSG_HEX_FILL_PLAYOUT
#define OSG_HEX_FILL_PLAYOUT 1
#endif
template <class G>
__global__ void __launch_bounds__(kBlock)
k_rollout_hexfill(typename G::Params p, const typename G::word_t* base, int64_t n, uint64_t seed, int64_t index_offset,
                  int n_rollouts, int group, double* sum_returns) {
  const int64_t total = n * group;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t item = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; item < total; item += stride) {
    const int64_t root = item / group;
    const typename G::State s = G::load(p, base, n, root);
    double acc = 0.0;
    for (int r = static_cast<int>(item - root * group); r < n_rollouts; r += group) {
      if (G::terminal(p, s)) {   // a finished root: Returns() as it stands
        acc += G::result(s) == 1 ? 1.0 : -1.0;
        continue;
      }
      Rng rng(seed, static_cast<uint64_t>(index_offset + root), static_cast<uint64_t>(r));
      acc += G::fill_playout_winner(p, s, rng) == 0 ? 1.0 : -1.0;
    }
    sum_returns[item * 2] = acc;
    sum_returns[item * 2 + 1] = -acc + 0.0;
  }
}

template <class G>   // (a template so that the discarded branch is not instantiated for the other games)
void launch_rollout_hexfill(const typename G::Params& p, const void* words, int64_t n, uint64_t seed, int64_t index_offset,
                            int n_rollouts, int group, double* d_part, unsigned blocks, hipStream_t st) {
  if constexpr (is_hex<G>::value)
    k_rollout_hexfill<G><<<dim3(blocks), dim3(kBlock), 0, st>>>(p, static_cast<const typename G::word_t*>(words), n, seed,
                                                              index_offset, n_rollouts, group, d_part);
}

// Sums the `group` share slots of every root: sum_returns [n, P] and, optionally, the ply counts [n].
__global__ void __launch_bounds__(kBlock)
k_rollout_fold(const double* __restrict__ part, const int32_t* __restrict__ part_steps, int64_t n, int num_players,
               int group, double* __restrict__ sum_returns, int32_t* __restrict__ steps_out) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;  // (root, player)
  if (k >= n * num_players) return;
  const int64_t root = k / num_players;
  const int q = static_cast<int>(k - root * num_players);
  double v = 0.0;
  for (int j = 0; j < group; ++j) v += part[(root * group + j) * num_players + q];
  sum_returns[k] = v;
  if (steps_out && q == 0) {
    int32_t t = 0;
    for (int j = 0; j < group; ++j) t += part_steps[root * group + j];
    steps_out[root] = t;
  }
}

// ---------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------
struct Staged {  // an argument that may live on the host: staged through ctx scratch
  void* dev = nullptr;
};

int stage_in(osg_ctx* ctx, const void* ptr, size_t bytes, int on_host, size_t scratch_offset, const void** dev) {
  if (!on_host) { *dev = ptr; return OSG_OK; }
  void* scratch = nullptr;
  int rc = osg_ctx_scratch(ctx, scratch_offset + bytes, &scratch);
  if (rc) return rc;
  void* d = static_cast<char*>(scratch) + scratch_offset;
  OSG_HIP(hipMemcpyAsync(d, ptr, bytes, hipMemcpyHostToDevice, ctx->stream));
  *dev = d;
  return OSG_OK;
}

size_t align_up(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

int check_illegal(osg_ctx* ctx, int64_t* h_illegal) {
  unsigned long long count = 0;
  OSG_HIP(hipMemcpyAsync(&count, ctx->d_illegal, sizeof(count), hipMemcpyDeviceToHost, ctx->stream));
  OSG_HIP(hipStreamSynchronize(ctx->stream));
  if (count) OSG_HIP(hipMemsetAsync(ctx->d_illegal, 0, sizeof(count), ctx->stream));
  if (h_illegal) { *h_illegal = static_cast<int64_t>(count); return OSG_OK; }
  if (count) return set_error(OSG_ERR_ILLEGAL, std::to_string(count) + " illegal action(s) applied (or out-of-range gather indices)");
  return OSG_OK;
}

}  // namespace

int osg_ctx_scratch(osg_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->scratch_bytes) {
    // Grow-only; make sure no queued kernel still reads the old block.
    OSG_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_scratch) OSG_HIP(hipFree(ctx->d_scratch));
    ctx->d_scratch = nullptr;
    ctx->scratch_bytes = 0;
    size_t want = bytes + bytes / 2;
    OSG_HIP(hipMalloc(&ctx->d_scratch, want));
    ctx->scratch_bytes = want;
  }
  *out = ctx->d_scratch;
  return OSG_OK;
}

// ---------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------
extern "C" {

int osg_ctx_create(int device, void* stream, int own_stream, osg_ctx** out) {
  if (!out) return set_error(OSG_ERR_INVALID, "null out");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return set_error(OSG_ERR_HIP, "no HIP device visible: the MI355X path has no CPU fallback");
  if (device < 0 || device >= count) return set_error(OSG_ERR_INVALID, "bad device index");
  OSG_HIP(hipSetDevice(device));
  osg_ctx* ctx = new osg_ctx;
  ctx->device = device;
  if (!own_stream) {
    ctx->stream = static_cast<hipStream_t>(stream);
  } else {
    OSG_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  // [0] illegal-apply counter, [1 ...] the partial counter slots of k_random_steps
  OSG_HIP(hipMalloc(&ctx->d_illegal, sizeof(unsigned long long) * (1 + 2 * kCounterSlots)));
  OSG_HIP(hipMemsetAsync(ctx->d_illegal, 0, sizeof(unsigned long long) * (1 + 2 * kCounterSlots), ctx->stream));
  *out = ctx;
  return OSG_OK;
}

}  // extern "C"
namespace osg {
void ctx_retain(osg_ctx* ctx) { __atomic_add_fetch(&ctx->refs, 1, __ATOMIC_RELAXED); }
void ctx_release(osg_ctx* ctx) {
  if (__atomic_sub_fetch(&ctx->refs, 1, __ATOMIC_ACQ_REL) != 0) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_illegal) (void)hipFree(ctx->d_illegal);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  if (ctx->d_mcts_pool) (void)hipFree(ctx->d_mcts_pool);
  if (ctx->d_mcts_logs) (void)hipFree(ctx->d_mcts_logs);
  if (ctx->d_mcts_queue) (void)hipFree(ctx->d_mcts_queue);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}
}  // namespace osg
// One thread builds the position from its cells with the game's own rules and stores it in the batch's layout.
template <class G>
__global__ void k_set_cells(typename G::Params P, typename G::word_t* words, int64_t n, int64_t index,
                            const unsigned char* cells, int n_cells, int* err) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  typename G::State s;
  const int e = G::from_cells(P, cells, n_cells, s);
  *err = e;
  if (e == 0) G::store(P, words, n, index, s);
}

extern "C" {

int osg_ctx_destroy(osg_ctx* ctx) {
  if (!ctx) return OSG_OK;
  if (ctx->closed) return set_error(OSG_ERR_INVALID, "osg_ctx_destroy: context already destroyed");
  ctx->closed = true;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  osg::ctx_release(ctx);
  return OSG_OK;
}

int osg_ctx_synchronize(osg_ctx* ctx) {
  OSG_HIP(hipStreamSynchronize(ctx->stream));
  return check_illegal(ctx, nullptr);
}
void* osg_ctx_stream(osg_ctx* ctx) { return ctx->stream; }

int osg_ctx_set_stream(osg_ctx* ctx, void* stream) {
  if (!ctx || ctx->closed) return set_error(OSG_ERR_INVALID, "osg_ctx_set_stream: bad context");
  if (ctx->own_stream) return set_error(OSG_ERR_INVALID, "osg_ctx_set_stream: the context owns its stream");
  ctx->stream = static_cast<hipStream_t>(stream);
  return OSG_OK;
}

int osg_ctx_trim(osg_ctx* ctx) {
  if (!ctx || ctx->closed) return set_error(OSG_ERR_INVALID, "osg_ctx_trim: bad context");
  OSG_HIP(hipSetDevice(ctx->device));
  OSG_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->d_mcts_pool) OSG_HIP(hipFree(ctx->d_mcts_pool));
  ctx->d_mcts_pool = nullptr;
  ctx->mcts_pool_bytes = 0;
  if (ctx->d_scratch) OSG_HIP(hipFree(ctx->d_scratch));
  ctx->d_scratch = nullptr;
  ctx->scratch_bytes = 0;
  if (ctx->d_mcts_queue) OSG_HIP(hipFree(ctx->d_mcts_queue));
  ctx->d_mcts_queue = nullptr;
  ctx->mcts_queue_roots = 0;
  return OSG_OK;
}

int osg_batch_create(osg_ctx* ctx, const char* game_string, int64_t n, osg_batch** out) {
  if (!ctx || !out || n <= 0) return set_error(OSG_ERR_INVALID, "osg_batch_create: bad argument");
  if (ctx->closed) return set_error(OSG_ERR_INVALID, "osg_batch_create: the context was destroyed");
  osg_batch* b = new osg_batch;
  int rc = parse_game(game_string, &b->spec);
  if (rc) { delete b; return rc; }
  b->ctx = ctx;
  b->n = n;
  b->bytes = static_cast<size_t>(n) * b->spec.desc.state_words * b->spec.desc.state_word_bytes;
  hipError_t e = hipMalloc(&b->d_words, b->bytes);
  if (e != hipSuccess) { delete b; return set_error(OSG_ERR_NOMEM, hipGetErrorString(e)); }
  rc = osg_batch_reset(b);
  if (rc) { hipFree(b->d_words); delete b; return rc; }
  osg::ctx_retain(ctx);
  *out = b;
  return OSG_OK;
}

int osg_batch_destroy(osg_batch* b) {
  if (!b) return OSG_OK;
  (void)hipStreamSynchronize(b->ctx->stream);
  (void)hipFree(b->d_words);
  osg::ctx_release(b->ctx);
  delete b;
  return OSG_OK;
}
int64_t osg_batch_size(const osg_batch* b) { return b->n; }
int osg_batch_describe(const osg_batch* b, osg_game_desc* out) { *out = b->spec.desc; return OSG_OK; }
void* osg_batch_device_ptr(osg_batch* b) { return b->d_words; }

int osg_batch_reset(osg_batch* b) {
  OSG_DISPATCH_WIDE(b->spec, k_init<G><<<dim3(grid_for(b->n)), dim3(kBlock), 0, b->ctx->stream>>>(P,
                                            static_cast<typename G::word_t*>(b->d_words), b->n));
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

static bool same_game(const osg_batch* a, const osg_batch* b) {
  return strcmp(a->spec.desc.canonical, b->spec.desc.canonical) == 0 &&
         a->spec.desc.state_words == b->spec.desc.state_words;
}

int osg_batch_copy(osg_batch* dst, const osg_batch* src) {
  if (!same_game(dst, src) || dst->n != src->n) return set_error(OSG_ERR_INVALID, "osg_batch_copy: shape mismatch");
  OSG_HIP(hipMemcpyAsync(dst->d_words, src->d_words, src->bytes, hipMemcpyDeviceToDevice, dst->ctx->stream));
  return OSG_OK;
}

int osg_batch_gather(osg_batch* dst, const osg_batch* src, const int64_t* index, int on_host) {
  if (!same_game(dst, src)) return set_error(OSG_ERR_INVALID, "osg_batch_gather: different games");
  if (!index) return set_error(OSG_ERR_INVALID, "osg_batch_gather: null index");
  if (on_host)
    for (int64_t i = 0; i < dst->n; ++i)
      if (index[i] < 0 || index[i] >= src->n) return set_error(OSG_ERR_INVALID, "osg_batch_gather: index out of range");
  const void* d_index = nullptr;
  int rc = stage_in(dst->ctx, index, sizeof(int64_t) * dst->n, on_host, 0, &d_index);
  if (rc) return rc;
  OSG_DISPATCH_WIDE(dst->spec, k_gather<G><<<dim3(grid_for(dst->n)), dim3(kBlock), 0, dst->ctx->stream>>>(P,
                                              static_cast<typename G::word_t*>(dst->d_words), dst->n,
                                              static_cast<const typename G::word_t*>(src->d_words), src->n,
                                              static_cast<const int64_t*>(d_index), dst->ctx->d_illegal));
  OSG_HIP(hipGetLastError());
  if (on_host) OSG_HIP(hipStreamSynchronize(dst->ctx->stream));
  return OSG_OK;
}

int osg_copy_bytes(osg_ctx* ctx, void* d_dst, const void* d_src, int64_t bytes) {
  if (!ctx || !d_dst || !d_src || bytes < 0 || (bytes & 15) || (reinterpret_cast<uintptr_t>(d_dst) & 15) ||
      (reinterpret_cast<uintptr_t>(d_src) & 15))
    return set_error(OSG_ERR_INVALID, "osg_copy_bytes: null / unaligned argument");
  const int64_t n16 = bytes / 16;
  if (n16 == 0) return OSG_OK;
  k_copy16<<<dim3(static_cast<unsigned>((n16 + 255) / 256)), dim3(256), 0, ctx->stream>>>(
      static_cast<const uint4*>(d_src), static_cast<uint4*>(d_dst), n16);
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int osg_batch_download(const osg_batch* b, void* h_words) {
  OSG_HIP(hipMemcpyAsync(h_words, b->d_words, b->bytes, hipMemcpyDeviceToHost, b->ctx->stream));
  OSG_HIP(hipStreamSynchronize(b->ctx->stream));
  return OSG_OK;
}
int osg_batch_upload(osg_batch* b, const void* h_words) {
  OSG_HIP(hipMemcpyAsync(b->d_words, h_words, b->bytes, hipMemcpyHostToDevice, b->ctx->stream));
  OSG_HIP(hipStreamSynchronize(b->ctx->stream));
  return OSG_OK;
}

int osg_batch_set_cells(osg_batch* b, int64_t index, const char* cells, int n_cells) {
  if (!b || !cells || index < 0 || index >= b->n || n_cells <= 0 || n_cells > 4096)
    return osg::set_error(OSG_ERR_INVALID, "osg_batch_set_cells: bad argument");
  osg_ctx* ctx = b->ctx;
  void* scratch;
  if (int rc = osg_ctx_scratch(ctx, 4096 + sizeof(int), &scratch)) return rc;
  unsigned char* d_cells = static_cast<unsigned char*>(scratch);
  int* d_err = reinterpret_cast<int*>(d_cells + 4096);
  OSG_HIP(hipMemcpyAsync(d_cells, cells, static_cast<size_t>(n_cells), hipMemcpyHostToDevice, ctx->stream));
  const osg::GameSpec& spec = b->spec;
  switch (spec.desc.game_kind) {
    case osg::kTtt:
      k_set_cells<osg::Ttt><<<dim3(1), dim3(64), 0, ctx->stream>>>(spec.ttt, static_cast<osg::Ttt::word_t*>(b->d_words), b->n, index,
                                                                   d_cells, n_cells, d_err);
      break;
    case osg::kC4:
      if (spec.c4_std)
        k_set_cells<osg::C4Std><<<dim3(1), dim3(64), 0, ctx->stream>>>(spec.c4, static_cast<osg::C4Std::word_t*>(b->d_words), b->n, index,
                                                                       d_cells, n_cells, d_err);
      else if (spec.c4_wide)
        k_set_cells<osg::C4Wide><<<dim3(1), dim3(64), 0, ctx->stream>>>(spec.c4, static_cast<osg::C4Wide::word_t*>(b->d_words), b->n, index,
                                                                        d_cells, n_cells, d_err);
      else
        k_set_cells<osg::C4><<<dim3(1), dim3(64), 0, ctx->stream>>>(spec.c4, static_cast<osg::C4::word_t*>(b->d_words), b->n, index,
                                                                    d_cells, n_cells, d_err);
      break;
    default:
      return osg::set_error(OSG_ERR_UNSUPPORTED, "osg_batch_set_cells: tic_tac_toe and connect_four positions (the games whose "
                                                 "reference State has a constructor from a board)");
  }
  OSG_HIP(hipGetLastError());
  int err = 0;
  OSG_HIP(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  OSG_HIP(hipStreamSynchronize(ctx->stream));
  switch (err) {
    case 0: return OSG_OK;
    case 1: return osg::set_error(OSG_ERR_INVALID, "osg_batch_set_cells: the board does not have the game's number of cells");
    case 2: return osg::set_error(OSG_ERR_INVALID, "osg_batch_set_cells: a cell is not one of '.', 'x', 'o'");
    case 3: return osg::set_error(OSG_ERR_INVALID, "Invalid board: gap in a column. Pieces must be stacked from the bottom with no gaps.");
    default: return osg::set_error(OSG_ERR_INVALID, "Invalid board state: both players have a winning line.");
  }
}

int osg_legal_mask(const osg_batch* b, uint32_t* mask, int on_host) {
  osg_ctx* ctx = b->ctx;
  const int W = b->spec.desc.mask_words;
  size_t bytes = sizeof(uint32_t) * W * b->n;
  uint32_t* d_mask = mask;
  if (on_host) {
    void* scratch;
    int rc = osg_ctx_scratch(ctx, bytes, &scratch);
    if (rc) return rc;
    d_mask = static_cast<uint32_t*>(scratch);
  }
  OSG_DISPATCH_WIDE(b->spec, k_legal_mask<G><<<dim3(grid_for(b->n)), dim3(kBlock), 0, ctx->stream>>>(P,
                                            static_cast<const typename G::word_t*>(b->d_words), b->n, d_mask, W));
  OSG_HIP(hipGetLastError());
  if (on_host) {
    OSG_HIP(hipMemcpyAsync(mask, d_mask, bytes, hipMemcpyDeviceToHost, ctx->stream));
    OSG_HIP(hipStreamSynchronize(ctx->stream));
  }
  return OSG_OK;
}

int osg_apply(osg_batch* b, const int32_t* actions, int on_host, int64_t* h_illegal) {
  osg_ctx* ctx = b->ctx;
  const void* d_actions = nullptr;
  int rc = stage_in(ctx, actions, sizeof(int32_t) * b->n, on_host, 0, &d_actions);
  if (rc) return rc;
  OSG_DISPATCH_WIDE(b->spec, k_apply<G><<<dim3(grid_for(b->n)), dim3(kBlock), 0, ctx->stream>>>(P,
                                            static_cast<typename G::word_t*>(b->d_words), b->n,
                                            static_cast<const int32_t*>(d_actions), ctx->d_illegal));
  OSG_HIP(hipGetLastError());
  if (on_host || h_illegal) return check_illegal(ctx, h_illegal);
  return OSG_OK;
}

int osg_status_query(const osg_batch* b, int8_t* cur_player, uint8_t* terminal, double* returns, int on_host) {
  osg_ctx* ctx = b->ctx;
  const int P_ = b->spec.desc.num_players;
  int8_t* d_cur = cur_player;
  uint8_t* d_term = terminal;
  double* d_ret = returns;
  size_t off_term = align_up(b->n), off_ret = off_term + align_up(b->n);
  if (on_host) {
    void* scratch;
    int rc = osg_ctx_scratch(ctx, off_ret + sizeof(double) * P_ * b->n, &scratch);
    if (rc) return rc;
    char* sc = static_cast<char*>(scratch);
    d_cur = cur_player ? reinterpret_cast<int8_t*>(sc) : nullptr;
    d_term = terminal ? reinterpret_cast<uint8_t*>(sc + off_term) : nullptr;
    d_ret = returns ? reinterpret_cast<double*>(sc + off_ret) : nullptr;
  }
  OSG_DISPATCH_WIDE(b->spec, k_status<G><<<dim3(grid_for(b->n)), dim3(kBlock), 0, ctx->stream>>>(P,
                                            static_cast<const typename G::word_t*>(b->d_words), b->n, P_, d_cur,
                                            d_term, d_ret));
  OSG_HIP(hipGetLastError());
  if (on_host) {
    if (cur_player) OSG_HIP(hipMemcpyAsync(cur_player, d_cur, b->n, hipMemcpyDeviceToHost, ctx->stream));
    if (terminal) OSG_HIP(hipMemcpyAsync(terminal, d_term, b->n, hipMemcpyDeviceToHost, ctx->stream));
    if (returns) OSG_HIP(hipMemcpyAsync(returns, d_ret, sizeof(double) * P_ * b->n, hipMemcpyDeviceToHost, ctx->stream));
    OSG_HIP(hipStreamSynchronize(ctx->stream));
  }
  return OSG_OK;
}

int osg_chance_probs(const osg_batch* b, double* probs, int on_host) {
  osg_ctx* ctx = b->ctx;
  const int C = b->spec.desc.max_chance_outcomes;
  if (C == 0) return OSG_OK;
  size_t bytes = sizeof(double) * C * b->n;
  double* d_probs = probs;
  if (on_host) {
    void* scratch;
    int rc = osg_ctx_scratch(ctx, bytes, &scratch);
    if (rc) return rc;
    d_probs = static_cast<double*>(scratch);
  }
  OSG_DISPATCH_WIDE(b->spec, k_chance_probs<G><<<dim3(grid_for(b->n)), dim3(kBlock), 0, ctx->stream>>>(P,
                                            static_cast<const typename G::word_t*>(b->d_words), b->n, C, d_probs));
  OSG_HIP(hipGetLastError());
  if (on_host) {
    OSG_HIP(hipMemcpyAsync(probs, d_probs, bytes, hipMemcpyDeviceToHost, ctx->stream));
    OSG_HIP(hipStreamSynchronize(ctx->stream));
  }
  return OSG_OK;
}

int osg_step(const osg_batch* src, osg_batch* dst, const uint8_t* d_actions, void* d_mask, uint8_t* d_status) {
  if (!same_game(dst, src) || dst->n != src->n) return set_error(OSG_ERR_INVALID, "osg_step: shape mismatch");
  if (src->spec.desc.num_distinct_actions > 255)
    return set_error(OSG_ERR_UNSUPPORTED, "osg_step: action ids travel as one byte here (0xFF = skip); games with more than 255 "
                                          "actions (hex above 15 x 15) step through osg_apply / osg_env_step (32-bit actions)");
  osg_ctx* ctx = dst->ctx;
  const int cmb = src->spec.desc.compact_mask_bytes;
  const int W = src->spec.desc.mask_words;
  const int64_t n = src->n;
  // d_mask == NULL ("do not write the successor's mask"): hex only, where the mask is ~occupied of the successor record
  if (!d_mask && !(src->spec.desc.game_kind == kHex && cmb == 4 * W && W == src->spec.hex_nw && W <= 4))
    return set_error(OSG_ERR_UNSUPPORTED, "osg_step: d_mask may be NULL only for hex boards of up to 128 cells (there the successor's "
                                          "mask is ~occupied of the record written); every other game's mask comes from the step itself");
  // the kernels that move several states per lane use 16-byte plane accesses: planes start 16-byte aligned when the
  // allocation does (hipMalloc: 256 B) and n x word size is a multiple of 16 — checked here, not assumed
  const bool planes16 = ((reinterpret_cast<uintptr_t>(src->d_words) | reinterpret_cast<uintptr_t>(dst->d_words)) & 15u) == 0;
  if (planes16 && src->spec.desc.game_kind == kC4 && src->spec.c4_std && (n & 1) == 0 &&
      ((reinterpret_cast<uintptr_t>(d_actions) | reinterpret_cast<uintptr_t>(d_mask) | reinterpret_cast<uintptr_t>(d_status)) & 1u) == 0) {
    k_step_c4std2<<<dim3(static_cast<unsigned>((n / 2 + kC4StepBlock - 1) / kC4StepBlock)), dim3(kC4StepBlock), 0, ctx->stream>>>(
        static_cast<const uint64_t*>(src->d_words), static_cast<uint64_t*>(dst->d_words), n, d_actions,
        static_cast<uint8_t*>(d_mask), d_status);
    OSG_HIP(hipGetLastError());
    return OSG_OK;
  }
  if (src->spec.desc.game_kind == kC4 && src->spec.c4_std) {  // odd batches / unaligned side arrays: one state per thread
    k_step_c4std<<<dim3(static_cast<unsigned>((n + kC4StepBlock - 1) / kC4StepBlock)), dim3(kC4StepBlock), 0, ctx->stream>>>(
        static_cast<const uint64_t*>(src->d_words), static_cast<uint64_t*>(dst->d_words), n, d_actions,
        static_cast<uint8_t*>(d_mask), d_status);
    OSG_HIP(hipGetLastError());
    return OSG_OK;
  }
  const bool aligned2 = ((reinterpret_cast<uintptr_t>(d_actions) | reinterpret_cast<uintptr_t>(d_mask) |
                          reinterpret_cast<uintptr_t>(d_status)) & 1u) == 0;
  if (planes16 && src->spec.desc.game_kind == kC4 && !src->spec.c4_wide && (n & 1) == 0 && aligned2) {
    const int64_t pairs = n / 2;
    k_step_c4x2<C4><<<dim3(static_cast<unsigned>((pairs + kC4StepBlock - 1) / kC4StepBlock)), dim3(kC4StepBlock), 0, ctx->stream>>>(
        src->spec.c4, static_cast<const uint64_t*>(src->d_words), static_cast<uint64_t*>(dst->d_words), n, d_actions,
        static_cast<uint8_t*>(d_mask), d_status);
    OSG_HIP(hipGetLastError());
    return OSG_OK;
  }
  // one- and two-word states: V states per thread, 16-byte accesses (needs n % V == 0 and aligned side arrays)
  const uintptr_t side = reinterpret_cast<uintptr_t>(d_actions) | reinterpret_cast<uintptr_t>(d_mask) | reinterpret_cast<uintptr_t>(d_status);
  const int kind = src->spec.desc.game_kind;
  if (planes16 && kind == kTtt && (n & 3) == 0 && (side & 7u) == 0 && cmb == 2) {  // (8 states per thread measured slower)
    k_step_vec<Ttt, uint16_t, 4, 1><<<dim3(grid_for(n / 4)), dim3(kBlock), 0, ctx->stream>>>(
        src->spec.ttt, static_cast<const uint32_t*>(src->d_words), static_cast<uint32_t*>(dst->d_words), n, d_actions,
        static_cast<uint16_t*>(d_mask), d_status);
    OSG_HIP(hipGetLastError());
    return OSG_OK;
  }
  if (planes16 && kind == kKuhn && (n & 1) == 0 && (side & 1u) == 0 && cmb == 1) {
    k_step_vec<Kuhn, uint8_t, 2, 1><<<dim3(grid_for(n / 2)), dim3(kBlock), 0, ctx->stream>>>(
        src->spec.kuhn, static_cast<const uint64_t*>(src->d_words), static_cast<uint64_t*>(dst->d_words), n, d_actions,
        static_cast<uint8_t*>(d_mask), d_status);
    OSG_HIP(hipGetLastError());
    return OSG_OK;
  }
  // leduc_poker: its step is bound by the rules' arithmetic more than by the width of the plane accesses — two states
  // per thread pay off only once the batch is many rounds of wavefronts (2^24 states: 106.6 vs 114.1 us), smaller
  // batches run faster with one state per thread and twice the wavefronts (2^20 states: 9.6 vs 8.9 us;
  // tools/probe_states_per_thread.py, tools/probe_kernels.py)
  if (planes16 && kind == kLeduc && !src->spec.leduc_big && n >= (int64_t{1} << 22) && (n & 1) == 0 && (side & 1u) == 0 && cmb == 1) {
    k_step_vec<Leduc, uint8_t, 2, 2><<<dim3(grid_for(n / 2)), dim3(kBlock), 0, ctx->stream>>>(
        src->spec.leduc, static_cast<const uint64_t*>(src->d_words), static_cast<uint64_t*>(dst->d_words), n, d_actions,
        static_cast<uint8_t*>(d_mask), d_status);
    OSG_HIP(hipGetLastError());
    return OSG_OK;
  }
  // hex: V states per thread (16-byte plane accesses with V = 4); the mask rows are the [n, NW] u32 output
  if (kind == kHex && cmb == 4 * W && W == src->spec.hex_nw && W <= 4) {   // (the big boards: one state per thread, below)
    // OSG_HEX_STEP="<states per thread>:<non-temporal 0|1>" overrides the choice (a tuning knob; results do not depend on it)
    int v = 2, nt = n >= (int64_t{1} << 22) ? 1 : 0;
    if (const char* e = std::getenv("OSG_HEX_STEP")) {
      int ev = 0, ent = 0;
      if (std::sscanf(e, "%d:%d", &ev, &ent) == 2 && (ev == 1 || ev == 2)) { v = ev; nt = ent ? 1 : 0; }
    }
    while (v > 1 && ((n % v) != 0 || (side & static_cast<uintptr_t>(v - 1)) != 0 ||
                     (reinterpret_cast<uintptr_t>(d_mask) & static_cast<uintptr_t>(4 * v - 1)) != 0 || !planes16))
      v >>= 1;
    if (v > 1) {
      const auto* s32 = static_cast<const uint32_t*>(src->d_words);
      auto* d32 = static_cast<uint32_t*>(dst->d_words);
      auto* m32 = static_cast<uint32_t*>(d_mask);
#define OSG_HEXVEC(NWV, VV, NTV, MASKV, FOLDV, member)                                                                              \
  k_step_hexvec<NWV, VV, NTV, MASKV, FOLDV><<<dim3(grid_for(n / VV)), dim3(kBlock), 0, ctx->stream>>>(src->spec.member, s32, d32, n, \
                                                                                                    d_actions, m32, d_status)
#define OSG_HEXVEC_MASK(NWV, FOLDV, member)                                   \
  do {                                                                        \
    if (nt && m32) OSG_HEXVEC(NWV, 2, true, true, FOLDV, member);             \
    else if (nt) OSG_HEXVEC(NWV, 2, true, false, FOLDV, member);              \
    else if (m32) OSG_HEXVEC(NWV, 2, false, true, FOLDV, member);             \
    else OSG_HEXVEC(NWV, 2, false, false, FOLDV, member);                     \
  } while (0)
#define OSG_HEXVEC_NW(NWV, member)                                                                          \
  do {                                                                                                      \
    if (src->spec.hex_fold) OSG_HEXVEC_MASK(NWV, true, member); else OSG_HEXVEC_MASK(NWV, false, member);   \
  } while (0)
      switch (src->spec.hex_nw) {
        case 1: OSG_HEXVEC_NW(1, hex1); break;
        case 2: OSG_HEXVEC_NW(2, hex2); break;
        case 3: OSG_HEXVEC_NW(3, hex3); break;
        default: OSG_HEXVEC_NW(4, hex4); break;
      }
#undef OSG_HEXVEC_MASK
#undef OSG_HEXVEC_NW
#undef OSG_HEXVEC
      OSG_HIP(hipGetLastError());
      return OSG_OK;
    }
  }
  if (!d_mask)
    return set_error(OSG_ERR_UNSUPPORTED, "osg_step: d_mask may be NULL only for hex boards of up to 128 cells stepped two states per "
                                          "thread (an even batch, 2-byte aligned side arrays): there the successor's mask is ~occupied "
                                          "of the record written; every other game's mask is computed by the step itself");
  if (cmb == 1) {
    OSG_DISPATCH_WIDE(src->spec, k_step<G, uint8_t><<<dim3(grid_for(n)), dim3(kBlock), 0, ctx->stream>>>(P, static_cast<const typename G::word_t*>(src->d_words),
                                                static_cast<typename G::word_t*>(dst->d_words), n, d_actions,
                                                static_cast<uint8_t*>(d_mask), 1, d_status));
  } else if (cmb == 2) {
    OSG_DISPATCH_WIDE(src->spec, k_step<G, uint16_t><<<dim3(grid_for(n)), dim3(kBlock), 0, ctx->stream>>>(P, static_cast<const typename G::word_t*>(src->d_words),
                                                static_cast<typename G::word_t*>(dst->d_words), n, d_actions,
                                                static_cast<uint16_t*>(d_mask), 1, d_status));
  } else {
    OSG_DISPATCH_WIDE(src->spec, k_step<G, uint32_t><<<dim3(grid_for(n)), dim3(kBlock), 0, ctx->stream>>>(P, static_cast<const typename G::word_t*>(src->d_words),
                                                static_cast<typename G::word_t*>(dst->d_words), n, d_actions,
                                                static_cast<uint32_t*>(d_mask), W, d_status));
  }
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int osg_observation(const osg_batch* b, int player, int which, float* out, int on_host) {
  osg_ctx* ctx = b->ctx;
  const osg_game_desc& d = b->spec.desc;
  const int size = which == 0 ? d.obs_size : d.info_size;
  if (size <= 0) return set_error(OSG_ERR_INVALID, "this game provides no such tensor");
  if (player < -1 || player >= d.num_players)
    return set_error(OSG_ERR_INVALID, "player id out of range");  // SPIEL_CHECK_GE/LT, spiel.cc:914-915
  const int64_t total = b->n * size;
  float* d_out = out;
  if (on_host) {
    void* scratch;
    int rc = osg_ctx_scratch(ctx, sizeof(float) * total, &scratch);
    if (rc) return rc;
    d_out = static_cast<float*>(scratch);
  }
  // OSG_OBS_FORM=0: the span-per-wavefront kernels of round 3 (A/B: tools/probe_obs_forms.py); 1: the piece form
  // with plain stores; default 2: the piece form with non-temporal stores.
  static const int obs_form = std::getenv("OSG_OBS_FORM") ? std::atoi(std::getenv("OSG_OBS_FORM")) : 2;
  const bool aligned16 = (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0;
  // (below ~2^24 floats the launch is a few hundred workgroups: the span kernels' finer grid fills the chip better —
  // [2^16, 126]: 7.4 vs 8.4 us, hex(9) [2^14, 729]: 11.9 vs 29.5 us; from [2^20, 27] on the piece form is ahead)
  const bool pieces = obs_form != 0 && aligned16 && total < (int64_t{1} << 32) && total >= (int64_t{1} << 24);
  const bool nt = obs_form != 1;
  // OSG_OBS_LDS=0: every piece builds its rows' images itself (A/B); 4 / 8: images staged in LDS, that many spans per
  // workgroup; default: 8 for tic_tac_toe (0.77 vs 0.70 with 4 and 0.75 without), 4 for leduc_poker (information rows 0.83
  // vs 0.78 with 8 and 0.80 without) — tools/probe_obs_lds.py
  static const int obs_lds_env = std::getenv("OSG_OBS_LDS") ? std::atoi(std::getenv("OSG_OBS_LDS")) : -1;
  const int obs_lds = obs_lds_env >= 0 ? obs_lds_env : (b->spec.desc.game_kind == kTtt ? 8 : 4);
  const unsigned piece_grid = static_cast<unsigned>(((total + 1023) / 1024 + 3) / 4);   // four 1 KiB-piece spans per workgroup
  if (pieces && b->spec.desc.game_kind == kC4 && b->spec.c4_std) {
#define OSG_C4P(NT, EGO) k_observation_c4std_pieces<NT, EGO, 4><<<dim3(piece_grid), dim3(kPieceBlock), 0, ctx->stream>>>( \
      b->spec.c4, static_cast<const uint64_t*>(b->d_words), static_cast<uint32_t>(b->n), static_cast<uint32_t>(total), player, d_out)
    if (b->spec.c4.ego) { if (nt) OSG_C4P(true, true); else OSG_C4P(false, true); }
    else { if (nt) OSG_C4P(true, false); else OSG_C4P(false, false); }
#undef OSG_C4P
  } else if (pieces && obs_lds && (b->spec.desc.game_kind == kTtt || (b->spec.desc.game_kind == kLeduc && d.num_players == 2))) {
    // the rows' images staged in LDS (k_observation_row_pieces_lds)
    uint32_t magic = 0, shift = 0, cmagic = 0, cshift = 0;
    const uint32_t cap = 1024u / static_cast<uint32_t>(size) + 3u;
    if (!find_div_magic(static_cast<uint32_t>(size), static_cast<uint32_t>(size) + 1024u, &magic, &shift) ||
        !find_div_magic(cap, 8u * cap, &cmagic, &cshift))
      return set_error(OSG_ERR_INVALID, "osg_observation: no multiply-shift pair for this row size");
    const FastDiv fd = make_fast_div(static_cast<uint32_t>(size));
    const uint32_t nn = static_cast<uint32_t>(b->n), tot = static_cast<uint32_t>(total);
#define OSG_ROWL(F, f, SP) do {                                                                                              \
      const unsigned g = static_cast<unsigned>(((total + 1023) / 1024 + (SP) - 1) / (SP));                                     \
      const size_t lds = sizeof(typename F::Img) * (SP) * cap;                                                                 \
      if (nt) k_observation_row_pieces_lds<F, true, SP><<<dim3(g), dim3(kPieceBlock), lds, ctx->stream>>>(f, nn, fd, magic, shift, cap, cmagic, cshift, tot, d_out); \
      else k_observation_row_pieces_lds<F, false, SP><<<dim3(g), dim3(kPieceBlock), lds, ctx->stream>>>(f, nn, fd, magic, shift, cap, cmagic, cshift, tot, d_out);  \
    } while (0)
    if (b->spec.desc.game_kind == kTtt) {
      TttPieces f{static_cast<const uint32_t*>(b->d_words)};
      if (obs_lds == 8) OSG_ROWL(TttPieces, f, 8); else OSG_ROWL(TttPieces, f, 4);
    } else {
      const int K = b->spec.leduc.iso ? b->spec.leduc.cards / 2 : b->spec.leduc.cards;
      if (which == 0) {
        Leduc2Pieces<0> f{static_cast<const uint64_t*>(b->d_words), nn, player, K, b->spec.leduc};
        if (obs_lds == 8) OSG_ROWL(Leduc2Pieces<0>, f, 8); else OSG_ROWL(Leduc2Pieces<0>, f, 4);
      } else {
        Leduc2Pieces<1> f{static_cast<const uint64_t*>(b->d_words), nn, player, K, b->spec.leduc};
        if (obs_lds == 8) OSG_ROWL(Leduc2Pieces<1>, f, 8); else OSG_ROWL(Leduc2Pieces<1>, f, 4);
      }
    }
#undef OSG_ROWL
  } else if (pieces && b->spec.desc.game_kind == kTtt) {
    uint32_t magic = 0, shift = 0;
    find_div_magic(static_cast<uint32_t>(size), static_cast<uint32_t>(size) + 1024u, &magic, &shift);
    TttPieces f{static_cast<const uint32_t*>(b->d_words)};
    // eight spans per thread here (27-float rows: 0.74 with four, 0.76 with eight; the round-3 kernel: 0.69)
    const unsigned g8 = (piece_grid + 1) / 2;
    if (nt) k_observation_row_pieces<TttPieces, true, 8><<<dim3(g8), dim3(kPieceBlock), 0, ctx->stream>>>(
        f, static_cast<uint32_t>(b->n), make_fast_div(static_cast<uint32_t>(size)), magic, shift, static_cast<uint32_t>(total), d_out);
    else k_observation_row_pieces<TttPieces, false, 8><<<dim3(g8), dim3(kPieceBlock), 0, ctx->stream>>>(
        f, static_cast<uint32_t>(b->n), make_fast_div(static_cast<uint32_t>(size)), magic, shift, static_cast<uint32_t>(total), d_out);
  } else if (pieces && b->spec.desc.game_kind == kKuhn && d.num_players == 2) {
    uint32_t magic = 0, shift = 0;
    find_div_magic(static_cast<uint32_t>(size), static_cast<uint32_t>(size) + 1024u, &magic, &shift);
#define OSG_KUHNP(W) do {                                                                                               \
      Kuhn2Pieces<W> f{static_cast<const uint64_t*>(b->d_words), player};                                               \
      if (nt) k_observation_row_pieces<Kuhn2Pieces<W>, true, 4><<<dim3(piece_grid), dim3(kPieceBlock), 0, ctx->stream>>>( \
          f, static_cast<uint32_t>(b->n), make_fast_div(static_cast<uint32_t>(size)), magic, shift, static_cast<uint32_t>(total), d_out); \
      else k_observation_row_pieces<Kuhn2Pieces<W>, false, 4><<<dim3(piece_grid), dim3(kPieceBlock), 0, ctx->stream>>>(  \
          f, static_cast<uint32_t>(b->n), make_fast_div(static_cast<uint32_t>(size)), magic, shift, static_cast<uint32_t>(total), d_out); \
    } while (0)
    if (which == 0) OSG_KUHNP(0); else OSG_KUHNP(1);
#undef OSG_KUHNP
  } else if (pieces && b->spec.desc.game_kind == kLeduc && d.num_players == 2) {
    uint32_t magic = 0, shift = 0;
    find_div_magic(static_cast<uint32_t>(size), static_cast<uint32_t>(size) + 1024u, &magic, &shift);
    const int K = b->spec.leduc.iso ? b->spec.leduc.cards / 2 : b->spec.leduc.cards;
#define OSG_LEDUCP(W) do {                                                                                              \
      Leduc2Pieces<W> f{static_cast<const uint64_t*>(b->d_words), static_cast<uint32_t>(b->n), player, K, b->spec.leduc}; \
      if (nt) k_observation_row_pieces<Leduc2Pieces<W>, true, 4><<<dim3(piece_grid), dim3(kPieceBlock), 0, ctx->stream>>>( \
          f, static_cast<uint32_t>(b->n), make_fast_div(static_cast<uint32_t>(size)), magic, shift, static_cast<uint32_t>(total), d_out); \
      else k_observation_row_pieces<Leduc2Pieces<W>, false, 4><<<dim3(piece_grid), dim3(kPieceBlock), 0, ctx->stream>>>(  \
          f, static_cast<uint32_t>(b->n), make_fast_div(static_cast<uint32_t>(size)), magic, shift, static_cast<uint32_t>(total), d_out); \
    } while (0)
    if (which == 0) OSG_LEDUCP(0); else OSG_LEDUCP(1);
#undef OSG_LEDUCP
  } else if (pieces && b->spec.desc.game_kind == kHex && which == 0 && d.obs_shape[0] == 9 &&
             d.obs_shape[1] * d.obs_shape[2] >= 29) {   // (a span of 4096 floats then touches at most 18 states)
    const uint32_t cells = static_cast<uint32_t>(d.obs_shape[1] * d.obs_shape[2]);
    uint32_t cm = 0, cs = 0, lm = 0, ls = 0;
    if (!find_div_magic(cells, 9u * cells, &cm, &cs) || !find_div_magic(9u * cells, 9u * cells + kHexLdsSpan, &lm, &ls))
      return set_error(OSG_ERR_INVALID, "osg_observation: no multiply-shift pair for this hex board");
    const unsigned g = static_cast<unsigned>(((total + kHexLdsSpan - 1) / kHexLdsSpan + 3) / 4);
#define OSG_HEXL(NW, NT) k_observation_hex_pieces_lds<NW, NT, 4><<<dim3(g), dim3(kPieceBlock), 0, ctx->stream>>>(               \
      static_cast<const uint32_t*>(b->d_words), static_cast<uint32_t>(b->n), cells, cm, cs, make_fast_div(9u * cells), lm, ls,    \
      static_cast<uint32_t>(total), d_out, b->spec.hex_fold ? 0x07FFFFFFu : 0xFFFFFFFFu)
    switch (b->spec.hex_nw) {
      case 1: if (nt) OSG_HEXL(1, true); else OSG_HEXL(1, false); break;
      case 2: if (nt) OSG_HEXL(2, true); else OSG_HEXL(2, false); break;
      case 3: if (nt) OSG_HEXL(3, true); else OSG_HEXL(3, false); break;
      case 4: if (nt) OSG_HEXL(4, true); else OSG_HEXL(4, false); break;
      case 6: if (nt) OSG_HEXL(6, true); else OSG_HEXL(6, false); break;
      case 8: if (nt) OSG_HEXL(8, true); else OSG_HEXL(8, false); break;
      default: if (nt) OSG_HEXL(12, true); else OSG_HEXL(12, false); break;
    }
#undef OSG_HEXL
  } else if (b->spec.desc.game_kind == kC4 && b->spec.c4_std) {
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0 && b->n >= (int64_t{1} << 22))
      k_observation_c4std_planes<true><<<dim3(static_cast<unsigned>((b->n * 3 + kC4ObsBlock - 1) / kC4ObsBlock)),
                                         dim3(kC4ObsBlock), 0, ctx->stream>>>(
          b->spec.c4, static_cast<const uint64_t*>(b->d_words), b->n, player, d_out);
    else if ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0)
      k_observation_c4std_planes<false><<<dim3(static_cast<unsigned>((b->n * 3 + kC4ObsBlock - 1) / kC4ObsBlock)),
                                          dim3(kC4ObsBlock), 0, ctx->stream>>>(
          b->spec.c4, static_cast<const uint64_t*>(b->d_words), b->n, player, d_out);
    else
      k_observation_c4std<<<dim3(grid_for(b->n * 18)), dim3(kBlock), 0, ctx->stream>>>(
          b->spec.c4, static_cast<const uint64_t*>(b->d_words), b->n, player, d_out);
  } else if (b->spec.desc.game_kind == kHex && which == 0 && d.obs_shape[0] == 9 && b->spec.hex_nw <= 4 &&
             (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0) {   // (its LDS stage is 256 B per cell: the big boards go below)
    const size_t shmem = sizeof(float) * kHexObsBlock * static_cast<size_t>(d.obs_shape[1] * d.obs_shape[2]);
    const unsigned grid = static_cast<unsigned>((b->n * 9 + kHexObsBlock - 1) / kHexObsBlock);
#define OSG_HEX_OBS(NW, member)                                                                        \
  k_observation_hex_planes<HexT<NW>><<<dim3(grid), dim3(kHexObsBlock), shmem, ctx->stream>>>(          \
      b->spec.member, static_cast<const uint32_t*>(b->d_words), b->n, 9, d_out)
    switch (b->spec.hex_nw) {
      case 1: OSG_HEX_OBS(1, hex1); break;
      case 2: OSG_HEX_OBS(2, hex2); break;
      case 3: OSG_HEX_OBS(3, hex3); break;
      default: OSG_HEX_OBS(4, hex4); break;
    }
#undef OSG_HEX_OBS
  } else if (size <= kRowsMaxSize && b->spec.desc.game_kind != kHex && (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0) {
    // short rows: one lane per state, LDS-staged aligned float4 stores
    // states per lane: two for the shortest rows (measured at 2^24 states, 1 / 2 / 4 per lane: kuhn [n, 7] 124.8 / 103.5 /
    // 104.1 us, [n, 11] 144.2 / 128.6 / 161.3; from 16 floats per row on one is best: leduc [n, 16] 193 / 198 / 274,
    // tic_tac_toe [n, 27] 338 / 445 / 678, leduc [n, 30] 352 / 499 / 921 — profiles/r03_obs_rows_per_lane.log)
    const int kr = size <= 12 ? 2 : 1;
    const size_t shmem = sizeof(float) * (kRowsBlock / 64) * 64 * kr * static_cast<size_t>(size | 1);
    const unsigned grid = static_cast<unsigned>((b->n + kRowsBlock * kr - 1) / (kRowsBlock * kr));
#define OSG_ROWS(KR)                                                                                               \
  OSG_DISPATCH_WIDE(b->spec, k_observation_rows<G, KR><<<dim3(grid), dim3(kRowsBlock), shmem, ctx->stream>>>(           \
                            P, static_cast<const typename G::word_t*>(b->d_words), b->n, size, player, which, d_out))
    if (kr == 2) OSG_ROWS(2);
    else OSG_ROWS(1);
#undef OSG_ROWS
  } else {
    // Segment = one tensor plane for hex's 9-plane layout (the cursor's mask is per plane), else the row.
    int seg_len = size;
    if (b->spec.desc.game_kind == kHex && d.obs_shape[0] == 9) seg_len = d.obs_shape[1] * d.obs_shape[2];
    const bool wide = size >= 64 || b->spec.desc.game_kind == kLeduc;  // 16+ floats per lane: long rows, or a
                                                                      // bit-packed state worth decoding once
    // (32 floats per lane was measured too: fewer, fuller chunks for hex(9) but 15 % slower — the
    // 28 KiB LDS tile per workgroup costs more occupancy than the fuller chunks give back.)
    const int F = wide ? 16 : 4;
    const int cps = (seg_len + F - 1) / F;
    const int64_t lanes = b->n * (size / seg_len) * cps;
#define OSG_OBS_LAUNCH(FF)                                                                                          \
  OSG_DISPATCH_WIDE(b->spec, k_observation<G, FF><<<dim3(grid_for(lanes)), dim3(kBlock), 0, ctx->stream>>>(P,            \
                                            static_cast<const typename G::word_t*>(b->d_words), b->n, size, seg_len, \
                                            cps, player, which, d_out))
    if (F == 16) OSG_OBS_LAUNCH(16);
    else OSG_OBS_LAUNCH(4);
#undef OSG_OBS_LAUNCH
  }
  OSG_HIP(hipGetLastError());
  if (on_host) {
    OSG_HIP(hipMemcpyAsync(out, d_out, sizeof(float) * total, hipMemcpyDeviceToHost, ctx->stream));
    OSG_HIP(hipStreamSynchronize(ctx->stream));
  }
  return OSG_OK;
}

int osg_random_steps(osg_batch* b, uint64_t seed, int64_t index_offset, int steps, unsigned long long* d_counters) {
  osg_ctx* ctx = b->ctx;
  unsigned long long* partials = ctx->d_illegal + 1;
  int64_t blocks = (b->n + kBlock - 1) / kBlock;
#ifndef OSG_RS_BLOCKS
#define OSG_RS_BLOCKS 4096
#endif
  if (blocks > OSG_RS_BLOCKS) blocks = OSG_RS_BLOCKS;  // 16 workgroups per CU (4096 measured 4 % faster than 2048), grid-strided beyond
  OSG_DISPATCH_WIDE(b->spec, k_random_steps<G><<<dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, ctx->stream>>>(P,
                                            static_cast<typename G::word_t*>(b->d_words), b->n, seed, index_offset,
                                            steps, partials));
  k_fold_counters<<<dim3(1), dim3(64), 0, ctx->stream>>>(partials, d_counters);
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int osg_synth_batch(osg_batch* b, uint64_t seed, int64_t index_offset, int depth_mod, uint8_t* d_actions,
                    int32_t* d_depth) {
  if (!b) return set_error(OSG_ERR_INVALID, "osg_synth_batch: null batch");
  if (depth_mod < 1 || depth_mod > b->spec.desc.max_game_length)
    return set_error(OSG_ERR_INVALID, "osg_synth_batch: depth_mod must lie in [1, MaxGameLength()]");
  if (int rc = refuse_endless_playouts(b->spec, "osg_synth_batch")) return rc;
  if (d_actions && b->spec.desc.num_distinct_actions > 255)
    return set_error(OSG_ERR_UNSUPPORTED, "osg_synth_batch: d_actions holds one byte per action; pass NULL for games with more "
                                          "than 255 actions");
  osg_ctx* ctx = b->ctx;
  OSG_DISPATCH_WIDE(b->spec, k_synth<G><<<dim3(grid_for(b->n)), dim3(kBlock), 0, ctx->stream>>>(P,
                                     static_cast<typename G::word_t*>(b->d_words), b->n, seed, index_offset, depth_mod,
                                     d_actions, d_depth));
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int osg_rollout(const osg_batch* roots, uint64_t seed, int64_t index_offset, int n_rollouts, double* sum_returns,
                int32_t* steps, int on_host) {
  osg_ctx* ctx = roots->ctx;
  const int P_ = roots->spec.desc.num_players;
  const int64_t n = roots->n;
  if (n_rollouts <= 0) return set_error(OSG_ERR_INVALID, "n_rollouts must be positive");
  if (int rc = refuse_endless_playouts(roots->spec, "osg_rollout")) return rc;
  // Lanes per root: enough shares to fill the chip (8 waves per SIMD = 2^19 lanes), no more.
#ifndef OSG_ROLLOUT_LANES_LOG2
#define OSG_ROLLOUT_LANES_LOG2 19
#endif
  int64_t group = ((int64_t{1} << OSG_ROLLOUT_LANES_LOG2) + n - 1) / std::max<int64_t>(n, 1);
  if (group > n_rollouts) group = n_rollouts;
  if (group < 1) group = 1;
  // scratch: [sums | steps] when the results go to the host, then the per-share slots when group > 1
  const size_t ret_bytes = sizeof(double) * P_ * n, step_bytes = sizeof(int32_t) * n;
  const size_t off_steps = align_up(ret_bytes), off_part = on_host ? align_up(off_steps + step_bytes) : 0;
  const size_t part_bytes = group > 1 ? ret_bytes * group : 0, off_part_steps = align_up(off_part + part_bytes);
  const size_t scratch_bytes = group > 1 ? off_part_steps + step_bytes * group : (on_host ? off_steps + step_bytes : 0);
  char* scratch = nullptr;
  if (scratch_bytes) {
    void* ptr;
    int rc = osg_ctx_scratch(ctx, scratch_bytes, &ptr);
    if (rc) return rc;
    scratch = static_cast<char*>(ptr);
  }
  double* d_sum = on_host ? reinterpret_cast<double*>(scratch) : sum_returns;
  int32_t* d_steps = steps ? (on_host ? reinterpret_cast<int32_t*>(scratch + off_steps) : steps) : nullptr;
  double* d_part = group > 1 ? reinterpret_cast<double*>(scratch + off_part) : d_sum;
  int32_t* d_part_steps = d_steps ? (group > 1 ? reinterpret_cast<int32_t*>(scratch + off_part_steps) : d_steps) : nullptr;
  const int64_t total = n * group;
  // Persistent grid: at most 8 blocks per CU x 256 CUs, grid-strided beyond that.
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
  if (OSG_HEX_FILL_PLAYOUT && !steps && roots->spec.desc.game_kind == kHex) {
    OSG_DISPATCH_WIDE(roots->spec, launch_rollout_hexfill<G>(P, roots->d_words, n, seed, index_offset, n_rollouts,
                                                             static_cast<int>(group), d_part, static_cast<unsigned>(blocks), ctx->stream));
  } else {
  OSG_DISPATCH_WIDE(roots->spec, k_rollout<G><<<dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, ctx->stream>>>(P,
                                                static_cast<const typename G::word_t*>(roots->d_words), n, P_, seed,
                                                index_offset, n_rollouts, static_cast<int>(group), d_part,
                                                d_part_steps));
  }
  if (group > 1)
    k_rollout_fold<<<dim3(grid_for(n * P_)), dim3(kBlock), 0, ctx->stream>>>(d_part, d_part_steps, n, P_,
                                                                             static_cast<int>(group), d_sum, d_steps);
  OSG_HIP(hipGetLastError());
  if (on_host) {
    OSG_HIP(hipMemcpyAsync(sum_returns, d_sum, ret_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (steps) OSG_HIP(hipMemcpyAsync(steps, d_steps, step_bytes, hipMemcpyDeviceToHost, ctx->stream));
    OSG_HIP(hipStreamSynchronize(ctx->stream));
  }
  return OSG_OK;
}

int osg_env_step(osg_batch* b, const int32_t* d_actions, uint8_t* d_should_reset, uint64_t seed, int64_t index_offset,
                 int64_t step_index, int8_t* d_cur_player, uint8_t* d_step_type, double* d_rewards, uint32_t* d_mask) {
  if (!b || !d_actions || !d_should_reset || !d_cur_player || !d_step_type || !d_rewards || !d_mask)
    return set_error(OSG_ERR_INVALID, "osg_env_step: null argument");
  osg_ctx* ctx = b->ctx;
  {  // two environments per thread where the layout allows it (two 64-bit planes, two players, one mask word)
    const uintptr_t side = reinterpret_cast<uintptr_t>(d_should_reset) | reinterpret_cast<uintptr_t>(d_cur_player) |
                           reinterpret_cast<uintptr_t>(d_step_type);
    const bool ok = (b->n & 1) == 0 && (reinterpret_cast<uintptr_t>(b->d_words) & 15u) == 0 && (side & 1u) == 0 &&
                    (reinterpret_cast<uintptr_t>(d_actions) & 7u) == 0 && (reinterpret_cast<uintptr_t>(d_mask) & 7u) == 0 &&
                    (reinterpret_cast<uintptr_t>(d_rewards) & 15u) == 0 && b->spec.desc.num_players == 2 &&
                    b->spec.desc.mask_words == 1 && !std::getenv("OSG_ENV_STEP_X1");
    const unsigned grid2 = static_cast<unsigned>(grid_for(b->n / 2));
    if (ok && b->spec.desc.game_kind == kC4 && !b->spec.c4_wide) {
      if (b->spec.c4_std)
        k_env_step_x2<C4Std><<<dim3(grid2), dim3(kBlock), 0, ctx->stream>>>(b->spec.c4, static_cast<uint64_t*>(b->d_words), b->n, d_actions,
                                                                          d_should_reset, seed, index_offset, step_index, d_cur_player,
                                                                          d_step_type, d_rewards, d_mask, ctx->d_illegal);
      else
        k_env_step_x2<C4><<<dim3(grid2), dim3(kBlock), 0, ctx->stream>>>(b->spec.c4, static_cast<uint64_t*>(b->d_words), b->n, d_actions,
                                                                       d_should_reset, seed, index_offset, step_index, d_cur_player,
                                                                       d_step_type, d_rewards, d_mask, ctx->d_illegal);
      OSG_HIP(hipGetLastError());
      return OSG_OK;
    }
  }
  OSG_DISPATCH_WIDE(b->spec, k_env_step<G><<<dim3(grid_for(b->n)), dim3(kBlock), 0, ctx->stream>>>(P,
                                            static_cast<typename G::word_t*>(b->d_words), b->n,
                                            b->spec.desc.num_players, d_actions, d_should_reset, seed, index_offset,
                                            step_index, d_cur_player, d_step_type, d_rewards, d_mask,
                                            b->spec.desc.mask_words, ctx->d_illegal));
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int osg_env_step_compact(osg_batch* b, const uint8_t* d_actions, uint8_t* d_flags, uint64_t seed, int64_t index_offset,
                         int64_t step_index, int8_t* d_rewards_x2, uint32_t* d_mask) {
  if (!b || !d_actions || !d_flags || !d_rewards_x2 || !d_mask) return set_error(OSG_ERR_INVALID, "osg_env_step_compact: null argument");
  if (b->spec.desc.num_distinct_actions > 255)
    return set_error(OSG_ERR_UNSUPPORTED, "osg_env_step_compact: action ids travel as one byte here (0xFF = leave); games with more than 255 "
                                          "actions (hex from 16 x 16) take osg_env_step");
  if (2.0 * b->spec.desc.max_utility > 127.0 || 2.0 * b->spec.desc.min_utility < -128.0)
    return set_error(OSG_ERR_UNSUPPORTED, "osg_env_step_compact: twice the game's returns do not fit a signed byte (use osg_env_step)");
  osg_ctx* ctx = b->ctx;
  {
    const bool ok = (b->n & 1) == 0 && (reinterpret_cast<uintptr_t>(b->d_words) & 15u) == 0 &&
                    ((reinterpret_cast<uintptr_t>(d_actions) | reinterpret_cast<uintptr_t>(d_flags)) & 1u) == 0 &&
                    (reinterpret_cast<uintptr_t>(d_rewards_x2) & 3u) == 0 && (reinterpret_cast<uintptr_t>(d_mask) & 7u) == 0 &&
                    b->spec.desc.num_players == 2 && b->spec.desc.mask_words == 1 && !std::getenv("OSG_ENV_STEP_X1");
    const unsigned grid2 = static_cast<unsigned>(grid_for(b->n / 2));
    if (ok && b->spec.desc.game_kind == kC4 && !b->spec.c4_wide) {
      if (b->spec.c4_std)
        k_env_step_compact_x2<C4Std><<<dim3(grid2), dim3(kBlock), 0, ctx->stream>>>(b->spec.c4, static_cast<uint64_t*>(b->d_words), b->n, d_actions, d_flags,
                                                                                  seed, index_offset, step_index, d_rewards_x2, d_mask, ctx->d_illegal);
      else
        k_env_step_compact_x2<C4><<<dim3(grid2), dim3(kBlock), 0, ctx->stream>>>(b->spec.c4, static_cast<uint64_t*>(b->d_words), b->n, d_actions, d_flags,
                                                                               seed, index_offset, step_index, d_rewards_x2, d_mask, ctx->d_illegal);
      OSG_HIP(hipGetLastError());
      return OSG_OK;
    }
  }
  OSG_DISPATCH_WIDE(b->spec, k_env_step_compact<G><<<dim3(grid_for(b->n)), dim3(kBlock), 0, ctx->stream>>>(P,
                                            static_cast<typename G::word_t*>(b->d_words), b->n, b->spec.desc.num_players, d_actions,
                                            d_flags, seed, index_offset, step_index, d_rewards_x2, d_mask, b->spec.desc.mask_words,
                                            ctx->d_illegal));
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

}  // extern "C"
