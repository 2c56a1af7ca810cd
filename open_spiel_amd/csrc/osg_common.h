// Shared device/host helpers: counter RNG, bit tricks, the fixed-width legal
// mask.  gfx950 only.
#ifndef OSG_COMMON_H_
#define OSG_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define OSG_HD __host__ __device__ __forceinline__
#define OSG_D __device__ __forceinline__

namespace osg {

constexpr int kChancePlayer = -1;
constexpr int kTerminalPlayer = -4;
constexpr int kMaxPlayers = 10;     // kuhn_poker supports up to 10
constexpr int kMaskWords = 4;       // 128 actions (hex 11x11 = 121)
// No game served here lasts longer than 130 plies (hex 11 x 11 with the swap move).  A random playout that has not
// ended after kMaxPlayoutPlies moves started from a state the rules cannot finish (an uploaded record that is
// neither terminal nor has a legal action): it is cut off — Returns() of a running game are zeros — instead of
// spinning on the device for ever.
constexpr int kMaxPlayoutPlies = 512;

// ---------------------------------------------------------------------------
// Counter-based RNG.  splitmix64 over a key mixed from (seed, stream, sub); the
// CPU oracle (oracle/spiel_oracle_core.cpp CounterRng) restates exactly this so
// device rollouts / searches / trajectories can be replayed bit for bit.
// ---------------------------------------------------------------------------
OSG_HD uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
struct Rng {
  uint64_t s;
  OSG_HD Rng(uint64_t seed, uint64_t stream, uint64_t sub) {
    uint64_t a = mix64(seed + 0x9E3779B97F4A7C15ULL);
    uint64_t b = mix64(a ^ (stream * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL));
    s = mix64(b ^ (sub * 0xA0761D6478BD642FULL + 0xE7037ED1A0B428DBULL));
  }
  // Sub-streams by a jump of the counter (the external-sampling traversals open one per subtree: a re-mixed key per
  // sub-stream cost a quarter of the kernel's throughput): the generator is counter + mixer, so a stream that starts at
  // s0 + id * kJump draws mix64 of another part of the counter space.  Two sub-streams of one s0 would meet only after
  // (id1 - id2) * kJump / kStep draws (mod 2^64) — more than 2^56 for every |id1 - id2| <= 128
  // (tests/test_synth_batch_cpu.py checks the constant).
  static constexpr uint64_t kJump = 0xD6E8FEB86659FD93ULL;
  OSG_HD void jump_to(uint64_t s0, uint64_t id) { s = s0 + id * kJump; }
  OSG_HD uint64_t next() {
    s += 0x9E3779B97F4A7C15ULL;
    return mix64(s);
  }
  // floor(hi32 * n / 2^32): uniform in [0, n) up to 2^-32 bias.
  OSG_HD uint32_t below(uint32_t n) {
    uint64_t hi = next() >> 32;
    return static_cast<uint32_t>((hi * n) >> 32);
  }
  OSG_HD double unit() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// ---------------------------------------------------------------------------
// Keyed orderings used by the wave-per-root search (osg_mcts_wave.hip) and
// restated by the oracle's replay of it.  A uniformly random ORDER of a set is
// obtained by giving every element an independent 64-bit key and sorting by
// key; the low byte carries the element id so keys of one set never tie.
//   * sibling order of a freshly expanded node (the reference's std::shuffle,
//     mcts.cc:294): key = order_key(order_base(seed, root), path hash, action)
//   * the random fill of a hex playout: key = fill_key(fill_base(seed, root,
//     rollout), cell)
// ---------------------------------------------------------------------------
constexpr uint64_t kOrderSalt = 0x6F726465725F6B79ULL;
constexpr uint64_t kFillSalt = 0x66696C6C5F6B6579ULL;
OSG_HD uint64_t path_hash_root() { return 0x243F6A8885A308D3ULL; }
// (a 32-bit chain in a 64-bit slot: order_key only ever reads the low word, and one 32-bit mixer per tree
// level is a third of the scalar work of a 64-bit one)
OSG_HD uint32_t mix32(uint32_t x);
OSG_HD uint64_t path_hash_child(uint64_t parent, int action) {
  return mix32(static_cast<uint32_t>(parent) ^ (static_cast<uint32_t>(action + 1) * 0x9E3779B1u));
}
OSG_HD uint64_t order_base(uint64_t seed, uint64_t root) {
  return mix64(mix64(seed ^ kOrderSalt) ^ (root * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL));
}
// 32-bit avalanche mixer (two multiply-xorshift rounds): the sibling order only needs distinct,
// well-scattered keys, and 32-bit multiplies are what the vector ALU is good at.
OSG_HD uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  return x ^ (x >> 16);
}
OSG_HD uint32_t order_key(uint64_t base, uint64_t parent_path_hash, int action) {
  const uint32_t h = mix32(static_cast<uint32_t>(base) ^ static_cast<uint32_t>(parent_path_hash) ^
                           (static_cast<uint32_t>(action + 1) * 0x9E3779B1u));
  return (h & ~0xFFu) | static_cast<uint32_t>(action & 0xFF);  // low byte = action: siblings never tie
}
// The random fill of a hex playout orders the empty cells by a 40-bit key: 32 mixed bits and the cell id
// (so keys never tie; two cells share their 32 random bits with probability ~8e-7 per playout, and then the
// lower cell id goes first).  One 32-bit mixer per (root, playout) for the base, one per cell for the key.
// (in two parts so that a kernel can keep the per-root word and run the per-playout mixer where it likes)
OSG_HD uint32_t fill_root(uint64_t seed, uint64_t root) {
  const uint64_t a = mix64(mix64(seed ^ kFillSalt) ^ (root * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL));
  return static_cast<uint32_t>(a) ^ static_cast<uint32_t>(a >> 32);
}
OSG_HD uint64_t fill_base_of(uint32_t root_word, uint64_t sub) {
  return mix32(root_word ^ (static_cast<uint32_t>(sub) * 0x9E3779B1u) ^ (static_cast<uint32_t>(sub >> 32) * 0x85EBCA6Bu));
}
OSG_HD uint64_t fill_base(uint64_t seed, uint64_t root, uint64_t sub) { return fill_base_of(fill_root(seed, root), sub); }
OSG_HD uint64_t fill_key(uint64_t base, int cell) {
  const uint32_t h = mix32(static_cast<uint32_t>(base) ^ (static_cast<uint32_t>(cell + 1) * 0x9E3779B1u));
  return (static_cast<uint64_t>(h) << 8) | static_cast<uint64_t>(cell & 0xFF);
}
constexpr int kFillKeyBits = 40;

// ---------------------------------------------------------------------------
// Legal-action mask: up to 128 actions, bit a of word a/32.
// ---------------------------------------------------------------------------
template <int W>
struct MaskT {
  uint32_t w[W];
  OSG_HD MaskT() {
#pragma unroll
    for (int k = 0; k < W; ++k) w[k] = 0u;
  }
  // No dynamic indexing of w[]: a runtime index would demote the mask from
  // VGPRs to LDS/scratch (measured: 3x on the connect_four step kernel).
  OSG_HD void set(int a) {
    const uint32_t bit = 1u << (a & 31);
    const int i = a >> 5;
#pragma unroll
    for (int k = 0; k < W; ++k) w[k] |= (i == k) ? bit : 0u;
  }
  OSG_HD bool test(int a) const {
    const int i = a >> 5;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) v |= (i == k) ? w[k] : 0u;
    return (v >> (a & 31)) & 1u;
  }
  OSG_HD bool any() const {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) v |= w[k];
    return v != 0;
  }
  OSG_HD int count() const {
    int c = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) c += __builtin_popcount(w[k]);
    return c;
  }
};
// The mask of every game but the hex boards above 128 actions (kMaskWords words); those carry MaskT<NW>.
using Mask = MaskT<kMaskWords>;

// Index of the k-th (0-based) set bit of a 32-bit word; k < popcount(x).
OSG_HD int select32(uint32_t x, int k) {
  int pos = 0;
#pragma unroll
  for (int width = 16; width >= 1; width >>= 1) {
    uint32_t low = x & ((1u << width) - 1u);
    int c = __builtin_popcount(low);
    if (k >= c) {
      k -= c;
      x >>= width;
      pos += width;
    } else {
      x = low;
    }
  }
  return pos;
}
// k-th legal action of a mask (actions ascending, like LegalActions()[k]).
template <int W>
OSG_HD int select_action(const MaskT<W>& m, int k) {
  int base = 0;
#pragma unroll
  for (int i = 0; i < W; ++i) {
    int c = __builtin_popcount(m.w[i]);
    if (k < c) return base + select32(m.w[i], k);
    k -= c;
    base += 32;
  }
  return -1;
}

}  // namespace osg
#endif  // OSG_COMMON_H_
