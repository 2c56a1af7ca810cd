// Micro-benchmark: variants of the connect_four fused step kernel (states per thread,
// block size, non-temporal accesses).  Not part of the library; used to pick the
// shipped configuration.  hipcc --offload-arch=gfx950 -O3 -I../open_spiel_amd/csrc step_sweep.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "osg_common.h"
#include "osg_game_boards.h"
#include "osg_c4_step.h"
using namespace osg;
using G = C4Std;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint8_t enc(bool terminal, bool illegal, int cur, int outcome) {
  uint8_t v = illegal ? 0x40 : 0;
  if (terminal) return v | 0x80 | (uint8_t)(outcome & 7);
  return v | (uint8_t)((cur + 1) & 15);
}
// open columns of the standard 6x7 board by one multiply: the top cells sit at bits 5 + 7c; after >> 5 they
// are at 7c, and multiplying by sum_k 2^(36 - 6k) moves bit 7k to 36 + k with no two partial products on
// the same bit (7(i-i') = 6(k-k') has no solution with |k-k'| <= 6), hence no carries.
__device__ __forceinline__ uint32_t open_mul(uint64_t x, uint64_t o) {
  const uint64_t top = (1ull << 5) | (1ull << 12) | (1ull << 19) | (1ull << 26) | (1ull << 33) | (1ull << 40) | (1ull << 47);
  const uint64_t y = (~(x | o) & top) >> 5;
  const uint64_t M = (1ull << 36) | (1ull << 30) | (1ull << 24) | (1ull << 18) | (1ull << 12) | (1ull << 6) | 1ull;
  return (uint32_t)((y * M) >> 36) & 0x7Fu;
}
template <int V>
__device__ __forceinline__ void step_v(const G::Params& p, uint64_t& x, uint64_t& o, int a, uint32_t& m, uint32_t& st) {
  if (V == 5) { const uint32_t r = c4_fused_step(x, o, (uint32_t)a); m = r & 0xFF; st = r >> 8; return; }  // round 2: osg_c4_step.h
  if (V == 3) {  // the round-1 logic (osg_kernels.hip k_step_c4x2): result flags in plane 0's top byte
    G::State s = G::unpack(x, o);
    bool term = G::terminal(p, s);
    bool illegal = false;
    uint32_t open = G::open_columns(p, s);
    int to_move = G::plies(s) & 1;
    if (a != 0xFF) {
      if (!term && a < 32 && ((open >> a) & 1u)) {
        G::apply(p, s, a);
        term = G::terminal(p, s);
        open = G::open_columns(p, s);
        to_move ^= 1;
      } else illegal = true;
    }
    x = G::pack0(s); o = s.o;
    m = term ? 0u : open;
    st = enc(term, illegal, to_move, term ? G::outcome_code(p, s) : 0);
    return;
  }
  if (V == 2) { x ^= (uint64_t)a; o += 1; m = (uint32_t)x & 0x7F; st = (uint32_t)o & 0x7F; return; }
  G::State s{x, o, 0u};
  const int plies0 = G::plies(s);
  const int last = 1 - (plies0 & 1);              // only the player who moved last can own a line
  const bool win_last = G::line(p, last ? s.o : s.x);
  uint32_t open = open_mul(s.x, s.o);
  bool term = win_last | (open == 0);
  int outcome = win_last ? last : 2;
  bool illegal = false;
  if (a != 0xFF) {
    if (!term && a < 7 && ((open >> a) & 1u)) {
      const int mover = plies0 & 1;
      G::apply(p, s, a);
      const bool win = G::line(p, mover ? s.o : s.x);
      open = open_mul(s.x, s.o);
      term = win | (open == 0);
      outcome = win ? mover : 2;
    } else illegal = true;
  }
  x = s.x; o = s.o;
  m = term ? 0u : open;
  st = enc(term, illegal, term ? 0 : (G::plies(s) & 1), term ? outcome : 0);
}
__device__ __forceinline__ void step_one(const G::Params& p, uint64_t& x, uint64_t& o, int a, uint32_t& m, uint32_t& st) {
  G::State s{x, o, 0u};
  const bool win_x = G::line(p, s.x), win_o = G::line(p, s.o);
  bool term = win_x | win_o | G::full(p, s);
  int outcome = win_x ? 0 : (win_o ? 1 : 2);
  bool illegal = false;
  if (a != 0xFF) {
    const uint32_t open = term ? 0u : G::open_columns(p, s);
    if (a < 32 && ((open >> a) & 1u)) {
      const int mover = G::plies(s) & 1;
      G::apply(p, s, a);
      const bool win = G::line(p, mover ? s.o : s.x);
      term = win | G::full(p, s);
      outcome = win ? mover : 2;
    } else illegal = true;
  }
  x = s.x; o = s.o;
  m = term ? 0u : G::open_columns(p, s);
  st = enc(term, illegal, term ? 0 : (G::plies(s) & 1), term ? outcome : 0);
}

template <typename T> __device__ __forceinline__ T ld(const T* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <typename T> __device__ __forceinline__ void stg(T* p, T v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }

// S states per thread (S = 2, 4, 8): S/2 16-byte accesses per plane, S bytes of action/mask/status.
template <int S, int BLOCK, bool NT, int V = 0>
__global__ void __launch_bounds__(BLOCK) k(G::Params p, const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, int64_t n,
                                           const uint8_t* __restrict__ act, uint8_t* __restrict__ mask, uint8_t* __restrict__ status) {
  const int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * S;
  if (i >= n) return;
  uint64_t x[S], o[S];
  typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;
#pragma unroll
  for (int j = 0; j < S / 2; ++j) {
    u64x2 a = ld(reinterpret_cast<const u64x2*>(src + i) + j, NT);
    u64x2 b = ld(reinterpret_cast<const u64x2*>(src + n + i) + j, NT);
    x[2 * j] = a.x; x[2 * j + 1] = a.y; o[2 * j] = b.x; o[2 * j + 1] = b.y;
  }
  uint64_t av;
  if (S == 2) av = ld(reinterpret_cast<const uint16_t*>(act + i), NT);
  else if (S == 4) av = ld(reinterpret_cast<const uint32_t*>(act + i), NT);
  else av = ld(reinterpret_cast<const uint64_t*>(act + i), NT);
  uint64_t mv = 0, sv = 0;
#pragma unroll
  for (int j = 0; j < S; ++j) {
    uint32_t m, st;
    if (V == 0) step_one(p, x[j], o[j], (int)((av >> (8 * j)) & 0xFF), m, st); else step_v<V>(p, x[j], o[j], (int)((av >> (8 * j)) & 0xFF), m, st);
    mv |= (uint64_t)(m & 0xFF) << (8 * j);
    sv |= (uint64_t)st << (8 * j);
  }
#pragma unroll
  for (int j = 0; j < S / 2; ++j) {
    u64x2 a, b; a.x = x[2 * j]; a.y = x[2 * j + 1]; b.x = o[2 * j]; b.y = o[2 * j + 1];
    stg(reinterpret_cast<u64x2*>(dst + i) + j, a, NT);
    stg(reinterpret_cast<u64x2*>(dst + n + i) + j, b, NT);
  }
  if (S == 2) { stg(reinterpret_cast<uint16_t*>(mask + i), (uint16_t)mv, NT); stg(reinterpret_cast<uint16_t*>(status + i), (uint16_t)sv, NT); }
  else if (S == 4) { stg(reinterpret_cast<uint32_t*>(mask + i), (uint32_t)mv, NT); stg(reinterpret_cast<uint32_t*>(status + i), (uint32_t)sv, NT); }
  else { stg(reinterpret_cast<uint64_t*>(mask + i), mv, NT); stg(reinterpret_cast<uint64_t*>(status + i), sv, NT); }
}

// plain copy of the same bytes (the ceiling for this access pattern)
template <int BLOCK> __global__ void __launch_bounds__(BLOCK) k_copy(const uint4* __restrict__ s, uint4* __restrict__ d, int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n16) d[i] = s[i];
}

template <int S, int BLOCK, bool NT, int V = 0>
void run(const char* name, G::Params p, uint64_t* src, uint64_t* dst, int64_t n, uint8_t* act, uint8_t* mask, uint8_t* st, int iters) {
  const int64_t threads = n / S;
  dim3 grid((unsigned)((threads + BLOCK - 1) / BLOCK)), block(BLOCK);
  for (int i = 0; i < 50; ++i) k<S, BLOCK, NT, V><<<grid, block>>>(p, src, dst, n, act, mask, st);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) k<S, BLOCK, NT, V><<<grid, block>>>(p, src, dst, n, act, mask, st);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = ms * 1e3 / iters;
  printf("%-28s %8.3f us/launch  %7.1f GB/s algorithmic (35 B/state)  %.1f%% of 8 TB/s\n", name, us, 35.0 * n / us / 1e3, 35.0 * n / us / 1e3 / 80.0);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : (1 << 20);
  const int iters = 2000;
  G::Params p{}; p.words = 2; p.rows = 6; p.cols = 7; p.k = 4; p.ego = 0;
  uint64_t *src, *dst; uint8_t *act, *mask, *st;
  uint8_t*& st8 = st;
  CK(hipMalloc(&src, 16 * n)); CK(hipMalloc(&dst, 16 * n)); CK(hipMalloc(&act, n)); CK(hipMalloc(&mask, n)); CK(hipMalloc(&st, n));
  // mid-game-ish random positions are not needed for timing: the kernel has no data-dependent loops;
  // use sparse random boards with legal random actions.
  std::vector<uint64_t> h(2 * n, 0); std::vector<uint8_t> ha(n);
  uint64_t z = 12345;
  for (int64_t i = 0; i < n; ++i) { z = mix64(z + i); ha[i] = z % 7; int c1 = (z >> 8) % 7, c2 = (z >> 16) % 7; h[i] = 1ull << (c1 * 7); if (c2 != c1) h[n + i] = 1ull << (c2 * 7); }
  CK(hipMemcpy(src, h.data(), 16 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(act, ha.data(), n, hipMemcpyHostToDevice));
  printf("n = %lld states\n", (long long)n);
  run<2, 256, false>("S=2 B=256", p, src, dst, n, act, mask, st, iters);
  run<2, 512, false>("S=2 B=512", p, src, dst, n, act, mask, st, iters);
  run<2, 1024, false>("S=2 B=1024", p, src, dst, n, act, mask, st, iters);
  run<4, 256, false>("S=4 B=256", p, src, dst, n, act, mask, st, iters);
  run<4, 512, false>("S=4 B=512", p, src, dst, n, act, mask, st, iters);
  run<4, 128, false>("S=4 B=128", p, src, dst, n, act, mask, st, iters);
  run<8, 256, false>("S=8 B=256", p, src, dst, n, act, mask, st, iters);
  run<8, 128, false>("S=8 B=128", p, src, dst, n, act, mask, st, iters);
  run<8, 64, false>("S=8 B=64", p, src, dst, n, act, mask, st, iters);
  run<2, 256, false, 3>("S=2 B=256 stored-result", p, src, dst, n, act, mask, st, iters);
  run<2, 256, false, 5>("S=2 B=256 fused-step r2", p, src, dst, n, act, mask, st, iters);
  run<2, 512, false, 5>("S=2 B=512 fused-step r2", p, src, dst, n, act, mask, st, iters);
  run<2, 128, false, 5>("S=2 B=128 fused-step r2", p, src, dst, n, act, mask, st, iters);
  run<2, 512, false, 3>("S=2 B=512 stored-result", p, src, dst, n, act, mask, st, iters);
  run<4, 256, false, 3>("S=4 B=256 stored-result", p, src, dst, n, act, mask, st, iters);
  run<2, 256, true, 3>("S=2 B=256 stored-result nt", p, src, dst, n, act, mask, st, iters);
  run<2, 256, false, 1>("S=2 B=256 cheap-alu", p, src, dst, n, act, mask, st, iters);
  run<2, 1024, false, 1>("S=2 B=1024 cheap-alu", p, src, dst, n, act, mask, st, iters);
  run<4, 256, false, 1>("S=4 B=256 cheap-alu", p, src, dst, n, act, mask, st, iters);
  run<1 * 2, 256, false, 2>("S=2 B=256 memory-only", p, src, dst, n, act, mask, st, iters);
  run<4, 256, false, 2>("S=4 B=256 memory-only", p, src, dst, n, act, mask, st, iters);
  run<2, 256, true>("S=2 B=256 nt", p, src, dst, n, act, mask, st, iters);
  run<4, 256, true>("S=4 B=256 nt", p, src, dst, n, act, mask, st, iters);
  run<8, 256, true>("S=8 B=256 nt", p, src, dst, n, act, mask, st, iters);
  {  // copy ceiling: 35 B/state moved as 16-byte words (17.5 B in + 17.5 B out)
    const int64_t n16 = (35 * n / 2) / 16;
    uint4 *a, *b; CK(hipMalloc(&a, n16 * 16)); CK(hipMalloc(&b, n16 * 16));
    for (int i = 0; i < 50; ++i) k_copy<256><<<dim3((unsigned)((n16 + 255) / 256)), dim3(256)>>>(a, b, n16);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) k_copy<256><<<dim3((unsigned)((n16 + 255) / 256)), dim3(256)>>>(a, b, n16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("%-28s %8.3f us/launch  %7.1f GB/s (same bytes, plain uint4 copy)\n", "copy ceiling", us, 35.0 * n / us / 1e3);
    // empty kernel launch floor
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) k_copy<256><<<dim3(1), dim3(256)>>>(a, b, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.3f us/launch\n", "empty launch floor", ms * 1e3 / iters);
  }
  {  // hipGraph replay of the shipped logic: does a pre-built graph shorten the per-launch floor?
    hipStream_t st; CK(hipStreamCreate(&st));
    const int per_graph = 200;
    dim3 grid((unsigned)((n / 2 + 255) / 256)), block(256);
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < per_graph; ++i) k<2, 256, false, 3><<<grid, block, 0, st>>>(p, src, dst, n, act, mask, st8);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.3f us/launch (hipGraph of %d kernel nodes)\n", "S=2 B=256 stored-result graph", ms * 1e3 / (10 * per_graph), per_graph);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 2000; ++i) k<2, 256, false, 3><<<grid, block, 0, st>>>(p, src, dst, n, act, mask, st8);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.3f us/launch (plain launches, non-default stream)\n", "S=2 B=256 stored-result", ms * 1e3 / 2000);
  }
  {  // two (four) streams, each stepping its own share of the batch: do the boundaries between dependent launches overlap?
    for (int ns = 2; ns <= 4; ns *= 2) {
      hipStream_t ss[4];
      for (int i = 0; i < ns; ++i) CK(hipStreamCreate(&ss[i]));
      const int64_t h = n / ns;
      dim3 grid((unsigned)((h / 2 + 255) / 256)), block(256);
      auto round = [&]() {
        for (int i = 0; i < ns; ++i)
          k<2, 256, false, 5><<<grid, block, 0, ss[i]>>>(p, src + 2 * h * i, dst + 2 * h * i, h, act + h * i, mask + h * i, st8 + h * i);
      };
      for (int i = 0; i < 50; ++i) round();
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 2000; ++i) round();
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
      printf("%d streams x 1/%d batch        %8.3f us per full-batch step (wall)  %7.1f GB/s algorithmic\n", ns, ns, us, 35.0 * n / us / 1e3);
    }
    {
      dim3 grid((unsigned)((n / 2 + 255) / 256)), block(256);
      for (int i = 0; i < 50; ++i) k<2, 256, false, 5><<<grid, block>>>(p, src, dst, n, act, mask, st8);
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 2000; ++i) k<2, 256, false, 5><<<grid, block>>>(p, src, dst, n, act, mask, st8);
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
      printf("1 stream, whole batch         %8.3f us per full-batch step (wall)\n", us);
    }
  }
  return 0;
}
