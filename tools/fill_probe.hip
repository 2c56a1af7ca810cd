// The write-only ceiling of the chip: what a kernel that ONLY stores (the tensor packers write 42-729 floats per state
// and read 8-52 bytes) can reach.  float4 stores, ordinary and non-temporal, grid-stride and one-piece-per-thread,
// at the sizes of the tensor rows in tools/probe_kernels.py; hipMemsetAsync beside them.
//   hipcc --offload-arch=gfx950 -O3 tools/fill_probe.hip -o tools/fill_probe && ./tools/fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool kNt>
__device__ inline void st4(float4* p, float4 v) {
  if (kNt) {
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
  } else {
    *p = v;
  }
}
// one 16-byte piece per thread
template <bool kNt>
__global__ void __launch_bounds__(256) k_fill1(float4* out, int64_t n16, float v) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n16) st4<kNt>(out + i, make_float4(v, v, v, v));
}
// a wavefront owns a contiguous span of kPieces KiB (the tensor kernels' shape: 64 lanes x 16 B per store instruction)
template <bool kNt, int kPieces>
__global__ void __launch_bounds__(256) k_fill_span(float4* out, int64_t n16, float v) {
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  float4* base = out + wave * (64 * kPieces);
#pragma unroll
  for (int j = 0; j < kPieces; ++j) {
    const int64_t i = wave * (64 * kPieces) + lane + 64 * j;
    if (i < n16) st4<kNt>(base + lane + 64 * j, make_float4(v, v, v, v));
  }
}
// span stores with a throttle: at most kWait stores of a lane in flight (s_waitcnt vmcnt), and the launch's dynamic LDS
// caps the wavefronts per CU — how many stores does the chip want in flight?
template <int kPieces, int kWait>
__global__ void __launch_bounds__(256) k_fill_throttle(float4* out, int64_t n16, float v) {
  extern __shared__ float lds_cap[];
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  float4* base = out + wave * (64 * kPieces);
  if (v == 12345.f) lds_cap[threadIdx.x] = v;   // keep the allocation
#pragma unroll
  for (int j = 0; j < kPieces; ++j) {
    const int64_t i = wave * (64 * kPieces) + lane + 64 * j;
    if (i < n16) st4<false>(base + lane + 64 * j, make_float4(v, v, v, v));
    if (kWait == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kWait == 2 && (j & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kWait == 4 && (j & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
// one piece per thread with other group sizes / a shifted group -> chunk map: is the 0.86 a property of WHICH XCD writes
// which 4 KiB granule (groups go to XCDs round-robin: with 256 threads, XCD k writes the granules = k mod 8)?
template <int kBlock>
__global__ void __launch_bounds__(kBlock) k_fill1_b(float4* out, int64_t n16, float v, int shift) {
  int64_t b = blockIdx.x + shift;
  if (b >= gridDim.x) b -= gridDim.x;
  const int64_t i = b * kBlock + threadIdx.x;
  if (i < n16) st4<false>(out + i, make_float4(v, v, v, v));
}
// kPieces stores per lane, but every store of group b goes to a granule (kGran KiB) with the same index mod 8 as b:
// granule = (b % 8) + 8 * ((b / 8) * kPieces' + j)
template <int kPieces, int kGranKiB>
__global__ void __launch_bounds__(256) k_fill_xcd(float4* out, int64_t n16, float v) {
  // a 256-thread group stores 4 KiB per instruction round; a granule of kGranKiB takes kGranKiB / 4 rounds (>= 1)
  constexpr int kRoundsPerGran = kGranKiB >= 4 ? kGranKiB / 4 : 1;
  const int64_t b = blockIdx.x;
  const int64_t slot = b & 7, row = b >> 3;
#pragma unroll
  for (int j = 0; j < kPieces; ++j) {
    const int64_t gran = slot + 8 * (row * (kPieces / kRoundsPerGran) + j / kRoundsPerGran);
    const int64_t i = gran * (kGranKiB * 64) + (j % kRoundsPerGran) * 256 + threadIdx.x;
    if (i < n16) st4<false>(out + i, make_float4(v, v, v, v));
  }
}
// the workgroup owns a contiguous region of W x kPieces KiB; at store j its W wavefronts write W CONSECUTIVE KiB
// (wavefront w writes piece w + W * j), so the chip's in-flight stores stay dense while each lane still does kPieces
template <bool kNt, int kPieces, int kBlock>
__global__ void __launch_bounds__(kBlock) k_fill_wg(float4* out, int64_t n16, float v) {
  constexpr int W = kBlock / 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * (kBlock * kPieces);
#pragma unroll
  for (int j = 0; j < kPieces; ++j) {
    const int64_t i = base + (wave + W * j) * 64 + lane;
    if (i < n16) st4<kNt>(out + i, make_float4(v, v, v, v));
  }
}
// grid-interleaved: thread t writes pieces t, t + T, t + 2T ... (T = all threads of the launch)
template <bool kNt, int kPieces>
__global__ void __launch_bounds__(256) k_fill_grid(float4* out, int64_t n16, float v) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x, T = static_cast<int64_t>(gridDim.x) * 256;
#pragma unroll
  for (int j = 0; j < kPieces; ++j) {
    const int64_t i = t + T * j;
    if (i < n16) st4<kNt>(out + i, make_float4(v, v, v, v));
  }
}
// read 16 B per 512 B written (the c4 tensor's ratio), so that the read stream is in the picture as well
template <bool kNt>
__global__ void __launch_bounds__(256) k_expand(const float4* in, float4* out, int64_t n_in) {
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t i = wave * 64 + lane;
  if (i >= n_in) return;
  const float4 x = in[i];
  float4* base = out + wave * (64 * 32);
#pragma unroll
  for (int j = 0; j < 32; ++j) st4<kNt>(base + lane + 64 * j, make_float4(x.x + j, x.y, x.z, x.w));
}

template <class F>
static double time_us(F launch, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / iters;
}

int main() {
  const int64_t sizes[] = {int64_t{528} << 20, int64_t{8456} << 20};  // [2^20,42], [2^20,126], 2 GiB, [2^24,126]
  for (int64_t bytes : sizes) {
    float4* buf;
    CHECK(hipMalloc(&buf, bytes + (1 << 20)));
    const int64_t n16 = bytes / 16;
    const int iters = bytes > (int64_t{1} << 31) ? 10 : 40;
    auto report = [&](const char* name, double us) {
      printf("%-34s %8.1f MB  %9.1f us  %7.1f GB/s  %.3f of 8 TB/s\n", name, bytes / 1e6, us, bytes / us / 1e3, bytes / us / 8e6);
      fflush(stdout);
    };
    const unsigned g1 = static_cast<unsigned>((n16 + 255) / 256);
    report("fill, one piece/thread", time_us([&] { k_fill1<false><<<g1, 256>>>(buf, n16, 1.f); }, iters));
    report("fill, one piece/thread, nt", time_us([&] { k_fill1<true><<<g1, 256>>>(buf, n16, 1.f); }, iters));
    const unsigned g11 = static_cast<unsigned>((n16 + 256 * 11 - 1) / (256 * 11));
    report("fill, 11 KiB per wavefront", time_us([&] { k_fill_span<false, 11><<<g11, 256>>>(buf, n16, 1.f); }, iters));
    report("fill, 11 KiB per wavefront, nt", time_us([&] { k_fill_span<true, 11><<<g11, 256>>>(buf, n16, 1.f); }, iters));
    const unsigned g32 = static_cast<unsigned>((n16 + 256 * 32 - 1) / (256 * 32));
    report("fill, 32 KiB per wavefront, nt", time_us([&] { k_fill_span<true, 32><<<g32, 256>>>(buf, n16, 1.f); }, iters));
    const unsigned g2 = static_cast<unsigned>((n16 + 256 * 2 - 1) / (256 * 2)), g4 = static_cast<unsigned>((n16 + 256 * 4 - 1) / (256 * 4));
    report("fill, 2 KiB per wavefront", time_us([&] { k_fill_span<false, 2><<<g2, 256>>>(buf, n16, 1.f); }, iters));
    report("fill, 4 KiB per wavefront", time_us([&] { k_fill_span<false, 4><<<g4, 256>>>(buf, n16, 1.f); }, iters));
    report("fill, 32 KiB per wavefront", time_us([&] { k_fill_span<false, 32><<<g32, 256>>>(buf, n16, 1.f); }, iters));
    report("wg-interleaved 11, 256 threads", time_us([&] { k_fill_wg<false, 11, 256><<<g11, 256>>>(buf, n16, 1.f); }, iters));
    report("wg-interleaved 11, 256 thr, nt", time_us([&] { k_fill_wg<true, 11, 256><<<g11, 256>>>(buf, n16, 1.f); }, iters));
    report("wg-interleaved 11, 128 threads", time_us([&] { k_fill_wg<false, 11, 128><<<static_cast<unsigned>((n16 + 128 * 11 - 1) / (128 * 11)), 128>>>(buf, n16, 1.f); }, iters));
    report("wg-interleaved 11, 1024 threads", time_us([&] { k_fill_wg<false, 11, 1024><<<static_cast<unsigned>((n16 + 1024 * 11 - 1) / (1024 * 11)), 1024>>>(buf, n16, 1.f); }, iters));
    report("wg-interleaved 32, 256 threads", time_us([&] { k_fill_wg<false, 32, 256><<<g32, 256>>>(buf, n16, 1.f); }, iters));
    report("grid-interleaved 11", time_us([&] { k_fill_grid<false, 11><<<g11, 256>>>(buf, n16, 1.f); }, iters));
    report("grid-interleaved 11, nt", time_us([&] { k_fill_grid<true, 11><<<g11, 256>>>(buf, n16, 1.f); }, iters));
    report("grid-interleaved 4", time_us([&] { k_fill_grid<false, 4><<<g4, 256>>>(buf, n16, 1.f); }, iters));
    report("one piece, 64 threads", time_us([&] { k_fill1_b<64><<<static_cast<unsigned>((n16 + 63) / 64), 64>>>(buf, n16, 1.f, 0); }, iters));
    report("one piece, 128 threads", time_us([&] { k_fill1_b<128><<<static_cast<unsigned>((n16 + 127) / 128), 128>>>(buf, n16, 1.f, 0); }, iters));
    report("one piece, 512 threads", time_us([&] { k_fill1_b<512><<<static_cast<unsigned>((n16 + 511) / 512), 512>>>(buf, n16, 1.f, 0); }, iters));
    report("one piece, 1024 threads", time_us([&] { k_fill1_b<1024><<<static_cast<unsigned>((n16 + 1023) / 1024), 1024>>>(buf, n16, 1.f, 0); }, iters));
    for (int shift : {1, 2, 3, 4, 5, 8}) {
      char name[96];
      snprintf(name, sizeof name, "one piece, 256 thr, chunk = b + %d", shift);
      report(name, time_us([&] { k_fill1_b<256><<<g1, 256>>>(buf, n16, 1.f, shift); }, iters));
    }
    report("xcd-aligned 4 KiB granules, 8 stores", time_us([&] { k_fill_xcd<8, 4><<<static_cast<unsigned>((n16 + 2047) / 2048), 256>>>(buf, n16, 1.f); }, iters));
    report("xcd-aligned 4 KiB granules, 16 stores", time_us([&] { k_fill_xcd<16, 4><<<static_cast<unsigned>((n16 + 4095) / 4096), 256>>>(buf, n16, 1.f); }, iters));
    report("xcd-aligned 8 KiB granules, 8 stores", time_us([&] { k_fill_xcd<8, 8><<<static_cast<unsigned>((n16 + 2047) / 2048), 256>>>(buf, n16, 1.f); }, iters));
    report("xcd-aligned 16 KiB granules, 8 stores", time_us([&] { k_fill_xcd<8, 16><<<static_cast<unsigned>((n16 + 2047) / 2048), 256>>>(buf, n16, 1.f); }, iters));
    for (int lds_kb : {0}) {   // 256-thread groups: 8 / 8 / 4 / 2 per CU at 160 KB of LDS -> 8 / 8 / 4 / 2 waves per SIMD ... (0 and 20: no cap)
      char name[96];
      snprintf(name, sizeof name, "span 11, %d KB LDS per group", lds_kb);
      report(name, time_us([&] { k_fill_throttle<11, 0><<<g11, 256, lds_kb * 1024>>>(buf, n16, 1.f); }, iters));
      snprintf(name, sizeof name, "span 11, wait every store, %d KB", lds_kb);
      report(name, time_us([&] { k_fill_throttle<11, 1><<<g11, 256, lds_kb * 1024>>>(buf, n16, 1.f); }, iters));
      snprintf(name, sizeof name, "span 11, wait every 4, %d KB", lds_kb);
      report(name, time_us([&] { k_fill_throttle<11, 4><<<g11, 256, lds_kb * 1024>>>(buf, n16, 1.f); }, iters));
    }
    report("hipMemsetAsync", time_us([&] { hipMemsetAsync(buf, 0, bytes, 0); }, iters));
    float4* in;
    const int64_t n_in = n16 / 32;
    CHECK(hipMalloc(&in, n_in * 16 + 4096));
    hipMemset(in, 0, n_in * 16);
    const unsigned ge = static_cast<unsigned>((n_in + 255) / 256);
    report("expand 16 B -> 512 B / lane", time_us([&] { k_expand<false><<<ge, 256>>>(in, buf, n_in); }, iters));
    report("expand 16 B -> 512 B / lane, nt", time_us([&] { k_expand<true><<<ge, 256>>>(in, buf, n_in); }, iters));
    hipFree(in);
    hipFree(buf);
  }
  return 0;
}
