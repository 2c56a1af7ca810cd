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
  const int64_t sizes[] = {int64_t{176} << 20, int64_t{528} << 20, int64_t{2} << 30, int64_t{8456} << 20};  // [2^20,42], [2^20,126], 2 GiB, [2^24,126]
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
