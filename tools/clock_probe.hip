// What clock does the chip actually hold under a saturating load, and what are the issue rates?
// Every SIMD gets 8 wavefronts, each running a dependent chain of N instructions of one kind; with the
// SIMD saturated, cycles = 8 * N * (issue interval), so the measured time gives interval / clock.
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe && ./tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIters = 1 << 16;   // x 16 instructions per iteration

__global__ void __launch_bounds__(256) k_valu(uint32_t* out, uint32_t seed) {
  uint32_t v = threadIdx.x + seed;
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(seed));
  }
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
__global__ void __launch_bounds__(256) k_mul(uint32_t* out, uint32_t seed) {
  uint32_t v = threadIdx.x + seed;
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v) : "v"(seed));
  }
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
__global__ void __launch_bounds__(256) k_salu(uint32_t* out, uint32_t seed) {
  uint32_t s = seed + blockIdx.x;
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(seed) : "scc");
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_mix(uint32_t* out, uint32_t seed) {  // VALU and SALU interleaved
  uint32_t v = threadIdx.x + seed, s = seed;
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(seed));
      asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(seed) : "scc");
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = v + s;
}

__global__ void __launch_bounds__(256) k_shr64(uint32_t* out, uint32_t seed) {
  uint64_t v = (static_cast<uint64_t>(threadIdx.x + seed) << 32) | 0xFFFFFFFFull;
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(v));
  }
  out[blockIdx.x * 256 + threadIdx.x] = static_cast<uint32_t>(v);
}
__global__ void __launch_bounds__(256) k_alignbit(uint32_t* out, uint32_t seed) {
  uint32_t v = threadIdx.x + seed;
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(v) : "v"(seed));
  }
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
__global__ void __launch_bounds__(256) k_mul24(uint32_t* out, uint32_t seed) {
  uint32_t v = threadIdx.x + seed;
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v) : "v"(seed));
  }
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
__global__ void __launch_bounds__(256) k_add64(uint32_t* out, uint32_t seed) {
  uint64_t v = threadIdx.x + seed;
  const uint64_t w = seed;
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(v) : "v"(w));
  }
  out[blockIdx.x * 256 + threadIdx.x] = static_cast<uint32_t>(v);
}

template <class K>
double time_ms(K kernel, uint32_t* out, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  kernel<<<blocks, 256>>>(out, 1u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  kernel<<<blocks, 256>>>(out, 3u);
  hipEventRecord(b);
  hipEventSynchronize(b);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) printf("  (launch error: %s)\n", hipGetErrorString(err));
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
  uint32_t* out;
  hipMalloc(&out, sizeof(uint32_t) * blocks * 256);
  const double n = static_cast<double>(kIters) * 16;
  printf("%s: %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  struct { const char* name; double ms; double per_simd; } rows[] = {
      {"v_add_u32 (dependent, 8 waves/SIMD)", time_ms(k_valu, out, blocks), 8 * n},
      {"v_mul_lo_u32 (dependent, 8 waves/SIMD)", time_ms(k_mul, out, blocks), 8 * n},
      {"s_add_u32 (dependent, 8 waves/SIMD)", time_ms(k_salu, out, blocks), 8 * n},
      {"v_add + s_add interleaved (8 + 8 per 16)", time_ms(k_mix, out, blocks), 8 * n},
      {"v_lshrrev_b64 (dependent, 8 waves/SIMD)", time_ms(k_shr64, out, blocks), 8 * n},
      {"v_alignbit_b32 (dependent, 8 waves/SIMD)", time_ms(k_alignbit, out, blocks), 8 * n},
      {"v_mul_u32_u24 (dependent, 8 waves/SIMD)", time_ms(k_mul24, out, blocks), 8 * n},
      {"v_lshl_add_u64 (dependent, 8 waves/SIMD)", time_ms(k_add64, out, blocks), 8 * n},
  };
  for (auto& r : rows)
    printf("%-44s %8.3f ms  -> %.3f ns per instruction per SIMD (= issue interval / clock); at 4 cycles: %.2f GHz\n",
           r.name, r.ms, r.ms * 1e6 / r.per_simd, 4.0 / (r.ms * 1e6 / r.per_simd));
  // one wave per SIMD: the dependent-issue latency of a single wave
  const double one = time_ms(k_valu, out, cus);
  printf("%-44s %8.3f ms  -> %.3f ns per instruction (single wave per SIMD)\n", "v_add_u32, 1 wave/SIMD", one, one * 1e6 / n);
  return 0;
}
