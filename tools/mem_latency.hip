// mem_latency.hip -- how long is ONE dependent load on MI355X in the shape the finite-width re-slice
// issues them?  A wavefront holds 16 replica groups of 4 lanes; every group chases pointers inside
// its own REGION-byte piece of a big buffer (the replica's node blocks), one 16-byte load per step,
// each step depending on the previous one.  Reported: shader cycles (s_memtime) and nanoseconds per
// step, for a few footprints and occupancies.  The walk over a contraction tree is exactly this
// chain, so cycles/step here is the floor for cycles/node there.
//
//   hipcc -O3 --offload-arch=gfx950 -o build_variants/mem_latency tools/mem_latency.hip
//   build_variants/mem_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// every 128-byte slot of a region holds the number of the next slot (a random cycle per region)
__global__ void fill_kernel(int4* buf, int64_t n_regions, int slots, uint32_t seed) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_regions) return;
  // a permutation cycle: slot i -> (i * a + c) mod slots with a odd and slots a power of two
  const uint32_t a = ((seed + (uint32_t)r * 2654435761u) | 1u) * 4u + 1u, c = (uint32_t)r * 40503u | 1u;
  for (int i = 0; i < slots; ++i) {
    int4 v;
    v.x = (int)((i * a + c) & (uint32_t)(slots - 1));
    v.y = v.z = v.w = 0;
    buf[(r * slots + i) * 8] = v;  // 128-byte slots
  }
}

__global__ __launch_bounds__(256) void chase_kernel(const int4* buf, int64_t n_regions, int slots, int steps,
                                                    unsigned long long* cycles, int* sink) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (r >= n_regions) return;
  const int4* base = buf + r * slots * 8;
  int x = (int)(r & (slots - 1));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
    const int4 v = base[(int64_t)x * 8];
    x = v.x;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) atomicAdd(cycles, t1 - t0);
  if (x == -1) *sink = 1;
}

int main() {
  const int steps = 2000;
  unsigned long long* d_cycles;
  int* d_sink;
  CHECK(hipMalloc(&d_cycles, 8));
  CHECK(hipMalloc(&d_sink, 4));
  printf("%10s %10s %10s %12s %12s %10s\n", "regions", "region_KB", "GB", "cycles/step", "ns/step", "waves/SIMD");
  const int slots_list[] = {512, 1024};            // x 128 B = 64 KB / 128 KB per replica
  const int64_t regions_list[] = {1024, 16384, 32768, 65536, 131072};
  for (int slots : slots_list) {
    for (int64_t regions : regions_list) {
      const size_t bytes = (size_t)regions * slots * 128;
      if (bytes > (size_t)20 << 30) continue;
      int4* buf;
      CHECK(hipMalloc(&buf, bytes));
      hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((regions + 255) / 256)), dim3(256), 0, 0, buf, regions, slots, 12345u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemset(d_cycles, 0, 8));
      const int64_t threads = regions * 4;
      hipEvent_t a, b;
      CHECK(hipEventCreate(&a));
      CHECK(hipEventCreate(&b));
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(chase_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, buf, regions, slots, steps,
                         d_cycles, d_sink);
      CHECK(hipEventRecord(b));
      CHECK(hipDeviceSynchronize());
      float ms;
      CHECK(hipEventElapsedTime(&ms, a, b));
      unsigned long long cyc;
      CHECK(hipMemcpy(&cyc, d_cycles, 8, hipMemcpyDeviceToHost));
      const double waves = (double)threads / 64.0;
      // resident waves per SIMD: 1024 SIMDs, at most 8 per SIMD (no LDS, few registers)
      const double per_simd = waves / 1024.0 < 8 ? waves / 1024.0 : 8;
      const double rounds = waves / 1024.0 / per_simd;
      printf("%10lld %10d %10.2f %12.0f %12.0f %10.1f\n", (long long)regions, slots * 128 / 1024, bytes / 1e9,
             (double)cyc / waves / steps, ms * 1e6 / steps / rounds, per_simd);
      CHECK(hipFree(buf));
    }
  }
  return 0;
}
