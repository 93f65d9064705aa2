// few_lanes.hip -- does a wavefront with few active lanes run slower when many such wavefronts share the chip?
//
// The LDS-resident sweep kernels (csrc/sa_small.h) measured up to twice fewer iterations per second per wavefront when
// 512 or 1024 one-wavefront blocks of 1-4 replicas (4-16 active lanes) were resident than with 256 of them, and no such
// loss with 16 replicas per wavefront (profiles/experiments_r05.md).  This probe takes the sweep kernel out of the
// picture: one-wavefront blocks run a fixed number of iterations of (a) a dependent chain of vector instructions, (b)
// the same with a divergent branch structure -- three sections, each entered by a subset of the lane groups, skipped
// when none needs it -- and (c) the same with 39 KiB of LDS per block (so that four blocks fill a CU as the small-tree
// kernel's do); `active` lane groups of four lanes work, the others idle.  Printed: nanoseconds and shader cycles
// (s_memtime) per iteration against the number of blocks and of active groups.
//
//   hipcc -O3 --offload-arch=gfx950 -o few_lanes tools/few_lanes.hip && ./few_lanes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void probe(uint32_t* out, unsigned long long* cyc, int iters, int active) {
  __shared__ uint32_t lds[MODE == 2 ? 9856 : 64];
  const int tid = threadIdx.x, gib = tid >> 2;
  uint32_t x = tid * 2654435761u + blockIdx.x, y = 1;
  lds[tid] = x;
  const bool on = gib < active;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (on) {
    for (int i = 0; i < iters; ++i) {
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 200; ++k) { x = x * 1664525u + 1013904223u; y ^= x >> 7; }
      } else {
        // three sections; a group is "in" a section on its own schedule (as replicas are at their own places of a sweep)
        const uint32_t ph = (uint32_t)(i + gib * 5);
        if (ph % 5u < 2u) {
#pragma unroll
          for (int k = 0; k < 40; ++k) { x = x * 1664525u + 1013904223u; y ^= x >> 7; }
          lds[tid] = x;
        }
        {
#pragma unroll
          for (int k = 0; k < 100; ++k) { x = x * 1664525u + 1013904223u; y ^= x >> 7; }
          if (y & 1u) {
#pragma unroll
            for (int k = 0; k < 10; ++k) y = y * 22695477u + 1u;
            lds[tid] = y;
          }
          x += lds[tid ^ 1];
        }
        if (ph % 7u == 0u) {
#pragma unroll
          for (int k = 0; k < 60; ++k) { x = x * 1664525u + 1013904223u; y ^= x >> 7; }
          x += lds[tid ^ 2];
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + tid] = x ^ y;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* what, uint32_t* out, unsigned long long* cyc, int iters) {
  printf("## %s\n| blocks | active groups per wavefront | ns / iteration | s_memtime ticks / iteration | ticks per ns |\n|---|---|---|---|---|\n", what);
  for (int blocks : {64, 256, 512, 1024}) {
    for (int active : {1, 2, 4, 16}) {
      hipEvent_t a, b;
      CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
      hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters / 10, active);  // warm-up
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, active);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, a, b));
      unsigned long long h[1024];
      CHECK(hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
      double c = 0;
      for (int i = 0; i < blocks; ++i) c += (double)h[i];
      c /= blocks;
      printf("| %d | %d | %.1f | %.1f | %.2f |\n", blocks, active, ms * 1e6 / iters, c / iters, c / (ms * 1e6));
      CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    }
  }
}

int main() {
  uint32_t* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, 1024 * 64 * 4));
  CHECK(hipMalloc(&cyc, 1024 * 8));
  const int iters = 20000;
  run<0>("a dependent chain of 400 vector instructions per iteration", out, cyc, iters);
  run<1>("three sections behind divergent branches", out, cyc, iters);
  run<2>("the same with 39 KiB of LDS per block (four blocks fill a CU)", out, cyc, iters);
  return 0;
}
