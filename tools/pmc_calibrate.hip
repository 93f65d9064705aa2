// pmc_calibrate.hip -- known byte counts in the access patterns of the sweep kernels, to calibrate what rocprofv3's
// memory-side counters (FETCH_SIZE, WRITE_SIZE, TCC_EA0_RDREQ[_32B], TCC_EA0_WRREQ[_64B]) report for them.
// /opt/skills/guides/MI355X_MICROARCH.md ("HBM"): FETCH_SIZE tallies a 128-byte request at 64 bytes -- established for
// wide streaming reads; "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern".  The sweep kernels read and write RANDOM 32- / 64- / 128-byte pieces (groups of 4 lanes), so that is what
// runs here, one kernel per pattern and size, each moving exactly N_ACCESSES * size bytes of an 8-GiB buffer
// (past the 256-MiB Infinity Cache).  tools/pmc_calibrate.py runs it under rocprofv3 and tabulates bytes counted /
// bytes moved per counter.
//
//   hipcc -O3 --offload-arch=gfx950 -o build_variants/pmc_calibrate tools/pmc_calibrate.hip
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

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

constexpr int BLOCKS = 2048, ITERS = 256;  // x 64 groups per block = 33.5 M accesses per kernel

// groups of 4 lanes read one random, naturally aligned GRAIN-byte piece per iteration (4 in flight)
template <int GRAIN>
__global__ __launch_bounds__(256) void cal_random_read(const uint8_t* buf, uint64_t n_grains, uint64_t* sink) {
  const int lane = threadIdx.x & 3;
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  constexpr int WPL = GRAIN / 32;  // 64-bit words per lane
  uint64_t acc = 0;
  for (int it = 0; it < ITERS; it += 4) {
    uint64_t v[4][WPL];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t g = mix(gid * 0x9e3779b97f4a7c15ull + (uint64_t)(it + u)) % n_grains;
      const uint64_t* p = reinterpret_cast<const uint64_t*>(buf + g * GRAIN + lane * (GRAIN / 4));
#pragma unroll
      for (int w = 0; w < WPL; ++w) v[u][w] = p[w];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int w = 0; w < WPL; ++w) acc += v[u][w];
  }
  if (acc == 0x1234567) sink[0] = acc;
}

// ... write one (GRAIN = 4: lane 0 alone writes 4 bytes into a random 128-byte block -- a parent-link update)
template <int GRAIN>
__global__ __launch_bounds__(256) void cal_random_write(uint8_t* buf, uint64_t n_grains) {
  const int lane = threadIdx.x & 3;
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  for (int it = 0; it < ITERS; ++it) {
    const uint64_t g = mix(gid * 0x9e3779b97f4a7c15ull + (uint64_t)it) % n_grains;
    if constexpr (GRAIN == 4) {
      if (lane == 0) *reinterpret_cast<uint32_t*>(buf + g * 128 + 8) = (uint32_t)g;
    } else {
      constexpr int WPL = GRAIN / 32;
      uint64_t* p = reinterpret_cast<uint64_t*>(buf + g * GRAIN + lane * (GRAIN / 4));
#pragma unroll
      for (int w = 0; w < WPL; ++w) p[w] = g + w;
    }
  }
}

// wide streaming read / write, 16 bytes per lane: the guide's calibrated case
__global__ __launch_bounds__(256) void cal_stream_read(const uint4* buf, uint64_t n16, uint64_t* sink) {
  uint64_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = buf[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 0x1234567) sink[0] = acc;
}
__global__ __launch_bounds__(256) void cal_stream_write(uint4* buf, uint64_t n16) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
    buf[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 8;
  const size_t bytes = gib << 30;
  uint8_t* buf;
  uint64_t* sink;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipDeviceSynchronize());
  const double acc = (double)BLOCKS * 64 * ITERS;
  const uint64_t n16 = (uint64_t)2 << 26;  // 2 GiB streamed
  printf("accesses per random kernel: %.0f; streamed bytes: %.0f\n", acc, (double)n16 * 16);
  cal_random_read<32><<<BLOCKS, 256>>>(buf, bytes / 32, sink);
  cal_random_read<64><<<BLOCKS, 256>>>(buf, bytes / 64, sink);
  cal_random_read<128><<<BLOCKS, 256>>>(buf, bytes / 128, sink);
  cal_random_write<4><<<BLOCKS, 256>>>(buf, bytes / 128);
  cal_random_write<32><<<BLOCKS, 256>>>(buf, bytes / 32);
  cal_random_write<64><<<BLOCKS, 256>>>(buf, bytes / 64);
  cal_random_write<128><<<BLOCKS, 256>>>(buf, bytes / 128);
  cal_stream_read<<<4096, 256>>>(reinterpret_cast<const uint4*>(buf), n16, sink);
  cal_stream_write<<<4096, 256>>>(reinterpret_cast<uint4*>(buf), n16);
  CHECK(hipDeviceSynchronize());
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}
