// hbm_random.hip -- what does MI355X HBM deliver for the access pattern of the sweep kernel?
//
// The kernel touches, per move, a few RANDOM 128-byte node blocks of a multi-GB working set (no
// reuse: 65536 replicas x ~70 KB); each block is accessed by the 4 lanes of one replica group.
// This microbenchmark issues exactly that: groups of 4 lanes read (and optionally write back) random
// GRAIN-byte pieces (32 / 64 / 128 B, naturally aligned) of a large buffer, many independent requests
// in flight, full occupancy.  The GB/s it reaches is the practical roofline for the pattern, to set
// beside the 8 TB/s streaming peak used in bench.py's `roofline` object.
//
//   hipcc -O3 --offload-arch=gfx950 -o hbm_random tools/hbm_random.hip && ./hbm_random
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

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

// GRAIN bytes per group access, 4 lanes per group -> GRAIN/4 bytes per lane (8, 16 or 32).
// MODE 0: read only; 1: read then write the same piece back; 2: write only.
template <int GRAIN, int MODE, int UNROLL>
__global__ __launch_bounds__(256) void rnd_kernel(uint8_t* buf, uint64_t n_grains, int iters, uint64_t* sink) {
  const int lane = threadIdx.x & 3;
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  constexpr int BPL = GRAIN / 4;  // bytes per lane
  constexpr int WPL = BPL / 8;    // 64-bit words per lane
  uint64_t acc = 0;
  for (int it = 0; it < iters; it += UNROLL) {
    uint64_t v[UNROLL][WPL > 0 ? WPL : 1];
    uint64_t* p[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint64_t g = mix(gid * 0x9e3779b97f4a7c15ull + (uint64_t)(it + u)) % n_grains;
      p[u] = reinterpret_cast<uint64_t*>(buf + g * GRAIN + lane * BPL);
      if (MODE != 2) {
#pragma unroll
        for (int w = 0; w < WPL; ++w) v[u][w] = p[u][w];
      } else {
#pragma unroll
        for (int w = 0; w < WPL; ++w) v[u][w] = g + w;
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int w = 0; w < WPL; ++w) {
        acc += v[u][w];
        if (MODE != 0) p[u][w] = v[u][w] + 1;
      }
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

// The memory side of ONE move of the sweep kernel and nothing else (tnco_amd/csrc/sa_sweep.h):
//   read  32 B  header of an ancestor            (random 128-B node block X)
//   read 128 B  legs + partial cost of a sibling (random block Y)
//   write 32 B header (+ 96 B legs when the move is accepted: ACC of 4 moves) of block Z
//   write  4 B  parent field of two further blocks U, V (accepted moves only)
// PARENTS = 0: the two parent writes are left out (what a layout without per-node parent pointers
// would cost); 1: 4-byte writes; 2: every write is a whole aligned 64-byte piece (the parent field
// together with the rest of the first half of the block, and the header together with legs 0..3).
// moves/s of this kernel = the memory system's ceiling for the pattern.
// WIDE = 1: the block of an accepted move written by TWO instructions of 64 bytes per group (16 bytes per lane)
// instead of FOUR of 32 (the kernel's word-interleaved lanes: one 32-byte sector per instruction) -- fewer write
// requests for the same bytes.
template <int UNROLL, int ACC, int PARENTS, int WIDE = 0>
__global__ __launch_bounds__(256) void move_pattern_kernel(uint8_t* buf, uint64_t n_lines, int iters, uint64_t* sink) {
  const int lane = threadIdx.x & 3;
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  uint64_t acc = 0;
  for (int it = 0; it < iters; it += UNROLL) {
    uint64_t hx[UNROLL], y[UNROLL][4];
    uint64_t base[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      base[u] = mix(gid * 0x9e3779b97f4a7c15ull + (uint64_t)(it + u));
      const uint64_t X = base[u] % n_lines, Y = mix(base[u] + 1) % n_lines;
      hx[u] = *reinterpret_cast<const uint64_t*>(buf + X * 128 + lane * 8);
#pragma unroll
      for (int w = 0; w < 4; ++w) y[u][w] = *reinterpret_cast<const uint64_t*>(buf + Y * 128 + w * 32 + lane * 8);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint64_t Z = mix(base[u] + 2) % n_lines, U = mix(base[u] + 3) % n_lines, V = mix(base[u] + 4) % n_lines;
      const uint64_t s = hx[u] + y[u][0] + y[u][1] + y[u][2] + y[u][3];
      acc += s;
      const bool accepted = ((it + u) & 3) < ACC;
      if (WIDE) {
        if (accepted) {
          *reinterpret_cast<uint4*>(buf + Z * 128 + lane * 16) = make_uint4((uint32_t)s, (uint32_t)(s >> 32), 1, 2);
          *reinterpret_cast<uint4*>(buf + Z * 128 + 64 + lane * 16) = make_uint4((uint32_t)s, (uint32_t)(s >> 32), 3, 4);
        } else if (lane < 2) {
          *reinterpret_cast<uint4*>(buf + Z * 128 + lane * 16) = make_uint4((uint32_t)s, (uint32_t)(s >> 32), 1, 2);
        }
      } else {
        *reinterpret_cast<uint64_t*>(buf + Z * 128 + lane * 8) = s;
        if (PARENTS == 2) *reinterpret_cast<uint64_t*>(buf + Z * 128 + 32 + lane * 8) = s + 1;
      }
      if (accepted) {
        if (!WIDE) {
#pragma unroll
          for (int w = (PARENTS == 2 ? 2 : 1); w < 4; ++w)
            *reinterpret_cast<uint64_t*>(buf + Z * 128 + w * 32 + lane * 8) = s + w;
        }
        if (PARENTS == 1 && lane == 0) {
          *reinterpret_cast<uint32_t*>(buf + U * 128 + 8) = (uint32_t)s;
          *reinterpret_cast<uint32_t*>(buf + V * 128 + 8) = (uint32_t)s + 1;
        }
        if (PARENTS == 2) {
          *reinterpret_cast<uint4*>(buf + U * 128 + lane * 16) = make_uint4((uint32_t)s, 1, 2, 3);
          *reinterpret_cast<uint4*>(buf + V * 128 + lane * 16) = make_uint4((uint32_t)s, 4, 5, 6);
        }
      }
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

// The memory side of one move of the FINITE-WIDTH sweep kernel in the split layout (headers 32 B each in one array, legs
// 128 B each in another): read A's header, C's header (CHDR = 1; 0: a parent that carried its children's partial sums would
// make this read unnecessary -- profiles/design_notes_r01_r05.md section 9, "next"), C's legs line; write B's header (a 32-byte partial sector) and, for
// an accepted move (ACC of 4), B's legs line and two 4-byte parent words.
template <int UNROLL, int ACC, int CHDR>
__global__ __launch_bounds__(256) void fw_pattern_kernel(uint8_t* hdr, uint8_t* legs, uint64_t n_nodes, int iters, uint64_t* sink) {
  const int lane = threadIdx.x & 3;
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  uint64_t acc = 0;
  for (int it = 0; it < iters; it += UNROLL) {
    uint64_t ha[UNROLL], hc[UNROLL], y[UNROLL][4], base[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      base[u] = mix(gid * 0x9e3779b97f4a7c15ull + (uint64_t)(it + u));
      const uint64_t A = base[u] % n_nodes, C = mix(base[u] + 1) % n_nodes;
      ha[u] = *reinterpret_cast<const uint64_t*>(hdr + A * 32 + lane * 8);
      hc[u] = 0;
      if (CHDR) hc[u] = *reinterpret_cast<const uint64_t*>(hdr + C * 32 + lane * 8);
#pragma unroll
      for (int w = 0; w < 4; ++w) y[u][w] = *reinterpret_cast<const uint64_t*>(legs + C * 128 + w * 32 + lane * 8);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint64_t B = mix(base[u] + 2) % n_nodes, U = mix(base[u] + 3) % n_nodes, V = mix(base[u] + 4) % n_nodes;
      const uint64_t sres = ha[u] + hc[u] + y[u][0] + y[u][1] + y[u][2] + y[u][3];
      acc += sres;
      *reinterpret_cast<uint64_t*>(hdr + B * 32 + lane * 8) = sres;
      if (((it + u) & 3) < ACC) {
#pragma unroll
        for (int w = 0; w < 4; ++w) *reinterpret_cast<uint64_t*>(legs + B * 128 + w * 32 + lane * 8) = sres + w;
        if (lane == 0) {
          *reinterpret_cast<uint32_t*>(hdr + U * 32 + 8) = (uint32_t)sres;
          *reinterpret_cast<uint32_t*>(hdr + V * 32 + 8) = (uint32_t)sres + 1;
        }
      }
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}
template <int UNROLL, int ACC, int CHDR>
static void run_fw_pattern(uint8_t* buf, size_t bytes, uint64_t* sink, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 1024;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const uint64_t n_nodes = bytes / 160;  // headers in the first fifth of the buffer, legs behind them
  uint8_t* legs = buf + n_nodes * 32;
  fw_pattern_kernel<UNROLL, ACC, CHDR><<<blocks, 256>>>(buf, legs, n_nodes, 64, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  fw_pattern_kernel<UNROLL, ACC, CHDR><<<blocks, 256>>>(buf, legs, n_nodes, iters, sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  printf("finite-width move pattern (split layout), accepted %d/4, C's header %s, in-flight/group %d, waves/SIMD %d : %6.2f G moves/s\n",
         ACC, CHDR ? "read    " : "not read", UNROLL, waves_per_simd, (double)blocks * 64 * iters / ms * 1e-6);
}

template <int UNROLL, int ACC, int PARENTS, int WIDE = 0>
static void run_pattern(uint8_t* buf, size_t bytes, uint64_t* sink, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;
  const int iters = 1024;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const uint64_t n_lines = bytes / 128;
  move_pattern_kernel<UNROLL, ACC, PARENTS, WIDE><<<blocks, 256>>>(buf, n_lines, 64, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  move_pattern_kernel<UNROLL, ACC, PARENTS, WIDE><<<blocks, 256>>>(buf, n_lines, iters, sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double moves = (double)blocks * 64 * iters;
  printf("move pattern, accepted %d/4, parent writes %s, block written in %s, in-flight/group %d, waves/SIMD %d : %6.2f G moves/s\n", ACC,
         PARENTS == 0 ? "no " : (PARENTS == 1 ? "4 B" : "64B"), WIDE ? "2 x 64 B" : "4 x 32 B", UNROLL, waves_per_simd, moves / ms * 1e-6);
}

// lane 0 of every group writes 4 bytes into a random 128-B block (a parent-pointer update)
template <int UNROLL>
__global__ __launch_bounds__(256) void tiny_write_kernel(uint8_t* buf, uint64_t n_lines, int iters) {
  const int lane = threadIdx.x & 3;
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  for (int it = 0; it < iters; ++it) {
    const uint64_t g = mix(gid * 0x9e3779b97f4a7c15ull + (uint64_t)it) % n_lines;
    if (lane == 0) *reinterpret_cast<uint32_t*>(buf + g * 128 + 8) = (uint32_t)g;
  }
}
static void run_tiny(uint8_t* buf, size_t bytes, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 2048;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  tiny_write_kernel<1><<<blocks, 256>>>(buf, bytes / 128, 64);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  tiny_write_kernel<1><<<blocks, 256>>>(buf, bytes / 128, iters);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  printf("grain   4 B  write (1 lane)                    waves/SIMD %d : %7.1f G accesses/s\n", waves_per_simd,
         (double)blocks * 64 * iters / ms * 1e-6);
}

template <int GRAIN, int MODE, int UNROLL>
static void run(uint8_t* buf, size_t bytes, uint64_t* sink, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  const int iters = 2048;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const uint64_t n_grains = bytes / GRAIN;
  rnd_kernel<GRAIN, MODE, UNROLL><<<blocks, 256>>>(buf, n_grains, 64, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  rnd_kernel<GRAIN, MODE, UNROLL><<<blocks, 256>>>(buf, n_grains, iters, sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double groups = (double)blocks * 64;
  const double acc = groups * iters;
  const double bytes_moved = acc * GRAIN * (MODE == 1 ? 2 : 1);
  printf("grain %3d B  %-10s  in-flight/group %d  waves/SIMD %d : %7.1f G accesses/s  %8.1f GB/s\n", GRAIN,
         MODE == 0 ? "read" : (MODE == 1 ? "read+write" : "write"), UNROLL, waves_per_simd, acc / ms * 1e-6,
         bytes_moved / ms * 1e-6);
}

int main(int argc, char** argv) {
  size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 8;
  size_t bytes = gib << 30;
  uint8_t* buf;
  uint64_t* sink;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, bytes));
  printf("working set %zu GiB, groups of 4 lanes, random naturally aligned pieces\n", gib);
  for (int w : {3, 8}) {
    run<128, 0, 4>(buf, bytes, sink, w);
    run<128, 0, 8>(buf, bytes, sink, w);
    run<64, 0, 8>(buf, bytes, sink, w);
    run<32, 0, 8>(buf, bytes, sink, w);
    run<128, 1, 4>(buf, bytes, sink, w);
    run<128, 2, 8>(buf, bytes, sink, w);
    run<32, 1, 8>(buf, bytes, sink, w);
    run<32, 2, 8>(buf, bytes, sink, w);
    run<64, 2, 8>(buf, bytes, sink, w);
    run_tiny(buf, bytes, w);
    run_pattern<4, 3, 1>(buf, bytes, sink, w);
    run_pattern<4, 2, 1>(buf, bytes, sink, w);
    run_pattern<4, 2, 1, 1>(buf, bytes, sink, w);
    run_pattern<4, 3, 1, 1>(buf, bytes, sink, w);
    run_pattern<4, 3, 0>(buf, bytes, sink, w);
    run_pattern<4, 3, 2>(buf, bytes, sink, w);
    run_pattern<4, 4, 2>(buf, bytes, sink, w);
    run_pattern<4, 0, 2>(buf, bytes, sink, w);
    run_fw_pattern<4, 3, 1>(buf, bytes, sink, w);
    run_fw_pattern<4, 3, 0>(buf, bytes, sink, w);
  }
  // streaming reference: same kernel shape, consecutive grains
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}
