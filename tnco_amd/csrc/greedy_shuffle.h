// greedy_shuffle.h -- CPython's Random(seed).shuffle on the device: the generator's state in memory or in LDS
// (part of greedy_device.hip, the only file that includes it: everything lives in its unnamed namespace)
#pragma once
#include "greedy_wave.h"

namespace tnco {
namespace {

// ---------------------------------------------------------------------------------------------
// CPython's generator, one lane per tree
// ---------------------------------------------------------------------------------------------
struct ShuffleParams {
  int32_t n;
  int64_t R;
  const uint32_t* seeds;
  uint64_t* draws;      // [R] or NULL (in: outputs to skip, out: outputs consumed)
  const uint32_t* mt0;  // [624] init_genrand(19650218)
  uint32_t* mt;         // [624][R]
  uint16_t* perm;       // [R][n]
};

__global__ __launch_bounds__(64) void py_shuffle_kernel(const ShuffleParams p) {
  const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (r >= p.R) return;
  const int64_t S = p.R;
  uint32_t* mt = p.mt + r;
  const uint32_t seed = p.seeds[r];
  // Modules/_randommodule.c init_by_array(key = {seed})
  for (int i = 0; i < 624; ++i) mt[i * S] = p.mt0[i];
  uint32_t prev = p.mt0[0];
  int i = 1;
  for (int k = 624; k; --k) {
    const uint32_t cur = (mt[i * S] ^ ((prev ^ (prev >> 30)) * 1664525u)) + seed;  // + key[0] + 0
    mt[i * S] = cur;
    prev = cur;
    if (++i >= 624) {
      mt[0] = cur;
      i = 1;
    }
  }
  for (int k = 623; k; --k) {
    const uint32_t cur = (mt[i * S] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
    mt[i * S] = cur;
    prev = cur;
    if (++i >= 624) {
      mt[0] = cur;
      i = 1;
    }
  }
  mt[0] = 0x80000000u;
  int idx = 624;
  uint64_t used = 0;
  auto next = [&]() -> uint32_t {
    if (idx >= 624) {  // genrand_uint32: the whole state at once
      uint32_t cur = mt[0];
      for (int k = 0; k < 624; ++k) {
        const uint32_t nx = mt[((k + 1) % 624) * S];
        const uint32_t y = (cur & 0x80000000u) | (nx & 0x7fffffffu);
        mt[k * S] = mt[((k + 397) % 624) * S] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        cur = nx;
      }
      idx = 0;
    }
    uint32_t y = mt[(idx++) * S];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    ++used;
    return y;
  };
  if (p.draws)
    for (uint64_t k = p.draws[r]; k; --k) (void)next();
  // Lib/random.py shuffle: for i in reversed(range(1, n)): j = _randbelow(i + 1)
  uint16_t* perm = p.perm + r * (int64_t)p.n;
  for (int t = 0; t < p.n; ++t) perm[t] = (uint16_t)t;
  for (int t = p.n - 1; t >= 1; --t) {
    const uint32_t bound = (uint32_t)t + 1u;
    const int k = 32 - __clz(bound);  // bound.bit_length()
    uint32_t j = next() >> (32 - k);
    while (j >= bound) j = next() >> (32 - k);
    const uint16_t a = perm[t], b = perm[j];
    perm[t] = b;
    perm[j] = a;
  }
  if (p.draws) p.draws[r] = used;
}

// The same generator with the state of a tree IN LDS: TPW trees per workgroup (624 words + n positions each),
// one lane per tree for the serial chains -- seeding is 1 247 dependent steps, the shuffle n - 1 -- which now
// wait for LDS instead of memory (py_shuffle_kernel: 11.4 ms for 65 536 x 512, a third of the graph form's
// greedy; this one: see profiles/r04_greedy_graph.md).  Idle lanes help with the copies in and out.
template <int TPW>
__global__ __launch_bounds__(64) void py_shuffle_lds_kernel(const ShuffleParams p) {
  extern __shared__ uint32_t shuf_lds[];
  typedef __attribute__((address_space(3))) uint32_t* l32;
  typedef __attribute__((address_space(3))) uint16_t* l16;
  l32 mtl = (l32)shuf_lds;               // [624][TPW]
  l16 pm = (l16)(mtl + 624 * TPW);        // [n][TPW]
  const int lane = threadIdx.x, n = p.n;
  const int64_t r0 = (int64_t)blockIdx.x * TPW;
  const int64_t r = r0 + lane;
  const bool mine = lane < TPW && r < p.R;
  for (int c = lane; c < n * TPW; c += 64) pm[c] = (uint16_t)(c / TPW);
  __syncthreads();
  if (mine) {
#define TNCO_MT(i) mtl[(i) * TPW + lane]
    const uint32_t seed = p.seeds[r];
    // Modules/_randommodule.c init_by_array(key = {seed})
    // (init_genrand(19650218)'s word i is made on the way: a second chain beside the first, no table to copy in)
    uint32_t g0 = 19650218u, prev = g0;
    for (int i = 1; i < 624; ++i) {
      g0 = 1812433253u * (g0 ^ (g0 >> 30)) + (uint32_t)i;
      const uint32_t cur = (g0 ^ ((prev ^ (prev >> 30)) * 1664525u)) + seed;  // + key[0] + 0
      TNCO_MT(i) = cur;
      prev = cur;
    }
    {  // the 624th step: position 1 again, after the wrap
      const uint32_t cur = (TNCO_MT(1) ^ ((prev ^ (prev >> 30)) * 1664525u)) + seed;
      TNCO_MT(1) = cur;
      prev = cur;
    }
    int i = 2;
    uint32_t own = TNCO_MT(i);  // (the word of the next step is requested a step ahead)
    for (int k = 623; k; --k) {
      const int inext = i + 1 >= 624 ? 1 : i + 1;
      const uint32_t nxt = TNCO_MT(inext);
      const uint32_t cur = (own ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
      TNCO_MT(i) = cur;
      prev = cur;
      if (i + 1 >= 624) TNCO_MT(0) = cur;
      own = inext == 1 && k > 1 ? TNCO_MT(1) : nxt;
      i = inext;
    }
    TNCO_MT(0) = 0x80000000u;
    // genrand_uint32 regenerates all 624 words when they are used up; here word k is made when it is drawn --
    // the same values (a pass over the state reads word k + 1 before it is renewed and word k + 397 mod 624
    // renewed or not exactly as the batch does), and the lanes of a wavefront, which reject different numbers
    // of draws, never wait for each other's 624-step passes.
    int idx = 0;
    uint64_t used = 0;
    auto next = [&]() -> uint32_t {
      const int k1 = idx + 1 >= 624 ? 0 : idx + 1, km = idx + 397 >= 624 ? idx + 397 - 624 : idx + 397;
      const uint32_t a = TNCO_MT(idx), b = TNCO_MT(k1), m = TNCO_MT(km);
      const uint32_t yy = (a & 0x80000000u) | (b & 0x7fffffffu);
      uint32_t y = m ^ (yy >> 1) ^ ((yy & 1u) ? 0x9908b0dfu : 0u);
      TNCO_MT(idx) = y;
      idx = k1;
      y ^= y >> 11;
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= y >> 18;
      ++used;
      return y;
    };
    if (p.draws)
      for (uint64_t k = p.draws[r]; k; --k) (void)next();
    // Lib/random.py shuffle: for i in reversed(range(1, n)): j = _randbelow(i + 1)
    for (int t = n - 1; t >= 1; --t) {
      const uint32_t bound = (uint32_t)t + 1u;
      const int k = 32 - __clz(bound);  // bound.bit_length()
      uint32_t j = next() >> (32 - k);
      while (j >= bound) j = next() >> (32 - k);
      const uint16_t a = pm[t * TPW + lane], b = pm[(int)j * TPW + lane];
      pm[t * TPW + lane] = b;
      pm[(int)j * TPW + lane] = a;
    }
    if (p.draws) p.draws[r] = used;
#undef TNCO_MT
  }
  __syncthreads();
  for (int q = 0; q < TPW && r0 + q < p.R; ++q) {
    uint16_t* out = p.perm + (r0 + q) * (int64_t)n;
    for (int t = lane; t < n; t += 64) out[t] = pm[t * TPW + q];
  }
}


}  // namespace
}  // namespace tnco
