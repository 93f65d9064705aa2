// greedy_device.hip -- the reference's initial contraction trees drawn ON THE DEVICE: the batched
// twin of host_greedy.cpp (same trees, tree for tree; tests/test_gpu_greedy.py).
//
//   tnco/utils/tn.py:189-230: Random(seed).shuffle of the component's tensors (CPython's MT19937,
//   init_by_array, _randbelow), then opt_einsum's greedy path finder with every dimension 2
//   (paths.ssa_greedy_optimize, _simple_chooser, 'memory-removed'; restated, "parity unpinned" like
//   the host version: opt_einsum is not pinned by the reference and absent here).
//
// Why a kernel: for 65 536 runs of a 512-tensor network the host version takes 3 s on the GPU box's 16
// threads -- seven times the 1 000 SA sweeps that follow.
//
// Two forms (round 4).  Networks without hyper-indices -- every circuit and graph network the benchmarks
// use -- take py_shuffle_lds_kernel + greedy_graph_kernel (below: the greedy over a multigraph, a tree's
// whole state in LDS and registers; 65 536 trees of 512 tensors in 2.6 + 35-41 ms).  The others take the
// set form of round 2 (11 + 177 ms on the same network):
// py_shuffle_kernel: ONE LANE per tree (seeding and shuffling are a serial chain per
// tree, 2 500 dependent steps; the generator's state is a column of a [624][R] array, so the lanes of
// a wavefront read and write whole lines).  greedy_kernel: ONE WAVEFRONT per tree, a persistent grid:
//   * an index set = W 64-bit words, word x in lane x (W <= 64); |set| = a wave reduction;
//   * opt_einsum's candidate queue ordered by (cost, id2, id1) with cost = 2^|k12| - 2^|k1| - 2^|k2|
//     becomes ONE 64-bit key per candidate (greedy_key.h: the cost in non-adjacent form, exact), the
//     queue a flat array in LDS, cell c owned by lane c % 64: every lane keeps the minimum of its
//     cells, a pop is a wave-min over the lanes + a rescan of the winner's cells, a push one LDS
//     store into the cell just popped -- no sift-down chains, never more cells than initial candidates;
//   * the neighbours of the tensor just made (the keys sharing a contractible dim with it) are a
//     bitset over the slots, word x in lane x; the candidates of one push are evaluated one per lane;
//   * index sets are content-addressed slots (a stale queue entry revives when an equal set
//     reappears, as with the frozenset keys of the published code): an open-addressing table in LDS.
// Limits of this path (the host version takes the rest): n_inds <= 2040, n_leaves <= 2000, every
// index held by at most GREEDY_MAXH tensors, LDS need <= 64 KB.  A tree that ends with more than one
// tensor (outer products left) is flagged and redone on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/tnco_hip.h"
#include "greedy_key.h"

namespace tnco {
namespace {

constexpr int GREEDY_MAXH = 6;
constexpr int GREEDY_LCAP = 256;  // neighbours of one tensor the kernel handles (more: the tree goes to the host)

// -DTNCO_GREEDY_PROF: shader-clock ticks per section of greedy_kernel, summed per wavefront (diagnostic build)
#ifdef TNCO_GREEDY_PROF
#define GP_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_[i] += t_ - pt_; pt_ = t_; } while (0)
#else
#define GP_T(i)
#endif
constexpr uint32_t DEAD = 0xFFFFu;
constexpr uint64_t KMAX = ~0ull;

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

// ---------------------------------------------------------------------------------------------
// the greedy path finder, one wavefront per tree
// ---------------------------------------------------------------------------------------------
struct GreedyParams {
  int32_t n, I, W, NW, SMAX, Q, TS;  // TS: table size (a power of two)
  int64_t R;
  const uint64_t* leaf;     // [n][W] index sets of the tensors, original order
  const uint64_t* output;   // [W]
  const int32_t* hoff;      // CSR: index -> tensors holding it
  const int32_t* holders;
  const uint16_t* perm;     // [R][n]
  // scratch, one set per resident wavefront (G = gridDim.x)
  uint64_t* keys;           // [G][SMAX][W]   slot -> index set
  uint64_t* nbr;            // [G][SMAX][NW]  slot -> live slots sharing a contractible dim
  uint64_t* arena;          // [G][Q][W]      queued candidate -> its result set
  int32_t* path;            // [G][n][2]      ssa path
  uint16_t* slot_of_leaf;   // [G][n]
  int32_t* links;           // [R][3][2n - 1] out
  int32_t* status;          // [R] out: 0 = done, else redo on the host
  unsigned long long* prof; // [G][16] (TNCO_GREEDY_PROF)
};

__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int o) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int lane) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, lane), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}
// wave reductions: four DPP steps inside each row of 16 lanes (xor 1, xor 2, half mirror, mirror), then
// the four row results through readlane -- no LDS crossbar round trips
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp64(uint64_t v) {
  return ((uint64_t)dpp<CTRL>((uint32_t)(v >> 32)) << 32) | dpp<CTRL>((uint32_t)v);
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, int lane) {
  return ((uint64_t)rdlane((uint32_t)(v >> 32), lane) << 32) | rdlane((uint32_t)v, lane);
}
// (every DPP read is evaluated ONCE, with all lanes active: a read from a lane that a branch has switched
//  off returns nothing)
template <int CTRL>
__device__ __forceinline__ uint64_t min_step(uint64_t v) {
  const uint64_t t = dpp64<CTRL>(v);
  return t < v ? t : v;
}
__device__ __forceinline__ uint32_t wsum(uint32_t v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return rdlane(v, 0) + rdlane(v, 16) + rdlane(v, 32) + rdlane(v, 48);
}
__device__ __forceinline__ uint64_t wmin64(uint64_t v) {
  v = min_step<0xB1>(v);
  v = min_step<0x4E>(v);
  v = min_step<0x141>(v);
  v = min_step<0x140>(v);
  const uint64_t a = rdlane64(v, 0), b = rdlane64(v, 16), c = rdlane64(v, 32), d = rdlane64(v, 48);
  const uint64_t ab = b < a ? b : a, cd = d < c ? d : c;
  return cd < ab ? cd : ab;
}
__device__ __forceinline__ uint64_t wxor64(uint64_t v) {
  v ^= dpp64<0xB1>(v);
  v ^= dpp64<0x4E>(v);
  v ^= dpp64<0x141>(v);
  v ^= dpp64<0x140>(v);
  return rdlane64(v, 0) ^ rdlane64(v, 16) ^ rdlane64(v, 32) ^ rdlane64(v, 48);
}
// exclusive prefix sum over the lanes
__device__ __forceinline__ uint32_t wscan_excl(uint32_t v, int lane) {
  uint32_t s = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)s, o);
    if (lane >= o) s += t;
  }
  return s - v;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

typedef __attribute__((address_space(3))) volatile uint64_t* lds_u64;
typedef __attribute__((address_space(3))) volatile uint16_t* lds_u16;
typedef __attribute__((address_space(3))) volatile int16_t* lds_i16;

__global__ __launch_bounds__(64) void greedy_kernel(const GreedyParams p) {
  extern __shared__ uint64_t lds_raw[];
  const int lane = threadIdx.x;
  const int n = p.n, I = p.I, W = p.W, NW = p.NW, SMAX = p.SMAX, Q = p.Q, TS = p.TS;
  const int QC = (Q + 63) & ~63;  // queue cells (a multiple of 64)
  // LDS: queue keys [QC] | broadcast rows a, output, ref2, ref3 [4 W] | ssa [SMAX] | holders per dim [I] |
  //      neighbour list [GREEDY_LCAP] | slot of an ssa id [2n] | table [TS]
  lds_u64 hk = (lds_u64)lds_raw;
  lds_u64 bc = hk + QC;
  lds_u16 ssa = (lds_u16)(bc + 4 * W);
  lds_u16 cnt = ssa + SMAX;
  lds_u16 lst = cnt + I;
  lds_u16 sid = lst + GREEDY_LCAP;  // ssa id -> slot [2n]
  lds_i16 table = (lds_i16)(sid + 2 * n);

  const int g = blockIdx.x;
  uint64_t* keys = p.keys + (size_t)g * SMAX * W;
  uint64_t* nbr = p.nbr + (size_t)g * SMAX * NW;
  uint64_t* arena = p.arena + (size_t)g * Q * W;
  int32_t* path = p.path + (size_t)g * n * 2;
  uint16_t* slot_of_leaf = p.slot_of_leaf + (size_t)g * n;
  const bool inw = lane < W;

#ifdef TNCO_GREEDY_PROF
  unsigned long long prof_[16] = {0}, pt_ = __builtin_amdgcn_s_memtime();
#endif
  for (int64_t r = g; r < p.R; r += gridDim.x) {
    const uint16_t* perm = p.perm + r * (int64_t)n;
    int status = 0;
    // ---- clear ----
    for (int c = lane; c < QC; c += 64) hk[c] = KMAX;
    for (int c = lane; c < TS; c += 64) table[c] = -1;
    for (int c = lane; c < I; c += 64) cnt[c] = 0;
    for (int c = lane; c < SMAX; c += 64) ssa[c] = DEAD;
    for (size_t c = lane; c < (size_t)SMAX * NW; c += 64) nbr[c] = 0;
    uint64_t out = inw ? p.output[lane] : 0;
    {  // dims common to all inputs join the output
      uint64_t all = inw ? ~0ull : 0;
      for (int t = 0; t < n; ++t) all &= inw ? p.leaf[(size_t)t * W + lane] : 0;
      out |= all;
    }
    __syncthreads();
    GP_T(0);
    int next_ssa = n, nslots = 0, step = 0, n_alive = 0;
    // content-addressed slot of an index set (word x in lane x): >= 0 found, else -1 and *cell = the free cell
    auto hash_of = [&](uint64_t m) -> uint32_t {
      uint64_t h = inw ? (m + 0x9E3779B97F4A7C15ull * (uint64_t)(lane + 1)) * 0xff51afd7ed558ccdull : 0;
      h ^= h >> 29;
      h = wxor64(h);
      h *= 0xc4ceb9fe1a85ec53ull;
      h ^= h >> 32;
      return (uint32_t)h;
    };
    // a table cell: slot (12 bits: SMAX <= 4008) | 4 bits of the hash << 12; 0xFFFF = free.  A set is only
    // compared with the stored one (a read from memory) when those four bits agree.
    int tag = 0;
    auto find_slot = [&](uint64_t m, int& cell) -> int {
      const uint32_t hh = hash_of(m);
      tag = (int)(hh >> 28);
      for (uint32_t h = hh & (uint32_t)(TS - 1);; h = (h + 1) & (uint32_t)(TS - 1)) {
        const int e = uni((int)(uint16_t)table[h]);
        if (e == 0xFFFF) {
          cell = (int)h;
          return -1;
        }
        if ((e >> 12) != tag) continue;
        const int s = e & 0xFFF;
        const uint64_t ks = inw ? keys[(size_t)s * W + lane] : 0;
        if (__all(ks == m)) return s;
      }
    };
    auto new_slot = [&](uint64_t m, int cell) -> int {  // (right after the find_slot(m) that found `cell` free)
      const int s = nslots++;
      if (s >= SMAX) {
        status = 4;
        return 0;
      }
      if (inw) keys[(size_t)s * W + lane] = m;
      if (lane == 0) table[cell] = (int16_t)(uint16_t)(s | (tag << 12));
      return s;
    };
    // ---- the inputs in shuffled order; equal index sets are multiplied at once ----
    for (int t0 = 0; t0 < n; t0 += 64) {
      const int pv = t0 + lane < n ? (int)perm[t0 + lane] : 0;  // (64 positions per read; the next index set travels
      const int nt = n - t0 < 64 ? n - t0 : 64;                  //  while this one is looked up)
      const int lf0 = uni(__shfl(pv, 0));  // (all lanes take part in the shuffles)
      uint64_t m_next = inw ? p.leaf[(size_t)lf0 * W + lane] : 0;
      for (int i = 0; i < nt; ++i) {
        const int t = t0 + i;
        const int lf = uni(__shfl(pv, i));
        const uint64_t m = m_next;
        const int lf1 = uni(__shfl(pv, i + 1 < nt ? i + 1 : i));
        if (i + 1 < nt) m_next = inw ? p.leaf[(size_t)lf1 * W + lane] : 0;
        int cell = 0;
        int s = find_slot(m, cell);
        const bool alive = s >= 0 && ssa[s] != DEAD;
        if (alive) {
          if (lane == 0) {
            path[2 * step] = ssa[s];
            path[2 * step + 1] = t;
            ssa[s] = (uint16_t)next_ssa;
            sid[next_ssa] = (uint16_t)s;
          }
          ++step;
          ++next_ssa;
        } else {
          if (s < 0) s = new_slot(m, cell);
          if (lane == 0) {
            ssa[s] = (uint16_t)t;
            sid[t] = (uint16_t)s;
          }
          ++n_alive;
        }
        if (lane == 0) slot_of_leaf[lf] = (uint16_t)s;
        __syncthreads();
      }
    }
    GP_T(1);
    // ---- per contractible dim: its holders (slots, by ssa id), the counts, the neighbour sets ----
    if (inw) {
      bc[W + lane] = out;
      bc[2 * W + lane] = 0;
      bc[3 * W + lane] = 0;
    }
    __syncthreads();
    // the (deduplicated, ssa-ordered) slots holding dim d; returns their number
    auto dim_slots = [&](int d, int (&sl)[GREEDY_MAXH], int (&id)[GREEDY_MAXH]) -> int {
      const int h0 = p.hoff[d], m = p.hoff[d + 1] - h0;
      int cntu = 0;
#pragma unroll
      for (int i = 0; i < GREEDY_MAXH; ++i) {
        sl[i] = -1;
        id[i] = 0x7FFFFFFF;
        if (i < m) {
          const int s = slot_of_leaf[p.holders[h0 + i]];
          bool dup = false;
#pragma unroll
          for (int j = 0; j < GREEDY_MAXH; ++j)
            if (j < i && sl[j] == s) dup = true;
          if (!dup) {
            sl[i] = s;
            id[i] = ssa[s];
            ++cntu;
          }
        }
      }
      // order by ssa id (absent entries last): odd-even transposition over GREEDY_MAXH
#pragma unroll
      for (int pass = 0; pass < GREEDY_MAXH; ++pass) {
#pragma unroll
        for (int i = pass & 1; i + 1 < GREEDY_MAXH; i += 2) {
          if (id[i] > id[i + 1]) {
            const int ti = id[i], ts = sl[i];
            id[i] = id[i + 1]; sl[i] = sl[i + 1];
            id[i + 1] = ti; sl[i + 1] = ts;
          }
        }
      }
      return cntu;
    };
    for (int d0 = 0; d0 < I; d0 += 64) {
      const int d = d0 + lane;
      if (d < I && !((bc[W + (d >> 6)] >> (d & 63)) & 1ull)) {
        int sl[GREEDY_MAXH], id[GREEDY_MAXH];
        const int m = dim_slots(d, sl, id);
        cnt[d] = (uint16_t)m;
        if (m >= 2) atomicOr((unsigned long long*)(uint64_t*)(bc + 2 * W + (d >> 6)), 1ull << (d & 63));
        if (m >= 3) atomicOr((unsigned long long*)(uint64_t*)(bc + 3 * W + (d >> 6)), 1ull << (d & 63));
#pragma unroll
        for (int i = 0; i < GREEDY_MAXH; ++i)
#pragma unroll
          for (int j = 0; j < GREEDY_MAXH; ++j)
            if (i < m && j < m && i != j)
              atomicOr((unsigned long long*)&nbr[(size_t)sl[i] * NW + (sl[j] >> 6)], 1ull << (sl[j] & 63));
      }
    }
    __threadfence();
    __syncthreads();
    GP_T(2);
    uint64_t ref2 = inw ? bc[2 * W + lane] : 0, ref3 = inw ? bc[3 * W + lane] : 0;
    // |result| of contracting slots s1, s2 under the current counts (this lane alone: W words)
    // (|k1|, |k2| are counted along: no table of sizes)
    auto size12_of = [&](int s1, int s2, int& f1, int& f2) -> int {
      int c = 0;
      f1 = 0;
      f2 = 0;
      for (int x = 0; x < W; ++x) {
        const uint64_t a = keys[(size_t)s1 * W + x], b = keys[(size_t)s2 * W + x];
        const uint64_t either = a | b, two = a & b, one = either & ~two;
        c += __popcll((either & bc[W + x]) | (two & bc[3 * W + x]) | (one & bc[2 * W + x]));
        f1 += __popcll(a);
        f2 += __popcll(b);
      }
      return c;
    };
    // ---- initial candidates: per dim, each holder against the later ones, the cheapest pushed ----
    int count = 0;  // queue cells used
    for (int d0 = 0; d0 < I; d0 += 64) {
      const int d = d0 + lane;
      int sl[GREEDY_MAXH], id[GREEDY_MAXH];
      int m = 0;
      if (d < I && !((bc[W + (d >> 6)] >> (d & 63)) & 1ull)) m = dim_slots(d, sl, id);
      const int mine = m >= 2 ? m - 1 : 0;
      const int base = count + (int)wscan_excl((uint32_t)mine, lane);
      count += (int)wsum((uint32_t)mine);
      if (count > Q) {
        status = 5;
        break;
      }
#pragma unroll
      for (int i = 0; i + 1 < GREEDY_MAXH; ++i) {
        if (i + 1 < m) {
          uint64_t bestk = KMAX;
          int bj = i + 1;
#pragma unroll
          for (int j = 1; j < GREEDY_MAXH; ++j) {
            if (j > i && j < m) {
              int f1, f2;
              const int s12 = size12_of(sl[i], sl[j], f1, f2);
              const uint64_t k = greedy_cand_key(s12, f1, f2, id[j], id[i]);
              if (k < bestk) {
                bestk = k;
                bj = j;
              }
            }
          }
          int sj = sl[1];
#pragma unroll
          for (int j = 1; j < GREEDY_MAXH; ++j)
            if (j == bj) sj = sl[j];
          const int seq = base + i;
          uint64_t* row = arena + (size_t)seq * W;
          for (int x = 0; x < W; ++x) {
            const uint64_t a = keys[(size_t)sl[i] * W + x], b = keys[(size_t)sj * W + x];
            const uint64_t either = a | b, two = a & b, one = either & ~two;
            row[x] = (either & bc[W + x]) | (two & bc[3 * W + x]) | (one & bc[2 * W + x]);
          }
          hk[seq] = bestk;
        }
      }
    }
    __syncthreads();
    // every lane: the minimum of its cells (four reads in flight)
    uint64_t lkey = KMAX;
    int lrow = 0;
    auto rescan = [&]() {
      lkey = KMAX;
      lrow = 0;
      const __attribute__((address_space(3))) uint64_t* cells = (const __attribute__((address_space(3))) uint64_t*)hk + lane;
      const int rows = (count + 63) >> 6;
      for (int row = 0; row < rows; row += 4) {
        uint64_t k[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) k[q] = row + q < rows ? cells[(row + q) * 64] : KMAX;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (k[q] < lkey) {
            lkey = k[q];
            lrow = row + q;
          }
      }
    };
    rescan();
    GP_T(3);
    // ---- the greedy loop ----
    const bool innw = lane < NW;
    while (status == 0) {
      const uint64_t best = wmin64(lkey);
      if (best == KMAX) break;
      const int wl = __ffsll((unsigned long long)__ballot(lkey == best)) - 1;
      const int seq = uni(__shfl(lrow, wl)) * 64 + wl;
      if (lane == wl) {
        hk[seq] = KMAX;
        rescan();
      }
      // (the key carries the ssa ids at push time; an id belongs to one slot for good)
      const int s1 = uni((int)sid[(int)(best & 0x3FFFu)]), s2 = uni((int)sid[(int)((best >> 14) & 0x3FFFu)]);
      const int id1 = ssa[s1], id2 = ssa[s2];
      GP_T(4);
      if (id1 == (int)DEAD || id2 == (int)DEAD) continue;  // obsolete
      // everything this contraction reads from memory, requested together
      const uint64_t k12 = inw ? arena[(size_t)seq * W + lane] : 0;
      const uint64_t a = inw ? keys[(size_t)s1 * W + lane] : 0, b = inw ? keys[(size_t)s2 * W + lane] : 0;
      uint64_t un = innw ? (nbr[(size_t)s1 * NW + lane] | nbr[(size_t)s2 * NW + lane]) : 0;
      if (lane == 0) {
        ssa[s1] = DEAD;
        ssa[s2] = DEAD;
        path[2 * step] = id1;
        path[2 * step + 1] = id2;
      }
      ++step;
      __syncthreads();
      int cell = 0;
      int s12 = find_slot(k12, cell);
      const bool merged = s12 >= 0 && ssa[s12] != DEAD;  // an equal index set is live: multiplied with it
      if (merged) {
        if (lane == 0) {
          path[2 * step] = ssa[s12];
          path[2 * step + 1] = next_ssa;
        }
        ++step;
        ++next_ssa;
        n_alive -= 2;
      } else {
        if (s12 < 0) s12 = new_slot(k12, cell);
        if (status) break;
        n_alive -= 1;
      }
      const int id12 = next_ssa++;
      if (lane == 0) {
        ssa[s12] = (uint16_t)id12;
        sid[id12] = (uint16_t)s12;
      }
      GP_T(5);
      uint64_t n12m = (merged && innw) ? nbr[(size_t)s12 * NW + lane] : 0;
      // (the live equal set keeps its neighbours, minus the two tensors that just left)
      if (lane == (s1 >> 6)) n12m &= ~(1ull << (s1 & 63));
      if (lane == (s2 >> 6)) n12m &= ~(1ull << (s2 & 63));
      // holders per dim: only shared dims and dropped dims change their number
      {
        uint64_t u = (merged ? (a | b) : ((a & b) | ((a ^ b) & ~k12))) & ~out;
        while (u) {
          const int bit = __ffsll((unsigned long long)u) - 1;
          u &= u - 1;
          const int d = lane * 64 + bit;
          const int dec = (int)((a >> bit) & 1ull) + (int)((b >> bit) & 1ull) - (merged ? 0 : (int)((k12 >> bit) & 1ull));
          const int c = (int)cnt[d] - dec;
          cnt[d] = (uint16_t)c;
          const uint64_t m1 = 1ull << bit;
          ref2 = c >= 2 ? (ref2 | m1) : (ref2 & ~m1);
          ref3 = c >= 3 ? (ref3 | m1) : (ref3 & ~m1);
        }
      }
      GP_T(6);
      // neighbours: those of k1 and of k2 (every one of them shares a dim the result keeps)
      if (lane == (s1 >> 6)) un &= ~(1ull << (s1 & 63));
      if (lane == (s2 >> 6)) un &= ~(1ull << (s2 & 63));
      if (lane == (s12 >> 6)) un &= ~(1ull << (s12 & 63));
      const uint64_t n12 = un | n12m;
      if (innw) nbr[(size_t)s12 * NW + lane] = n12;
      {
        const int w1 = s1 >> 6, w2 = s2 >> 6, w12 = s12 >> 6;
        const uint64_t m1 = 1ull << (s1 & 63), m2 = 1ull << (s2 & 63), m12 = 1ull << (s12 & 63);
        uint64_t u = un;
        while (u) {
          const int y = lane * 64 + __ffsll((unsigned long long)u) - 1;
          u &= u - 1;
          uint64_t* ry = nbr + (size_t)y * NW;
          uint64_t v1 = ry[w1], v2 = ry[w2], v12 = ry[w12];  // (three reads in flight; equal words: equal values)
          v1 &= ~m1;
          if (w2 == w1) v1 &= ~m2;
          if (w12 == w1) v1 |= m12;
          ry[w1] = v1;
          if (w2 != w1) {
            v2 &= ~m2;
            if (w12 == w2) v2 |= m12;
            ry[w2] = v2;
          }
          if (w12 != w1 && w12 != w2) ry[w12] = v12 | m12;
        }
      }
      GP_T(7);
      // push the cheapest (k12, neighbour)
      const uint32_t pc = (uint32_t)__popcll(n12);
      const int total = (int)wsum(pc);
      if (total > GREEDY_LCAP) {
        status = 7;
        break;
      }
      if (total > 0) {
        {
          int at = (int)wscan_excl(pc, lane);
          uint64_t u = n12;
          while (u) {
            lst[at++] = (uint16_t)(lane * 64 + __ffsll((unsigned long long)u) - 1);
            u &= u - 1;
          }
        }
        __syncthreads();
        GP_T(8);
        // one neighbour per lane.  Per word the result keeps, of the legs the neighbour b does NOT hold,
        // P = k12 & (output | ref2), and of those it holds Q = output | (k12 & ref3) | (~k12 & ref2):
        // |k12'| = popcount(b ? Q : P), two masks per push; their words come from their lanes by readlane
        const uint64_t pmask = k12 & (out | ref2), qmask = out | (k12 & ref3) | (~k12 & ref2);
        const int f12 = (int)wsum(inw ? (uint32_t)__popcll(k12) : 0u);
        uint64_t bestk = KMAX;
        int bests = 0;
        for (int j0 = 0; j0 < total; j0 += 64) {
          const int j = j0 + lane;
          const bool valid = j < total;
          const int s = lst[valid ? j : 0];
          const uint64_t* ks = keys + (size_t)s * W;
          int c = 0, fs = 0;
          for (int x0 = 0; x0 < W; x0 += 16) {  // (sixteen words requested together: one memory latency per neighbour)
            uint64_t bx[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) bx[q] = x0 + q < W ? ks[x0 + q] : 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              if (x0 + q < W) {
                const int x = x0 + q;
                c += __popcll((bx[q] & rdlane64(qmask, x)) | (~bx[q] & rdlane64(pmask, x)));
                fs += __popcll(bx[q]);
              }
            }
          }
          const int ids = ssa[s];
          const uint64_t k = greedy_cand_key(c, f12, fs, ids > id12 ? ids : id12, ids > id12 ? id12 : ids);
          if (valid && k < bestk) {
            bestk = k;
            bests = s;
          }
        }
        GP_T(9);
        // (keys of distinct neighbours differ in an id: exactly one lane holds the minimum)
        const uint64_t wk = wmin64(bestk);
        const int sbest = uni(__shfl(bests, __ffsll((unsigned long long)__ballot(bestk == wk)) - 1));
        const uint64_t bb = inw ? keys[(size_t)sbest * W + lane] : 0;
        const uint64_t res = (bb & qmask) | (~bb & pmask);
        // into the cell (and the arena row) of the candidate popped in this iteration: the queue never
        // holds more than the initial candidates
        if (inw) arena[(size_t)seq * W + lane] = res;
        if (lane == wl) {
          hk[seq] = wk;
          if (wk < lkey) {
            lkey = wk;
            lrow = seq >> 6;
          }
        }
      }
      __syncthreads();
      GP_T(10);
    }
    if (status == 0 && (n_alive != 1 || step != n - 1)) status = 3;  // outer products left: the host's
    // ---- ssa path -> links ----
    const int N = 2 * n - 1;
    int32_t* lk = p.links + r * 3 * (int64_t)N;
    for (int c = lane; c < 3 * N; c += 64) lk[c] = -1;
    __threadfence();
    __syncthreads();
    if (status == 0) {
      int bad = 0;
      for (int s = lane; s < n - 1; s += 64) {
        int x = path[2 * s], y = path[2 * s + 1];
        x = x < n ? (int)perm[x] : x;
        y = y < n ? (int)perm[y] : y;
        const int z = n + s;
        if (x < 0 || y < 0 || x >= z || y >= z || x == y) {
          bad = 1;
          continue;
        }
        lk[z] = x < y ? x : y;
        lk[N + z] = x < y ? y : x;
        if (atomicExch(&lk[2 * N + x], z) != -1) bad = 1;
        if (atomicExch(&lk[2 * N + y], z) != -1) bad = 1;
      }
      if (__any(bad)) status = 6;
    }
    if (lane == 0) p.status[r] = status;
    __threadfence();
    __syncthreads();
    GP_T(11);
  }
#ifdef TNCO_GREEDY_PROF
  if (lane == 0)
    for (int i = 0; i < 16; ++i) p.prof[(size_t)g * 16 + i] = prof_[i];
#endif
}

// ---------------------------------------------------------------------------------------------
// the same greedy on a MULTIGRAPH: one wavefront per tree, the whole tree in LDS
// ---------------------------------------------------------------------------------------------
// Where no index is a hyper-index (a contractible index has exactly two holders, an output index one),
// the published algorithm never needs an index SET (tools/greedy_graph_model.py: the model, checked
// against the set form): a tensor is its ssa id, its number of legs and a list of (neighbour, shared legs);
// contracting u and v makes z = the next id with
//     |z| = kept(u) + kept(v) - 2 w(u, v),      list(z) = list(u) + list(v) without each other, equal
// neighbours joined -- kept = the legs that are output legs or have a live partner (all of them but for an
// input's dangling legs).  A dead index set cannot come back (the leg a contraction removed is gone for
// good) and two live tensors with equal sets are an isolated pair, so a slot IS an ssa id; the stored result
// of a queued candidate is what contracting its two tensors gives as long as both are alive.
//   * lists are never updated in place: an entry names the tensor its leg went to when the list was made,
//     and link[] (the parent of a dead id) leads from there to the live tensor that holds it now -- a
//     union-find whose paths the look-ups shorten; the entries of u and v resolve in parallel, one per lane;
//   * equal neighbours are joined through a byte per id (mark: the lane that speaks for the id);
//   * lists live in ONE arena of L + 256 entries (L = entries of the inputs; the live entries only ever
//     get fewer): when it is full the live lists move to its front, in place;
//   * the queue is greedy_kernel's (64-bit keys, cell c owned by lane c % 64, a push goes into the cell
//     just popped) but IN REGISTERS, ROWS cells per lane; the cost keys of the initial candidates do not
//     depend on the shuffle and come from the host.
// 512 tensors / 768 indices: 12 KB of LDS per tree (13 trees per CU), no memory traffic but the links written out.
constexpr int GRAPH_MAXLIST = 255;  // entries of u and v together (more: the tree goes to the host)
constexpr int GRAPH_SHORT = 8;      // a list this short moves through registers when the arena is compacted
struct GraphParams {
  int32_t n, E, L, CAP;
  int32_t dangling;          // 1: some tensor has a leg with a single holder that is no output leg (kept < legs)
  int64_t R;
  const uint16_t* perm;      // [R][n]
  // the network in the ORIGINAL numbering of its tensors (shared by all trees)
  const uint16_t* t_off;     // [n + 1] neighbour lists, CSR
  const uint16_t* t_nbr;     // [L]
  const uint8_t* t_mult;     // [L] legs shared with that neighbour
  const uint8_t* t_fp;       // [n] legs
  const uint8_t* t_kp;       // [n] legs that are output legs or have two holders
  const uint32_t* e_ends;    // [E] the two holders of a contractible index: a | b << 16
  const uint64_t* e_key;     // [E] greedy_cost_key of contracting them << 28
  int32_t* links;            // [R][3][2n - 1] out
  int32_t* status;           // [R] out
  unsigned long long* prof;  // [G][16] (TNCO_GREEDY_PROF)
};

typedef __attribute__((address_space(3))) volatile uint8_t* lds_u8;
typedef __attribute__((address_space(3))) volatile uint32_t* lds_u32;

__host__ __device__ inline size_t graph_lds_bytes(int n, int CAP) { return 256 + (size_t)13 * n + (size_t)3 * CAP + 16; }  // (rec 8n, perm 2n, mark 2n, kept n)

// LDS traffic of ONE wavefront needs no barrier: the LDS unit takes a wavefront's instructions in order.  (A
// __syncthreads() would also wait for the links on their way to memory -- a microsecond per contraction.)
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t wmin32(uint32_t v) {
  uint32_t t = dpp<0xB1>(v);
  v = t < v ? t : v;
  t = dpp<0x4E>(v);
  v = t < v ? t : v;
  t = dpp<0x141>(v);
  v = t < v ? t : v;
  t = dpp<0x140>(v);
  v = t < v ? t : v;
  const uint32_t a = rdlane(v, 0), b = rdlane(v, 16), c = rdlane(v, 32), d = rdlane(v, 48);
  const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}
// wave minimum of 64-bit keys as two 32-bit ones (high words, then the low words of the lanes that hold the minimum)
__device__ __forceinline__ uint64_t wmin64_2(uint64_t k) {
  const uint32_t hi = (uint32_t)(k >> 32);
  const uint32_t mh = wmin32(hi);
  const uint32_t ml = wmin32(hi == mh ? (uint32_t)k : 0xFFFFFFFFu);
  return ((uint64_t)mh << 32) | ml;
}

// ROWS: queue cells per lane (the queue is 64 ROWS cells IN REGISTERS, cell (row, lane); E <= 64 ROWS)
template <int ROWS>
__global__ __launch_bounds__(64) void greedy_graph_kernel(const GraphParams p) {
  extern __shared__ uint64_t lds_raw[];
  const int lane = threadIdx.x;
  const int n = p.n, E = p.E, CAP = p.CAP, N2 = 2 * n, N = 2 * n - 1;
  const bool dang = p.dangling != 0;
  // LDS: acc [64] | rec [2n] | perm [n] | list ids [CAP] || mark [2n] | kept [n] | list mult [CAP]
  lds_u32 acc = (lds_u32)lds_raw;
  // rec of a live id: 0x8000 | offset of its list, legs << 16, list length << 24; of a dead id: its parent
  lds_u32 rec = acc + 64;
  lds_u16 permL = (lds_u16)(rec + N2);
  lds_u16 ids = permL + n;
  lds_u8 mark = (lds_u8)(ids + CAP);
  lds_u8 kp8 = mark + N2;
  lds_u8 mult = kp8 + n;
  lds_u16 inv = (lds_u16)mark;  // tensor -> position in the shuffled order (set-up only)
  const int g = blockIdx.x;
#ifdef TNCO_GREEDY_PROF
  unsigned long long prof_[16] = {0}, pt_ = __builtin_amdgcn_s_memtime();
#endif
  // The queue: cell[q] of lane l is cell 64 q + l; every lane knows the smallest of its cells (lkey, in row lrow).
  // (Keeping the second smallest as well, so that the scan over the rows runs only when a lane is hit twice,
  // was measured: the same instruction count -- the write into a row chosen at run time costs what the scan does.)
  uint64_t cell[ROWS];
  uint64_t lkey = KMAX;
  int lrow = 0;
  auto rescan = [&]() {
    lkey = cell[0];
    lrow = 0;
#pragma unroll
    for (int q = 1; q < ROWS; ++q)
      if (cell[q] < lkey) {
        lkey = cell[q];
        lrow = q;
      }
  };
  // the smallest cell of the lanes in `who` becomes x
  auto replace_min = [&](bool who, uint64_t x) {
#pragma unroll
    for (int q = 0; q < ROWS; ++q)
      if (who && q == lrow) cell[q] = x;
    rescan();
  };
  for (int64_t r = g; r < p.R; r += gridDim.x) {
    const uint16_t* perm = p.perm + r * (int64_t)n;
    int32_t* lk = p.links + r * 3 * (int64_t)N;
    int status = 0;
    // ---- the inputs in shuffled order: position t is ssa id t ----
    for (int t = lane; t < n; t += 64) {
      const int T = perm[t];
      permL[t] = (uint16_t)T;
      inv[T] = (uint16_t)t;
      lk[t] = -1;
      lk[N + t] = -1;
    }
    wsync();
    int bump = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
      const int t = t0 + lane;
      const bool valid = t < n;
      const int T = valid ? (int)permL[t] : 0;
      const int o = p.t_off[T], l = valid ? (int)p.t_off[T + 1] - o : 0;
      const int dst = bump + (int)wscan_excl((uint32_t)l, lane);
      bump += (int)wsum((uint32_t)l);
      for (int j = 0; j < l; ++j) {
        ids[dst + j] = inv[p.t_nbr[o + j]];
        mult[dst + j] = p.t_mult[o + j];
      }
      if (valid) {
        rec[t] = 0x8000u | (uint32_t)dst | ((uint32_t)p.t_fp[T] << 16) | ((uint32_t)l << 24);
        kp8[t] = p.t_kp[T];
      }
    }
    // ---- one candidate per contractible index ----
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
      const int e = q * 64 + lane;
      uint64_t k = KMAX;
      if (e < E) {
        const uint32_t ends = p.e_ends[e];
        const int x = inv[ends & 0xFFFFu], y = inv[ends >> 16];
        k = p.e_key[e] | ((uint64_t)(x > y ? x : y) << 14) | (uint64_t)(x > y ? y : x);
      }
      cell[q] = k;
    }
    wsync();
    rescan();
    GP_T(0);
    int z = n;
    // The live lists moved to the front of the arena, in place.  Lists lie in the order of their ids, a
    // list never moves up: per 64 ids, the short lists go through registers (all read before any is
    // written), the long ones are copied one at a time by the whole wavefront in between.
    auto compact = [&]() {
      int pos = 0;
      for (int x0 = 0; x0 < z; x0 += 64) {
        const int x = x0 + lane;
        const uint32_t rx = x < z ? (uint32_t)rec[x] : 0u;
        const bool live = (rx & 0x8000u) != 0;
        if (!__any(live)) continue;
        const int l = live ? (int)(rx >> 24) : 0, off = (int)(rx & 0x7FFFu);
        const int d = pos + (int)wscan_excl((uint32_t)l, lane);
        pos += (int)wsum((uint32_t)l);
        uint32_t ent[GRAPH_SHORT];
#pragma unroll
        for (int q = 0; q < GRAPH_SHORT; ++q) ent[q] = q < l ? ((uint32_t)ids[off + q] | ((uint32_t)mult[off + q] << 16)) : 0u;
        wsync();
        unsigned long long lb = __ballot(l > GRAPH_SHORT);
        while (lb) {
          const int a = __ffsll(lb) - 1;
          lb &= lb - 1;
          const int la = rdlane((uint32_t)l, a), oa = rdlane((uint32_t)off, a), da = rdlane((uint32_t)d, a);
          if (da == oa) continue;
          for (int k = 0; k < la; k += 64) {
            const bool valid = k + lane < la;
            const uint32_t vi = valid ? (uint32_t)ids[oa + k + lane] : 0u, vm = valid ? (uint32_t)mult[oa + k + lane] : 0u;
            wsync();
            if (valid) {
              ids[da + k + lane] = (uint16_t)vi;
              mult[da + k + lane] = (uint8_t)vm;
            }
            wsync();
          }
        }
        if (l <= GRAPH_SHORT && d != off) {
#pragma unroll
          for (int q = 0; q < GRAPH_SHORT; ++q)
            if (q < l) {
              ids[d + q] = (uint16_t)ent[q];
              mult[d + q] = (uint8_t)(ent[q] >> 16);
            }
        }
        if (live) rec[x] = (rx & 0xFFFF8000u) | (uint32_t)d;
        wsync();
      }
      bump = pos;
    };
    // ---- the greedy loop (z == N: every candidate left is obsolete) ----
    while (status == 0 && z < N) {
      const uint64_t best = wmin64_2(lkey);
      if (best == KMAX) break;
      const int u = (int)(best & 0x3FFFu), v = (int)((best >> 14) & 0x3FFFu);  // (u < v)
      const uint32_t ru = (uint32_t)uni((int)rec[u]), rv = (uint32_t)uni((int)rec[v]);
      if (!(ru & rv & 0x8000u)) {
        // obsolete -- and so may be the minima of other lanes: every lane looks at its own and drops it
        const bool any = lkey != KMAX;
        const uint32_t mu = (uint32_t)rec[any ? (int)(lkey & 0x3FFFu) : 0], mv = (uint32_t)rec[any ? (int)((lkey >> 14) & 0x3FFFu) : 0];
        replace_min(any && !(mu & mv & 0x8000u), KMAX);
        continue;
      }
      const int wl = __ffsll((unsigned long long)__ballot(lkey == best)) - 1;
      GP_T(1);
      const int lu = (int)(ru >> 24), lv = (int)(rv >> 24), total = lu + lv;
      if (total > GRAPH_MAXLIST) {
        status = 7;
        break;
      }
      int ou = (int)(ru & 0x7FFFu), ov = (int)(rv & 0x7FFFu);
      if (bump + total > CAP) {
        compact();
        ou = uni((int)rec[u]) & 0x7FFF;
        ov = uni((int)rec[v]) & 0x7FFF;
      }
      const int fu = (int)((ru >> 16) & 0xFFu), fv = (int)((rv >> 16) & 0xFFu);
      const int ku = (dang && u < n) ? uni((int)kp8[u]) : fu, kv = (dang && v < n) ? uni((int)kp8[v]) : fv;
      const int dst = bump;
      if (lane == 0) {
        rec[u] = (uint32_t)z;
        rec[v] = (uint32_t)z;
        rec[z] = 0x8000u | (uint32_t)dst;
      }
      const int xs = u < n ? (int)permL[u] : u, ys = v < n ? (int)permL[v] : v;  // (for the links, at the end of the step)
      wsync();
      GP_T(2);
      int nz = 0;
      uint32_t sh = 0;
      bool over = false, lead1 = false;
      int y1 = 0;
      uint32_t w1 = 0;
      for (int c0 = 0; c0 < total; c0 += 64) {
        const int j = c0 + lane;
        const bool valid = j < total;
        const int src = j < lu ? ou + j : ov + (j - lu);
        const int e = valid ? (int)ids[src] : 0;
        const uint32_t m = valid ? (uint32_t)mult[src] : 0u;
        int rr = e;
        if (valid)
          for (;;) {  // up to the live tensor that holds the leg now
            const uint32_t rx = rec[rr];
            if (rx & 0x8000u) break;
            rr = (int)rx;
          }
        if (valid && rr != e) rec[e] = (uint32_t)rr;  // (e is dead: a shorter way up for the next look-up)
        const bool cand = valid && rr != z;
        sh += (valid && rr == z) ? m : 0u;
        // an entry of an earlier chunk?  (positions < nz are checked against the list: a stale mark is harmless)
        const int p0 = (cand && c0 > 0) ? (int)mark[rr] : 0;
        const bool old = cand && c0 > 0 && p0 < nz && (int)ids[dst + p0] == rr;
        if (cand) mark[rr] = (uint8_t)lane;
        wsync();
        const int q = cand ? (int)mark[rr] : lane;  // the lane that speaks for this neighbour
        const bool leader = cand && q == lane;
        uint32_t w = m;
        if (__any(cand && !leader)) {  // equal neighbours: their shared legs add up in the speaker's cell
          acc[lane] = leader ? m : 0u;
          wsync();
          if (cand && !leader) atomicAdd((unsigned int*)(uint32_t*)(acc + q), m);
          wsync();
          w = acc[lane];
        }
        const bool app = leader && !old;
        const unsigned long long ab = __ballot(app);
        const int pos = nz + __popcll(ab & ((1ull << lane) - 1ull));
        if (leader && old) {
          w += mult[dst + p0];
          mult[dst + p0] = (uint8_t)w;
          mark[rr] = (uint8_t)p0;
        }
        if (app) {
          ids[dst + pos] = (uint16_t)rr;
          mult[dst + pos] = (uint8_t)w;
          if (total > 64) mark[rr] = (uint8_t)pos;
        }
        over |= leader && w > 255u;
        nz += __popcll(ab);
        lead1 = app;  // (total <= 64: the speakers hold the new list in registers)
        y1 = rr;
        w1 = w;
        wsync();
      }
      GP_T(3);
      const int fz = ku + kv - (int)wsum(sh);  // (a shared leg is in u's list and in v's: sh = 2 w(u, v))
      if (__any(over) || fz > 255) {
        status = 9;
        break;
      }
      if (lane == 0) rec[z] = 0x8000u | (uint32_t)dst | ((uint32_t)fz << 16) | ((uint32_t)nz << 24);
      bump += nz;
      // ---- the cheapest (z, neighbour) into the cell just popped ----
      uint64_t bestk = KMAX;
      if (total <= 64) {
        const int y = lead1 ? y1 : 0;
        const int fy = (int)(((uint32_t)rec[y] >> 16) & 0xFFu), ky = (dang && y < n) ? (int)kp8[y] : fy;
        const uint64_t k = greedy_cand_key(fz + ky - 2 * (int)w1, fz, fy, z, y);
        if (lead1) bestk = k;
      } else {
        for (int j0 = 0; j0 < nz; j0 += 64) {
          const int j = j0 + lane;
          const bool valid = j < nz;
          const int y = ids[dst + (valid ? j : 0)];
          const int w = mult[dst + (valid ? j : 0)];
          const int fy = (int)(((uint32_t)rec[y] >> 16) & 0xFFu), ky = (dang && y < n) ? (int)kp8[y] : fy;
          const uint64_t k = greedy_cand_key(fz + ky - 2 * w, fz, fy, z, y);
          if (valid && k < bestk) bestk = k;
        }
      }
      const uint64_t wk = nz > 0 ? wmin64_2(bestk) : KMAX;
      replace_min(lane == wl, wk);
      if (lane == 0) {
        lk[z] = xs < ys ? xs : ys;
        lk[N + z] = xs < ys ? ys : xs;
        lk[2 * N + xs] = z;
        lk[2 * N + ys] = z;
      }
      ++z;
      wsync();
      GP_T(4);
    }
    if (status == 0 && z != N) status = 3;  // not one tensor left: the host's (outer products)
    if (lane == 0) {
      lk[2 * N + N - 1] = -1;
      p.status[r] = status;
    }
    wsync();
    GP_T(5);
  }
#ifdef TNCO_GREEDY_PROF
  if (lane == 0)
    for (int i = 0; i < 16; ++i) p.prof[(size_t)g * 16 + i] = prof_[i];
#endif
}

// Device memory of a call: ONE block, kept between calls (hipFree of the ~2.5 GB a 65 536-tree batch uses
// took 0.1 s -- as long as half of the kernel).  tnco_hip_greedy_device_release() gives it back.
struct Pool {
  void* base = nullptr;
  size_t bytes = 0;
  int device = -1;
  std::mutex mu;
};
Pool g_pool;

struct DevBufs {
  bool dry = true;   // first pass: only add up the sizes
  size_t off = 0;
  char* base = nullptr;
  template <typename T>
  hipError_t alloc(T** p, size_t count) {
    const size_t nb = (std::max<size_t>(count * sizeof(T), 16) + 255) & ~(size_t)255;
    *p = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += nb;
    return hipSuccess;
  }
  // second pass: the block (from the pool if it is large enough and on this device)
  hipError_t commit(int device) {
    const size_t need = off;
    if (g_pool.base == nullptr || g_pool.bytes < need || g_pool.device != device) {
      if (g_pool.base) (void)hipFree(g_pool.base);
      g_pool.base = nullptr;
      g_pool.bytes = 0;
      const hipError_t e = hipMalloc(&g_pool.base, need);
      if (e != hipSuccess) {
        g_pool.base = nullptr;
        return e;
      }
      g_pool.bytes = need;
      g_pool.device = device;
    }
    base = static_cast<char*>(g_pool.base);
    off = 0;
    dry = false;
    return hipSuccess;
  }
};

void py_init_genrand(uint32_t* mt, uint32_t s) {
  mt[0] = s;
  for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

// The network as a multigraph, if it is one (greedy_graph_kernel's tables); false: the set form's.
struct GraphHost {
  int E = 0, L = 0, CAP = 0;
  std::vector<uint16_t> t_off, t_nbr;
  std::vector<uint8_t> t_mult, t_fp, t_kp;
  std::vector<uint32_t> e_ends;
  std::vector<uint64_t> e_key;
};
bool build_graph(int n, int I, const int32_t* off, const int32_t* holders, const uint64_t* output_mask, GraphHost& g) {
  if (n < 3 || n > 2040) return false;
  if (const char* e = std::getenv("TNCO_HIP_GREEDY_GRAPH"))  // =0: the set form for every network (tests compare the two)
    if (std::atoi(e) == 0) return false;
  std::vector<int> fp((size_t)n, 0), kp((size_t)n, 0);
  std::vector<std::pair<uint32_t, uint32_t>> edges;  // (a << 16 | b, index), a < b
  for (int i = 0; i < I; ++i) {
    const int m = off[i + 1] - off[i];
    const bool is_out = output_mask && ((output_mask[i >> 6] >> (i & 63)) & 1ull);
    if (m > 2 || (is_out && m > 1)) return false;  // a hyper-index
    for (int k = off[i]; k < off[i + 1]; ++k) {
      const int t = holders[k];
      if (t < 0 || t >= n) return false;
      fp[t] += 1;
      if (is_out || m == 2) kp[t] += 1;
    }
    if (m == 2) {
      const int a = std::min(holders[off[i]], holders[off[i] + 1]), b = std::max(holders[off[i]], holders[off[i] + 1]);
      if (a == b) return false;
      edges.emplace_back(((uint32_t)a << 16) | (uint32_t)b, (uint32_t)i);
    }
  }
  for (int t = 0; t < n; ++t)
    if (fp[t] > 255) return false;
  g.E = (int)edges.size();
  if (g.E < 1) return false;
  // shared legs per pair of tensors
  std::vector<std::pair<uint32_t, uint32_t>> byp(edges);
  std::sort(byp.begin(), byp.end());
  std::vector<std::vector<std::pair<int, int>>> adj((size_t)n);
  std::vector<int> w_of((size_t)I, 0);
  for (size_t i = 0; i < byp.size();) {
    size_t j = i;
    while (j < byp.size() && byp[j].first == byp[i].first) ++j;
    const int a = (int)(byp[i].first >> 16), b = (int)(byp[i].first & 0xFFFFu), w = (int)(j - i);
    if (w == fp[a] && w == fp[b]) return false;  // equal index sets among the inputs
    adj[a].emplace_back(b, w);
    adj[b].emplace_back(a, w);
    for (size_t k = i; k < j; ++k) w_of[byp[k].second] = w;
    i = j;
  }
  g.t_off.assign((size_t)n + 1, 0);
  g.t_fp.resize((size_t)n);
  g.t_kp.resize((size_t)n);
  size_t L = 0;
  for (int t = 0; t < n; ++t) L += adj[t].size();
  if (L > 12000) return false;
  g.L = (int)L;
  g.CAP = std::max((int)L + 256, n);
  for (int t = 0; t < n; ++t) {
    g.t_off[t + 1] = (uint16_t)(g.t_off[t] + adj[t].size());
    g.t_fp[t] = (uint8_t)fp[t];
    g.t_kp[t] = (uint8_t)kp[t];
    for (auto& e : adj[t]) {
      g.t_nbr.push_back((uint16_t)e.first);
      g.t_mult.push_back((uint8_t)e.second);
    }
  }
  g.t_nbr.resize(std::max<size_t>(g.t_nbr.size(), 1));
  g.t_mult.resize(std::max<size_t>(g.t_mult.size(), 1));
  for (auto& e : edges) {  // (in index order, as the set form queues them: the order does not matter)
    const int a = (int)(e.first >> 16), b = (int)(e.first & 0xFFFFu);
    g.e_ends.push_back((uint32_t)a | ((uint32_t)b << 16));
    g.e_key.push_back(greedy_cost_key(kp[a] + kp[b] - 2 * w_of[e.second], fp[a], fp[b]) << 28);
  }
  return g.E <= 64 * 24 && graph_lds_bytes(n, g.CAP) <= 64 * 1024;
}

size_t lds_bytes(int W, int SMAX, int I, int QC, int TS) {
  return (size_t)QC * 8 + (size_t)4 * W * 8 + (size_t)SMAX * 2 + (size_t)GREEDY_LCAP * 2 + (size_t)(SMAX - 8) * 2 + (size_t)I * 2 + (size_t)TS * 2 + 16;
}

}  // namespace
}  // namespace tnco

using namespace tnco;

#define G_TRY(expr)                      \
  do {                                   \
    const hipError_t e_ = (expr);        \
    if (e_ != hipSuccess) return TNCO_HIP_ERUNTIME; \
  } while (0)

// gives the device memory kept between calls back to the driver
extern "C" void tnco_hip_greedy_device_release(void) {
  std::lock_guard<std::mutex> lock(g_pool.mu);
  if (g_pool.base) {
    (void)hipSetDevice(g_pool.device);
    (void)hipFree(g_pool.base);
  }
  g_pool.base = nullptr;
  g_pool.bytes = 0;
  g_pool.device = -1;
}

static int64_t g_last_redone = -1;
// trees of the last tnco_hip_greedy_trees_device call that the host version did (-1: the whole batch)
extern "C" int64_t tnco_hip_greedy_device_redone(void) { return g_last_redone; }

// 1 when tnco_hip_greedy_trees_device takes this network itself (else it hands the batch to the host version)
extern "C" int tnco_hip_greedy_device_supported(int32_t n_leaves, int32_t n_inds, const int32_t* holders_off) {
  if (n_leaves < 3 || n_leaves > 2000 || n_inds < 1 || n_inds > GREEDY_KEY_MAX_EXP || !holders_off) return 0;
  int64_t q = 0;  // candidates queued at most: the initial ones (a push takes the cell of the pop before it)
  for (int32_t i = 0; i < n_inds; ++i) {
    const int32_t m = holders_off[i + 1] - holders_off[i];
    if (m > GREEDY_MAXH) return 0;
    q += m > 1 ? m - 1 : 0;
  }
  const int W = (n_inds + 63) / 64, SMAX = 2 * n_leaves + 8;
  int TS = 64;
  while (4 * TS < 5 * SMAX) TS <<= 1;
  const int QC = (int)((q + 63) & ~63ll);
  return lds_bytes(W, SMAX, n_inds, QC, TS) <= 64 * 1024 ? 1 : 0;
}

extern "C" int tnco_hip_greedy_trees_device(int32_t device, int32_t n_leaves, int32_t n_inds, const int32_t* holders_off,
                                            const int32_t* holders, const uint64_t* output_mask, int64_t n_replicas,
                                            const uint32_t* seeds, uint64_t* draws, int32_t* links_out,
                                            int32_t** links_device, int32_t n_threads) {
  if (n_leaves < 1 || n_inds < 0 || !holders_off || !holders || !seeds || (!links_out && !links_device) || n_replicas < 0)
    return TNCO_HIP_EINVAL;
  if (links_device) *links_device = nullptr;
  if (n_replicas == 0) return TNCO_HIP_OK;
  g_last_redone = -1;
  if (!tnco_hip_greedy_device_supported(n_leaves, n_inds, holders_off)) {
    // the host version; the trees go to the device afterwards if the caller wants them there
    const size_t cnt = (size_t)n_replicas * 3 * (2 * (size_t)n_leaves - 1);
    std::vector<int32_t> tmp;
    if (!links_out) tmp.resize(cnt);
    int32_t* host = links_out ? links_out : tmp.data();
    const int rc = tnco_hip_greedy_trees(n_leaves, n_inds, holders_off, holders, output_mask, n_replicas, seeds, draws,
                                         host, n_threads);
    if (rc || !links_device) return rc;
    G_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> pool_lock(g_pool.mu);
    DevBufs db;
    int32_t* dl = nullptr;
    for (int pass = 0; pass < 2; ++pass) {
      G_TRY(db.alloc(&dl, cnt));
      if (pass == 0) G_TRY(db.commit(device));
    }
    G_TRY(hipMemcpy(dl, host, cnt * 4, hipMemcpyHostToDevice));
    *links_device = dl;
    return TNCO_HIP_OK;
  }
  const auto t_start = std::chrono::steady_clock::now();
  G_TRY(hipSetDevice(device));
  const int n = n_leaves, I = n_inds, W = (I + 63) / 64, SMAX = 2 * n + 8, NW = (SMAX + 63) / 64;
  const int64_t R = n_replicas, N = 2 * (int64_t)n - 1;
  int64_t q = 0;
  for (int i = 0; i < I; ++i) q += std::max(0, holders_off[i + 1] - holders_off[i] - 1);
  const int Q = (int)q, QC = (Q + 63) & ~63;
  int TS = 64;
  while (4 * TS < 5 * SMAX) TS <<= 1;
  std::vector<uint64_t> leaf((size_t)n * W, 0), outm((size_t)W, 0);
  for (int i = 0; i < I; ++i)
    for (int k = holders_off[i]; k < holders_off[i + 1]; ++k) {
      const int t = holders[k];
      if (t < 0 || t >= n) return TNCO_HIP_EINVAL;
      leaf[(size_t)t * W + (i >> 6)] |= 1ull << (i & 63);
    }
  if (output_mask)
    for (int x = 0; x < W; ++x) outm[x] = output_mask[x];
  uint32_t mt0[624];
  py_init_genrand(mt0, 19650218u);

  int cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  GraphHost gh;
  const bool graph = build_graph(n, I, holders_off, holders, output_mask, gh);
  const size_t lds = graph ? graph_lds_bytes(n, gh.CAP) : lds_bytes(W, SMAX, I, QC, TS);
  int per_cu = (int)std::max<size_t>(1, std::min<size_t>(16, (size_t)(160 * 1024) / lds));
  if (per_cu > 8) per_cu &= ~3;  // (whole wavefronts per SIMD: 13 per CU measured a third slower than 12)
  if (const char* e = std::getenv("TNCO_HIP_GREEDY_PER_CU")) per_cu = std::max(1, std::min(per_cu, std::atoi(e)));  // (experiment: occupancy sensitivity)
  const int G = (int)std::min<int64_t>(R, (int64_t)cus * per_cu);

  std::lock_guard<std::mutex> pool_lock(g_pool.mu);
  DevBufs db;
  ShuffleParams sp{};
  GreedyParams gp{};
  uint32_t *d_seeds, *d_mt0, *d_mt;
  uint64_t *d_draws = nullptr, *d_leaf, *d_out;
  int32_t *d_hoff, *d_hold;
  uint16_t* d_perm;
  uint16_t *d_toff = nullptr, *d_tnbr = nullptr;
  uint8_t *d_tmult = nullptr, *d_tfp = nullptr, *d_tkp = nullptr;
  uint32_t* d_eends = nullptr;
  uint64_t* d_ekey = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    G_TRY(db.alloc(&d_seeds, (size_t)R));
    G_TRY(db.alloc(&d_mt0, 624));
    G_TRY(db.alloc(&d_mt, (size_t)624 * R));
    G_TRY(db.alloc(&d_perm, (size_t)R * n));
    if (draws) G_TRY(db.alloc(&d_draws, (size_t)R));
    G_TRY(db.alloc(&d_leaf, leaf.size()));
    G_TRY(db.alloc(&d_out, outm.size()));
    G_TRY(db.alloc(&d_hoff, (size_t)I + 1));
    G_TRY(db.alloc(&d_hold, (size_t)std::max(1, holders_off[I])));
    if (!graph) {
      G_TRY(db.alloc(&gp.keys, (size_t)G * SMAX * W));
      G_TRY(db.alloc(&gp.nbr, (size_t)G * SMAX * NW));
      G_TRY(db.alloc(&gp.arena, (size_t)G * Q * W));
      G_TRY(db.alloc(&gp.path, (size_t)G * n * 2));
      G_TRY(db.alloc(&gp.slot_of_leaf, (size_t)G * n));
    } else {
      G_TRY(db.alloc(&d_toff, gh.t_off.size()));
      G_TRY(db.alloc(&d_tnbr, gh.t_nbr.size()));
      G_TRY(db.alloc(&d_tmult, gh.t_mult.size()));
      G_TRY(db.alloc(&d_tfp, gh.t_fp.size()));
      G_TRY(db.alloc(&d_tkp, gh.t_kp.size()));
      G_TRY(db.alloc(&d_eends, gh.e_ends.size()));
      G_TRY(db.alloc(&d_ekey, gh.e_key.size()));
    }
    G_TRY(db.alloc(&gp.links, (size_t)R * 3 * N));
    G_TRY(db.alloc(&gp.status, (size_t)R));
    G_TRY(db.alloc(&gp.prof, (size_t)G * 16));
    if (pass == 0) G_TRY(db.commit(device));
  }
  G_TRY(hipMemcpy(d_seeds, seeds, (size_t)R * 4, hipMemcpyHostToDevice));
  G_TRY(hipMemcpy(d_mt0, mt0, sizeof(mt0), hipMemcpyHostToDevice));
  if (draws) G_TRY(hipMemcpy(d_draws, draws, (size_t)R * 8, hipMemcpyHostToDevice));
  G_TRY(hipMemcpy(d_leaf, leaf.data(), leaf.size() * 8, hipMemcpyHostToDevice));
  G_TRY(hipMemcpy(d_out, outm.data(), outm.size() * 8, hipMemcpyHostToDevice));
  G_TRY(hipMemcpy(d_hoff, holders_off, ((size_t)I + 1) * 4, hipMemcpyHostToDevice));
  if (holders_off[I] > 0) G_TRY(hipMemcpy(d_hold, holders, (size_t)holders_off[I] * 4, hipMemcpyHostToDevice));
  if (graph) {
    G_TRY(hipMemcpy(d_toff, gh.t_off.data(), gh.t_off.size() * 2, hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_tnbr, gh.t_nbr.data(), gh.t_nbr.size() * 2, hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_tmult, gh.t_mult.data(), gh.t_mult.size(), hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_tfp, gh.t_fp.data(), gh.t_fp.size(), hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_tkp, gh.t_kp.data(), gh.t_kp.size(), hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_eends, gh.e_ends.data(), gh.e_ends.size() * 4, hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_ekey, gh.e_key.data(), gh.e_key.size() * 8, hipMemcpyHostToDevice));
  }

  const bool dbg = std::getenv("TNCO_HIP_GREEDY_DEBUG") != nullptr;
  if (dbg) {
    G_TRY(hipDeviceSynchronize());
    std::fprintf(stderr, "greedy_device: set-up (allocations, inputs to the device) %.1f ms\n",
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  }
  hipEvent_t ev[4];
  if (dbg) {
    for (auto& e : ev) G_TRY(hipEventCreate(&e));
    G_TRY(hipEventRecord(ev[0], 0));
  }
  sp.n = n; sp.R = R; sp.seeds = d_seeds; sp.draws = d_draws; sp.mt0 = d_mt0; sp.mt = d_mt; sp.perm = d_perm;
  {
    // the state in LDS where 4+ trees fit a workgroup's 64 KB; TNCO_HIP_SHUFFLE_LDS=0: the state in memory (tests)
    const size_t per_tree = (size_t)624 * 4 + (size_t)n * 2;
    const char* e = std::getenv("TNCO_HIP_SHUFFLE_LDS");
    const bool in_lds = !(e && std::atoi(e) == 0) && 4 * per_tree <= 64 * 1024;
    if (!in_lds) {
      hipLaunchKernelGGL(py_shuffle_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, 0, sp);
    } else if (16 * per_tree <= 64 * 1024) {
      hipLaunchKernelGGL(py_shuffle_lds_kernel<16>, dim3((unsigned)((R + 15) / 16)), dim3(64), 16 * per_tree, 0, sp);
    } else if (8 * per_tree <= 64 * 1024) {
      hipLaunchKernelGGL(py_shuffle_lds_kernel<8>, dim3((unsigned)((R + 7) / 8)), dim3(64), 8 * per_tree, 0, sp);
    } else {
      hipLaunchKernelGGL(py_shuffle_lds_kernel<4>, dim3((unsigned)((R + 3) / 4)), dim3(64), 4 * per_tree, 0, sp);
    }
  }
  G_TRY(hipGetLastError());
  if (dbg) G_TRY(hipEventRecord(ev[1], 0));
  gp.n = n; gp.I = I; gp.W = W; gp.NW = NW; gp.SMAX = SMAX; gp.Q = Q; gp.TS = TS; gp.R = R;
  gp.leaf = d_leaf; gp.output = d_out; gp.hoff = d_hoff; gp.holders = d_hold; gp.perm = d_perm;
  if (graph) {
    GraphParams qp{};
    qp.n = n; qp.E = gh.E; qp.L = gh.L; qp.CAP = gh.CAP; qp.R = R; qp.perm = d_perm;
    qp.dangling = 0;
    for (int t = 0; t < n; ++t)
      if (gh.t_kp[t] != gh.t_fp[t]) qp.dangling = 1;
    qp.t_off = d_toff; qp.t_nbr = d_tnbr; qp.t_mult = d_tmult; qp.t_fp = d_tfp; qp.t_kp = d_tkp;
    qp.e_ends = d_eends; qp.e_key = d_ekey; qp.links = gp.links; qp.status = gp.status; qp.prof = gp.prof;
    const int rows = (gh.E + 63) / 64;
#define TNCO_GRAPH_LAUNCH(RW) hipLaunchKernelGGL(greedy_graph_kernel<RW>, dim3((unsigned)G), dim3(64), lds, 0, qp)
    if (rows <= 4) TNCO_GRAPH_LAUNCH(4);
    else if (rows <= 8) TNCO_GRAPH_LAUNCH(8);
    else if (rows <= 12) TNCO_GRAPH_LAUNCH(12);
    else if (rows <= 16) TNCO_GRAPH_LAUNCH(16);
    else TNCO_GRAPH_LAUNCH(24);
#undef TNCO_GRAPH_LAUNCH
  } else {
    hipLaunchKernelGGL(greedy_kernel, dim3((unsigned)G), dim3(64), lds, 0, gp);
  }
  G_TRY(hipGetLastError());
  if (dbg) G_TRY(hipEventRecord(ev[2], 0));
  G_TRY(hipDeviceSynchronize());
  if (links_out) G_TRY(hipMemcpy(links_out, gp.links, (size_t)R * 3 * N * 4, hipMemcpyDeviceToHost));
  if (dbg) {
    G_TRY(hipEventRecord(ev[3], 0));
    G_TRY(hipEventSynchronize(ev[3]));
    float a = 0, b = 0, c = 0;
    (void)hipEventElapsedTime(&a, ev[0], ev[1]);
    (void)hipEventElapsedTime(&b, ev[1], ev[2]);
    (void)hipEventElapsedTime(&c, ev[2], ev[3]);
    std::fprintf(stderr, "greedy_device: shuffle kernel %.1f ms, greedy kernel (%s form) %.1f ms (%d wavefronts, %zu B LDS each), "
                 "links to the host %.1f ms\n", a, graph ? "graph" : "set", b, G, lds, c);
    for (auto& e : ev) (void)hipEventDestroy(e);
#ifdef TNCO_GREEDY_PROF
    std::vector<unsigned long long> pr((size_t)G * 16);
    G_TRY(hipMemcpy(pr.data(), gp.prof, pr.size() * 8, hipMemcpyDeviceToHost));
    const char* nmg[12] = {"set-up", "pop", "bookkeeping, links", "lists joined", "candidates + push", "end", "-", "-", "-", "-", "-", "-"};
    const char* nms[12] = {"clear", "inputs", "dims: counts, neighbours", "dims: candidates", "pop", "slot of the result", "holders per dim", "neighbour rows", "list + broadcast", "evaluate", "winner + push", "links"};
    const char** nm = graph ? nmg : nms;
    double tot = 0;
    for (size_t i = 0; i < pr.size(); ++i) tot += (double)pr[i];
    for (int i = 0; i < 12; ++i) {
      double v = 0;
      for (int g2 = 0; g2 < G; ++g2) v += (double)pr[(size_t)g2 * 16 + i];
      std::fprintf(stderr, "  %-26s %5.1f %%  %9.0f ticks per tree\n", nm[i], 100 * v / tot, v / (double)R);
    }
#endif
  }
  std::vector<int32_t> status((size_t)R);
  G_TRY(hipMemcpy(status.data(), gp.status, (size_t)R * 4, hipMemcpyDeviceToHost));
  std::vector<uint64_t> draws_in;
  if (draws) {
    draws_in.assign(draws, draws + R);
    G_TRY(hipMemcpy(draws, d_draws, (size_t)R * 8, hipMemcpyDeviceToHost));
  }
  // the trees the kernel left to the host (outer products at the end, limits)
  std::vector<int64_t> redo;
  for (int64_t r = 0; r < R; ++r)
    if (status[r] != 0) redo.push_back(r);
  g_last_redone = (int64_t)redo.size();
  if (!redo.empty()) {
    if (std::getenv("TNCO_HIP_GREEDY_DEBUG"))
      for (size_t k = 0; k < std::min<size_t>(redo.size(), 8); ++k)
        std::fprintf(stderr, "greedy_device: tree %lld status %d\n", (long long)redo[k], status[redo[k]]);
    std::vector<uint32_t> s2(redo.size());
    std::vector<uint64_t> d2(redo.size());
    std::vector<int32_t> l2(redo.size() * 3 * (size_t)N);
    for (size_t k = 0; k < redo.size(); ++k) {
      s2[k] = seeds[redo[k]];
      if (draws) d2[k] = draws_in[redo[k]];
    }
    const int rc = tnco_hip_greedy_trees(n_leaves, n_inds, holders_off, holders, output_mask, (int64_t)redo.size(),
                                         s2.data(), draws ? d2.data() : nullptr, l2.data(), n_threads);
    if (rc) return rc;
    for (size_t k = 0; k < redo.size(); ++k) {
      if (links_out) std::memcpy(links_out + redo[k] * 3 * N, l2.data() + k * 3 * (size_t)N, (size_t)3 * N * 4);
      G_TRY(hipMemcpy(gp.links + redo[k] * 3 * N, l2.data() + k * 3 * (size_t)N, (size_t)3 * N * 4, hipMemcpyHostToDevice));
      if (draws) draws[redo[k]] = d2[k];
    }
  }
  if (dbg)
    std::fprintf(stderr, "greedy_device: %.1f ms from entry to the return (before the buffers are freed)\n",
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  if (links_device) *links_device = gp.links;
  return TNCO_HIP_OK;
}

// device -> host copy of a buffer this library handed out (the trees of tnco_hip_greedy_trees_device)
extern "C" int tnco_hip_copy_to_host(void* dst, const void* device_src, uint64_t bytes) {
  if (!dst || !device_src) return TNCO_HIP_EINVAL;
  G_TRY(hipMemcpy(dst, device_src, (size_t)bytes, hipMemcpyDeviceToHost));
  return TNCO_HIP_OK;
}
