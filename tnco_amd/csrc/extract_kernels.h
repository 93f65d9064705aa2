// extract_kernels.h -- device side of the result read-back: the k best
// replicas (replaces `sorted(results)` of tnco/app/infinite_memory/sa.py:257 for the head of the
// list), their best trees and get_contraction (include/tnco/utils.hpp:53-71) of each, so that the host
// receives ONE buffer instead of R replica records + 3-4 copies per tree.
#pragma once
#include "sa_kernels.h"

namespace tnco {

// ---- k smallest (min_total_cost, replica) ------------------------------------------------------
// Keys: the cost's bit pattern (costs are >= 0, so the patterns order like the values) and the
// replica id for ties.  One pass: every block sorts TOPK_CHUNK keys in LDS (bitonic) and keeps the
// first `keep`; passes repeat on what was kept until one block is left.
constexpr int TOPK_CHUNK = 4096;

struct TopKey {
  unsigned long long cost;
  uint32_t id;
};
__device__ __forceinline__ bool key_less(const TopKey& a, const TopKey& b) {
  return a.cost < b.cost || (a.cost == b.cost && a.id < b.id);
}

// src_rs != NULL: first pass, keys from the replica records; else from (src_cost, src_id)[count]
static __global__ __launch_bounds__(256) void topk_pass_kernel(const ReplicaState* src_rs, const unsigned long long* src_cost,
                                                               const uint32_t* src_id, int64_t count, int keep,
                                                               unsigned long long* dst_cost, uint32_t* dst_id) {
  __shared__ unsigned long long kc[TOPK_CHUNK];
  __shared__ uint32_t ki[TOPK_CHUNK];
  const int64_t base = (int64_t)blockIdx.x * TOPK_CHUNK;
  for (int i = threadIdx.x; i < TOPK_CHUNK; i += 256) {
    const int64_t g = base + i;
    unsigned long long c = ~0ull;
    uint32_t id = ~0u;
    if (g < count) {
      if (src_rs) {
        c = (unsigned long long)__double_as_longlong(src_rs[g].min_cost);
        id = (uint32_t)g;
      } else {
        c = src_cost[g];
        id = src_id[g];
      }
    }
    kc[i] = c;
    ki[i] = id;
  }
  __syncthreads();
  for (int size = 2; size <= TOPK_CHUNK; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < TOPK_CHUNK / 2; t += 256) {
        const int lo = 2 * t - (t & (stride - 1));  // index with bit `stride` clear
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const TopKey a{kc[lo], ki[lo]}, b{kc[hi], ki[hi]};
        if (key_less(b, a) == up) {
          kc[lo] = b.cost; ki[lo] = b.id;
          kc[hi] = a.cost; ki[hi] = a.id;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < keep; i += 256) {
    dst_cost[(int64_t)blockIdx.x * keep + i] = kc[i];
    dst_id[(int64_t)blockIdx.x * keep + i] = ki[i];
  }
}

// min over replicas of min_total_cost into *out (one block; feeds the RCCL all-reduce from device memory)
static __global__ __launch_bounds__(1024) void min_cost_kernel(const ReplicaState* rs, int64_t R, double* out) {
  __shared__ unsigned long long sm[1024];
  unsigned long long m = ~0ull;
  for (int64_t r = threadIdx.x; r < R; r += 1024) {
    const unsigned long long c = (unsigned long long)__double_as_longlong(rs[r].min_cost);
    m = c < m ? c : m;
  }
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = sm[threadIdx.x + s] < sm[threadIdx.x] ? sm[threadIdx.x + s] : sm[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = __longlong_as_double((long long)sm[0]);
}

// min_total_cost / total_cost of every replica as two dense arrays (R x 8 bytes each cross PCIe
// instead of the 128-byte replica records)
static __global__ __launch_bounds__(256) void gather_costs_kernel(const Params P, double* total, double* mn) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= P.R) return;
  if (mn) mn[r] = P.rs[r].min_cost;
  if (total)
    total[r] = reinterpret_cast<const NodeRec*>(P.blocks + r * P.RB + (int64_t)(P.n - 2) * P.BS)->partial;
}

// slices / min_slices of k replicas: out[q][0..W) <- src[ids[q]][which][0..W)
static __global__ __launch_bounds__(256) void gather_slices_kernel(const uint64_t* src, int LK, int W, const int64_t* ids,
                                                                   int64_t k, uint64_t* out_cur, uint64_t* out_min) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= k * W) return;
  const int64_t q = i / W;
  const int w = (int)(i % W);
  const uint64_t* s = src + ids[q] * 2 * (int64_t)LK;
  if (out_cur) out_cur[i] = s[w];
  if (out_min) out_min[i] = s[LK + w];
}

// ---- trees of k replicas -----------------------------------------------------------------------
// One wavefront per requested replica.  which = 1: min_ctree = checkpoint + rotations jlog[0, jmin)
// (Tree::swap_with_nn, include/tnco/tree.hpp:141-192, replayed by lane 0 on links held in LDS, the
// log read 64 entries at a time by the whole wavefront); which = 0: the current tree from the node
// blocks.  Then the post-order of include/tnco/utils.hpp:34-51 and get_contraction (:53-71) from the
// same LDS copy.  Work area per replica (LDS when it fits, else global scratch): left[N] right[N]
// parent[N] stack[N] visited[N].
template <bool IN_LDS>
static __global__ __launch_bounds__(64) void gather_trees_kernel(const Params P, const int64_t* ids, const int which,
                                                                 int32_t* out_links, int32_t* out_con,
                                                                 int32_t* scratch) {
  extern __shared__ int32_t lds_area[];
  __shared__ int32_t chunk[64];
  const int q = blockIdx.x, lane = threadIdx.x;
  const int64_t r = ids[q];
  const int n = P.n, N = P.N;
  int32_t* L = IN_LDS ? lds_area : scratch + (int64_t)q * 5 * N;
  int32_t *Rr = L + N, *Pp = L + 2 * N, *stack = L + 3 * N, *visited = L + 4 * N;
  if (which == 1) {
    const Links* src = P.minlinks + r * (int64_t)N;
    for (int i = lane; i < N; i += 64) {
      const Links o = src[i];
      L[i] = o.left; Rr[i] = o.right; Pp[i] = o.parent;
    }
    __syncthreads();
    const int32_t* lg = P.jlog + r * (int64_t)P.jcap;
    const uint32_t m = P.rs[r].jmin;
    for (uint32_t k0 = 0; k0 < m; k0 += 64) {
      if (k0 + lane < m) chunk[lane] = lg[k0 + lane];
      __syncthreads();
      if (lane == 0) {
        const uint32_t cnt = m - k0 < 64u ? m - k0 : 64u;
        for (uint32_t k = 0; k < cnt; ++k) {
          const int D = chunk[k];
          const int B = Pp[D];
          const int A = Pp[B];
          const int C = (L[A] == B) ? Rr[A] : L[A];
          if (L[A] != C) Rr[A] = D; else L[A] = D;
          if (L[B] != D) Rr[B] = C; else L[B] = C;
          Pp[C] = B;
          Pp[D] = A;
        }
      }
      __syncthreads();
    }
  } else {
    const uint8_t* blk = P.blocks + r * P.RB;
    const int32_t* lp = P.lpar + r * (int64_t)n * LPS;
    for (int i = lane; i < N; i += 64) {
      if (i < n) {
        L[i] = -1; Rr[i] = -1; Pp[i] = lp[(int64_t)i * LPS];
      } else {
        const NodeRec* hd = reinterpret_cast<const NodeRec*>(blk + (int64_t)(i - n) * P.BS);
        L[i] = hd->left; Rr[i] = hd->right; Pp[i] = hd->parent;
      }
    }
    __syncthreads();
  }
  int32_t* ol = out_links + (int64_t)q * 3 * N;
  for (int i = lane; i < N; i += 64) {
    ol[i] = L[i]; ol[N + i] = Rr[i]; ol[2 * (int64_t)N + i] = Pp[i];
    visited[i] = 0;
  }
  __syncthreads();
  if (out_con && lane == 0) {
    int32_t* oc = out_con + (int64_t)q * 3 * (n - 1);
    int sp = 1, cnt = 0;
    stack[0] = N - 1;
    while (sp > 0) {
      const int pos = stack[sp - 1];
      const int l = L[pos];
      if (visited[pos] || l < 0) {
        --sp;
        if (l >= 0) {
          oc[3 * cnt] = l; oc[3 * cnt + 1] = Rr[pos]; oc[3 * cnt + 2] = pos;
          ++cnt;
        }
      } else {
        visited[pos] = 1;
        stack[sp] = Rr[pos];
        stack[sp + 1] = l;
        sp += 2;
      }
    }
  }
}

}  // namespace tnco
