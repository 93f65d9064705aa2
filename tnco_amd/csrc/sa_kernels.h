// sa_kernels.h -- gfx950 device code: batched simulated annealing of contraction trees.
//
// One replica (one annealing run of the reference,
// tnco/app/infinite_memory/sa.py:166-234) is owned by a GROUP of L = 2^LOG2L
// adjacent lanes of a wavefront; lane w of the group holds word w of every leg
// bitmask, so a mask is one coalesced L*8-byte row and the set operations of
// include/tnco/optimize/infinite_memory/optimizer.hpp:147,171-172 are one VALU
// op per lane.  Popcounts / intersects are reduced across the group with DPP
// (quad_perm / row_half_mirror / row_mirror), never through memory.  Scalar
// state (links, costs) is computed redundantly by every lane of the group and
// stored by lane 0.  A wavefront therefore advances 64/L replicas, each at its
// own position of its own leaf->root walk: the sweep loop is flattened so that
// groups do not wait for each other at sweep boundaries.
//
// Memory is replica-major: all arrays of one replica are contiguous, node
// records are 32 B (links + contraction cost + partial cost), masks of
// internal nodes are rows padded to L words (one 128-B line at L = 16).  Leaf
// masks never change and are shared by all replicas.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tnco {

struct __attribute__((aligned(32))) NodeRec {
  int32_t left, right, parent, pad;
  double ccost;    // CostCache::contraction_cost
  double partial;  // CostCache::partial_cost
};
struct __attribute__((aligned(16))) Links {
  int32_t left, right, parent, pad;
};
struct __attribute__((aligned(128))) ReplicaState {
  double min_cost;  // min_total_cost
  double init_total;
  unsigned long long n_moves, n_accepted, n_improved;
  int32_t mti;     // outputs consumed in the current MT generation, 0..624
  int32_t mtw;     // state words already twisted in the current generation
  int32_t status;  // 0 ok, else validity code
  int32_t jinvalid;               // journal overflowed: next improvement takes a full copy
  unsigned long long n_randpick;  // moves whose (D, E) order was drawn at random
  // rotation journal (ring of JCAP entries): min_ctree == minlinks + entries [jhead, jmin)
  uint32_t jhead, jmin, jtail, pad0;
  unsigned long long n_fullcopy;
  unsigned long long pad1[5];
};
static_assert(sizeof(NodeRec) == 32, "NodeRec");
static_assert(sizeof(ReplicaState) == 128, "ReplicaState");

// One accepted rotation (Tree::swap_with_nn, include/tnco/tree.hpp:176-184), fully resolved so
// that replaying it needs no loads: A.child[slotC] = E; B.child[slotE] = C; C.parent = B;
// E.parent = A.  Bit 30 of a = slotC (1: right), bit 30 of b = slotE.
struct __attribute__((aligned(16))) JEntry {
  int32_t a, b, c, e;
};
constexpr int JCAP = 256;  // journal capacity per replica (power of two)

struct Params {
  int32_t n, N, I, W;
  int64_t R;
  NodeRec* rec;              // [R][N]
  uint64_t* imask;           // [R][n-1][L]   legs of internal nodes
  uint64_t* hyper;           // [R][n-1][L]   HyperCache (NULL when the TN has no hyper legs)
  uint32_t* mt;              // [R][624]
  ReplicaState* rs;          // [R]
  Links* minlinks;           // [R][N]        min_ctree checkpoint (links only; legs re-derived on read)
  JEntry* journal;           // [R][JCAP]     rotations accepted since the checkpoint
  const uint64_t* leafmask;  // [n][L]
  const uint64_t* outmask;   // [L]
  int32_t cost_mode;         // 0: uniform dims = 2^log2d; 1: uniform dims table; 2: per-index dims
  int32_t log2d;
  const double* ctab;        // [I+1]  d^k in cost_type (mode 1)
  const double* dimsd;       // [L*64] dims in cost_type (mode 2)
  const uint64_t* sparse;    // [L] or NULL
  double n_projs;            // (cost_type)n_projs
  int32_t f32;               // cost_type float32
  int32_t disable_shared;
};

// ---------------------------------------------------------------------------
// group reductions
// ---------------------------------------------------------------------------
#define TNCO_DPP_XOR1 0xB1        // quad_perm [1,0,3,2]
#define TNCO_DPP_XOR2 0x4E        // quad_perm [2,3,0,1]
#define TNCO_DPP_HALF_MIRROR 0x141
#define TNCO_DPP_MIRROR 0x140

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}

// Sum of v over the 2^LOG2L lanes of the group, result in every lane.
template <int LOG2L>
__device__ __forceinline__ uint32_t gsum(uint32_t v) {
  if constexpr (LOG2L >= 1) v += dpp<TNCO_DPP_XOR1>(v);
  if constexpr (LOG2L >= 2) v += dpp<TNCO_DPP_XOR2>(v);
  if constexpr (LOG2L >= 3) v += dpp<TNCO_DPP_HALF_MIRROR>(v);
  if constexpr (LOG2L >= 4) v += dpp<TNCO_DPP_MIRROR>(v);
  if constexpr (LOG2L >= 5) v += (uint32_t)__shfl_xor((int)v, 16);
  if constexpr (LOG2L >= 6) v += (uint32_t)__shfl_xor((int)v, 32);
  return v;
}

template <int LOG2L>
__device__ __forceinline__ bool gany(bool p) {
  return gsum<LOG2L>(p ? 1u : 0u) != 0u;
}

__device__ __forceinline__ double rnd_cost(double x, int f32) {
  return f32 ? (double)(float)x : x;
}

// 2^e as cost_type (exact; overflow -> inf like std::pow).
__device__ __forceinline__ double pow2_cost(int e, int f32) {
  double v = (e > 1023) ? __builtin_huge_val() : __hiloint2double((1023 + e) << 20, 0);
  return f32 ? (double)(float)v : v;
}

// ---------------------------------------------------------------------------
// cost model, generic path (cost modes 0/1/2, optional sparse legs, f32/f64)
// include/tnco/optimize/infinite_memory/cost_model/simple.hpp:37-55,
// simple_sparse_inds.hpp:37-49.
// ---------------------------------------------------------------------------
template <int LOG2L>
__device__ __forceinline__ double seq_product(const Params& P, uint64_t u, int gbase) {
  // running product in cost_type over ascending set bits (Bitset::visit order)
  double c = 1.0;
  for (int w = 0; w < P.W; ++w) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)u, gbase + w);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(u >> 32), gbase + w);
    uint64_t x = ((uint64_t)hi << 32) | lo;
    while (x) {
      const int b = __ffsll((unsigned long long)x) - 1;
      c = rnd_cost(c * P.dimsd[w * 64 + b], P.f32);
      x &= x - 1;
    }
  }
  return c;
}

template <int LOG2L>
__device__ __forceinline__ double uniform_cost(const Params& P, int pc) {
  return P.cost_mode == 0 ? pow2_cost(P.log2d * pc, P.f32) : P.ctab[pc];
}

// cost of contracting two tensors whose leg union is `u` (this lane's word).
template <int LOG2L>
__device__ __forceinline__ double generic_cost(const Params& P, uint64_t u, int lig, int gbase) {
  if (P.sparse == nullptr) {
    if (P.cost_mode <= 1) return uniform_cost<LOG2L>(P, (int)gsum<LOG2L>((uint32_t)__popcll(u)));
    return seq_product<LOG2L>(P, u, gbase);
  }
  const uint64_t s = P.sparse[lig];
  double c1, c2;
  if (P.cost_mode <= 1) {
    const uint32_t v = gsum<LOG2L>((uint32_t)__popcll(u & ~s) | ((uint32_t)__popcll(u & s) << 16));
    c1 = uniform_cost<LOG2L>(P, (int)(v & 0xffffu));
    c2 = uniform_cost<LOG2L>(P, (int)(v >> 16));
  } else {
    c1 = seq_product<LOG2L>(P, u & ~s, gbase);
    c2 = seq_product<LOG2L>(P, u & s, gbase);
  }
  return rnd_cost(c1 * (c2 < P.n_projs ? c2 : P.n_projs), P.f32);
}

// Acceptance probability: include/tnco/optimize/prob/base.hpp:32-52,
// greedy.hpp:33-47, mh.hpp:35-64.
__device__ __forceinline__ double accept_prob(int kind, double beta, double delta, double old_cost,
                                              int f32) {
  if (kind == 0) return 1.0;
  if (kind == 1) return delta <= 0 ? 1.0 : 0.0;
  if (delta <= 0) return 1.0;
  if (old_cost == 0) return 0.0;
  const double x = rnd_cost(1.0 + rnd_cost(delta / old_cost, f32), f32);
  return rnd_cost(pow(x, -beta), f32);
}

// ---------------------------------------------------------------------------
// Best-tree bookkeeping.  The reference deep-copies the whole tree on every
// improvement (`min_ctree = ctree`, optimizer.hpp:198-201).  Here the best tree
// is a checkpoint (minlinks) plus a prefix [jhead, jmin) of a ring journal of
// accepted rotations; an improvement only moves jmin.  Entries are applied to
// the checkpoint when the ring needs room and at the end of every launch.
// ---------------------------------------------------------------------------
template <int LOG2L>
__device__ __forceinline__ void journal_replay(Links* __restrict__ ml, const JEntry* __restrict__ jr,
                                               uint32_t from, uint32_t to, int lig, int gbase) {
  constexpr int L = 1 << LOG2L;
  for (uint32_t k0 = from; k0 != to;) {
    const uint32_t left = to - k0;
    const int cnt = left < (uint32_t)L ? (int)left : L;
    JEntry e{0, 0, 0, 0};
    if (lig < cnt) e = jr[(k0 + (uint32_t)lig) & (JCAP - 1)];
    for (int j = 0; j < cnt; ++j) {
      const int a = __shfl(e.a, gbase + j), b = __shfl(e.b, gbase + j);
      const int c = __shfl(e.c, gbase + j), ee = __shfl(e.e, gbase + j);
      if (lig == 0) {
        const int A = a & 0x3fffffff, B = b & 0x3fffffff;
        if (a & 0x40000000) ml[A].right = ee; else ml[A].left = ee;
        if (b & 0x40000000) ml[B].right = c; else ml[B].left = c;
        ml[c].parent = B;
        ml[ee].parent = A;
      }
    }
    k0 += (uint32_t)cnt;
  }
}


// ---------------------------------------------------------------------------
// mt19937 seeding: one thread per replica (random.tcc:326-343).
// ---------------------------------------------------------------------------
__global__ void mt_seed_kernel(uint32_t* mt, ReplicaState* rs, const uint32_t* seeds, int64_t R) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  uint32_t* st = mt + r * 624;
  uint32_t x = seeds[r];
  st[0] = x;
  for (int i = 1; i < 624; ++i) {
    x ^= x >> 30;
    x *= 1812433253u;
    x += (uint32_t)i;
    st[i] = x;
  }
  rs[r].mti = 624;
  rs[r].mtw = 624;
}

// ---------------------------------------------------------------------------
// Cache construction / validation: CostCache + HyperCache constructors
// (include/tnco/optimize/infinite_memory/utils.hpp:31-57,76-92), get_cost
// (:102-116), ContractionTree::is_valid (include/tnco/ctree.hpp:101-152), over
// the traverse order of include/tnco/utils.hpp:34-51.
//
// links come from `in_links` ([3][N] int32 per replica) when not NULL, else
// from src_rec (NodeRec) / src_links (Links).  Results go to the out_* arrays
// (which may alias P's own arrays).  scratch: 4*N int32 per replica.
// ---------------------------------------------------------------------------
struct BuildArgs {
  const int32_t* in_links; int64_t in_links_stride;
  const NodeRec* src_rec;      // [R][N] or NULL
  const Links* src_links;      // [R][N] or NULL
  const uint64_t* in_masks; int64_t in_masks_stride;  // optional explicit legs [N][W]
  NodeRec* out_rec;            // [R][N]
  uint64_t* out_imask;         // [R][n-1][L]
  uint64_t* out_hyper;         // [R][n-1][L] (HYPER)
  int32_t* scratch;            // [R][4N]
  double* out_total;           // [R] partial[root]
  double* out_sum;             // [R] get_cost() sum
  int32_t* out_status;         // [R]
  int64_t r0;                  // first replica handled by this launch
  int64_t count;
};

template <int LOG2L, bool HYPER>
__global__ __launch_bounds__(256) void build_kernel(const Params P, const BuildArgs a) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;
  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gbase = (tid & 63) & ~(L - 1);
  const int64_t q = (int64_t)blockIdx.x * GPB + (tid >> LOG2L);
  if (q >= a.count) return;
  const int64_t r = a.r0 + q;
  const bool lane0 = lig == 0;
  const int n = P.n, N = P.N;

  NodeRec* rec = a.out_rec + q * (int64_t)N;
  uint64_t* imask = a.out_imask + q * (int64_t)(n - 1) * L;
  uint64_t* hyper = HYPER ? a.out_hyper + q * (int64_t)(n - 1) * L : nullptr;
  int32_t* stack = a.scratch + q * 4 * (int64_t)N;
  int32_t* order = stack + N;
  int32_t* visited = order + N;
  int status = 0;

  // -- links --------------------------------------------------------------
  auto src_left = [&](int i) -> int {
    if (a.in_links) return a.in_links[r * a.in_links_stride + i];
    if (a.src_rec) return a.src_rec[r * (int64_t)N + i].left;
    return a.src_links[r * (int64_t)N + i].left;
  };
  auto src_right = [&](int i) -> int {
    if (a.in_links) return a.in_links[r * a.in_links_stride + N + i];
    if (a.src_rec) return a.src_rec[r * (int64_t)N + i].right;
    return a.src_links[r * (int64_t)N + i].right;
  };
  auto src_parent = [&](int i) -> int {
    if (a.in_links) return a.in_links[r * a.in_links_stride + 2 * (int64_t)N + i];
    if (a.src_rec) return a.src_rec[r * (int64_t)N + i].parent;
    return a.src_links[r * (int64_t)N + i].parent;
  };
  for (int i = lig; i < N; i += L) {
    NodeRec o;
    o.left = src_left(i); o.right = src_right(i); o.parent = src_parent(i); o.pad = 0;
    o.ccost = 0; o.partial = 0;
    rec[i] = o;
    visited[i] = 0;
  }
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // -- traverse (utils.hpp:34-51), every lane of the group redundantly -----
  int sp = 0, k = 0;
  if (lane0) stack[0] = N - 1;
  sp = 1;
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  while (sp > 0) {
    const int pos = stack[sp - 1];
    const int l = rec[pos].left;
    if (visited[pos] || l < 0) {
      --sp;
      if (lane0) order[k] = pos;
      ++k;
    } else {
      const int rr = rec[pos].right;
      if (lane0) { visited[pos] = 1; stack[sp] = rr; stack[sp + 1] = l; }
      sp += 2;
    }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  auto get_mask = [&](int x) -> uint64_t {
    return x < n ? P.leafmask[(int64_t)x * L + lig] : imask[(int64_t)(x - n) * L + lig];
  };

  // -- legs of internal nodes ---------------------------------------------
  if (a.in_masks) {
    const uint64_t* im = a.in_masks + r * a.in_masks_stride;
    for (int p = n; p < N; ++p)
      imask[(int64_t)(p - n) * L + lig] = (lig < P.W) ? im[(int64_t)p * P.W + lig] : 0ull;
    // leaves must be the shared leaf table
    for (int p = 0; p < n; ++p) {
      const uint64_t x = (lig < P.W) ? im[(int64_t)p * P.W + lig] : 0ull;
      if (gany<LOG2L>(x != P.leafmask[(int64_t)p * L + lig])) status = 12;
    }
  } else if constexpr (!HYPER) {
    for (int i = 0; i < N; ++i) {
      const int p = order[i];
      const int l = rec[p].left;
      if (l < 0) continue;
      imask[(int64_t)(p - n) * L + lig] = get_mask(l) ^ get_mask(rec[p].right);
    }
  } else {
    // union of leaves below (in imask), legs held outside (in hyper), then
    // legs = (a ^ b) | (a & b & outside)   [tnco/ctree.py:163-189]
    for (int i = 0; i < N; ++i) {
      const int p = order[i];
      const int l = rec[p].left;
      if (l < 0) continue;
      imask[(int64_t)(p - n) * L + lig] = get_mask(l) | get_mask(rec[p].right);
    }
    for (int i = N - 1; i >= 0; --i) {
      const int p = order[i];
      const int l = rec[p].left;
      if (l < 0) continue;
      const int rr = rec[p].right;
      const uint64_t op = (p == N - 1) ? P.outmask[lig] : hyper[(int64_t)(p - n) * L + lig];
      const uint64_t ul = get_mask(l), ur = get_mask(rr);
      if (l >= n) hyper[(int64_t)(l - n) * L + lig] = op | ur;
      if (rr >= n) hyper[(int64_t)(rr - n) * L + lig] = op | ul;
    }
    if (lig == 0) {}  // (root's outside is the output mask, read above)
    for (int i = 0; i < N; ++i) {
      const int p = order[i];
      const int l = rec[p].left;
      if (l < 0) continue;
      const uint64_t ia = get_mask(l), ib = get_mask(rec[p].right);
      const uint64_t op = (p == N - 1) ? P.outmask[lig] : hyper[(int64_t)(p - n) * L + lig];
      imask[(int64_t)(p - n) * L + lig] = (ia ^ ib) | (ia & ib & op);
    }
  }

  // -- validity, hyper cache, cost caches ----------------------------------
  double sum = 0.0;
  for (int i = 0; i < N; ++i) {
    const int p = order[i];
    const int l = rec[p].left;
    if (l < 0) continue;
    const int rr = rec[p].right;
    const uint64_t ia = get_mask(l), ib = get_mask(rr), ip = imask[(int64_t)(p - n) * L + lig];
    if (!P.disable_shared && !gany<LOG2L>((ia & ib) != 0)) status = status ? status : 10;
    if (gany<LOG2L>((((ia ^ ib) & ~ip) | (ip & ~(ia | ib))) != 0)) status = status ? status : 11;
    if constexpr (HYPER) hyper[(int64_t)(p - n) * L + lig] = ip & ia & ib;
    const double c = generic_cost<LOG2L>(P, ia | ib, lig, gbase);
    const double pl = l < n ? 0.0 : rec[l].partial, pr = rr < n ? 0.0 : rec[rr].partial;
    const double part = rnd_cost(rnd_cost(c + pl, P.f32) + pr, P.f32);  // utils.hpp:54
    sum = rnd_cost(sum + c, P.f32);                                     // utils.hpp:112
    if (lane0) { rec[p].ccost = c; rec[p].partial = part; }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (lane0) {
    a.out_total[q] = rec[N - 1].partial;
    a.out_sum[q] = sum;
    a.out_status[q] = status;
  }
}

// Compare a rebuilt cache set with the live one (is_valid,
// infinite_memory/optimizer.hpp:223-251, is_logclose include/tnco/utils.hpp:78-87).
template <int LOG2L, bool HYPER>
__global__ __launch_bounds__(256) void compare_kernel(const Params P, const BuildArgs a, double atol,
                                                      int32_t* out_bad) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;
  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int64_t q = (int64_t)blockIdx.x * GPB + (tid >> LOG2L);
  if (q >= a.count) return;
  const int64_t r = a.r0 + q;
  const int n = P.n, N = P.N;
  const NodeRec* ref = a.out_rec + q * (int64_t)N;
  const NodeRec* cur = P.rec + r * (int64_t)N;
  int bad = a.out_status[q];
  auto logclose = [&](double x, double y) -> bool {
    if (x < 0 || y < 0) return false;
    if (x == 0 || y == 0) return x == y;
    return fabs(log(x) - log(y)) <= atol;
  };
  for (int i = lig; i < N; i += L) {
    if (!logclose(ref[i].ccost, cur[i].ccost)) bad = bad ? bad : 31;
    if (!logclose(ref[i].partial, cur[i].partial)) bad = bad ? bad : 32;
    if (ref[i].left != cur[i].left || ref[i].right != cur[i].right || ref[i].parent != cur[i].parent)
      bad = bad ? bad : 2;
    if (cur[i].left >= 0 && (cur[cur[i].left].parent != i || cur[cur[i].right].parent != i))
      bad = bad ? bad : 8;
  }
  for (int64_t j = lig; j < (int64_t)(n - 1) * L; j += L) {
    if (a.out_imask[q * (int64_t)(n - 1) * L + j] != P.imask[r * (int64_t)(n - 1) * L + j])
      bad = bad ? bad : 34;
    if constexpr (HYPER)
      if (a.out_hyper[q * (int64_t)(n - 1) * L + j] != P.hyper[r * (int64_t)(n - 1) * L + j])
        bad = bad ? bad : 33;
  }
  bad = (int)gsum<LOG2L>((uint32_t)(bad != 0)) ? (bad ? bad : 99) : 0;
  if (lig == 0) out_bad[q] = bad;
}

}  // namespace tnco
