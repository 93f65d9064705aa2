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
struct __attribute__((aligned(64))) ReplicaState {
  double min_cost;  // min_total_cost
  double init_total;
  unsigned long long n_moves, n_accepted, n_improved;
  int32_t mti;     // outputs consumed in the current MT generation, 0..624
  int32_t mtw;     // state words already twisted in the current generation
  int32_t status;  // 0 ok, else validity code
  int32_t pad;
  unsigned long long n_randpick;  // moves whose (D, E) order was drawn at random
};
static_assert(sizeof(NodeRec) == 32, "NodeRec");
static_assert(sizeof(ReplicaState) == 64, "ReplicaState");

struct Params {
  int32_t n, N, I, W;
  int64_t R;
  NodeRec* rec;              // [R][N]
  uint64_t* imask;           // [R][n-1][L]   legs of internal nodes
  uint64_t* hyper;           // [R][n-1][L]   HyperCache (NULL when the TN has no hyper legs)
  uint32_t* mt;              // [R][624]
  ReplicaState* rs;          // [R]
  Links* minlinks;           // [R][N]        min_ctree (links only; legs re-derived on read)
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
// std::mt19937, generated lazily in blocks of 16 outputs per group.
// State words live in HBM ([624] per replica); the tempered outputs of the
// current block live in a 64-byte LDS slot of the group.
// (libstdc++ random.tcc:396-471; seeding :326-343.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mt_temper(uint32_t z) {
  z ^= (z >> 11);
  z ^= (z << 7) & 0x9d2c5680u;
  z ^= (z << 15) & 0xefc60000u;
  z ^= (z >> 18);
  return z;
}

template <int LOG2L>
struct Rng {
  uint32_t* st;            // replica's 624 state words (HBM)
  volatile uint32_t* buf;  // group's 16-word LDS slot
  int mti, mtw, cur_blk, lig;

  __device__ __forceinline__ void refill(int blk) {
    constexpr int L = 1 << LOG2L;
    const bool twist = (blk * 16) >= mtw;
    for (int t = lig; t < 16; t += L) {
      const int k = blk * 16 + t;
      uint32_t v;
      if (twist) {
        const uint32_t a = st[k];
        const int k1 = (k + 1 == 624) ? 0 : k + 1;
        const uint32_t b = st[k1];
        int km = k + 397;
        if (km >= 624) km -= 624;
        const uint32_t c = st[km];
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        v = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        st[k] = v;
      } else {
        v = st[k];
      }
      buf[t] = mt_temper(v);
      // passes must stay in ascending order: a later pass overwrites words an
      // earlier pass read as "old" (cross-lane write-after-read).
      __asm__ volatile("" ::: "memory");
    }
    if (twist) mtw = blk * 16 + 16;
    cur_blk = blk;
  }

  __device__ __forceinline__ uint32_t next() {
    if (mti >= 624) {
      mti = 0;
      mtw = 0;
      cur_blk = -1;
    }
    const int blk = mti >> 4;
    if (blk != cur_blk) refill(blk);
    const uint32_t v = buf[mti & 15];
    ++mti;
    return v;
  }

  // std::uniform_real_distribution<double>{} == generate_canonical<double,53>
  // (random.tcc:3348-3380): low word first, one rounding, scale by 2^-64.
  __device__ __forceinline__ double uniform01() {
    const uint32_t x1 = next();
    const uint32_t x2 = next();
    double s = (double)x1 + (double)x2 * 4294967296.0;
    double r = s * 5.421010862427522170037e-20;  // 2^-64
    if (r >= 1.0) r = 0.99999999999999988897769753748;  // nextafter(1, 0)
    return r;
  }
};

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
// The sweep kernel: n_steps calls of Optimizer::update
// (include/tnco/optimize/infinite_memory/optimizer.hpp:90-221) per replica.
// ---------------------------------------------------------------------------
template <int LOG2L, bool HYPER, bool GENERIC>
__global__ __launch_bounds__(256) void sa_run_kernel(const Params P, const double* __restrict__ betas,
                                                     const int64_t n_steps, const int prob_kind) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;  // groups (replicas) per block
  __shared__ uint32_t rngbuf[GPB * 16];

  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gib = tid >> LOG2L;
  const int gbase = (tid & 63) & ~(L - 1);  // first lane of the group inside the wave
  const int64_t r = (int64_t)blockIdx.x * GPB + gib;
  if (r >= P.R || n_steps <= 0) return;
  const bool lane0 = (lig == 0);

  const int n = P.n, N = P.N;
  NodeRec* __restrict__ rec = P.rec + r * (int64_t)N;
  uint64_t* __restrict__ imask = P.imask + r * (int64_t)(n - 1) * L;
  uint64_t* __restrict__ hyper = HYPER ? P.hyper + r * (int64_t)(n - 1) * L : nullptr;
  const uint64_t* __restrict__ leafmask = P.leafmask;
  ReplicaState* rs = P.rs + r;

  Rng<LOG2L> rng;
  rng.st = P.mt + r * 624;
  rng.buf = rngbuf + gib * 16;
  rng.mti = rs->mti;
  rng.mtw = rs->mtw;
  rng.cur_blk = -1;
  rng.lig = lig;

  double min_cost = rs->min_cost;
  unsigned long long n_moves = 0, n_acc = 0, n_impr = 0, n_rpick = 0;
  const int f32 = GENERIC ? P.f32 : 0;
  const int log2d = P.log2d;
  const bool disable_shared = P.disable_shared != 0;

  auto load_mask = [&](int x) -> uint64_t {
    return x < n ? leafmask[(int64_t)x * L + lig] : imask[(int64_t)(x - n) * L + lig];
  };
  auto load_partial = [&](int x) -> double { return x < n ? 0.0 : rec[x].partial; };

  // ---- carried state: B and what is known about its two children ----------
  int B, bl, br, bA;
  double ccB, partB, total, beta;
  uint64_t m0, m1, iB = 0, hB = 0;
  double p0, p1;

  auto start_sweep = [&](int64_t step) {
    beta = betas[step];
    // optimizer.hpp:103-112
    const uint32_t x = rng.next();
    const int leaf = (int)(x % (uint32_t)n);
    B = rec[leaf].parent;
    const NodeRec rb = rec[B];
    bl = rb.left;
    br = rb.right;
    bA = rb.parent;
    ccB = rb.ccost;
    partB = rb.partial;
    total = (B == N - 1) ? partB : rec[N - 1].partial;
    m0 = load_mask(bl);
    m1 = load_mask(br);
    p0 = load_partial(bl);
    p1 = load_partial(br);
    if constexpr (HYPER) {
      iB = imask[(int64_t)(B - n) * L + lig];
      hB = hyper[(int64_t)(B - n) * L + lig];
    }
  };

  int64_t step = 0;
  start_sweep(0);

  for (;;) {
    if (bA < 0) {
      // ---- B is the root: end of sweep (optimizer.hpp:194-201) ------------
      if (lane0) {
        NodeRec o;
        o.left = bl; o.right = br; o.parent = -1; o.pad = 0; o.ccost = ccB; o.partial = partB;
        rec[B] = o;
      }
      if (partB < min_cost) {
        min_cost = partB;
        ++n_impr;
        Links* __restrict__ ml = P.minlinks + r * (int64_t)N;
        for (int i = lig; i < N; i += L) ml[i] = *reinterpret_cast<const Links*>(&rec[i]);
      }
      ++step;
      if (step >= n_steps) break;
      start_sweep(step);
      if (bA < 0) continue;
    }

    // ---- one move evaluation (optimizer.hpp:117-192) -----------------------
    const int A = bA;
    const NodeRec ra = rec[A];
    int al = ra.left, ar = ra.right;
    const int aP = ra.parent;
    double ccA = ra.ccost;
    // get_ctree_nn, optimize/optimizer.hpp:121-144
    const bool c_is_right = (al == B);
    const int C = c_is_right ? ar : al;
    const uint64_t mC = load_mask(C);
    const double pC = load_partial(C);
    uint64_t iA = 0, hA = 0;
    if constexpr (HYPER) {
      iA = imask[(int64_t)(A - n) * L + lig];
      hA = hyper[(int64_t)(A - n) * L + lig];
    }
    const uint64_t hy = HYPER ? (hA | hB) : 0ull;
    // both candidate (D, E) assignments evaluated at once:
    //   cand0: D = child0, E = child1;  cand1: D = child1, E = child0
    const uint64_t nb0 = (m0 ^ mC) | hy;  // optimizer.hpp:147
    const uint64_t nb1 = (m1 ^ mC) | hy;
    bool inter0, inter1;
    int pcA0 = 0, pcB0 = 0, pcA1 = 0, pcB1 = 0;
    if constexpr (!GENERIC) {
      uint32_t w0 = (uint32_t)__popcll(nb0 | m1) | ((uint32_t)__popcll(m0 | mC) << 13) |
                    (((m0 & mC) != 0 ? 1u : 0u) << 26);
      uint32_t w1 = (uint32_t)__popcll(nb1 | m0) | ((uint32_t)__popcll(m1 | mC) << 13) |
                    (((m1 & mC) != 0 ? 1u : 0u) << 26);
      w0 = gsum<LOG2L>(w0);
      w1 = gsum<LOG2L>(w1);
      inter0 = (w0 >> 26) != 0;
      inter1 = (w1 >> 26) != 0;
      pcA0 = (int)(w0 & 0x1fffu); pcB0 = (int)((w0 >> 13) & 0x1fffu);
      pcA1 = (int)(w1 & 0x1fffu); pcB1 = (int)((w1 >> 13) & 0x1fffu);
    } else {
      const uint32_t w = gsum<LOG2L>(((m0 & mC) != 0 ? 1u : 0u) | (((m1 & mC) != 0 ? 1u : 0u) << 8));
      inter0 = (w & 0xffu) != 0;
      inter1 = (w >> 8) != 0;
    }
    bool pick0;  // true: (D, E) = (child0, child1)
    if (disable_shared || (inter0 && inter1)) {
      pick0 = (rng.next() & 1u) != 0;  // optimize/optimizer.hpp:139
      ++n_rpick;
    } else {
      pick0 = inter0;
    }
    const uint64_t mD = pick0 ? m0 : m1, mE = pick0 ? m1 : m0;
    const uint64_t newB = pick0 ? nb0 : nb1;
    const double pD = pick0 ? p0 : p1, pE = pick0 ? p1 : p0;
    const int E = pick0 ? br : bl;

    double nA, nB;  // optimizer.hpp:152-155
    if constexpr (!GENERIC) {
      nA = pow2_cost(log2d * (pick0 ? pcA0 : pcA1), 0);
      nB = pow2_cost(log2d * (pick0 ? pcB0 : pcB1), 0);
    } else {
      nA = generic_cost<LOG2L>(P, newB | mE, lig, gbase);
      nB = generic_cost<LOG2L>(P, mD | mC, lig, gbase);
    }
    const double delta = rnd_cost(rnd_cost(nB - ccB, f32) + rnd_cost(nA - ccA, f32), f32);  // :158
    ++n_moves;

    const double u = rng.uniform01();  // :162 (always drawn)
    const bool acc = u <= accept_prob(prob_kind, beta, delta, total, f32);

    double pEcur = pE, pCcur = pC;  // partials of B's other child / A's other child after the move
    uint64_t mBnow;                 // legs of B after the move
    if (acc) {
      ++n_acc;
      // Tree::swap_with_nn(E): include/tnco/tree.hpp:176-184
      if (pick0) br = C; else bl = C;
      if (c_is_right) ar = E; else al = E;
      if (lane0) {
        rec[C].parent = B;
        rec[E].parent = A;
      }
      imask[(int64_t)(B - n) * L + lig] = newB;  // :170
      if constexpr (HYPER) {
        hA = iA & newB & mE;  // :171
        hB = newB & mD & mC;  // :172
        hyper[(int64_t)(A - n) * L + lig] = hA;
        hyper[(int64_t)(B - n) * L + lig] = hB;
      }
      ccB = nB;
      ccA = nA;
      total = rnd_cost(total + delta, f32);  // :177
      pEcur = pC;
      pCcur = pE;
      mBnow = newB;
    } else {
      mBnow = HYPER ? iB : (m0 ^ m1);
    }
    // :185-188
    partB = rnd_cost(rnd_cost(pD + pEcur, f32) + ccB, f32);
    const double partA = rnd_cost(rnd_cost(partB + pCcur, f32) + ccA, f32);
    if (lane0) {
      NodeRec o;
      o.left = bl; o.right = br; o.parent = A; o.pad = 0; o.ccost = ccB; o.partial = partB;
      rec[B] = o;
    }
    // :191  B <- A, carrying what we already know about A's children
    const uint64_t mX = acc ? mE : mC;  // legs of A's other child
    if (c_is_right) { m0 = mBnow; p0 = partB; m1 = mX; p1 = pCcur; }
    else            { m1 = mBnow; p1 = partB; m0 = mX; p0 = pCcur; }
    B = A; bl = al; br = ar; bA = aP; ccB = ccA; partB = partA;
    if constexpr (HYPER) { iB = iA; hB = hA; }
  }

  if (lane0) {
    rs->min_cost = min_cost;
    rs->n_moves += n_moves;
    rs->n_accepted += n_acc;
    rs->n_improved += n_impr;
    rs->n_randpick += n_rpick;
    rs->mti = rng.mti;
    rs->mtw = rng.mtw;
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
