// sa_kernels.h -- gfx950 device code shared by the kernels: data layout, group reductions,
// cost models, cache construction / validation.
//
// One replica (one annealing run of the reference, tnco/app/infinite_memory/sa.py:166-234) is
// owned by a GROUP of L = 2^LOG2L adjacent lanes of a wavefront; lane j of the group holds the K
// words j, L+j, 2L+j, ... of every leg bitmask (L*K >= W = ceil(n_inds/64); one load instruction
// of the group covers L consecutive words = one contiguous piece of the block), so the set
// operations of include/tnco/optimize/infinite_memory/optimizer.hpp:147,171-172 are K VALU ops
// per lane and popcounts / intersects are reduced across the group with DPP, never through
// memory.  Scalar state is computed redundantly by every lane of the group and stored by lane 0:
// small groups (L = 4) keep that redundancy low and put 16 replicas in one wavefront.
//
// HBM layout, replica-major (everything of one replica is contiguous):
//   node block of internal node p (BS bytes, BS = 32 + 8*W rounded to 32; with or without hyper-indices):
//       [ left right parent pad | ccost | partial | legs: W words ]
//     so one move touches ONE line per node it reads or writes (512-leaf TN: W = 12, BS = 128 B
//     = exactly one 128-B line).  HBM here is bound by the number of random line activations,
//     not by bytes.
//   leaf parents: int32[n] (a leaf has no children, no cost, and its legs never change: the leg
//     masks of the leaves are one table shared by all replicas).
//   mt19937 state: 624 words.  best tree: checkpoint links + a log of rotations (below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tnco {

// Stride (in int32) of the leaf-parent array.  1: packed int32[n] per replica.  16 (experiment,
// -DTNCO_LPS=16): every leaf's parent is a 64-byte record of its own, rewritten whole, because HBM
// writes whole 64-byte pieces and a 4-byte update is a read-modify-write there
// (tools/hbm_random.hip).  Measured -1.5 % on the 512-leaf workload: the packed array of 65536
// replicas (128 MiB) lives in the 256-MiB Infinity Cache, the 2-GiB record array does not.
#ifndef TNCO_LPS
#define TNCO_LPS 1
#endif
constexpr int LPS = TNCO_LPS;

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
  int32_t jinvalid;               // rotation log overflowed: next improvement takes a full copy
  unsigned long long n_randpick;  // moves whose (D, E) order was drawn at random
  // min_ctree == checkpoint (minlinks) + rotations jlog[0, jmin) ; jlog holds jtail entries
  uint32_t jmin, jtail, pad0, pad2;
  unsigned long long n_fullcopy;
  unsigned long long pad1[5];
};
static_assert(sizeof(NodeRec) == 32, "NodeRec");
static_assert(sizeof(ReplicaState) == 128, "ReplicaState");

struct Params {
  int32_t n, N, I, W;
  int32_t BS;                // bytes between the HEADERS of consecutive nodes (unified layout: the node block)
  // Two layouts of a replica's region (RB bytes, at blocks + r * RB):
  //   unified  [ header | legs (| hyper legs) ] per node, BS bytes: WOFF = 32, WS = BS -- one line per node
  //            at <= 12 mask words (the infinite-memory benchmark);
  //   split    [ all headers, 32 B each ][ all legs (| hyper legs), WS bytes each, 128-byte aligned ]: BS = 32 -- the
  //            finite-width optimizer: its node blocks (15 words: 160 B) straddled lines anyway, the walk of
  //            the re-slice reads the header array as ONE coalesced 17-KB piece instead of n - 1 random
  //            32-byte reads, and a kept re-slice rewrites it the same way.
  int32_t WS;                // bytes between the legs of consecutive nodes
  int32_t WOFF;              // byte offset of node n's legs from the replica's base
  int64_t RB;                // bytes per replica
  // Networks with hyper-indices.  hyper[p] = legs(p) & legs(c0) & legs(c1) (infinite_memory/utils.hpp:82-91) is a function
  // of legs the kernels hold anyway, so NO layout stores it (round 5: the blocks are those of a network without
  // hyper-indices; rounds 1-4 kept W more words per node and read / wrote them with every move).  build_kernel needs a
  // second mask per node while it derives the legs of a tree: BuildArgs::hyper_tmp (View::hyp).
  int32_t jcap;              // rotation-log entries per replica
  int64_t R;
  uint8_t* blocks;           // [R][RB]
  int32_t* lpar;             // [R][n][LPS]   parent of every leaf (one 64-byte record per leaf)
  uint32_t* mt;              // [R][624]
  uint32_t* mtshadow;        // [R][MT_SHADOW = 64]      generation-g words overwritten ahead of consumption
  ReplicaState* rs;          // [R]
  Links* minlinks;           // [R][N]        best-tree checkpoint (links only; legs re-derived on read)
  int32_t* jlog;             // [R][jcap]     node E of every accepted rotation since the checkpoint
  const uint64_t* leafmask;  // [n][L*K]
  const uint64_t* outmask;   // [L*K]
  int32_t cost_mode;         // 0: uniform dims = 2^log2d; 1: uniform dims table; 2: per-index dims
                             // (dims[p] = 2^a[p] * odd[p]: exponent classes + a sequential product over
                             // the odd parts); 3: per-index dims that are all powers of two
  int32_t log2d;
  const double* ctab;        // [64*W+1]  d^k in cost_type (mode 1)
  const double* dimsd;       // [L*K*64] ODD PARTS of the dims in cost_type (mode 2)
  const uint64_t* oddmask;   // [L*K] positions whose dimension is not a power of two (mode 2)
  int32_t odd_single;        // mode 2: every such dimension has the SAME odd part m; ctab[t] = the product of
                             // t factors m as the reference's loop rounds it
  const uint64_t* sparse;    // [L*K] or NULL
  double n_projs;            // (cost_type)n_projs
  int32_t f32;               // cost_type float32
  int32_t disable_shared;
  const uint64_t* dimclass;  // [n_dimclass][L*K] mode 3: legs whose dimension is 2^(j+1), j = 0..n_dimclass-1
  int32_t n_dimclass;
};

// K words of a leg mask held by one lane.
template <int K>
struct Mask {
  uint64_t w[K];
};
template <int K>
__device__ __forceinline__ Mask<K> mzero() {
  Mask<K> r;
#pragma unroll
  for (int k = 0; k < K; ++k) r.w[k] = 0;
  return r;
}
#define TNCO_MASK_OP(NAME, EXPR)                                                  \
  template <int K>                                                                \
  __device__ __forceinline__ Mask<K> NAME(const Mask<K>& a, const Mask<K>& b) {   \
    Mask<K> r;                                                                    \
    _Pragma("unroll") for (int k = 0; k < K; ++k) r.w[k] = (EXPR);                \
    return r;                                                                     \
  }
TNCO_MASK_OP(mand, a.w[k] & b.w[k])
TNCO_MASK_OP(mor, a.w[k] | b.w[k])
TNCO_MASK_OP(mxor, a.w[k] ^ b.w[k])
TNCO_MASK_OP(mandn, a.w[k] & ~b.w[k])
#undef TNCO_MASK_OP
template <int K>
__device__ __forceinline__ uint32_t mpopc(const Mask<K>& a) {
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) c += (uint32_t)__popcll(a.w[k]);
  return c;
}
template <int K>
__device__ __forceinline__ bool mnonzero(const Mask<K>& a) {
  uint64_t x = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) x |= a.w[k];
  return x != 0;
}
template <int K>
__device__ __forceinline__ bool mdiffer(const Mask<K>& a, const Mask<K>& b) {
  return mnonzero<K>(mxor<K>(a, b));
}
template <int K>
__device__ __forceinline__ Mask<K> msel(bool c, const Mask<K>& a, const Mask<K>& b) {
  Mask<K> r;
#pragma unroll
  for (int k = 0; k < K; ++k) r.w[k] = c ? a.w[k] : b.w[k];
  return r;
}

// Per-replica view of the node blocks.  UNI: layouts whose legs are WS == BS bytes apart only (the infinite-memory sweep
// kernel: one multiply per node, as before the split layout existed); else whatever Params says.
template <int LOG2L, int K, bool HYPER, bool UNI = false>
struct View {
  static constexpr int L = 1 << LOG2L;
  static constexpr int LK = L * K;
  uint8_t* blk;
  int32_t* lpar;
  const uint64_t* leafmask;
  uint64_t* hyp;  // hyper legs of internal node p at hyp + (p - n) * W when they are not part of the blocks (build_kernel), else nullptr
  int n, BS, W, lig, WS, WOFF;

  __device__ __forceinline__ void init(const Params& P, uint8_t* blk_, int32_t* lpar_, int lig_, uint64_t* hyp_ = nullptr) {
    blk = blk_; lpar = lpar_; leafmask = P.leafmask; n = P.n; BS = P.BS; W = P.W; lig = lig_;
    WS = P.WS; WOFF = P.WOFF; hyp = hyp_;
  }
  __device__ __forceinline__ NodeRec* hdr(int p) const {
    return reinterpret_cast<NodeRec*>(blk + (int64_t)(p - n) * BS);
  }
  // word index held in slot k of this lane: k-major, so that the 4/8/16 lanes of a group access
  // L consecutive words (one contiguous 8L-byte piece of the line) per instruction
  __device__ __forceinline__ int widx(int k) const { return k * L + lig; }
  __device__ __forceinline__ uint64_t* words(int p) const {
    if constexpr (UNI) return reinterpret_cast<uint64_t*>(blk + (int64_t)(p - n) * BS + 32);
    return reinterpret_cast<uint64_t*>(blk + WOFF + (int64_t)(p - n) * WS);
  }
  // (networks with hyper-indices: a second mask per node in the caller's temporary array -- build_kernel)
  __device__ __forceinline__ uint64_t* hwords(int p) const { return hyp + (int64_t)(p - n) * W; }
  // Legs of node x (this lane's words).  The ADDRESS is selected (leaf table / node block), not the
  // value: one load per word, one definition -- two loads in the arms of a branch merge where the
  // arms meet, and the merge waits ("a loaded value has ONE definition"); several masks
  // requested back to back then really are in flight together.
  __device__ __forceinline__ Mask<K> mask(int x) const {
    const bool leaf = x < n;
    const uint64_t* s = leaf ? leafmask + (int64_t)x * LK : words(x);
    Mask<K> r;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = widx(k);
      r.w[k] = s[(leaf || i < W) ? i : W - 1];  // (a node block holds W words, a leaf row LK)
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (!leaf && widx(k) >= W) r.w[k] = 0ull;
    return r;
  }
  // Same value as mask(x), as ONE load sequence for leaves and internal nodes (the address is
  // selected, not the result): the destination registers have a single definition, so nothing has
  // to read them -- and wait for them -- before the landing fence of the sweep kernel.
  __device__ __forceinline__ Mask<K> mask_staged(int x) const {
    const bool leaf = x < n;
    const uint64_t* s = leaf ? leafmask + (int64_t)x * LK : words(x);
    Mask<K> r;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      r.w[k] = 0ull;
      if (leaf || widx(k) < W) r.w[k] = s[widx(k)];
    }
    return r;
  }
  // mask_staged into registers that live across iterations: only the loads -- a word that is not loaded (beyond W of an
  // internal node) keeps what it held, which is the zero it started from or the zero of a leaf row's padding
  __device__ __forceinline__ void mask_stage_into(Mask<K>& r, int x) const {
    const bool leaf = x < n;
    const uint64_t* s = leaf ? leafmask + (int64_t)x * LK : words(x);
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (leaf || widx(k) < W) r.w[k] = s[widx(k)];
  }
  __device__ __forceinline__ void set_mask(int p, const Mask<K>& v) const {
    uint64_t* s = words(p);
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (widx(k) < W) s[widx(k)] = v.w[k];
  }
  // words 0..L-1 only (slot 0 of every lane): with the 32-byte header, the first 64 bytes of a block
  __device__ __forceinline__ void set_mask_first(int p, const Mask<K>& v) const {
    if (widx(0) < W) words(p)[widx(0)] = v.w[0];
  }
  __device__ __forceinline__ Mask<K> hyper(int p) const {
    Mask<K> r = mzero<K>();
    if constexpr (HYPER) {
      const uint64_t* s = hwords(p);
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (widx(k) < W) r.w[k] = s[widx(k)];  // (a predicated load, not a select on the loaded value)
    }
    return r;
  }
  __device__ __forceinline__ void set_hyper(int p, const Mask<K>& v) const {
    if constexpr (HYPER) {
      if (hyp == nullptr) return;  // (no temporary array: nothing to keep)
      uint64_t* s = hwords(p);
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (widx(k) < W) s[widx(k)] = v.w[k];
    }
  }
  // (address selected, value fixed up afterwards: see mask())
  __device__ __forceinline__ double partial(int x) const {
    // (a leaf: any 8 bytes that are surely in the L2 -- the leaf table, shared by all replicas)
    const double* a = x < n ? reinterpret_cast<const double*>(leafmask) : &hdr(x)->partial;
    const double p = *a;
    return x < n ? 0.0 : p;
  }
  __device__ __forceinline__ int parent(int x) const {
    const int32_t* a = x < n ? lpar + (int64_t)x * LPS : &hdr(x)->parent;
    return *a;
  }
  __device__ __forceinline__ void set_parent(int x, int p) const {
    if (x < n) lpar[(int64_t)x * LPS] = p; else hdr(x)->parent = p;
  }
  // the same, called by ALL lanes of the group: a leaf's whole record is rewritten (16 bytes per lane)
  __device__ __forceinline__ void set_parent_group(int x, int p) const {
    if (x < n) {
      if constexpr (LPS == 16) {
        if (lig < 4) *reinterpret_cast<int4*>(lpar + (int64_t)x * LPS + 4 * lig) = make_int4(p, p, p, p);
      } else {
        if (lig == 0) lpar[(int64_t)x * LPS] = p;
      }
    } else if (lig == 0) {
      hdr(x)->parent = p;
    }
  }
  __device__ __forceinline__ int left(int x) const { return x < n ? -1 : hdr(x)->left; }
  __device__ __forceinline__ int right(int x) const { return x < n ? -1 : hdr(x)->right; }
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

// Maximum of v over the lanes of the group, result in every lane.
template <int LOG2L>
__device__ __forceinline__ uint32_t gmax(uint32_t v) {
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  if constexpr (LOG2L >= 1) v = mx(v, dpp<TNCO_DPP_XOR1>(v));
  if constexpr (LOG2L >= 2) v = mx(v, dpp<TNCO_DPP_XOR2>(v));
  if constexpr (LOG2L >= 3) v = mx(v, dpp<TNCO_DPP_HALF_MIRROR>(v));
  if constexpr (LOG2L >= 4) v = mx(v, dpp<TNCO_DPP_MIRROR>(v));
  if constexpr (LOG2L >= 5) v = mx(v, (uint32_t)__shfl_xor((int)v, 16));
  if constexpr (LOG2L >= 6) v = mx(v, (uint32_t)__shfl_xor((int)v, 32));
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
// cost model, generic path (cost modes 0/1/2/3, optional sparse legs, f32/f64)
// include/tnco/optimize/infinite_memory/cost_model/simple.hpp:37-55,
// simple_sparse_inds.hpp:37-49.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) const double lds_cdouble;

// Broadcast of lane j's 64-bit value to its group: DPP inside a quad (groups of <= 4 lanes), a
// permute through LDS hardware otherwise.
template <int LOG2L>
__device__ __forceinline__ uint64_t group_bcast(uint64_t x, int j, int gbase) {
  uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  if constexpr (LOG2L <= 2) {
    uint32_t l2, h2;
    switch (j) {
      case 0: l2 = dpp<0x00>(lo); h2 = dpp<0x00>(hi); break;  // quad_perm [0,0,0,0]
      case 1: l2 = dpp<0x55>(lo); h2 = dpp<0x55>(hi); break;
      case 2: l2 = dpp<0xAA>(lo); h2 = dpp<0xAA>(hi); break;
      default: l2 = dpp<0xFF>(lo); h2 = dpp<0xFF>(hi); break;
    }
    // (groups of 2 lanes: two groups share a quad -- lanes j and j + 2)
    if constexpr (LOG2L == 1) {
      const uint32_t l3 = j == 0 ? dpp<0xAA>(lo) : dpp<0xFF>(lo), h3 = j == 0 ? dpp<0xAA>(hi) : dpp<0xFF>(hi);
      const bool upper = (__lane_id() & 2) != 0;
      l2 = upper ? l3 : l2;
      h2 = upper ? h3 : h2;
    }
    lo = l2; hi = h2;
  } else {
    lo = (uint32_t)__shfl((int)lo, gbase + j);
    hi = (uint32_t)__shfl((int)hi, gbase + j);
  }
  return ((uint64_t)hi << 32) | lo;
}

// Where the cost models' tables are read from.  TabsGlobal: the arrays of Params in device memory (constructors,
// validation, the finite-width kernels).  TabsLds: copies in LDS / registers made once per launch by the sweep kernel
// -- a table read from memory inside its loop is a load that is consumed at once, and the `s_waitcnt vmcnt(0)` in front
// of the consumer also waits for every node the iteration has just requested: the general cost models ran 15-40 %
// below the fast path for that, not for their arithmetic (round 5).
struct TabsGlobal {
  const Params& P;
  lds_cdouble* sdims;  // the odd parts staged in LDS by the caller, or nullptr
  __device__ __forceinline__ double ctab(int i) const { return P.ctab[i]; }
  __device__ __forceinline__ double dimsd(int i) const { return sdims ? sdims[i] : P.dimsd[i]; }
  __device__ __forceinline__ uint64_t oddmask(int w) const { return P.oddmask[w]; }
  __device__ __forceinline__ uint64_t dimclass(int idx) const { return P.dimclass[idx]; }
  __device__ __forceinline__ uint64_t sparse(int k, int w) const { return P.sparse[w]; }
};
typedef __attribute__((address_space(3))) const uint64_t lds_cu64;
constexpr int TABS_MAXCLS = 32;  // exponent classes the LDS copy holds: dims up to 2^32 * odd (tnco_hip_create refuses larger ones)
template <int K>
struct TabsLds {
  const Params& P;
  lds_cdouble* tab;   // ctab (cost modes 1 and 2 with one odd part) or the odd parts (mode 2): never both in one mode
  lds_cu64* cls;      // [n_dimclass][L K]
  lds_cu64* odd;      // [L K]
  Mask<K> sp;         // this lane's words of the sparse mask
  __device__ __forceinline__ double ctab(int i) const { return tab[i]; }
  __device__ __forceinline__ double dimsd(int i) const { return tab[i]; }
  __device__ __forceinline__ uint64_t oddmask(int w) const { return odd[w]; }
  __device__ __forceinline__ uint64_t dimclass(int idx) const { return cls[idx]; }
  __device__ __forceinline__ uint64_t sparse(int k, int w) const { return sp.w[k]; }
};

// simple.hpp:51-53: the running product in cost_type over ascending set bits (Bitset::visit order;
// word k*L + j is slot k of lane j).  A dimension 2^a * m (m odd) multiplies the running product by m
// -- rounded -- and by 2^a -- exact: scaling by a power of two commutes with every rounding, and an
// overflow is reached by the scaled chain exactly when the reference's reaches it (the chain only
// grows).  So only the ODD parts are multiplied sequentially (dims that are powers of two drop out
// of the chain) and the powers of two are one masked popcount per exponent class, applied at the end.
template <int LOG2L, int K, class T>
__device__ __forceinline__ double seq_product(const Params& P, const T& tb, const Mask<K>& u, int lig, int gbase) {
  constexpr int L = 1 << LOG2L;
  if (P.odd_single) {
    // one odd part m for all of them (dims like {2, 3, 4, 6, 12}): the chain is m * m * ... rounded
    // after every factor, a function of the NUMBER of factors alone -- a table built on the host
    uint32_t et = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) et += (uint32_t)__popcll(u.w[k] & tb.oddmask(k * L + lig)) << 18;
    for (int j = 0; j < P.n_dimclass; ++j) {
      uint32_t cnt = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) cnt += (uint32_t)__popcll(u.w[k] & tb.dimclass(j * (L * K) + k * L + lig));
      et += (uint32_t)(j + 1) * cnt;
    }
    et = gsum<LOG2L>(et);
    return rnd_cost(ldexp(tb.ctab((int)(et >> 18)), (int)(et & 0x3ffffu)), P.f32);
  }
  double c = 1.0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint64_t mine = u.w[k] & tb.oddmask(k * L + lig);
    for (int j = 0; j < L; ++j) {
      const int w = k * L + j;
      if (tb.oddmask(w) == 0ull) continue;  // (uniform: no such dimension in this word)
      uint64_t x = group_bcast<LOG2L>(mine, j, gbase);
      while (x) {
        const int b = __ffsll((unsigned long long)x) - 1;
        // (one dependent table look-up per leg: from LDS when the caller staged the table there)
        c = rnd_cost(c * tb.dimsd(w * 64 + b), P.f32);
        x &= x - 1;
      }
    }
  }
  uint32_t e = 0;
  for (int j = 0; j < P.n_dimclass; ++j) {
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) cnt += (uint32_t)__popcll(u.w[k] & tb.dimclass(j * (L * K) + k * L + lig));
    e += (uint32_t)(j + 1) * cnt;
  }
  return rnd_cost(ldexp(c, (int)gsum<LOG2L>(e)), P.f32);
}

// Per-index dims that are all powers of two: every partial product of simple.hpp:51-53 is an exact
// power of two (or overflows to inf at the same point whatever the order), so the running product is
// 2^(sum of exponents): one masked popcount per exponent class instead of a loop over the legs.
template <int LOG2L, int K, class T>
__device__ __forceinline__ double pow2_product(const Params& P, const T& tb, const Mask<K>& u, int lig) {
  constexpr int L = 1 << LOG2L;
  uint32_t e = 0;
  for (int j = 0; j < P.n_dimclass; ++j) {
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) c += (uint32_t)__popcll(u.w[k] & tb.dimclass(j * (L * K) + k * L + lig));
    e += (uint32_t)(j + 1) * c;
  }
  return pow2_cost((int)gsum<LOG2L>(e), P.f32);
}

template <int LOG2L, int K, class T>
__device__ __forceinline__ double product_cost(const Params& P, const T& tb, const Mask<K>& u, int lig, int gbase) {
  return P.cost_mode == 3 ? pow2_product<LOG2L, K>(P, tb, u, lig) : seq_product<LOG2L, K>(P, tb, u, lig, gbase);
}

template <class T>
__device__ __forceinline__ double uniform_cost(const Params& P, const T& tb, int pc) {
  return P.cost_mode == 0 ? pow2_cost(P.log2d * pc, P.f32) : tb.ctab(pc);
}

// cost of contracting two tensors whose leg union is `u` (this lane's words).
template <int LOG2L, int K, class T>
__device__ __forceinline__ double generic_cost_t(const Params& P, const T& tb, const Mask<K>& u, int lig, int gbase) {
  if (P.sparse == nullptr) {
    if (P.cost_mode <= 1) return uniform_cost(P, tb, (int)gsum<LOG2L>(mpopc<K>(u)));
    return product_cost<LOG2L, K>(P, tb, u, lig, gbase);
  }
  Mask<K> s;
#pragma unroll
  for (int k = 0; k < K; ++k) s.w[k] = tb.sparse(k, k * (1 << LOG2L) + lig);
  double c1, c2;
  if (P.cost_mode <= 1) {
    const uint32_t v = gsum<LOG2L>(mpopc<K>(mandn<K>(u, s)) | (mpopc<K>(mand<K>(u, s)) << 16));
    c1 = uniform_cost(P, tb, (int)(v & 0xffffu));
    c2 = uniform_cost(P, tb, (int)(v >> 16));
  } else {
    c1 = product_cost<LOG2L, K>(P, tb, mandn<K>(u, s), lig, gbase);
    c2 = product_cost<LOG2L, K>(P, tb, mand<K>(u, s), lig, gbase);
  }
  return rnd_cost(c1 * (c2 < P.n_projs ? c2 : P.n_projs), P.f32);
}
template <int LOG2L, int K>
__device__ __forceinline__ double generic_cost(const Params& P, const Mask<K>& u, int lig, int gbase,
                                               lds_cdouble* sdims = nullptr) {
  return generic_cost_t<LOG2L, K>(P, TabsGlobal{P, sdims}, u, lig, gbase);
}

// ---------------------------------------------------------------------------
// mt19937 seeding: one thread per replica (random.tcc:326-343).
// ---------------------------------------------------------------------------
static __global__ void mt_seed_kernel(uint32_t* mt, ReplicaState* rs, const uint32_t* seeds, int64_t R) {
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
// (include/tnco/optimize/infinite_memory/utils.hpp:31-57,76-92), get_cost (:102-116),
// ContractionTree::is_valid (include/tnco/ctree.hpp:101-152), over the traverse order of
// include/tnco/utils.hpp:34-51.
//
// Links come from `in_links` ([3][N] int32 per replica) when not NULL, else from the live
// blocks of P (src_live) or from src_links (Links[N] per replica).  Results go to out_blocks /
// out_lpar (which may be P's own arrays).  scratch: 4*N int32 per replica.
// ---------------------------------------------------------------------------
struct BuildArgs {
  const int32_t* in_links; int64_t in_links_stride;
  int32_t src_live;            // read links from P.blocks / P.lpar
  const Links* src_links;      // [count][N] or NULL (indexed by q, not r)
  const uint64_t* in_masks; int64_t in_masks_stride;  // optional explicit legs [N][W]
  uint8_t* out_blocks;         // [count][RB]
  int32_t* out_lpar;           // [count][n][LPS]
  int32_t* scratch;            // [count][4N]
  double* out_total;           // [count] partial[root]
  double* out_sum;             // [count] get_cost() sum
  int32_t* out_status;         // [count]
  int64_t r0;                  // first replica handled by this launch
  int64_t count;
  // finite width: legs OR-ed into every contraction (sliced indices; finite_width/utils.hpp:36-47),
  // [LK] words per replica at cost_slices + r * cost_slices_stride; NULL: none
  const uint64_t* cost_slices; int64_t cost_slices_stride;
  // networks with hyper-indices: the second mask per node build_kernel works with (no layout stores hyper legs): [count][n - 1][W]
  uint64_t* hyper_tmp;
};

template <int LOG2L, int K, bool HYPER>
__global__ __launch_bounds__(256) void build_kernel(const Params P, const BuildArgs a) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;
  using M = Mask<K>;
  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gbase = (tid & 63) & ~(L - 1);
  const int64_t q = (int64_t)blockIdx.x * GPB + (tid >> LOG2L);
  if (q >= a.count) return;
  const int64_t r = a.r0 + q;
  const bool lane0 = lig == 0;
  const int n = P.n, N = P.N;

  View<LOG2L, K, HYPER> v;
  v.init(P, a.out_blocks + q * P.RB, a.out_lpar + q * (int64_t)n * LPS, lig,
         (HYPER && a.hyper_tmp) ? a.hyper_tmp + q * (int64_t)(n - 1) * P.W : nullptr);
  View<LOG2L, K, HYPER> live;
  live.init(P, P.blocks + r * P.RB, P.lpar + r * (int64_t)n * LPS, lig);
  int32_t* stack = a.scratch + q * 4 * (int64_t)N;
  int32_t* order = stack + N;
  int status = 0;
  M om;
#pragma unroll
  for (int k = 0; k < K; ++k) om.w[k] = P.outmask[v.widx(k)];

  // -- links --------------------------------------------------------------
  for (int i = lig; i < N; i += L) {
    int l, rr, p;
    if (a.in_links) {
      const int32_t* lk = a.in_links + r * a.in_links_stride;
      l = lk[i]; rr = lk[N + i]; p = lk[2 * (int64_t)N + i];
    } else if (a.src_live) {
      l = live.left(i); rr = live.right(i); p = live.parent(i);
    } else {
      const Links s = a.src_links[q * (int64_t)N + i];
      l = s.left; rr = s.right; p = s.parent;
    }
    if (i < n) {
      v.lpar[(int64_t)i * LPS] = p;
    } else {
      NodeRec o;
      o.left = l; o.right = rr; o.parent = p; o.pad = 0; o.ccost = 0; o.partial = 0;
      *v.hdr(i) = o;
    }
  }
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");

  M csl = mzero<K>();
  if (a.cost_slices) {
#pragma unroll
    for (int k = 0; k < K; ++k) csl.w[k] = a.cost_slices[r * a.cost_slices_stride + v.widx(k)];
  }
  double sum = 0.0;
  // validity (ctree.hpp:101-152), hyper cache, cost caches of one internal node whose children are done
  auto finish_node = [&](int p, int l, int rr, const M& ia, const M& ib, const M& ip) {
    const M uni = mor<K>(ia, ib);
    if (!P.disable_shared && !gany<LOG2L>(mnonzero<K>(mand<K>(ia, ib)))) status = status ? status : 10;
    if (gany<LOG2L>(mnonzero<K>(mor<K>(mandn<K>(mxor<K>(ia, ib), ip), mandn<K>(ip, uni))))) status = status ? status : 11;
    v.set_hyper(p, mand<K>(ip, mand<K>(ia, ib)));
    const double c = generic_cost<LOG2L, K>(P, mor<K>(uni, csl), lig, gbase);
    const double part = rnd_cost(rnd_cost(c + v.partial(l), P.f32) + v.partial(rr), P.f32);  // utils.hpp:54
    sum = rnd_cost(sum + c, P.f32);                                                          // utils.hpp:112
    if (lane0) { v.hdr(p)->ccost = c; v.hdr(p)->partial = part; }
  };
  // Without hyper-indices and with legs derived from the links, a node's legs, checks and costs need nothing but its
  // children's: everything is done when the traverse LEAVES the node -- one pass of dependent round trips over the
  // tree instead of three (65 536 trees of 512 leaves: 16 ms of every create(); each replica's time is its chain of
  // memory round trips, all replicas being resident at once).
  const bool fused = !HYPER && !a.in_masks;

  // -- traverse (utils.hpp:34-51), every lane of the group redundantly -----
  // The stack's first SCAP entries live in LDS (bit 31 of an entry: its children have been pushed -- no `visited`
  // array to consult), deeper ones in the replica's scratch: a step then costs the one header it needs instead of a
  // stack read, a flag read and the acknowledgement of the stores before the next step may look at the stack.
  constexpr int SCAP = (10240 / GPB) < 160 ? (10240 / GPB) : 160;
  __shared__ uint32_t sstack[GPB * SCAP];
  uint32_t* sst = sstack + (tid >> LOG2L) * SCAP;
  auto st_get = [&](int i) -> uint32_t { return i < SCAP ? sst[i] : (uint32_t)stack[i]; };
  auto st_set = [&](int i, uint32_t val) {
    if (i < SCAP) sst[i] = val;
    else if (lane0) stack[i] = (int32_t)val;
  };
  int sp = 1, cnt = 0;
  st_set(0, (uint32_t)(N - 1));
  while (sp > 0) {
    const uint32_t top = st_get(sp - 1);
    const int pos = (int)(top & 0x7FFFFFFFu);
    const int l = v.left(pos);
    bool wrote = sp > SCAP;
    if ((top >> 31) || l < 0) {
      --sp;
      if (lane0) order[cnt] = pos;
      ++cnt;
      if (fused && l >= 0) {
        const int rr = v.right(pos);
        const M ia = v.mask(l), ib = v.mask(rr), ip = mxor<K>(ia, ib);
        v.set_mask(pos, ip);
        finish_node(pos, l, rr, ia, ib, ip);
        wrote = true;  // (the parent's turn reads this node's partial cost, written by lane 0)
      }
    } else {
      const int rr = v.right(pos);
      st_set(sp - 1, top | 0x80000000u);
      st_set(sp, (uint32_t)rr);
      st_set(sp + 1, (uint32_t)l);
      sp += 2;
      wrote = wrote || sp > SCAP;
    }
    if (wrote) __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (`order`, for the passes below)

  // -- legs of internal nodes ---------------------------------------------
  if (a.in_masks) {
    const uint64_t* im = a.in_masks + r * a.in_masks_stride;
    for (int p = 0; p < N; ++p) {
      M x;
#pragma unroll
      for (int k = 0; k < K; ++k) x.w[k] = (v.widx(k) < P.W) ? im[(int64_t)p * P.W + v.widx(k)] : 0ull;
      if (p >= n) v.set_mask(p, x);
      else if (gany<LOG2L>(mdiffer<K>(x, v.mask(p)))) status = 12;  // leaves must be the shared table
    }
  } else if constexpr (!HYPER) {
    // (done while traversing)
  } else {
    // union of leaves below (in the legs slot), legs held outside (in the hyper slot), then
    // legs = (a ^ b) | (a & b & outside)   [tnco/ctree.py:163-189]
    for (int i = 0; i < N; ++i) {
      const int p = order[i];
      const int l = v.left(p);
      if (l < 0) continue;
      v.set_mask(p, mor<K>(v.mask(l), v.mask(v.right(p))));
    }
    for (int i = N - 1; i >= 0; --i) {
      const int p = order[i];
      const int l = v.left(p);
      if (l < 0) continue;
      const int rr = v.right(p);
      const M op = (p == N - 1) ? om : v.hyper(p);
      const M ul = v.mask(l), ur = v.mask(rr);
      if (l >= n) v.set_hyper(l, mor<K>(op, ur));
      if (rr >= n) v.set_hyper(rr, mor<K>(op, ul));
    }
    for (int i = 0; i < N; ++i) {
      const int p = order[i];
      const int l = v.left(p);
      if (l < 0) continue;
      const M ia = v.mask(l), ib = v.mask(v.right(p));
      const M op = (p == N - 1) ? om : v.hyper(p);
      v.set_mask(p, mor<K>(mxor<K>(ia, ib), mand<K>(mand<K>(ia, ib), op)));
    }
  }

  // -- validity, hyper cache, cost caches (the other cases: in traverse order, once the legs are there) --------
  if (!fused) {
    for (int i = 0; i < N; ++i) {
      const int p = order[i];
      const int l = v.left(p);
      if (l < 0) continue;
      const int rr = v.right(p);
      finish_node(p, l, rr, v.mask(l), v.mask(rr), v.mask(p));
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (lane0) {
    a.out_total[q] = v.hdr(N - 1)->partial;
    a.out_sum[q] = sum;
    a.out_status[q] = status;
  }
}

// Compare a rebuilt cache set with the live one (is_valid,
// infinite_memory/optimizer.hpp:223-251, is_logclose include/tnco/utils.hpp:78-87).
template <int LOG2L, int K, bool HYPER>
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
  View<LOG2L, K, HYPER> ref, cur;
  ref.init(P, a.out_blocks + q * P.RB, a.out_lpar + q * (int64_t)n * LPS, lig);
  cur.init(P, P.blocks + r * P.RB, P.lpar + r * (int64_t)n * LPS, lig);
  int bad = a.out_status[q];
  auto logclose = [&](double x, double y) -> bool {
    if (x < 0 || y < 0) return false;
    if (x == 0 || y == 0) return x == y;
    return fabs(log(x) - log(y)) <= atol;
  };
  for (int i = lig; i < N; i += L) {
    if (ref.parent(i) != cur.parent(i)) bad = bad ? bad : 2;
    if (i >= n) {
      const NodeRec x = *ref.hdr(i), y = *cur.hdr(i);
      if (!logclose(x.ccost, y.ccost)) bad = bad ? bad : 31;
      if (!logclose(x.partial, y.partial)) bad = bad ? bad : 32;
      if (x.left != y.left || x.right != y.right) bad = bad ? bad : 2;
      if (cur.parent(y.left) != i || cur.parent(y.right) != i) bad = bad ? bad : 8;
    }
  }
  for (int p = n; p < N; ++p) {
    if (mdiffer<K>(ref.mask(p), cur.mask(p))) bad = bad ? bad : 34;
  }
  const uint32_t anybad = gsum<LOG2L>((uint32_t)(bad != 0));
  if (lig == 0) out_bad[q] = anybad ? (bad ? bad : 99) : 0;
}

// Best tree of every replica = checkpoint + rotations jlog[0, jmin): apply
// Tree::swap_with_nn (include/tnco/tree.hpp:141-192) to a copy.  One thread per replica
// (diagnostic path: validate()).
static __global__ void materialize_min_kernel(const Params P, Links* out, int64_t r0, int64_t count) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= count) return;
  const int64_t r = r0 + q;
  const int N = P.N;
  Links* t = out + q * (int64_t)N;
  const Links* src = P.minlinks + r * (int64_t)N;
  for (int i = 0; i < N; ++i) t[i] = src[i];
  const int32_t* lg = P.jlog + r * (int64_t)P.jcap;
  const uint32_t m = P.rs[r].jmin;
  for (uint32_t k = 0; k < m; ++k) {
    const int D = lg[k];
    const int B = t[D].parent;
    const int A = t[B].parent;
    const int C = (t[A].left == B) ? t[A].right : t[A].left;
    if (t[A].left != C) t[A].right = D; else t[A].left = D;
    if (t[B].left != D) t[B].right = C; else t[B].left = C;
    t[C].parent = B;
    t[D].parent = A;
  }
}

}  // namespace tnco
