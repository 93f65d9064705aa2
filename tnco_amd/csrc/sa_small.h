// sa_small.h -- the sweep kernel for SMALL trees (BASELINE config 2: 64 leaves; the latency regime of README.md): every
// replica's whole tree lives in LDS for the duration of a launch, HBM is touched when the launch starts and ends, by
// the mt19937 state stream, by one beta per sweep and by the rotation log (one 64-byte piece per 16 accepted moves).
//
// Same algorithm, same arithmetic and the same draws as sa_run_kernel (Optimizer::update,
// include/tnco/optimize/infinite_memory/optimizer.hpp:90-221): what differs is where the operands come from.  With the
// node blocks in HBM the kernel is a state machine that overlaps the memory round trips of 192 replicas per CU, one
// landing fence per iteration, and a wavefront runs the code of every state its sixteen replicas are in (2.7 microseconds
// per move and replica in a small batch).  Here an operand is one LDS access (~130 cycles) away and the loop is the plain
// walk, every replica of a wavefront at its own place of its own sweep: per iteration [generator] [one move] [end of a
// sweep + begin of the next].  What bounds it is the instruction
// stream of the wavefront -- 16 replicas, one wavefront per SIMD because LDS holds 64 replicas of 64 leaves per CU: ~3250
// cycles = 1.35 microseconds per move and replica whatever the number of replicas (tools/stage_cycles.py:
// profiles/r05_small_stage_cycles.txt).  Hence the rules of the loop:
//   * every global load is issued at ONE point of the iteration and consumed there one iteration later (vmcnt retires in
//     order: a wait elsewhere waits for the youngest load of sixteen replicas);
//   * the operands of a move (A's record, the legs and partial cost of B's sibling) are read during the previous
//     iteration, the draws of an iteration together at its top: no LDS round trip inside a dependent chain of the move;
//   * values by selection, stores under as few branches as possible (a divergent branch costs all sixteen replicas).
//   * a wavefront with fewer replicas than lane groups (a small batch is spread over the chip's wavefront slots; big trees
//     leave few seats per block) runs the spare groups as SHADOWS of its replicas: a CU with fewer than 64 active lanes
//     runs its wavefronts slower and one after the other (tools/few_lanes.hip), shadows cost no instruction.
// Measured (profiles/r05_small_tree_ab.txt): x2.4 ... x3.3 the HBM kernel up to 16 384 replicas of 64 leaves (config 2:
// 3.0e9 move-evals/s against 1.1e9), 1.05e10 from there on (the HBM kernel: 9.0-9.7e9 at 65 536 replicas); above 64 leaves a
// CU holds 32 replicas and the kernel wins while two rounds of blocks hold them.  The host picks accordingly
// (tnco_hip_create).  History: round 2 had a first version as an opt-in and measured only the 65 536-replica case (6.8e9
// against 7.4e9), round 3 removed it, round 5 re-measured it where it can win and rebuilt the loop around the rules above.
//
// Two kernels around one loop (small_sweeps<Store>), differing in where a block keeps its trees (the stores below):
//   sa_small_kernel<NI, 64>: up to 128 leaves and 2 mask words -- byte links, one 32-byte record per node
//     [ left right parent - | cost exponent | partial cost f64 | legs: 2 words ], the leaf legs a table of the block; static
//     LDS, 16 replicas per block, 64 (<= 64 leaves) or 32 replicas per CU.  Picked for every batch up to 64 leaves, beyond
//     while two rounds of blocks hold the batch.
//   sa_lds_kernel<K, HYPER>: any other tree of up to 16 mask words, with or without hyper-indices, whose whole BATCH fits the
//     CUs' LDS at once (the latency regime of the larger networks: a 512-leaf tree of 12 words is 58 KiB, two per CU, 512
//     replicas) -- 16-bit links, arrays of [links | exponent], partial costs, legs; the leaf legs as a table while that is
//     small, else as lists of at most 32 index positions, expanded where needed; dynamic LDS carved by the host (LdsPlan).
//     x1.6 ... x2.8 the HBM kernel per replica (profiles/r05_small_tree_ab.txt).
// Conditions of both: the fast cost path (uniform power-of-two dims, float64 cost, no sparse legs), four lanes per replica,
// a log that starts at the checkpoint (no min_links given).  A contraction cost is 2^e exactly -- or +inf, e >= 1024 -- so
// the exponent is what the records keep.
#pragma once
#include "sa_sweep.h"

namespace tnco {

// Stage timing (diagnostic builds, tools/stage_cycles.py): -DTNCO_PROFILE=1 the sections of an iteration
// [generator | move | sweep end | sweep begin], -DTNCO_PROFILE=2 the parts of a move [operands | costs | acceptance | update].
#if defined(TNCO_PROFILE) && TNCO_PROFILE == 2
#define SMALL_PROF_C(i) do { if ((i) == 0) { pt_[0] = pt_[1] = pt_[2] = pt_[3] = pt_[4] = __builtin_amdgcn_s_memtime(); } } while (0)
#define SMALL_PROF_M(i) pt_[i] = __builtin_amdgcn_s_memtime()
#elif defined(TNCO_PROFILE)
#define SMALL_PROF_C(i) pt_[i] = __builtin_amdgcn_s_memtime()
#define SMALL_PROF_M(i)
#else
#define SMALL_PROF_C(i)
#define SMALL_PROF_M(i)
#endif

struct __attribute__((aligned(32))) SmallRec {
  uint32_t links;   // left | right << 8 | parent << 16   (0xFF: null)
  uint32_t cexp;    // exponent of the contraction cost
  double partial;
  uint64_t legs[2];
};
static_assert(sizeof(SmallRec) == 32, "SmallRec");
constexpr int SMALL_MAX_INTERNAL = 127;  // (links are bytes, 0xFF = null: at most 254 nodes)
constexpr uint32_t SMALL_RNG_LOW = 12;  // draws left in a replica's ring below which the wavefront refills (see the loop)
constexpr int SMALL_TPB = 64;            // one wavefront = 16 replicas per block: few replicas still spread over the CUs
// LDS of a block of sa_small_kernel<NI, SMALL_TPB> (the __shared__ arrays below) and the replicas one CU (160 KiB) holds
constexpr int small_lds_bytes(int NI) {
  return (SMALL_TPB / 4) * (NI * 32 + (NI + 1) + 64 * 4 + 16 * 4) + (NI + 1) * 16;
}
constexpr int small_replicas_per_cu(int ni) { return (160 * 1024 / small_lds_bytes(ni <= 63 ? 63 : 127)) * (SMALL_TPB / 4); }
static_assert(small_replicas_per_cu(63) == 64 && small_replicas_per_cu(127) == 32, "LDS budget of the small-tree kernel");

// accept_move (sa_sweep.h: same rule, same filter, same margin, same decision) without its early returns: one branch, taken
// when some replica of the wavefront needs the double-precision pow.
__device__ __forceinline__ bool small_accept(int kind, double beta, double delta, double total, double u) {
  const bool yes = kind == 0 || delta <= 0;        // base.hpp; greedy.hpp / mh.hpp: p = 1
  const bool zero = kind == 1 || total == 0;       // greedy.hpp: p = 0; mh.hpp:55-57
  const double x = 1.0 + delta / total;
  const float uf = (float)u, xf = (float)x, bf = (float)beta;
  const float lu = __log2f(uf), lx = __log2f(xf);
  const float lp = -bf * lx;
  const float margin = (fabsf(lp) + fabsf(lu)) * 2e-6f + fabsf(bf) * 3e-7f + 1e-5f;
  const bool ok = uf > 1e-30f && xf < 1e30f && fabsf(lp) < 1e30f && beta >= 0.0;
  const bool sure_yes = ok && lu < lp - margin, sure_no = ok && lu > lp + margin;
  bool acc = yes || (zero ? u <= 0.0 : sure_yes);
  if (!yes && !zero && !sure_yes && !sure_no) acc = accept_exact(x, beta, u, 0);
  return acc;
}

// x % n through the reciprocal (n < 2^16, x < 2^32: the quotient estimate is off by one at most)
__device__ __forceinline__ uint32_t small_mod(uint32_t x, uint32_t n, double inv_n) {
  const uint32_t q = (uint32_t)((double)x * inv_n);
  uint32_t r = x - q * n;
  r = (int32_t)r < 0 ? r + n : r;
  return r >= n ? r - n : r;
}

__device__ __forceinline__ uint32_t small_exp_of(double c) {
  return (uint32_t)((__double2hiint(c) >> 20) & 0x7ff) - 1023u;  // c = 2^e exactly (or +inf -> 1024)
}

// ---------------------------------------------------------------------------------------------------------------
// Where a block keeps its replicas' trees.  Two layouts, one loop (small_sweeps below):
//   SmallStore<NI>: up to 128 leaves and 2 mask words -- links are bytes, a node is one 32-byte record, the leaf legs a
//     table of the block; static LDS, 16 replicas per block.  <63>: 38.5 KiB, four blocks per CU; <127>: 72.5 KiB, two.
//   WideStore<K>: any tree whose replicas fit -- 16-bit links, up to 16 mask words (4 lanes x K words), separate arrays
//     for [links | cost exponent], partial costs and legs, the leaf legs as a table while that is small (<= 16 KiB), else
//     as lists of index positions (at most 32 per leaf; expanded where a leaf's legs are needed); dynamic LDS carved by
//     the host (LdsPlan), 1 ... 16 replicas per block: a 512-leaf tree of 12 words is 58 KiB, two per CU.
// ---------------------------------------------------------------------------------------------------------------
typedef TNCO_LDS volatile uint64_t lvu64;
typedef TNCO_LDS volatile uint16_t lvu16;
typedef TNCO_LDS volatile uint8_t lvu8;
typedef TNCO_LDS volatile double lvf64;

template <int NI_>
struct SmallStore {
  static constexpr int K = 1, NI = NI_, GPB = SMALL_TPB / 4, NULLID = 0xFF;
  TNCO_LDS volatile SmallRec* rec;
  lvu8* lpar;
  lvu64* leaf;
  int n, lig;
  // links | cost exponent << 32
  static __device__ __forceinline__ uint64_t pack(int l, int r, int p, uint32_t e) {
    return (uint64_t)((uint32_t)l | ((uint32_t)r << 8) | ((uint32_t)(p & 0xFF) << 16)) | ((uint64_t)e << 32);
  }
  static __device__ __forceinline__ int left_of(uint64_t h) { return (int)((uint32_t)h & 0xFF); }
  static __device__ __forceinline__ int right_of(uint64_t h) { return (int)(((uint32_t)h >> 8) & 0xFF); }
  static __device__ __forceinline__ int parent_of(uint64_t h) {
    const int p = (int)(((uint32_t)h >> 16) & 0xFF);
    return p == NULLID ? -1 : p;
  }
  static __device__ __forceinline__ uint32_t exp_of(uint64_t h) { return (uint32_t)(h >> 32); }
  __device__ __forceinline__ uint64_t head(int i) const { return *(lvu64*)&rec[i]; }
  __device__ __forceinline__ void set_head(int i, uint64_t h) const { *(lvu64*)&rec[i] = h; }
  __device__ __forceinline__ double partial(int i) const { return rec[i].partial; }
  __device__ __forceinline__ void set_partial(int i, double v) const { rec[i].partial = v; }
  // (no branch around the read: the address is selected -- a divergent branch costs the sixteen replicas of the wavefront
  // more than a read does; lanes 2, 3 hold nothing)
  __device__ __forceinline__ Mask<1> legs(int x) const {
    Mask<1> m;
    lvu64* a = x < n ? leaf + 2 * x + (lig & 1) : (lvu64*)&rec[x - n].legs[lig & 1];
    const uint64_t v = *a;
    m.w[0] = lig < 2 ? v : 0ull;
    return m;
  }
  __device__ __forceinline__ void set_legs(int i, const Mask<1>& m) const {
    if (lig < 2) rec[i].legs[lig] = m.w[0];
  }
  __device__ __forceinline__ int leaf_parent(int x) const { return lpar[x]; }
  __device__ __forceinline__ void set_parent(int x, int p) const {  // (lane 0 of the group calls)
    lvu8* a = x < n ? lpar + x : (lvu8*)&rec[x - n].links + 2;
    *a = (uint8_t)p;
  }
};

// The host's carve of a block's dynamic LDS for WideStore (bytes; tnco_hip_create).
struct LdsPlan {
  int seats;        // replicas per block (1 ... 16): the lane groups beyond them shadow them (small_sweeps)
  int leaf_stride;  // 64-bit words per leaf in the index lists: four 16-bit index positions each (1 ... 8); 0: a table of legs
  int leaf_words;   // 64-bit words of the leaf lists / table
  int seat0, seat_stride;                       // first seat, bytes per seat
  int o_part, o_legs, o_lpar, o_ring, o_jb;     // inside a seat ([links | exponent] first)
  int total;        // bytes of the block
  int blocks_per_cu;
};

template <int K_>
struct WideStore {
  static constexpr int K = K_, NULLID = 0xFFFF;
  lvu64* hdr;       // left | right << 16 | parent << 32 | cost exponent << 48
  lvf64* part;
  lvu64* leg;       // [node][4 K]: word k * 4 + lane of node i at i * 4 K + k * 4 + lane (the lanes' k-major layout)
  lvu16* lpar;
  lvu64* leaf;      // [leaf][stride]: 4 x stride 16-bit index positions, 0xFFFF: none; stride 0: [leaf][4 K] the legs themselves
  int n, lig, stride;
  static __device__ __forceinline__ uint64_t pack(int l, int r, int p, uint32_t e) {
    return (uint64_t)(uint32_t)l | ((uint64_t)(uint32_t)r << 16) | ((uint64_t)((uint32_t)p & 0xFFFFu) << 32) | ((uint64_t)e << 48);
  }
  static __device__ __forceinline__ int left_of(uint64_t h) { return (int)(h & 0xFFFF); }
  static __device__ __forceinline__ int right_of(uint64_t h) { return (int)((h >> 16) & 0xFFFF); }
  static __device__ __forceinline__ int parent_of(uint64_t h) {
    const int p = (int)((h >> 32) & 0xFFFF);
    return p == NULLID ? -1 : p;
  }
  static __device__ __forceinline__ uint32_t exp_of(uint64_t h) { return (uint32_t)(h >> 48); }
  __device__ __forceinline__ uint64_t head(int i) const { return hdr[i]; }
  __device__ __forceinline__ void set_head(int i, uint64_t h) const { hdr[i] = h; }
  __device__ __forceinline__ double partial(int i) const { return part[i]; }
  __device__ __forceinline__ void set_partial(int i, double v) const { part[i] = v; }
  __device__ __forceinline__ Mask<K> legs(int x) const {
    Mask<K> m = mzero<K>();
    if (stride == 0) {  // the leaf legs as a table of rows like the nodes': one read sequence, the address selected
      lvu64* a = x < n ? leaf + x * (4 * K) : leg + (x - n) * (4 * K);
#pragma unroll
      for (int k = 0; k < K; ++k) m.w[k] = a[k * 4 + lig];
    } else if (x < n) {  // a leaf: its index positions, expanded into this lane's words
      for (int q = 0; q < stride; ++q) {
        const uint64_t v = leaf[x * stride + q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t idx = (uint32_t)(v >> (16 * j)) & 0xFFFFu;
          const int w = (int)(idx >> 6);
          const uint64_t bit = idx == 0xFFFFu ? 0ull : 1ull << (idx & 63u);
#pragma unroll
          for (int k = 0; k < K; ++k) m.w[k] |= w == k * 4 + lig ? bit : 0ull;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) m.w[k] = leg[(x - n) * (4 * K) + k * 4 + lig];
    }
    return m;
  }
  __device__ __forceinline__ void set_legs(int i, const Mask<K>& m) const {
#pragma unroll
    for (int k = 0; k < K; ++k) leg[i * (4 * K) + k * 4 + lig] = m.w[k];
  }
  __device__ __forceinline__ int leaf_parent(int x) const { return lpar[x]; }
  __device__ __forceinline__ void set_parent(int x, int p) const {
    lvu16* a = x < n ? lpar + x : (lvu16*)(hdr + (x - n)) + 2;
    *a = (uint16_t)p;
  }
};

// The sweeps of one replica group (4 lanes) on a tree held by store `st` (copied in by the caller, copied out by it).
// HYPER: networks with hyper-indices -- hyper[p] = legs(p) & legs(c0) & legs(c1) (infinite_memory/utils.hpp:82-91) is derived
// from the own legs of B and A, which are carried (and A's parent's read ahead) next to the children's, as in sa_run_kernel.
template <bool HYPER, class Store>
__device__ __forceinline__ void small_sweeps(const Params& P, const Store& st, const int64_t r, const int lig,
                                             lds_vu32* ringbuf, lds_vi32* jbuf, const double* __restrict__ betas,
                                             const int64_t n_steps, const int prob_kind, const bool master) {
  constexpr int LOG2L = 2, K = Store::K, L = 4;
  using M = Mask<K>;
  using R = Rng<LOG2L, 64>;
  const int n = P.n, N = P.N;
  // `master`: a wavefront with fewer replicas than lane groups runs the spare groups as SHADOWS of its replicas -- same
  // replica, same LDS seat, same reads, hence the same values and the same control flow, no store of their own.  A CU
  // with fewer than 64 active lanes runs its wavefronts markedly slower (and one after the other: tools/few_lanes.hip,
  // profiles/r05_few_lanes.txt); shadows cost no instruction and keep every lane active.
  const bool lane0 = lig == 0 && master;
  auto partial_of = [&](int x) -> double {  // (a leaf: 0; the read is unconditional, the address selected)
    const double v = st.partial(x < n ? 0 : x - n);
    return x < n ? 0.0 : v;
  };
  R rng;
  const ReplicaState* rs0 = P.rs + r;
  rng.init(P, r, ringbuf, rs0->mti, rs0->mtw, master ? lig : 64);  // (lane index 64: neither loads nor stores of the generator)
  double min_cost = rs0->min_cost;
  uint32_t jmin = rs0->jmin, jtail = rs0->jtail;
  bool jinvalid = rs0->jinvalid != 0;
  uint32_t n_moves = 0, n_acc = 0, n_impr = 0, n_full = 0, n_rpick = 0;
  const uint32_t jcap = (uint32_t)P.jcap;
  int32_t* jlog = P.jlog + r * (int64_t)P.jcap;
  auto jb = [&](int i) -> lds_vi32& { return jbuf[i]; };
  if (master && (jtail & 15u) != 0u) {
    const int4 t = *reinterpret_cast<const int4*>(jlog + (jtail & ~15u) + 4 * lig);
    jb(4 * lig + 0) = t.x; jb(4 * lig + 1) = t.y; jb(4 * lig + 2) = t.z; jb(4 * lig + 3) = t.w;
  }
  const int log2d = P.log2d;
  const bool disable_shared = P.disable_shared != 0;
  const int root = N - 1;

  // ---- carried state (as in sa_run_kernel): B, its children's legs / partial costs ----------------
  int step = 0;
  int B = -1, bl = 0, br = 0, A = -1;
  uint32_t eB = 0;
  double partB = 0, beta = 0;
  double total = st.partial(root - n);  // (optimizer.hpp:103: the root's partial cost where a sweep begins)
  const double inv_n = 1.0 / (double)n;
  M m0 = mzero<K>(), m1 = mzero<K>();
  double p0 = 0, p1 = 0;
  // ... and, read one iteration ahead: A's record (links | cost exponent << 32), the legs and the partial cost of B's sibling
  uint64_t hdA = 0;
  M mC = mzero<K>();
  [[maybe_unused]] M hB = mzero<K>(), hA = mzero<K>();  // (HYPER) the own legs of B and of A
  double pC = 0;
  bool active = true;
  // the beta of the NEXT sweep is loaded while this one runs (see the schedule below)
  double beta_next = betas[0], beta_tmp = 0;
  bool bpend = false;
  TNCO_PROF_DECL;  // (diagnostic builds: cycles of [generator | move | sweep end | sweep begin] per iteration)
  // (one loop for the wavefront, left when no replica of it has sweeps to do: a per-replica `break`
  // made the compiler run the sweeps of the sixteen replicas in step)
  while (__ballot(active) != 0ull) {  // (B < 0: between two sweeps)
    SMALL_PROF_C(0);
    // The generator and the schedule.  Every global load of the loop is issued HERE and consumed HERE one iteration
    // later: vmcnt retires in order, so a wait anywhere else waits for the youngest load of sixteen replicas (a beta
    // loaded where the sweep begins and used by its first move cost ~700 cycles per iteration).  A block of 16 draws is
    // requested by every replica that has room for one as soon as ANY replica of the wavefront runs low: the replicas
    // draw at about the same rate, so the wavefront executes this section in two iterations out of ~4.5 instead of
    // in every one (it was 28 % of the loop, profiles/r05_small_stage_cycles.txt).
    if (__ballot(active && (rng.pend || bpend)) != 0ull) {
      if (active && rng.pend) rng.produce();
      if (active && bpend) {
        beta_next = beta_tmp;
        bpend = false;
      }
    }
    if (__ballot(active && rng.avail() < SMALL_RNG_LOW) != 0ull) {
      if (active && rng.room()) rng.request();
    }
    if (active) {
    // (an iteration draws at most 4 numbers and a requested block arrives one iteration later: with SMALL_RNG_LOW = 12
    //  the ring never runs dry -- kept so that the plain next() below is backed whatever the constants)
    while (rng.avail() < 4u) {
      if (!rng.pend) rng.request();
      rng.produce();
    }
    // the draws this iteration may use -- a move's [pick] x1 x2, then the leaf of the next sweep -- read together, ahead
    // of their use: one LDS round trip instead of four, none of them inside a dependent chain
    const uint32_t d0 = rng.ring[rng.cons & (R::RING - 1)], d1 = rng.ring[(rng.cons + 1u) & (R::RING - 1)],
                   d2 = rng.ring[(rng.cons + 2u) & (R::RING - 1)], d3 = rng.ring[(rng.cons + 3u) & (R::RING - 1)];
    uint32_t drawn = 0;  // ... and how many of them the move took
    SMALL_PROF_C(1);
    if (A >= 0) {
      // ---- one move evaluation (optimizer.hpp:117-192); its operands -- A's record, the legs and the partial cost of
      // C, B's sibling -- were read during the previous iteration ---------------------------------------------------
      SMALL_PROF_M(0);
      int al = Store::left_of(hdA), ar = Store::right_of(hdA);
      const int aP = Store::parent_of(hdA);
      const uint32_t eA = Store::exp_of(hdA);
      const bool c_is_right = (al == B);
      const int C = c_is_right ? ar : al;
      // next iteration's operands, first half: the record of A's parent (nothing of it changes in this move)
      const uint64_t hdN = st.head(aP < 0 ? 0 : aP - n);
      // hyper[A] | hyper[B] (optimizer.hpp:145-147)
      const M hy = HYPER ? mand<K>(hB, mor<K>(mand<K>(hA, mC), mand<K>(m0, m1))) : mzero<K>();
      [[maybe_unused]] M hN = mzero<K>();
      if constexpr (HYPER) hN = st.legs(aP < 0 ? n : aP);  // (the own legs of A's parent: the next move's hA)
      uint32_t w0 = mpopc<K>(mor<K>(mor<K>(mxor<K>(m0, mC), hy), m1)) | (mpopc<K>(mor<K>(m0, mC)) << 13) |
                    ((mnonzero<K>(mand<K>(m0, mC)) ? 1u : 0u) << 26);
      uint32_t w1 = mpopc<K>(mor<K>(mor<K>(mxor<K>(m1, mC), hy), m0)) | (mpopc<K>(mor<K>(m1, mC)) << 13) |
                    ((mnonzero<K>(mand<K>(m1, mC)) ? 1u : 0u) << 26);
      SMALL_PROF_M(1);
      w0 = gsum<LOG2L>(w0);
      w1 = gsum<LOG2L>(w1);
      // ... second half: A's sibling under that parent (another subtree: untouched by this move)
      const int CN = Store::left_of(hdN) == A ? Store::right_of(hdN) : Store::left_of(hdN);
      const M mCN = st.legs(CN);
      const double pCN = partial_of(CN);
      const bool inter0 = (w0 >> 26) != 0, inter1 = (w1 >> 26) != 0;
      // get_ctree_nn, optimize/optimizer.hpp:128-144
      const bool rpick = disable_shared || (inter0 && inter1);
      const bool pick0 = rpick ? (d0 & 1u) != 0 : inter0;
      n_rpick += rpick ? 1u : 0u;
      const M mD = msel<K>(pick0, m0, m1), mE = msel<K>(pick0, m1, m0);
      const M newB = mor<K>(mxor<K>(mD, mC), hy);  // (optimizer.hpp:147)
      const double pD = pick0 ? p0 : p1, pE = pick0 ? p1 : p0;
      const int E = pick0 ? br : bl;
      const uint32_t enA = (uint32_t)log2d * ((pick0 ? w0 : w1) & 0x1fffu);
      const uint32_t enB = (uint32_t)log2d * (((pick0 ? w0 : w1) >> 13) & 0x1fffu);
      const double nA = pow2_cost((int)enA, 0), nB = pow2_cost((int)enB, 0);
      const double ccB0 = pow2_cost((int)eB, 0), ccA0 = pow2_cost((int)eA, 0);
      const double delta = (nB - ccB0) + (nA - ccA0);  // :158
      ++n_moves;
      SMALL_PROF_M(2);
      const uint32_t x1 = rpick ? d1 : d0, x2 = rpick ? d2 : d1;  // :162 generate_canonical<double, 53>
      drawn = rpick ? 3u : 2u;
      double u = ((double)x1 + (double)x2 * 4294967296.0) * 5.421010862427522170037e-20;
      if (u >= 1.0) u = 0.99999999999999988897769753748;
      const bool acc = small_accept(prob_kind, beta, delta, total, u);
      SMALL_PROF_M(3);
      // Tree::swap_with_nn(E), include/tnco/tree.hpp:176-184 -- the values by selection, the stores under one branch
      n_acc += acc ? 1u : 0u;
      br = (acc && pick0) ? C : br;
      bl = (acc && !pick0) ? C : bl;
      ar = (acc && c_is_right) ? E : ar;
      al = (acc && !c_is_right) ? E : al;
      const uint32_t eBn = acc ? enB : eB, eAn = acc ? enA : eA;
      const double ccB = acc ? nB : ccB0, ccA = acc ? nA : ccA0;
      total = acc ? total + delta : total;  // :177
      const double pEcur = acc ? pC : pE, pCcur = acc ? pE : pC;
      const M mBnow = msel<K>(acc, newB, HYPER ? hB : mxor<K>(m0, m1)), mX = msel<K>(acc, mE, mC);
      // :185-188
      partB = (pD + pEcur) + ccB;
      const double partA = (partB + pCcur) + ccA;
      // rotation log (best tree = checkpoint + log prefix): full -> the next improvement re-bases the checkpoint
      const bool logged = acc && !jinvalid && jtail != jcap;
      jinvalid = jinvalid || (acc && jtail == jcap);
      if (lane0) {
        if (acc) {
          st.set_parent(C, B);
          st.set_parent(E, A);
          if (logged) jb((int)(jtail & 15u)) = E;
        }
        // (A's own record is written when A is B -- in the next move, or where the sweep ends)
        st.set_head(B - n, Store::pack(bl, br, A, eBn));
        st.set_partial(B - n, partB);
      }
      if (acc && master) st.set_legs(B - n, newB);
      jtail += logged ? 1u : 0u;
      if (master && logged && (jtail & 15u) == 0u)  // a 64-byte piece of the log is complete
        *reinterpret_cast<int4*>(jlog + (jtail - 16u) + 4 * lig) =
            make_int4(jb(4 * lig + 0), jb(4 * lig + 1), jb(4 * lig + 2), jb(4 * lig + 3));
      // :191  B <- A
      if (c_is_right) { m0 = mBnow; m1 = mX; p0 = partB; p1 = pCcur; }
      else            { m1 = mBnow; m0 = mX; p1 = partB; p0 = pCcur; }
      B = A; bl = al; br = ar; eB = eAn; partB = partA;
      A = aP;
      hdA = hdN; mC = mCN; pC = pCN;
      if constexpr (HYPER) { hB = hA; hA = hN; }
      SMALL_PROF_M(4);
    }
    SMALL_PROF_C(2);
    SMALL_PROF_C(3);
    if (A < 0) {
      // ---- between two sweeps: the end of one (B is the root, optimizer.hpp:194-201) and the begin of the next
      // (:103-112) in one section -- the loads of the second overlap the stores of the first ------------------------
      const uint32_t x = drawn == 0u ? d0 : (drawn == 2u ? d2 : d3);
      if (B >= 0) {
        if (lane0) {
          st.set_head(B - n, Store::pack(bl, br, Store::NULLID, eB));
          st.set_partial(B - n, partB);
        }
        if (partB < min_cost) {
          min_cost = partB;
          ++n_impr;
          if (jinvalid) {  // the log overflowed: re-base the checkpoint on the current tree
            Links* __restrict__ ml = P.minlinks + r * (int64_t)N;
            for (int i = master ? lig : N; i < N; i += L) {
              Links o;
              if (i < n) {
                o.left = -1; o.right = -1; o.parent = st.leaf_parent(i);
              } else {
                const uint64_t hd = st.head(i - n);
                o.left = Store::left_of(hd); o.right = Store::right_of(hd);
                o.parent = Store::parent_of(hd);
              }
              o.pad = 0;
              ml[i] = o;
            }
            jtail = 0;
            jinvalid = false;
            ++n_full;
          }
          jmin = jtail;
        }
        total = partB;  // (the next sweep's total cost: the root's partial cost)
        B = -1;
        if (++step >= (int)n_steps) active = false;
      }
      SMALL_PROF_C(3);
      if (active) {
        // a random leaf, its parent is B; the total cost is the root's partial cost
        ++drawn;
        B = st.leaf_parent((int)small_mod(x, (uint32_t)n, inv_n));
        const uint64_t hd = st.head(B - n);
        partB = st.partial(B - n);
        bl = Store::left_of(hd); br = Store::right_of(hd);
        A = Store::parent_of(hd);
        eB = Store::exp_of(hd);
        beta = beta_next;
        beta_tmp = betas[step + 1 < (int)n_steps ? step + 1 : step];
        bpend = true;
        hdA = st.head(A < 0 ? 0 : A - n);  // the first move's operands
        m0 = st.legs(bl); m1 = st.legs(br);
        if constexpr (HYPER) { hB = st.legs(B); hA = st.legs(A < 0 ? n : A); }
        p0 = partial_of(bl); p1 = partial_of(br);
        const int C = Store::left_of(hdA) == B ? Store::right_of(hdA) : Store::left_of(hdA);
        mC = st.legs(C);
        pC = partial_of(C);
      }
    }
    rng.cons += drawn;
    SMALL_PROF_C(4);
    TNCO_PROF_ACC;
    }
  }

  if (master && (jtail & 15u) != 0u)
    *reinterpret_cast<int4*>(jlog + (jtail & ~15u) + 4 * lig) =
        make_int4(jb(4 * lig + 0), jb(4 * lig + 1), jb(4 * lig + 2), jb(4 * lig + 3));
  int mti, mtw;
  rng.finish(mti, mtw);
  if (lane0) {
    ReplicaState* rs = P.rs + r;
    rs->jmin = jmin; rs->jtail = jtail;
    rs->jinvalid = jinvalid ? 1 : 0;
    rs->n_fullcopy += n_full;
    rs->min_cost = min_cost;
    rs->n_moves += n_moves;
    rs->n_accepted += n_acc;
    rs->n_improved += n_impr;
    rs->n_randpick += n_rpick;
    rs->mti = mti;
    rs->mtw = mtw;
    TNCO_PROF_OUT(rs);
  }
}

// The kernels: copy the replicas' trees HBM -> LDS, sweep, copy back.
// <NI, TPB>: internal nodes the static arrays are sized for (n - 1 <= NI); TPB / 4 = 16 replicas per block.
template <int NI, int TPB>
__global__ __launch_bounds__(TPB) void sa_small_kernel(const Params P, const double* __restrict__ betas,
                                                       const int64_t n_steps, const int prob_kind, const int seats) {
  constexpr int L = 4, GPB = TPB / 4;
  using R = Rng<2, 64>;
  __shared__ SmallRec recbuf[GPB * NI];
  __shared__ uint8_t lparbuf[GPB * (NI + 1)];
  __shared__ uint64_t leafbuf[(NI + 1) * 2];
  __shared__ uint32_t rngbuf[GPB * R::RING];
  __shared__ int32_t jbuf[GPB * 16];

  const int tid = threadIdx.x;
  const int lig = tid & 3;
  // (`seats` replicas per block, 1 ... 16 -- a small batch is spread over the chip's wavefront slots --; lane groups beyond
  // them shadow the replicas: small_sweeps)
  const int gib = (tid >> 2) % seats;
  const bool master = (tid >> 2) < seats;
  const int64_t r = (int64_t)blockIdx.x * seats + gib;
  const int n = P.n, ni = n - 1;
  // the leaf legs, shared by the replicas of the block (words 0, 1 of every row of the padded table)
  for (int i = tid; i < n * 2; i += TPB) leafbuf[i] = P.leafmask[(int64_t)(i >> 1) * L + (i & 1)];
  __syncthreads();
  if (r >= P.R || n_steps <= 0) return;
  SmallStore<NI> st;
  st.rec = (TNCO_LDS volatile SmallRec*)recbuf + gib * NI;
  st.lpar = (lvu8*)lparbuf + gib * (NI + 1);
  st.leaf = (lvu64*)leafbuf;
  st.n = n; st.lig = lig;
  uint8_t* blk = P.blocks + r * P.RB;
  int32_t* lp = P.lpar + r * (int64_t)n * LPS;
  const int lig0 = master ? lig : (1 << 30);  // (shadows copy nothing)
  for (int i = lig0; i < ni; i += L) {
    const NodeRec* q = reinterpret_cast<const NodeRec*>(blk + (int64_t)i * P.BS);
    const uint64_t* lg = reinterpret_cast<const uint64_t*>(blk + (int64_t)i * P.BS + 32);
    st.set_head(i, SmallStore<NI>::pack(q->left, q->right, q->parent, small_exp_of(q->ccost)));
    st.rec[i].partial = q->partial;
    st.rec[i].legs[0] = lg[0];
    st.rec[i].legs[1] = P.W > 1 ? lg[1] : 0ull;
  }
  for (int i = lig0; i < n; i += L) st.lpar[i] = (uint8_t)lp[(int64_t)i * LPS];
  // (LDS operations of one wavefront are executed in order: no barrier between a group's own writes and reads)
  small_sweeps<false>(P, st, r, lig, (lds_vu32*)rngbuf + gib * R::RING, (lds_vi32*)jbuf + gib * 16, betas, n_steps, prob_kind, master);
  for (int i = lig0; i < ni; i += L) {
    const uint64_t hd = st.head(i);
    NodeRec o;
    o.left = SmallStore<NI>::left_of(hd); o.right = SmallStore<NI>::right_of(hd);
    o.parent = SmallStore<NI>::parent_of(hd);
    o.pad = 0;
    o.ccost = pow2_cost((int)SmallStore<NI>::exp_of(hd), 0);
    o.partial = st.rec[i].partial;
    *reinterpret_cast<NodeRec*>(blk + (int64_t)i * P.BS) = o;
    uint64_t* lg = reinterpret_cast<uint64_t*>(blk + (int64_t)i * P.BS + 32);
    lg[0] = st.rec[i].legs[0];
    if (P.W > 1) lg[1] = st.rec[i].legs[1];
  }
  for (int i = lig0; i < n; i += L) lp[(int64_t)i * LPS] = (int32_t)st.lpar[i];
}

// <K>: 4 lanes x K words per replica; the block's LDS as the host carved it (LdsPlan), leaf_idx = [n][4 x leaf_stride] index
// positions.
template <int K, bool HYPER>
__global__ __launch_bounds__(SMALL_TPB) void sa_lds_kernel(const Params P, const double* __restrict__ betas, const int64_t n_steps,
                                                           const int prob_kind, const LdsPlan pl,
                                                           const uint64_t* __restrict__ leaf_idx) {
  constexpr int L = 4;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  TNCO_LDS uint8_t* lds = (TNCO_LDS uint8_t*)lds_raw;
  const int tid = threadIdx.x;
  const int lig = tid & 3;
  const int gib = (tid >> 2) % pl.seats;   // (lane groups beyond the seats shadow the replicas: small_sweeps)
  const bool master = (tid >> 2) < pl.seats;
  const int64_t r = (int64_t)blockIdx.x * pl.seats + gib;
  const int n = P.n, ni = n - 1, W = P.W;
  for (int i = tid; i < pl.leaf_words; i += SMALL_TPB) ((lvu64*)lds)[i] = leaf_idx[i];
  __syncthreads();
  if (r >= P.R || n_steps <= 0) return;
  TNCO_LDS uint8_t* seat = lds + pl.seat0 + gib * pl.seat_stride;
  WideStore<K> st;
  st.hdr = (lvu64*)seat;
  st.part = (lvf64*)(seat + pl.o_part);
  st.leg = (lvu64*)(seat + pl.o_legs);
  st.lpar = (lvu16*)(seat + pl.o_lpar);
  st.leaf = (lvu64*)lds;
  st.n = n; st.lig = lig; st.stride = pl.leaf_stride;
  uint8_t* blk = P.blocks + r * P.RB;
  int32_t* lp = P.lpar + r * (int64_t)n * LPS;
  const int lig0 = master ? lig : (1 << 30);  // (shadows copy nothing)
  for (int i = lig0; i < ni; i += L) {
    const NodeRec* q = reinterpret_cast<const NodeRec*>(blk + (int64_t)i * P.BS);
    const uint64_t* lg = reinterpret_cast<const uint64_t*>(blk + (int64_t)i * P.BS + 32);
    st.set_head(i, WideStore<K>::pack(q->left, q->right, q->parent, small_exp_of(q->ccost)));
    st.part[i] = q->partial;
    for (int w = 0; w < 4 * K; ++w) st.leg[i * (4 * K) + w] = w < W ? lg[w] : 0ull;
  }
  for (int i = lig0; i < n; i += L) st.lpar[i] = (uint16_t)lp[(int64_t)i * LPS];
  small_sweeps<HYPER>(P, st, r, lig, (lds_vu32*)(seat + pl.o_ring), (lds_vi32*)(seat + pl.o_jb), betas, n_steps, prob_kind, master);
  for (int i = lig0; i < ni; i += L) {
    const uint64_t hd = st.head(i);
    NodeRec o;
    o.left = WideStore<K>::left_of(hd); o.right = WideStore<K>::right_of(hd);
    o.parent = WideStore<K>::parent_of(hd);
    o.pad = 0;
    o.ccost = pow2_cost((int)WideStore<K>::exp_of(hd), 0);
    o.partial = st.part[i];
    *reinterpret_cast<NodeRec*>(blk + (int64_t)i * P.BS) = o;
    uint64_t* lg = reinterpret_cast<uint64_t*>(blk + (int64_t)i * P.BS + 32);
    for (int w = 0; w < W; ++w) lg[w] = st.leg[i * (4 * K) + w];
  }
  for (int i = lig0; i < n; i += L) lp[(int64_t)i * LPS] = (int32_t)st.lpar[i];
}

}  // namespace tnco
