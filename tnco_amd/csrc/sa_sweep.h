// sa_sweep.h -- the sweep kernel: n_steps calls of Optimizer::update
// (include/tnco/optimize/infinite_memory/optimizer.hpp:90-221) per replica, for all replicas.
//
// Execution structure (gfx950).  A wavefront carries 64/L replicas (16 at L = 4), each at its own
// position of its own leaf->root walk.  The walk is a dependent pointer chase, and vmcnt retires
// loads and stores IN ORDER, so a load consumed after a store also waits for that store.  The
// kernel is therefore a per-replica STATE MACHINE in which no load is consumed in the iteration
// that issues it:
//
//     every iteration:   [per state: decide WHICH node header / node legs / 8 bytes are needed next]
//                        [one load sequence for all replicas, into staging registers]
//                        [MOVE replicas: one move evaluation on what landed in EARLIER iterations]
//                        [landing fence: wait for this iteration's loads]   <- the only wait
//                        [this iteration's stores]
//                        [per state: staging registers -> carried state]
//
//   states:  BEGIN   draw the leaf, request its parent B and its own legs -- the leaf is the child the walk comes up
//                    through                                               (optimizer.hpp:103)
//            GOT_B   request the header of B                              (:107)
//            GOT_HB  request legs + partial cost of the leaf's sibling, header of A, total cost (:112)
//            GOT_B1  request the block of C (sibling of B), header of parent(A), beta
//            MOVE    one move evaluation per iteration (:117-192), requesting the sibling and the
//                    grandparent header of the next level
//            END     B is the root: best-tree update (:198-201), then BEGIN of the next sweep
//
// Two rules keep the compiler from putting waits anywhere but the fence.  (1) Registers are
// tracked per wave, not per lane: a load issued for a replica in one state into a register that the
// code of another state reads later in the same iteration makes that code wait.  So loads only
// ever target the staging registers, which nothing reads before the fence.  (2) A loaded value
// must have ONE definition (no merge of "loaded here" and "loaded there"), or the merge copies --
// and waits -- right after the load: leaf / internal-node legs select the address, not the value.
// The mt19937 stream is produced the same way: a 16-word block (4 words per lane) is requested
// with the other loads and twisted + tempered into an LDS ring at the top of a later iteration.
//
// FW = true: the same state machine runs the moves of the memory-constrained optimizer
// (finite_width/greedy/optimizer.hpp:117-331 without the max_number_new_slices branch, which keeps
// the unstaged fw_move_kernel): the sliced-index mask is carried in registers and OR-ed into both
// contraction costs (:191-193), the cached width of a node travels with its header (the spare
// word), a move whose new B is wider than max_width once sliced draws no uniform and is not
// accepted (:188-201), an accepted one stores B's new width (:216).  The re-slice at the end of a
// sweep is a sequence of kernels of its own (fw_kernels.h: fw_order_kernel | fw_reslice_a_kernel | fw_tree_kernel |
// fw_reslice_b_kernel, or a walk kernel + fw_reslice_kernel): the host launches [moves][re-slice]...
#pragma once
#include "sa_kernels.h"
#include "fw_params.h"

namespace tnco {

#define TNCO_LANDED(x) __asm__ volatile("" : "+v"(x) : : "memory")

// LDS pointers must keep their address space: a generic (flat) pointer to LDS makes every access a
// flat_load / flat_store, which counts in vmcnt and drags an `s_waitcnt vmcnt(0)` behind it.
#define TNCO_LDS __attribute__((address_space(3)))
typedef TNCO_LDS volatile uint32_t lds_vu32;
typedef TNCO_LDS volatile int32_t lds_vi32;

__device__ __forceinline__ uint32_t mt_temper(uint32_t z) {
  z ^= (z >> 11);
  z ^= (z << 7) & 0x9d2c5680u;
  z ^= (z << 15) & 0xefc60000u;
  z ^= (z >> 18);
  return z;
}

// ---------------------------------------------------------------------------
// std::mt19937 (libstdc++ random.tcc:326-471) as a staged producer / consumer.
// Positions are VIRTUAL word indices vp = 624*generation + i, generation 0 being the one the
// state array held at kernel entry.  `cons` outputs have been drawn, `prod` produced into the
// ring (multiple of SB), state words below `tw` are twisted.  Blocks may be produced up to RING
// outputs ahead of consumption; when that crosses into the next generation the overwritten
// words are kept in a shadow so that the exported state is exactly libstdc++'s at `cons`.
// ---------------------------------------------------------------------------
constexpr int MT_SHADOW = 64;  // words of P.mtshadow per replica: the largest ring any kernel uses

template <int LOG2L, int RINGX = 0>
struct Rng {
  static constexpr int L = 1 << LOG2L;
  static constexpr int NL = L < 4 ? L : 4;         // lanes of the group that work on a block
  static constexpr int SB = 4 * NL;                // words per block: 4 per lane (one dwordx4); 624 % 16 == 0
  static constexpr int RING = RINGX ? RINGX : ((2 * SB > 16) ? 2 * SB : 16);
  static_assert(RING <= MT_SHADOW && (RING & (RING - 1)) == 0 && RING >= 2 * SB, "ring size");

  struct __attribute__((packed, aligned(4))) U4 { uint32_t x[4]; };  // 4 words at any word address

  uint32_t* mt_base;       // P.mt      (uniform; the replica's words start at 624 * r)
  uint32_t* sh_base;       // P.mtshadow (uniform; MT_SHADOW * r)
  uint32_t r32;            // replica index
  lds_vu32* ring;          // group's LDS ring
  int lig;
  __device__ __forceinline__ uint32_t* st() const { return mt_base + (uint64_t)r32 * 624u; }
  __device__ __forceinline__ uint32_t* shadow() const { return sh_base + (uint64_t)r32 * (uint32_t)MT_SHADOW; }
  uint32_t cons, prod, tw;
  bool pend, ptw;          // a block's inputs are in flight (for virtual position prod); it needs a twist
  // this lane's inputs: mt[k..k+3], mt[k+4], mt[k+397..k+400] (indices mod 624); written ONLY by
  // the loads of request(), so that the loads land directly in these registers
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 pa;                // one 128-bit register tuple: the dwordx4 load lands in place
  uint32_t pb, pc[4];

  __device__ __forceinline__ uint32_t gen_of_cons() const { return cons == 0 ? 0u : (cons - 1u) / 624u; }

  // A block is requested in one loop iteration and produced in a later one.  One 16-word block costs
  // ~3 memory requests (64 B aligned, 64 B unaligned, 64 B store): the state stream is the only
  // per-draw traffic, so wide blocks matter -- HBM here is bound by the NUMBER of requests.
  __device__ __forceinline__ void request() {
    const uint32_t k0 = prod % 624u;
    ptw = prod >= tw;
    if (lig < NL) {
      const int k = (int)k0 + 4 * lig;
      const uint32_t* s = st();
      pa = *reinterpret_cast<const u32x4*>(s + k);
      if (ptw) {
        pb = s[(k + 4 == 624) ? 0 : k + 4];
        if (k == 224) {  // 224 + 397 = 621: the only group of 4 that straddles the wrap
          pc[0] = s[621]; pc[1] = s[622]; pc[2] = s[623]; pc[3] = s[0];
        } else {
          int km = k + 397;
          if (km >= 624) km -= 624;
          const U4 c = *reinterpret_cast<const U4*>(s + km);
          pc[0] = c.x[0]; pc[1] = c.x[1]; pc[2] = c.x[2]; pc[3] = c.x[3];
        }
      }
    }
    pend = true;
  }
  // inputs landed: twist + temper into the ring and store the twisted words.  The store is issued
  // at the TOP of an iteration, as far from the next wait (the landing fence at its end) as the
  // loads of that iteration are, so it costs no extra wait and no registers across the body.
  __device__ __forceinline__ void produce() {
    if (lig < NL) {
      const uint32_t nx[4] = {pa[1], pa[2], pa[3], pb};
      uint32_t v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = pa[j];
        if (ptw) {
          const uint32_t y = (pa[j] & 0x80000000u) | (nx[j] & 0x7fffffffu);
          v[j] = pc[j] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        ring[(prod + (uint32_t)(4 * lig + j)) & (RING - 1)] = mt_temper(v[j]);
      }
      if (ptw) {
        const int sidx = (int)(prod % 624u) + 4 * lig;
        *reinterpret_cast<uint4*>(st() + sidx) = make_uint4(v[0], v[1], v[2], v[3]);
        // a block of the generation after the one being consumed: keep the old words (sidx < RING
        // <= MT_SHADOW: production runs at most RING words ahead of consumption)
        if ((prod / 624u) > gen_of_cons())
          *reinterpret_cast<uint4*>(shadow() + sidx) = make_uint4(pa[0], pa[1], pa[2], pa[3]);
      }
    }
    if (ptw) tw = prod + SB;
    prod += SB;
    pend = false;
  }
  __device__ __forceinline__ bool room() const { return !pend && (prod - cons) + SB <= (uint32_t)RING; }

  __device__ __forceinline__ void init(const Params& P, int64_t r, lds_vu32* ring_, int mti, int mtw,
                                       int lig_) {
    mt_base = P.mt; sh_base = P.mtshadow; r32 = (uint32_t)r; ring = ring_; lig = lig_;
    pend = false; ptw = false;
    pb = 0;
    pa = (u32x4)(0u);
#pragma unroll
    for (int j = 0; j < 4; ++j) pc[j] = 0;
    if (mti >= 624) { cons = 624; tw = 624; } else { cons = (uint32_t)mti; tw = (uint32_t)mtw; }
    prod = cons & ~(uint32_t)(SB - 1);
    while (room()) {  // synchronous prologue fill
      request();
      produce();
    }
  }
  __device__ __forceinline__ uint32_t next() {
    const uint32_t v = ring[cons & (RING - 1)];
    ++cons;
    return v;
  }
  // on-demand variant for code that is not staged (finite-width kernels): generate the next block
  // right now if the ring is empty
  __device__ __forceinline__ uint32_t next_sync() {
    if (cons == prod) {
      if (!pend) request();
      produce();
    }
    return next();
  }
  // issue the loads of the next block if there is room for one (its latency then overlaps whatever
  // comes before the draw that needs it)
  __device__ __forceinline__ void prefetch() {
    if (room()) request();
  }
  __device__ __forceinline__ uint32_t avail() const { return prod - cons; }
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
  // leave (state array, mti, mtw) exactly as libstdc++ would hold them after `cons` draws
  __device__ __forceinline__ void finish(int& mti, int& mtw) {
    if (pend) produce();  // inputs already requested: completing the block keeps `tw` consistent
    const uint32_t g = gen_of_cons();
    mti = (int)(cons - 624u * g);
    if (tw > 624u * (g + 1u)) {
      const uint32_t nw = tw - 624u * (g + 1u);  // words of generation g+1 twisted ahead
      for (uint32_t k = (uint32_t)lig; k < nw; k += L) st()[k] = shadow()[k];
      mtw = 624;
    } else {
      mtw = (int)(tw - 624u * g);
    }
  }
};

// The double-precision pow of the Metropolis rule, kept out of line: it is reached with
// probability ~1e-5 per uphill move and would otherwise dominate the kernel's register budget.
__device__ __attribute__((noinline)) bool accept_exact(double x, double beta, double u, int f32) {
  return u <= rnd_cost(pow(x, -beta), f32);
}

// ---------------------------------------------------------------------------
// `uniform <= prob(delta, total)` (optimizer.hpp:162) for the rules of
// include/tnco/optimize/prob/{base,greedy,mh}.hpp.  Metropolis: p = pow(1 + delta/total, -beta)
// (mh.hpp:52-58).  The comparison is first decided in the log2 domain with single-precision
// hardware logs and an error margin; only when u falls inside the margin is the double-precision pow
// evaluated, so the decision is the one `u <= pow(...)` gives.
//
// Margin (beta >= 0, x = 1 + delta/total >= 1, 0 < u < 1; outside these the exact path is taken):
//   lu = log2f((float)u), lx = log2f((float)x), lp = -beta * lx.
//   * rounding u, x to float: relative 2^-24 each -> |d lu|, |d lx| <= 1.4427 * 6.0e-8 = 8.6e-8
//     absolute (|d log2 t| = |dt / t| * log2 e), hence up to 8.6e-8 * beta on lp;
//   * v_log_f32: assumed |err| <= 2e-7 + 1.2e-7 * |result| (1 ulp of the result plus an absolute
//     floor near 1) -- an assumption about the instruction, not a documented bound;
//   * (float)beta and the product -beta * lx: relative 2^-24 each.
//   Sum: |lp_true - lp| + |lu_true - lu| <= (|lp| + |lu|) * 2.4e-7 + beta * 2.9e-7 + 2e-7, bounded
//   by the margin (|lp| + |lu|) * 2e-6 + beta * 3e-7 + 1e-5 used below.
// Cross-check: a build with -DTNCO_CHECK_ACCEPT evaluates the exact rule next to every filtered
// decision and traps on a difference (tools/fuzz_gpu.py and the parity tests run clean with it:
// profiles/r02_accept_check.txt).
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool accept_move(int kind, double beta, double delta, double total, double u,
                                            int f32) {
  if (kind == 0) return true;             // base.hpp: p = 1, u < 1
  if (delta <= 0) return true;            // greedy.hpp / mh.hpp: p = 1
  if (kind == 1) return u <= 0.0;         // greedy.hpp: p = 0
  if (total == 0) return u <= 0.0;        // mh.hpp:55-57
  const double x = rnd_cost(1.0 + rnd_cost(delta / total, f32), f32);
  const float uf = (float)u, xf = (float)x, bf = (float)beta;
  const float lu = __log2f(uf), lx = __log2f(xf);
  const float lp = -bf * lx;
  const float margin = (fabsf(lp) + fabsf(lu)) * 2e-6f + fabsf(bf) * 3e-7f + 1e-5f;
  if (uf > 1e-30f && xf < 1e30f && fabsf(lp) < 1e30f && beta >= 0.0) {
#ifdef TNCO_CHECK_ACCEPT
    if (lu < lp - margin || lu > lp + margin) {
      const bool fast = lu < lp - margin;
      if (fast != accept_exact(x, beta, u, f32)) __builtin_trap();
      return fast;
    }
#else
    if (lu < lp - margin) return true;
    if (lu > lp + margin) return false;
#endif
  }
  return accept_exact(x, beta, u, f32);
}

// ---------------------------------------------------------------------------
#ifndef TNCO_WAVES_PER_SIMD
#define TNCO_WAVES_PER_SIMD 3
#endif
enum : int { S_BEGIN = 0, S_GOT_B = 1, S_GOT_HB = 2, S_GOT_B1 = 3, S_GOT_HA = 4, S_MOVE = 5, S_END = 6 };

// Scalars of a replica's walk.  A LEAN instantiation (the finite-width moves) keeps them in LDS instead of registers, one
// slot per replica -- the lanes of a group hold the same values: the Metropolis operands, the partial costs of the children,
// the headers of A and of A's parent that wait for their move, the counters.  With them (and the sliced-index mask) out
// of the register file the moves need at most 192 VGPRs: two of their wavefronts leave room on a SIMD for one wavefront
// of the OTHER stream's re-slice kernel (127 VGPRs), which is bound by instruction issue while the moves wait on memory.
struct WalkScal {
  double total, beta, pP, pO, pC, raC, rnC;
  int raL, raR, raP, rnL, rnR, rnP, raW, rnW;
  uint32_t n_moves, n_acc;
};
// Rarely touched per-replica state lives in LDS, not in registers (VGPRs bound the occupancy).
struct ColdState {
  double min_cost;
  uint32_t jmin, n_impr, n_full, n_rpick;
};
typedef TNCO_LDS volatile ColdState lds_cold;

// Stage timing (diagnostic builds only, -DTNCO_PROFILE): shader cycles between five points of the
// loop body, accumulated per replica into ReplicaState::pad1 (tnco_hip_diag_stage_cycles).
#ifdef TNCO_PROFILE
#define TNCO_PROF_DECL unsigned long long pt_[5] = {0, 0, 0, 0, 0}, pa_[5] = {0, 0, 0, 0, 0}
#if TNCO_PROFILE == 2
// fine split of the "state branches" stage: [end-of-sweep block, addresses, load issue, move]
#define TNCO_PROF_T(i)                                           \
  do {                                                           \
    if ((i) == 1) pt_[0] = __builtin_amdgcn_s_memtime();         \
    if ((i) == 2) pt_[4] = __builtin_amdgcn_s_memtime();         \
  } while (0)
#define TNCO_PROF_F(i) pt_[i] = __builtin_amdgcn_s_memtime()
#else
#define TNCO_PROF_T(i) pt_[i] = __builtin_amdgcn_s_memtime()
#define TNCO_PROF_F(i)
#endif
#define TNCO_PROF_ACC                                            \
  do {                                                           \
    pa_[0] += pt_[1] - pt_[0]; pa_[1] += pt_[2] - pt_[1];        \
    pa_[2] += pt_[3] - pt_[2]; pa_[3] += pt_[4] - pt_[3];        \
    pa_[4] += 1;                                                 \
  } while (0)
#define TNCO_PROF_OUT(rs) \
  for (int k_ = 0; k_ < 5; ++k_) (rs)->pad1[k_] += pa_[k_]
#else
#define TNCO_PROF_DECL
#define TNCO_PROF_T(i)
#define TNCO_PROF_F(i)
#define TNCO_PROF_ACC
#define TNCO_PROF_OUT(rs)
#endif

// Threads per block of the sweep kernel.  Replicas never talk to each other, so a block is only a unit of
// dispatch: a block's resources go back to the dispatcher when its LAST wavefront ends.
#ifndef TNCO_SWEEP_THREADS
#define TNCO_SWEEP_THREADS 256
#endif
constexpr int SWT = TNCO_SWEEP_THREADS;
#ifndef TNCO_FW_STAGED_WAVES
#define TNCO_FW_STAGED_WAVES 2
#endif
// which instantiations keep WalkScal and the slices in LDS
#ifndef TNCO_LEAN_RULE
#define TNCO_LEAN_RULE (FW && !GENERIC && !HYPER && LOG2L == 2)
#endif
template <int LOG2L, int K, bool HYPER, bool GENERIC, bool FW = false, bool SPREAD = false>
// (hyper-indices: six more masks are carried -- at 3 wavefronts per SIMD the K = 3 kernel spilled 51 VGPRs)
#ifndef TNCO_HYPER_WAVES
#define TNCO_HYPER_WAVES 2
#endif
// (four mask words per lane: at three wavefronts per SIMD the kernel spills; at two, 1360 leaves
// (8 lanes x 4 words) run 3.25 -> 3.6e9 move-evals/s, 680 leaves the same within the +-3 % of the boxes)
// (the general cost models -- dims tables, per-index dims, sparse legs, float32 -- spill 80 bytes per lane at three
// wavefronts per SIMD; at two: +12 ... +18 % on every one of them, round 3, profiles/r03_other_configs.md)
#ifndef TNCO_GENERIC_WAVES
#define TNCO_GENERIC_WAVES 2
#endif
// (four words per lane WITHOUT hyper-indices: round 5 -- what is spilled at three wavefronts per SIMD sits outside the loop)
#ifndef TNCO_K4_WAVES
#define TNCO_K4_WAVES 2
#endif
__device__ __forceinline__ void sa_run_body(
    const Params& P, const double* __restrict__ betas, const int64_t n_steps, const int prob_kind,
    const FwParams& F, const int tail_last, const int block0) {
  // SPREAD (a batch smaller than the kernel's wavefront slots): fewer replicas per wavefront than 64 / L -- `block0` of
  // them, the host's choice --, the other lane groups SHADOW them: same replica, same reads, hence the same values and
  // control flow, no store of their own.  Sixteen replicas in sixteen states make a wavefront run every state's code
  // in every iteration -- 5 300 cycles at 512 leaves, of which the landing fence waits 13: a small batch is bound by this
  // instruction stream, not by memory --, and a CU wants 64 active lanes (tools/few_lanes.hip; csrc/sa_small.h).
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = SWT >> LOG2L;  // groups (replicas) per block
  using M = Mask<K>;
  using R = Rng<LOG2L>;
  __shared__ uint32_t rngbuf[GPB * R::RING];
  __shared__ ColdState coldbuf[GPB];
  __shared__ int32_t jbuf[GPB * 16];
  constexpr bool LEAN = TNCO_LEAN_RULE;
  __shared__ WalkScal scalbuf[LEAN ? GPB : 1];
  __shared__ uint64_t slbuf[LEAN && FW ? GPB * L * K : 1];  // (LEAN: the sliced indices, constant during a launch)
  // The general cost models read their tables from LDS (sa_kernels.h, TabsLds): the cost table d^k (uniform dims that
  // are not a power of two; the one-odd-part chain) or the odd parts of per-index dims -- one buffer, a cost mode uses
  // one of them --, the exponent classes, the odd-part mask; the sparse mask sits in registers.
  __shared__ double tabbuf[GENERIC ? (L * K * 64 + 1) : 1];
  __shared__ uint64_t clsbuf[GENERIC ? (TABS_MAXCLS * L * K) : 1];
  __shared__ uint64_t oddbuf[GENERIC ? (L * K) : 1];
  if constexpr (GENERIC) {
    const bool use_ctab = P.cost_mode == 1 || (P.cost_mode == 2 && P.odd_single);
    if (use_ctab) {
      for (int i = threadIdx.x; i < P.W * 64 + 1; i += SWT) tabbuf[i] = P.ctab[i];
    } else if (P.cost_mode == 2) {
      for (int i = threadIdx.x; i < L * K * 64; i += SWT) tabbuf[i] = P.dimsd[i];
    }
    if (P.cost_mode >= 2) {
      for (int i = threadIdx.x; i < P.n_dimclass * L * K; i += SWT) clsbuf[i] = P.dimclass[i];  // (n_dimclass <= TABS_MAXCLS)
      if (P.cost_mode == 2)
        for (int i = threadIdx.x; i < L * K; i += SWT) oddbuf[i] = P.oddmask[i];
    }
    __syncthreads();
  }

  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gbase = (tid & 63) & ~(L - 1);  // first lane of the group inside the wave
  // (block0: the launch covers the replicas of blocks block0 .. block0 + gridDim.x - 1 -- a handle may split a
  // step over several streams, launch_run_lk)
  const int seats = SPREAD ? block0 : (64 >> LOG2L);             // replicas per wavefront (SPREAD: `block0` carries them)
  const int gw = (tid & 63) >> LOG2L;                            // lane group inside the wavefront
  const bool master = !SPREAD || gw < seats;
  const int seat = SPREAD ? (gw & (seats - 1)) : gw;
  const int gib = SPREAD ? (tid >> 6) * (64 >> LOG2L) + seat : tid >> LOG2L;  // the replica's slot in the block's LDS arrays
  const int64_t r = SPREAD ? ((int64_t)blockIdx.x * (SWT / 64) + (tid >> 6)) * seats + seat
                           : ((int64_t)blockIdx.x + block0) * GPB + gib;
  if (r >= P.R || n_steps <= 0) return;
  const bool lane0 = (lig == 0) && master;
  const int slig = master ? lig : (1 << 20);  // (the lane index of loops that store)

  const int n = P.n, N = P.N;
  View<LOG2L, K, HYPER, !FW> v;  // (finite width: the split layout)
  // Hyper-indices.  The reference keeps hyper[p] = legs(p) & legs(c0) & legs(c1) per node (infinite_memory/utils.hpp:82-91)
  // and updates it with that very formula (optimizer.hpp:171-172), so it never has to be STORED: the own legs of B and A
  // sit in the line their headers come from, the children's legs are carried anyway.
  // The own legs of B / A are carried (hB / hA hold them), hyper legs derived; the node blocks are those of a network
  // without hyper-indices -- one line per node at <= 12 mask words (round 5; rounds 1-4 stored W more words per node
  // and read + wrote them with every move).
  constexpr bool HYD = HYPER;
  v.init(P, P.blocks + r * P.RB, nullptr, lig);
  auto lpar = [&]() -> int32_t* { return P.lpar + r * (int64_t)n * LPS; };
  lds_cold& cold = *((lds_cold*)coldbuf + gib);

  R rng;
  {
    const ReplicaState* rs = P.rs + r;
    rng.init(P, r, (lds_vu32*)rngbuf + gib * R::RING, rs->mti, rs->mtw, master ? lig : 64);  // (64: no loads, no stores)
    if (lane0) {
      cold.min_cost = rs->min_cost;
      cold.jmin = rs->jmin;
      cold.n_impr = 0; cold.n_full = 0; cold.n_rpick = 0;
    }
  }
  const uint32_t jcap = (uint32_t)P.jcap;
  uint32_t jtail = P.rs[r].jtail;
  bool jinvalid = P.rs[r].jinvalid != 0;
  // The tail of the rotation log is assembled in LDS and written 16 entries at a time: HBM writes
  // whole 64-byte pieces, a 4-byte append evicted on its own is a read-modify-write there.
  auto jb = [&](int i) -> lds_vi32& { return *((lds_vi32*)jbuf + gib * 16 + i); };
  // (recomputed at every use, opaquely: as a loop invariant it would be spilled and its reload
  // would put a vmcnt(0) wait into the store phase)
  auto jlog = [&]() -> int32_t* {
    uint32_t rr = rng.r32;
    __asm__ volatile("" : "+v"(rr));
    return P.jlog + (int64_t)rr * (int64_t)P.jcap;
  };
  if ((jtail & 15u) != 0u) {
    for (int q = slig; q < 4; q += L) {
      const int4 t = *reinterpret_cast<const int4*>(jlog() + (jtail & ~15u) + 4 * q);
      jb(4 * q + 0) = t.x; jb(4 * q + 1) = t.y; jb(4 * q + 2) = t.z; jb(4 * q + 3) = t.w;
    }
  }
  // finite width: the sliced indices (constant during this kernel), this lane's words
  M sl_regs = mzero<K>();
  typedef TNCO_LDS volatile uint64_t lds_vu64;
  [[maybe_unused]] lds_vu64* slw = (lds_vu64*)slbuf + (LEAN && FW ? gib * (L * K) : 0);
  [[maybe_unused]] bool impr_any = false;
  if constexpr (FW) {
    const uint64_t* s0 = F.slices + r * 2 * (int64_t)(L * K);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if constexpr (LEAN && FW) slw[v.widx(k)] = s0[v.widx(k)]; else sl_regs.w[k] = s0[v.widx(k)];
    }
  }
  [[maybe_unused]] TabsLds<K> tabs{P, (lds_cdouble*)tabbuf, (lds_cu64*)clsbuf, (lds_cu64*)oddbuf, mzero<K>()};
  if constexpr (GENERIC) {
    if (P.sparse != nullptr) {
#pragma unroll
      for (int k = 0; k < K; ++k) tabs.sp.w[k] = P.sparse[k * L + lig];
    }
  }
  const int f32 = GENERIC ? P.f32 : 0;
  const int log2d = P.log2d;
  const bool disable_shared = P.disable_shared != 0;
  const int nsteps32 = (int)n_steps;

  // ---- carried state: B and what is known about its two children ----------
  int B = 0, bl = 0, br = 0, A = -1;
  [[maybe_unused]] int wB = 0;  // finite width: the spare header word (cached width) of B (A's and parent(A)'s: S.raW, S.rnW)
  double ccB = 0, partB = 0;
  WalkScal scal_regs{0, 0, 0, 0, 0, 0, 0, -1, -1, -1, -1, -1, -1, 0, 0, 0, 0};
  TNCO_LDS volatile WalkScal* scal_lds = (TNCO_LDS volatile WalkScal*)scalbuf + (LEAN ? gib : 0);
  auto& S = *[&]() {
    if constexpr (LEAN) return scal_lds; else return &scal_regs;
  }();
  if constexpr (LEAN) {
    S.total = 0; S.beta = 0; S.pP = 0; S.pO = 0; S.pC = 0; S.raC = 0; S.rnC = 0;
    S.raL = -1; S.raR = -1; S.raP = -1; S.rnL = -1; S.rnR = -1; S.rnP = -1; S.raW = 0; S.rnW = 0; S.n_moves = 0; S.n_acc = 0;
  }
  // (hyper-indices: hB / hA = the OWN legs of B / A)
  // The legs of B's two children, by ROLE, not by slot: mP = the child the walk came up through (after a move: the old B,
  // with its new legs), mO = the other one; `pl`: mP is the LEFT child (child 0).  The slots only matter to the (D, E)
  // rule -- cand0 is D = child0 -- so a move ends with two plain assignments instead of a left / right select of four masks.
  M mP = mzero<K>(), mO = mzero<K>(), hB = mzero<K>();
  bool pl = true;
  // (S.pP / S.pO: the partial costs of the two children, by role like their legs; S.ra* / S.rn*: the headers of A and of
  //  A's parent, landed in earlier iterations, that wait for their move)
  M mC = mzero<K>(), hA = mzero<K>();
  int step = 0;
  int state = S_BEGIN;

  // What the store phase needs (written by END / MOVE, read behind the fence under the same conditions) and the staging
  // registers of the loads live ACROSS iterations: an iteration that does not write them keeps stale values nobody
  // reads, instead of re-initialising some forty registers per iteration (a third of this loop's VALU instructions were
  // moves).  The words of a mask beyond W are never loaded for an internal node and zero in the leaf table: they stay
  // zero.  gMp alone is reset: a leaf's partial cost is the zero it starts from.
  // (B's record is written from the carried B, A, bl, br, ccB, partB, wB themselves: the B <- A shift follows the stores)
  int stC = 0, stE = 0;
  [[maybe_unused]] double stW64 = 0;
  int x_al = 0, x_ar = 0, x_aP = -1;
  double x_ccA = 0, x_partA = 0, x_pCcur = 0;
  int gL = -1, gR = -1, gP = -1;
  [[maybe_unused]] int gW = 0;
  double gC = 0;
  M gM = mzero<K>(), gH = mzero<K>();
  uint32_t gXlo = 0, gXhi = 0;

  TNCO_PROF_DECL;
  for (;;) {
    TNCO_PROF_T(0);
    // ======================= mt19937: consume what landed =====================
    // (the next block is requested further down, with the other loads: registers are tracked per
    // wave, not per lane -- a block requested by one replica AHEAD of another replica's produce()
    // in program order would make that produce() wait for it)
    if (rng.pend) rng.produce();
    TNCO_PROF_T(1);

    // what the store phase needs
    bool acc = false, did_move = false, did_end = false, improved = false;

    if (state == S_END) {
      // ---- B is the root: end of sweep (optimizer.hpp:194-201) ------------
      did_end = true;
      // (finite width, a launch that ends in a re-slicing sweep: fw_reslice_kernel closes that sweep)
      const bool close_sweep = !FW || tail_last != 0 || step != nsteps32 - 1;
      if (close_sweep && partB < cold.min_cost) {
        if constexpr (FW) impr_any = true;  // min_slices := slices, once, at the end of the kernel
        if (lane0) {
          cold.min_cost = partB;
          cold.n_impr = cold.n_impr + 1;
        }
        if (jinvalid) {
          // the rotation log overflowed since the last best tree: take a full copy (rare).  The
          // root header of this sweep is still in registers: write it first.
          if (lane0) {
            NodeRec o;
            o.left = bl; o.right = br; o.parent = -1; o.pad = 0; o.ccost = ccB; o.partial = partB;
            *v.hdr(B) = o;
          }
          v.lpar = lpar();
          // (the sixteen replicas of the wavefront wait for this copy: four nodes per lane in flight)
          Links* __restrict__ ml = P.minlinks + r * (int64_t)N;
          for (int i0 = slig; i0 < N; i0 += 4 * L) {
            int4 h4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int i = i0 + q * L;
              h4[q] = make_int4(-1, -1, -1, 0);
              if (i < n) h4[q].z = v.lpar[(int64_t)i * LPS];
              else if (i < N) h4[q] = *reinterpret_cast<const int4*>(v.hdr(i));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int i = i0 + q * L;
              if (i < N) {
                Links o;
                o.left = h4[q].x; o.right = h4[q].y; o.parent = h4[q].z; o.pad = 0;
                ml[i] = o;
              }
            }
          }
          jtail = 0;
          jinvalid = false;
          if (lane0) cold.n_full = cold.n_full + 1;
        }
        improved = true;  // cold.jmin = jtail: in the store phase (nothing may read jtail's register here)
      }
      ++step;
      state = (step >= nsteps32) ? -1 : S_BEGIN;
    }

    TNCO_PROF_F(1);
    // ======================= what does each replica need next? ================
    // Every state asks for at most: one node header (hN), the legs + partial cost of one node
    // (x1), 8 bytes from anywhere (xa) and, with hyper-indices, the hyper legs of one node
    // (yN).  Only addresses are decided per state; the loads below are one sequence for all.
    int hN = -1, x1 = -1, yN = -1;
    const uint32_t* xa = nullptr;
    if (state == S_BEGIN) {
      // optimizer.hpp:103-107: a random leaf; its parent is B.  The leaf IS the child the walk comes up through: its legs
      // (the shared leaf table) are asked for right here, so that the walk starts one iteration earlier (round 6: four
      // iterations before the first move instead of five).  x_al keeps the leaf until B's header says which child it is.
      const uint32_t x = rng.next();
      x_al = (int)(x % (uint32_t)n);
      x1 = x_al;
      xa = reinterpret_cast<const uint32_t*>(lpar() + (int64_t)x_al * LPS);
    } else if (state == S_GOT_B) {
      hN = B;
      xa = reinterpret_cast<const uint32_t*>(&v.hdr(B)->partial);
    } else if (state == S_GOT_HB) {
      x1 = pl ? br : bl;  // the leaf's sibling
      hN = A;
      yN = B;
      xa = reinterpret_cast<const uint32_t*>(&v.hdr(N - 1)->partial);  // optimizer.hpp:112
    } else if (state == S_GOT_B1) {
      xa = reinterpret_cast<const uint32_t*>(betas + step);
      if (A >= 0) {
        hN = S.raP;
        yN = A;
        x1 = (S.raL == B) ? S.raR : S.raL;  // C, the sibling of B (A's header landed with the leaf's sibling)
      }
    } else if (state == S_MOVE) {
      if (S.raP >= 0) {
        x1 = (S.rnL == A) ? S.rnR : S.rnL;  // the sibling of A: C of the next move
        hN = S.rnP;
        yN = S.raP;
      }
    }

    TNCO_PROF_F(2);
    // ======================= requests (staging registers) ====================
    // Nothing below reads these registers before the landing fence.
    if (hN >= 0) {
      const NodeRec* q = v.hdr(hN);
      gL = q->left; gR = q->right; gP = q->parent; gC = q->ccost;
      if constexpr (FW) gW = q->pad;
    }
    double gMp = 0;
    if (x1 >= 0) {
      v.mask_stage_into(gM, x1);
      if (x1 >= n) gMp = v.hdr(x1)->partial;
    }
    if (xa != nullptr) {
      gXlo = xa[0];
      gXhi = xa[1];
    }
    if constexpr (HYPER) {  // own legs of node yN
      if (yN >= 0) v.mask_stage_into(gH, yN);
    }
    if (rng.room()) rng.request();
    TNCO_PROF_F(3);

    if (state == S_MOVE) {
      did_move = true;
      int al = S.raL, ar = S.raR;
      const int aP = S.raP;
      double ccA = S.raC;
      const bool c_is_right = (al == B);
      const int C = c_is_right ? ar : al;

      // ---- one move evaluation (optimizer.hpp:117-192) -----------------------
      // hyper[A] | hyper[B] (optimizer.hpp:145-147); derived: hyper[B] = B & c0 & c1, hyper[A] = A & B & C
      const M hy = HYD ? mand<K>(hB, mor<K>(mand<K>(hA, mC), mand<K>(mP, mO))) : mzero<K>();
      M sl = sl_regs;
      if constexpr (LEAN && FW) {
#pragma unroll
        for (int k = 0; k < K; ++k) sl.w[k] = slw[v.widx(k)];
      }
      // both candidate (D, E) assignments evaluated at once:
      //   cand0: D = child0, E = child1;  cand1: D = child1, E = child0
      // new legs of B: (D ^ C) | hyper_A | hyper_B   (optimizer.hpp:147)
      // PICK_FIRST (round 6; the finite-width moves, which are bound by their instruction stream): decide (D, E) from the
      // two intersect flags first, then price the ONE chosen candidate -- 786 -> 722 VALU instructions per iteration for
      // one more dependent DPP sum, +2.5 % on config 5 (profiles/experiments_r06.md).  The infinite-memory instantiations
      // (bound by memory requests; the headline one at the edge of three wavefronts per SIMD) keep pricing both at once.
      constexpr bool PICK_FIRST = !GENERIC && FW;
      // (the candidates by role: P = D is the path child, O = D is the other child; cand0 of the reference is D = child0)
      bool interP, interO;
      int pcAP = 0, pcBP = 0, pcAO = 0, pcBO = 0;
      if constexpr (!GENERIC && !PICK_FIRST) {
        // (finite width: both costs are over in1 | in2 | slices, finite_width/cost_model/simple.hpp:139-144;
        // sl is zero otherwise)
        uint32_t wP = mpopc<K>(mor<K>(mor<K>(mor<K>(mxor<K>(mP, mC), hy), mO), sl)) |
                      (mpopc<K>(mor<K>(mor<K>(mP, mC), sl)) << 13) | ((mnonzero<K>(mand<K>(mP, mC)) ? 1u : 0u) << 26);
        uint32_t wO = mpopc<K>(mor<K>(mor<K>(mor<K>(mxor<K>(mO, mC), hy), mP), sl)) |
                      (mpopc<K>(mor<K>(mor<K>(mO, mC), sl)) << 13) | ((mnonzero<K>(mand<K>(mO, mC)) ? 1u : 0u) << 26);
        wP = gsum<LOG2L>(wP);
        wO = gsum<LOG2L>(wO);
        interP = (wP >> 26) != 0;
        interO = (wO >> 26) != 0;
        pcAP = (int)(wP & 0x1fffu); pcBP = (int)((wP >> 13) & 0x1fffu);
        pcAO = (int)(wO & 0x1fffu); pcBO = (int)((wO >> 13) & 0x1fffu);
      } else {
        const uint32_t w = gsum<LOG2L>((mnonzero<K>(mand<K>(mP, mC)) ? 1u : 0u) |
                                       ((mnonzero<K>(mand<K>(mO, mC)) ? 1u : 0u) << 8));
        interP = (w & 0xffu) != 0;
        interO = (w >> 8) != 0;
      }
      bool pick0;  // true: (D, E) = (child0, child1)   -- get_ctree_nn, optimize/optimizer.hpp:128-144
      if (disable_shared || (interP && interO)) {
        pick0 = (rng.next() & 1u) != 0;  // optimize/optimizer.hpp:139
        if (lane0) cold.n_rpick = cold.n_rpick + 1;
      } else {
        pick0 = pl ? interP : interO;  // (inter0: child0 intersects C)
      }
      const bool pickP = pick0 == pl;  // D is the path child
      const M mD = msel<K>(pickP, mP, mO), mE = msel<K>(pickP, mO, mP);
      const M newB = mor<K>(mxor<K>(mD, mC), hy);
      const double pD = pickP ? S.pP : S.pO, pE = pickP ? S.pO : S.pP;
      const int E = pick0 ? br : bl;
      if constexpr (PICK_FIRST) {  // the chosen candidate's two costs: A over newB | E | slices, B over D | C | slices
        const uint32_t w = gsum<LOG2L>(mpopc<K>(mor<K>(mor<K>(newB, mE), sl)) | (mpopc<K>(mor<K>(mor<K>(mD, mC), sl)) << 13));
        pcAP = pcAO = (int)(w & 0x1fffu);
        pcBP = pcBO = (int)((w >> 13) & 0x1fffu);
      }

      // finite width (greedy/optimizer.hpp:174-190): the width of the new B (cached on accept) and
      // its width without the sliced indices, which gates the move
      [[maybe_unused]] double new_width_B = 0;
      bool fits = true;
      if constexpr (FW) {
        double sliced_width;
        if constexpr (!GENERIC) {
          // log2(d) * count (simple.hpp:43-47) with d = 2^k: whole numbers, exact in either width type, and
          // `k * count <= max_width` is a comparison of integers
          const uint32_t x = gsum<LOG2L>(mpopc<K>(newB) | (mpopc<K>(mandn<K>(newB, sl)) << 16));
          new_width_B = (double)(log2d * (int)(x & 0xffffu));
          sliced_width = (double)(log2d * (int)(x >> 16));
        } else if (F.log2dims == nullptr && P.sparse == nullptr) {
          const uint32_t x = gsum<LOG2L>(mpopc<K>(newB) | (mpopc<K>(mandn<K>(newB, sl)) << 16));
          new_width_B = fw_wr(F, F.log2d * (double)(x & 0xffffu));
          sliced_width = fw_wr(F, F.log2d * (double)(x >> 16));
        } else {
          new_width_B = fw_width<LOG2L, K>(P, F, newB, lig, gbase);
          sliced_width = fw_width<LOG2L, K>(P, F, mandn<K>(newB, sl), lig, gbase);
        }
        fits = sliced_width <= F.max_width;
      }
      double nA, nB;  // optimizer.hpp:152-155
      if constexpr (!GENERIC) {
        nA = pow2_cost(log2d * (pickP ? pcAP : pcAO), 0);
        nB = pow2_cost(log2d * (pickP ? pcBP : pcBO), 0);
      } else {
        nA = generic_cost_t<LOG2L, K>(P, tabs, mor<K>(mor<K>(newB, mE), sl), lig, gbase);
        nB = generic_cost_t<LOG2L, K>(P, tabs, mor<K>(mor<K>(mD, mC), sl), lig, gbase);
      }
      const double delta = rnd_cost(rnd_cost(nB - ccB, f32) + rnd_cost(nA - ccA, f32), f32);  // :158
      ++S.n_moves;

      // :162 (always drawn; finite width: only for a move that fits, greedy/optimizer.hpp:188-201)
      if (fits) {
        const double u = rng.uniform01();
        acc = accept_move(prob_kind, S.beta, delta, S.total, u, f32);
      }

      double pEcur = pE, pCcur = S.pC;  // partials of B's / A's other child after the move
      M mBnow, mX;                     // legs of B / of A's other child after the move
      stC = C; stE = E;
      if (acc) {
        ++S.n_acc;
        if constexpr (FW) {  // :216  width_B = new_width_B
          if (F.width_f32) wB = __float_as_int((float)new_width_B);
          stW64 = new_width_B;
        }
        // Tree::swap_with_nn(E): include/tnco/tree.hpp:176-184
        if (pick0) br = C; else bl = C;
        if (c_is_right) ar = E; else al = E;
        if constexpr (HYD) hB = newB;  // (B's own legs; A's stay)
        ccB = nB;
        ccA = nA;
        S.total = rnd_cost(S.total + delta, f32);  // :177
        pEcur = S.pC;
        pCcur = pE;
        mBnow = newB;
        mX = mE;
      } else {
        mBnow = HYD ? hB : mxor<K>(mP, mO);
        mX = mC;
      }
      // :185-188
      partB = rnd_cost(rnd_cost(pD + pEcur, f32) + ccB, f32);
      const double partA = rnd_cost(rnd_cost(partB + pCcur, f32) + ccA, f32);
      // what B's record becomes (written in the store phase); the B <- A shift (:191) follows it
      x_al = al; x_ar = ar; x_aP = aP; x_ccA = ccA; x_partA = partA; x_pCcur = pCcur;
      // :191, legs only (registers: the scalars follow after the store phase).  B becomes a child
      // of the next B: its legs, and those of A's other child, are what the next move starts from.
      mP = mBnow; mO = mX; pl = c_is_right;
    }

    TNCO_PROF_T(2);
    // ======================= landing fence ===================================
    // Everything requested above is needed before the first store below: vmcnt is in order, so
    // waiting for these loads later would also wait for the stores.
    TNCO_LANDED(gL); TNCO_LANDED(gR); TNCO_LANDED(gP); TNCO_LANDED(gC);
    if constexpr (FW) TNCO_LANDED(gW);
    TNCO_LANDED(gMp); TNCO_LANDED(gXlo); TNCO_LANDED(gXhi);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      TNCO_LANDED(gM.w[k]);
      if constexpr (HYPER) { TNCO_LANDED(gH.w[k]); }
    }
    TNCO_LANDED(rng.pb);
    TNCO_LANDED(rng.pa);
#pragma unroll
    for (int j = 0; j < 4; ++j) TNCO_LANDED(rng.pc[j]);

    TNCO_PROF_T(3);
    // ======================= store phase =====================================
    if (did_move && acc) {
      if (!jinvalid) {
        if (jtail == jcap) {
          jinvalid = true;  // log full: the next improvement re-bases the checkpoint
        } else {
          if (lane0) jb((int)(jtail & 15u)) = stE;
          ++jtail;
          if ((jtail & 15u) == 0u) {  // a 64-byte piece is complete: 16 bytes per lane
            for (int q = slig; q < 4; q += L)
              *reinterpret_cast<int4*>(jlog() + (jtail - 16u) + 4 * q) =
                  make_int4(jb(4 * q + 0), jb(4 * q + 1), jb(4 * q + 2), jb(4 * q + 3));
          }
        }
      }
      v.lpar = lpar();
      if (master) {
        v.set_parent_group(stC, B);
        v.set_parent_group(stE, A);
        v.set_mask(B, mP);  // :170 (accepted: B's legs are the new legs -- B is the path child of the next level)
      }
    }
    if (did_move || did_end) {
      if (lane0) {
        NodeRec o;
        o.left = bl; o.right = br; o.parent = A; o.pad = FW ? wB : 0; o.ccost = ccB; o.partial = partB;  // (END: A == -1)
        *v.hdr(B) = o;
        if constexpr (FW) {
          if (did_move && acc && !F.width_f32) F.width64[(int64_t)rng.r32 * N + B] = stW64;
        }
      }
      // Experiment (off): HBM writes whole 64-byte pieces and a shorter write is a read-modify-write
      // there (tools/hbm_random.hip), so complete the 32-byte header with legs 0..3, unchanged.
      // Measured -1.5 %: the block was read a few iterations ago and the L2 / Infinity Cache still
      // hold it, so the short write never reaches HBM on its own; the extra store only costs issue.
#ifdef TNCO_FIRST64
      if (!(did_move && acc))
        v.set_mask_first(B, did_move ? mP : (HYD ? hB : mxor<K>(mP, mO)));
#endif
      if (improved && lane0) cold.jmin = jtail;
      if (state < 0) break;
    }
    TNCO_PROF_T(4);
    TNCO_PROF_ACC;

    // ======================= what landed goes where ==========================
    if (did_move) {
      // :191  B <- A, carrying what is already known about A's children
      S.pP = partB; S.pO = x_pCcur;
      B = A; bl = x_al; br = x_ar; ccB = x_ccA; partB = x_partA;
      if constexpr (HYPER) { hB = hA; hA = gH; }
      A = x_aP;
      S.raL = S.rnL; S.raR = S.rnR; S.raP = S.rnP; S.raC = S.rnC;
      S.rnL = gL; S.rnR = gR; S.rnP = gP; S.rnC = gC;
      if constexpr (FW) { wB = S.raW; S.raW = S.rnW; S.rnW = gW; }
      mC = gM;
      S.pC = gMp;
      state = (A < 0) ? S_END : S_MOVE;
    } else if (state == S_BEGIN) {
      B = (int)gXlo;
      mP = gM; S.pP = gMp;  // (the leaf: the path child of the first move; its partial cost is the zero gMp starts from)
      state = S_GOT_B;
    } else if (state == S_GOT_B) {
      bl = gL; br = gR; A = gP; ccB = gC;
      if constexpr (FW) wB = gW;
      partB = __hiloint2double((int)gXhi, (int)gXlo);
      pl = gL == x_al;  // the leaf is B's left child
      state = S_GOT_HB;
    } else if (state == S_GOT_HB) {
      mO = gM; S.pO = gMp;
      S.total = __hiloint2double((int)gXhi, (int)gXlo);
      S.raL = gL; S.raR = gR; S.raP = gP; S.raC = gC;
      if constexpr (FW) S.raW = gW;
      if constexpr (HYPER) { hB = gH; }
      state = S_GOT_B1;
    } else if (state == S_GOT_B1) {
      mC = gM; S.pC = gMp;
      S.beta = __hiloint2double((int)gXhi, (int)gXlo);
      S.rnL = gL; S.rnR = gR; S.rnP = gP; S.rnC = gC;
      if constexpr (FW) S.rnW = gW;
      if constexpr (HYPER) { hA = gH; }
      state = (A < 0) ? S_END : S_MOVE;
    }
  }

  if ((jtail & 15u) != 0u) {  // the unfinished piece of the rotation log (entries past jtail: don't care)
    for (int q = slig; q < 4; q += L)
      *reinterpret_cast<int4*>(jlog() + (jtail & ~15u) + 4 * q) =
          make_int4(jb(4 * q + 0), jb(4 * q + 1), jb(4 * q + 2), jb(4 * q + 3));
  }
  if constexpr (FW) {
    if (impr_any) {  // :385-389  min_slices = slices (the slices did not change during these sweeps)
      uint64_t* s1 = F.slices + r * 2 * (int64_t)(L * K) + L * K;
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (master) s1[v.widx(k)] = (LEAN && FW) ? (uint64_t)slw[v.widx(k)] : sl_regs.w[k];
    }
  }
  int mti, mtw;
  rng.finish(mti, mtw);
  if (lane0) {
    ReplicaState* rs = P.rs + r;
    rs->jmin = cold.jmin; rs->jtail = jtail;
    rs->jinvalid = jinvalid ? 1 : 0;
    rs->n_fullcopy += cold.n_full;
    rs->min_cost = cold.min_cost;
    rs->n_moves += S.n_moves;
    rs->n_accepted += S.n_acc;
    rs->n_improved += cold.n_impr;
    rs->n_randpick += cold.n_rpick;
    rs->mti = mti;
    rs->mtw = mtw;
    if constexpr (!FW) { TNCO_PROF_OUT(rs); }
  }
}

// The kernels.  `sa_run_kernel`: every instantiation but the plain finite-width moves.
template <int LOG2L, int K, bool HYPER, bool GENERIC, bool FW = false, bool SPREAD = false>
__global__ __launch_bounds__(SWT, (LOG2L == 1 ? 2 : (FW ? TNCO_FW_STAGED_WAVES : (HYPER ? TNCO_HYPER_WAVES : (GENERIC ? TNCO_GENERIC_WAVES : (K >= 4 ? TNCO_K4_WAVES : TNCO_WAVES_PER_SIMD)))))) void sa_run_kernel(
    const Params P, const double* __restrict__ betas, const int64_t n_steps, const int prob_kind,
    const FwParams F, const int tail_last, const int block0) {
  sa_run_body<LOG2L, K, HYPER, GENERIC, FW, SPREAD>(P, betas, n_steps, prob_kind, F, tail_last, block0);
}

// `sa_run_fw_kernel`: the finite-width moves of the plain cost model in four lanes per replica (config 5) under a ceiling
// of 192 VGPRs -- a register attribute cannot depend on a template argument, hence a kernel of its own around the same
// body (its LEAN form: WalkScal and the slices in LDS; four registers spilled outside the loop).  Two of its wavefronts
// then leave a SIMD 128 registers: room for one wavefront of the OTHER stream's `fw_wave_kernel` (127), which is bound by
// instruction issue while these moves wait on memory -- the re-slice of one half of the batch runs in the shadow of the
// moves of the other (+3.9 % on config 5, same box, alternating: profiles/experiments_r06.md).  The attribute counts
// HALVES of the unified register file: 96 -> 192.
template <int LOG2L, int K, bool SPREAD = false>
__global__ __launch_bounds__(SWT, 2) __attribute__((amdgpu_num_vgpr(96))) void sa_run_fw_kernel(
    const Params P, const double* __restrict__ betas, const int64_t n_steps, const int prob_kind,
    const FwParams F, const int tail_last, const int block0) {
  sa_run_body<LOG2L, K, false, false, true, SPREAD>(P, betas, n_steps, prob_kind, F, tail_last, block0);
}
// which of the two a finite-width handle launches
template <int LOG2L, int K, bool HYPER, bool GENERIC, bool SPREAD>
__host__ inline auto fw_staged_kernel() {
  if constexpr (!HYPER && !GENERIC && LOG2L == 2) return &sa_run_fw_kernel<LOG2L, K, SPREAD>;
  else return &sa_run_kernel<LOG2L, K, HYPER, GENERIC, true, SPREAD>;
}

}  // namespace tnco
