// sa_sweep.h -- the sweep kernel: n_steps calls of Optimizer::update
// (include/tnco/optimize/infinite_memory/optimizer.hpp:90-221) per replica, for all replicas.
//
// Latency structure (gfx950): the leaf->root walk is a dependent pointer chase and vmcnt
// retires loads and stores IN ORDER, so a load issued after a store waits for the store's
// acknowledgement.  The loop is therefore software-pipelined by hand:
//   * the header of A(k+2) and the block of C(k+1) are requested at the TOP of move k, before
//     move k's stores, and consumed one / two moves later;
//   * the mt19937 inputs of the next 16-output block are requested when the current block is
//     generated;
// so that the only exposed memory latencies are at the start of a sweep.
#pragma once
#include "sa_kernels.h"

namespace tnco {

// ---------------------------------------------------------------------------
// std::mt19937, generated lazily in blocks of SB = min(L, 16) outputs per group.
// State words live in HBM ([624] per replica); the tempered outputs of the current block
// live in an LDS slot of the group; the raw inputs of the NEXT block are prefetched into
// registers.  (libstdc++ random.tcc:396-471; seeding :326-343.)
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
  static constexpr int L = 1 << LOG2L;
  static constexpr int LOG2SB = LOG2L < 4 ? LOG2L : 4;
  static constexpr int SB = 1 << LOG2SB;  // outputs per block
  static constexpr int NB = 624 / SB;

  uint32_t* st;            // replica's 624 state words (HBM)
  volatile uint32_t* buf;  // group's SB-word LDS slot
  int mti, mtw, cur_blk, lig;
  uint32_t pa, pb, pc;     // prefetched inputs of block pf_blk
  int pf_blk;

  __device__ __forceinline__ void init(uint32_t* st_, volatile uint32_t* buf_, int mti_, int mtw_, int lig_) {
    st = st_; buf = buf_; mti = mti_; mtw = mtw_; lig = lig_;
    cur_blk = -1; pf_blk = -1; pa = pb = pc = 0;
  }

  __device__ __forceinline__ void load_inputs(int blk, bool twist, uint32_t& a, uint32_t& b, uint32_t& c) {
    if (lig < SB) {
      const int k = blk * SB + lig;
      a = st[k];
      if (twist) {
        const int k1 = (k + 1 == 624) ? 0 : k + 1;
        int km = k + 397;
        if (km >= 624) km -= 624;
        b = st[k1];
        c = st[km];
      }
    }
  }

  __device__ __forceinline__ void refill(int blk) {
    const bool twist = (blk * SB) >= mtw;
    uint32_t a = pa, b = pb, c = pc;
    if (pf_blk != blk) load_inputs(blk, twist, a, b, c);
    if (lig < SB) {
      uint32_t v = a;
      if (twist) {
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        v = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        st[blk * SB + lig] = v;
      }
      buf[lig] = mt_temper(v);
    }
    if (twist) mtw = blk * SB + SB;
    cur_blk = blk;
    // request the next block's inputs now; they are consumed ~SB draws later.  The block after
    // the last one is block 0 of the next generation (always twisted, from the words just
    // completed).
    const int nb = (blk + 1 == NB) ? 0 : blk + 1;
    const bool ntwist = (blk + 1 == NB) ? true : ((nb * SB) >= mtw);
    load_inputs(nb, ntwist, pa, pb, pc);
    pf_blk = nb;
  }

  __device__ __forceinline__ uint32_t next() {
    if (mti >= 624) {
      mti = 0;
      mtw = 0;
      cur_blk = -1;
    }
    const int blk = mti >> LOG2SB;
    if (blk != cur_blk) refill(blk);
    const uint32_t v = buf[mti & (SB - 1)];
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

// The double-precision pow of the Metropolis rule, kept out of line: it is reached with
// probability ~1e-5 per uphill move and would otherwise dominate the kernel's register budget.
__device__ __attribute__((noinline)) bool accept_exact(double x, double beta, double u, int f32) {
  return u <= rnd_cost(pow(x, -beta), f32);
}

// ---------------------------------------------------------------------------
// `uniform <= prob(delta, total)` (optimizer.hpp:162) for the rules of
// include/tnco/optimize/prob/{base,greedy,mh}.hpp.  Metropolis: p = pow(1 + delta/total, -beta)
// (mh.hpp:52-58).  The comparison is first decided in the log2 domain with single-precision
// hardware logs and a rigorous error margin; only when u falls inside the margin is the
// double-precision pow evaluated, so the decision is always the one `u <= pow(...)` gives.
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
    if (lu < lp - margin) return true;
    if (lu > lp + margin) return false;
  }
  return accept_exact(x, beta, u, f32);
}

// ---------------------------------------------------------------------------
#ifndef TNCO_WAVES_PER_SIMD
#define TNCO_WAVES_PER_SIMD 4
#endif
// "this value is needed here": forces the wait for a staged load at a chosen program point and
// keeps memory operations from moving across it.
#define TNCO_LANDED(x) __asm__ volatile("" : "+v"(x) : : "memory")
template <int LOG2L, bool HYPER, bool GENERIC>
__global__ __launch_bounds__(256, TNCO_WAVES_PER_SIMD) void sa_run_kernel(
    const Params P, const double* __restrict__ betas, const int64_t n_steps, const int prob_kind) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;  // groups (replicas) per block
  __shared__ uint32_t rngbuf[GPB * Rng<LOG2L>::SB];

  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gib = tid >> LOG2L;
  const int gbase = (tid & 63) & ~(L - 1);  // first lane of the group inside the wave
  const int64_t r = (int64_t)blockIdx.x * GPB + gib;
  if (r >= P.R || n_steps <= 0) return;
  const bool lane0 = (lig == 0);

  const int n = P.n, N = P.N;
  View<LOG2L, HYPER> v;
  v.init(P, P.blocks + r * (int64_t)(n - 1) * P.BS, P.lpar + r * (int64_t)n, lig);
  ReplicaState* rs = P.rs + r;

  Rng<LOG2L> rng;
  rng.init(P.mt + r * 624, rngbuf + gib * Rng<LOG2L>::SB, rs->mti, rs->mtw, lig);

  double min_cost = rs->min_cost;
  uint32_t n_moves = 0, n_acc = 0, n_impr = 0, n_rpick = 0, n_full = 0;
  int32_t* __restrict__ jlog = P.jlog + r * (int64_t)P.jcap;
  const uint32_t jcap = (uint32_t)P.jcap;
  uint32_t jmin = rs->jmin, jtail = rs->jtail;
  bool jinvalid = rs->jinvalid != 0;
  const int f32 = GENERIC ? P.f32 : 0;
  const int log2d = P.log2d;
  const bool disable_shared = P.disable_shared != 0;

  // ---- carried state: B and what is known about its two children ----------
  int B, bl, br, A;
  double ccB, partB, total, beta;
  uint64_t m0, m1, iB = 0, hB = 0;
  double p0, p1;
  // ---- pipeline registers ---------------------------------------------------
  NodeRec recA;   // header of A            (valid when A >= 0)
  NodeRec recN;   // header of parent(A)    (valid when recA.parent >= 0)
  uint64_t mC = 0, iA = 0, hA = 0;  // legs of C, legs / hyper legs of A
  double pC = 0;

  auto start_sweep = [&](int64_t step) {
    beta = betas[step];
    // optimizer.hpp:103-112
    const uint32_t x = rng.next();
    const int leaf = (int)(x % (uint32_t)n);
    B = v.lpar[leaf];
    const NodeRec rb = *v.hdr(B);
    bl = rb.left;
    br = rb.right;
    A = rb.parent;
    ccB = rb.ccost;
    partB = rb.partial;
    total = (B == N - 1) ? partB : v.hdr(N - 1)->partial;
    if (A >= 0) recA = *v.hdr(A);
    m0 = v.mask(bl);
    m1 = v.mask(br);
    p0 = v.partial(bl);
    p1 = v.partial(br);
    if constexpr (HYPER) {
      iB = v.mask(B);
      hB = v.hyper(B);
    }
    if (A >= 0) {
      const int C = (recA.left == B) ? recA.right : recA.left;
      if (recA.parent >= 0) recN = *v.hdr(recA.parent);
      mC = v.mask(C);
      pC = v.partial(C);
      if constexpr (HYPER) {
        iA = v.mask(A);
        hA = v.hyper(A);
      }
    }
  };

  int64_t step = 0;
  start_sweep(0);

  for (;;) {
    if (A < 0) {
      // ---- B is the root: end of sweep (optimizer.hpp:194-201) ------------
      if (lane0) {
        NodeRec o;
        o.left = bl; o.right = br; o.parent = -1; o.pad = 0; o.ccost = ccB; o.partial = partB;
        *v.hdr(B) = o;
      }
      if constexpr (HYPER) v.set_hyper(B, hB);
      if (partB < min_cost) {
        min_cost = partB;
        ++n_impr;
        if (jinvalid) {
          // the rotation log overflowed since the last best tree: take a full copy
          Links* __restrict__ ml = P.minlinks + r * (int64_t)N;
          for (int i = lig; i < N; i += L) {
            Links o;
            o.left = v.left(i); o.right = v.right(i); o.parent = v.parent(i); o.pad = 0;
            ml[i] = o;
          }
          jtail = 0;
          jinvalid = false;
          ++n_full;
        }
        jmin = jtail;
      }
      ++step;
      if (step >= n_steps) break;
      start_sweep(step);
      if (A < 0) continue;
    }

    // ---- stage the NEXT move's operands before this move's stores -----------
    int al = recA.left, ar = recA.right;
    const int aP = recA.parent;
    double ccA = recA.ccost;
    const bool c_is_right = (al == B);
    const int C = c_is_right ? ar : al;
    NodeRec recNN{-1, -1, -1, 0, 0.0, 0.0};
    uint64_t mCn = 0, iAn = 0, hAn = 0;
    double pCn = 0;
    if (aP >= 0) {
      const int Cn = (recN.left == A) ? recN.right : recN.left;
      if (recN.parent >= 0) recNN = *v.hdr(recN.parent);
      mCn = v.mask(Cn);
      pCn = v.partial(Cn);
      if constexpr (HYPER) {
        iAn = v.mask(aP);
        hAn = v.hyper(aP);
      }
    }

    // ---- one move evaluation (optimizer.hpp:117-192) -----------------------
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
    bool pick0;  // true: (D, E) = (child0, child1)   -- get_ctree_nn, optimize/optimizer.hpp:128-144
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
    const bool acc = accept_move(prob_kind, beta, delta, total, u, f32);

    // vmcnt retires loads and stores in order and the compiler cannot count across the
    // divergent regions of this loop, so it would wait for the staged operands AFTER this move's
    // stores (i.e. for the stores' acknowledgements as well).  A fake use here, before the first
    // store of the move, makes it wait for them now: they have had the whole evaluation above
    // (plus the other waves of the SIMD) to arrive, and nothing younger is outstanding.
    TNCO_LANDED(recNN.left); TNCO_LANDED(recNN.right); TNCO_LANDED(recNN.parent); TNCO_LANDED(recNN.ccost);
    TNCO_LANDED(mCn); TNCO_LANDED(pCn);
    if constexpr (HYPER) { TNCO_LANDED(iAn); TNCO_LANDED(hAn); }
    TNCO_LANDED(rng.pa); TNCO_LANDED(rng.pb); TNCO_LANDED(rng.pc);

    double pEcur = pE, pCcur = pC;  // partials of B's other child / A's other child after the move
    uint64_t mBnow;                 // legs of B after the move
    if (acc) {
      ++n_acc;
      // Tree::swap_with_nn(E): include/tnco/tree.hpp:176-184
      if (pick0) br = C; else bl = C;
      if (c_is_right) ar = E; else al = E;
      if (!jinvalid) {
        if (jtail == jcap) {
          jinvalid = true;  // log full: the next improvement re-bases the checkpoint
        } else {
          if (lane0) jlog[jtail] = E;
          ++jtail;
        }
      }
      if (lane0) {
        v.set_parent(C, B);
        v.set_parent(E, A);
      }
      v.set_mask(B, newB);  // :170
      if constexpr (HYPER) {
        hA = iA & newB & mE;  // :171
        hB = newB & mD & mC;  // :172
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
      *v.hdr(B) = o;
    }
    if constexpr (HYPER) v.set_hyper(B, hB);  // B's hyper legs may also have changed one level below
    // :191  B <- A, carrying what we already know about A's children
    const uint64_t mX = acc ? mE : mC;  // legs of A's other child
    if (c_is_right) { m0 = mBnow; p0 = partB; m1 = mX; p1 = pCcur; }
    else            { m1 = mBnow; p1 = partB; m0 = mX; p0 = pCcur; }
    B = A; bl = al; br = ar; ccB = ccA; partB = partA;
    if constexpr (HYPER) { iB = iA; hB = hA; iA = iAn; hA = hAn; }
    A = aP;
    recA = recN;
    recN = recNN;
    mC = mCn;
    pC = pCn;
  }

  if (lane0) {
    rs->jmin = jmin; rs->jtail = jtail;
    rs->jinvalid = jinvalid ? 1 : 0;
    rs->n_fullcopy += n_full;
    rs->min_cost = min_cost;
    rs->n_moves += n_moves;
    rs->n_accepted += n_acc;
    rs->n_improved += n_impr;
    rs->n_randpick += n_rpick;
    rs->mti = rng.mti;
    rs->mtw = rng.mtw;
  }
}

}  // namespace tnco
