// fw_wave.h -- the re-slice of a replica in ONE WAVEFRONT (fw_wave_kernel) and its pieces: the too-wide tensors
// ordered by root-path keys, get_slices as a radix select over bit-sliced counts with the shuffle's permutation
// resolved per final position, the cost cache re-priced along path marks.  Part of fw_kernels.h (included there,
// between fw_reslice_a_kernel and fw_reslice_b_kernel, which take what this kernel leaves behind).
#pragma once

// ---------------------------------------------------------------------------------------------
// fw_wave_kernel's pieces.
//   * get_slices wants the TOO-WIDE tensors in post-order -- some tens.  The header array (one contiguous piece
//     in the split layout) is read once; every too-wide node gets the key of its root path (one bit per level,
//     0 = left; padded with ones; deeper first on ties): ascending keys ARE the post-order of
//     include/tnco/utils.hpp:34-51.  Ranks by counting.
//   * the re-priced CostCache (finite_width/utils.hpp:36-47) only needs children before parents: the node table
//     in LDS (8 bytes per node).  The new cost of a node is its old one times 2^(k * (joined - left)) over the
//     changed indices that are NOT among its children's legs -- and those that are sit on the paths from the
//     indices' holders up to where the paths meet: one lane per (index, holder) marks its path, then every node
//     prices itself.  The partial sums follow: every lane starts at its nodes with two leaf children, and the
//     LAST of two children to arrive at a parent (an LDS counter) goes on with the parent.  partial = (cost +
//     left) + right whatever the order of evaluation, so the sums are the reference's bit for bit.  A replica
//     that keeps the new slices rewrites its header array as whole lines.
// ---------------------------------------------------------------------------------------------
// BIG = false: up to 255 too-wide tensors per replica, 128 candidate legs per tensor -- the lean configuration (Sycamore-53:
// 30-60 too-wide tensors of up to ~90 legs); BIG = true: up to 1 023 and 512 -- networks sliced far below their natural width
// (1 000 tensors at 0.7 of the greedy width: hundreds of too-wide tensors of ~200 legs); more LDS and two more count planes.
template <bool BIG> constexpr int FWO_MAXW = BIG ? 1024 : 256;   // too-wide tensors the wavefront form orders (more: fw_reslice_a_kernel's traverse)
template <bool BIG> constexpr int FWS_MAXNP = BIG ? 512 : 128;   // candidate legs of one tensor (beyond 128: the sequential shuffle)
template <bool BIG> constexpr int FWS_NPL = BIG ? 10 : 8;        // planes of the bit-sliced too-wide counts
constexpr int FWT_JMAX = 32;     // internal nodes per lane at most: n - 1 <= 2048
constexpr int FWS_MINCAP = 16;   // too-wide tensors whose legs stay in LDS at least (the wavefront form sizes its LDS for that)

// std::mt19937 for one wavefront: outputs [.., hi) of the CURRENT generation are in the ring (256 entries, batches
// of 64 aligned to 64), words below `tw` of the state array are twisted.  A fill produces up to three batches in ONE
// memory round trip (a dependent round trip costs this kernel 5-10 us: everything it touches is cold): word i is
// twisted from words i, i + 1, i + 397 (mod 624), of which only i + 397 - 624 = i - 227 must be a NEW value, and
// that word lies before the 192 being produced.  (mti, mtw) in and out as Rng<> keeps them.
struct RngWave {
  uint32_t* s;
  lds_vu32* ring;
  int lane;
  uint32_t cons, tw, hi;
  bool pend;  // (fw_shuffle_lds's interface: nothing is ever in flight here)
  __device__ __forceinline__ void init(uint32_t* st, lds_vu32* ring_, int mti, int mtw, int lane_) {
    s = st; ring = ring_; lane = lane_; pend = false;
    if (mti >= 624) { cons = 624; tw = 624; } else { cons = (uint32_t)mti; tw = (uint32_t)mtw; }
    hi = cons >= 624 ? 624u : (cons & ~63u);
  }
  __device__ __forceinline__ void roll() { cons = 0; tw = 0; hi = 0; }
  // the (at most) three batches that start at hi
  __device__ __forceinline__ void fill() {
    const uint32_t k0 = hi, end = (k0 + 192u) < 624u ? (k0 + 192u) : 624u;
    uint32_t v[3], nx[3], far[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const uint32_t i = k0 + 64u * (uint32_t)b + (uint32_t)lane;
      v[b] = nx[b] = far[b] = 0u;
      if (i < end) {
        v[b] = s[i];
        if (i >= tw) {
          nx[b] = s[i + 1u == 624u ? 0u : i + 1u];
          far[b] = s[i + 397u >= 624u ? i + 397u - 624u : i + 397u];
        }
      }
    }
    // (every load of the wavefront above, every store below: word i + 1 is read before its lane rewrites it)
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const uint32_t i = k0 + 64u * (uint32_t)b + (uint32_t)lane;
      if (i < end) {
        uint32_t x = v[b];
        if (i >= tw) {
          const uint32_t y = (x & 0x80000000u) | (nx[b] & 0x7fffffffu);
          x = far[b] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
          s[i] = x;
        }
        ring[i & 255u] = mt_temper(x);
      }
    }
    if (end > tw) tw = end;
    hi = end;
  }
  __device__ __forceinline__ uint32_t next() {
    if (cons == 624u) roll();
    if (cons >= hi) fill();
    const uint32_t x = ring[cons & 255u];
    ++cons;
    return x;
  }
  // m <= 64 outputs from here on in the ring?  (false: the generation ends first)
  __device__ __forceinline__ bool ensure(uint32_t m) {
    if (cons == 624u) roll();
    if (cons + m > 624u) return false;
    while (hi < cons + m) fill();  // (hi - 256 <= cons - 64: nothing unconsumed is overwritten)
    return true;
  }
  __device__ __forceinline__ uint32_t peek(uint32_t k) const { return ring[(cons + k) & 255u]; }
  __device__ __forceinline__ void advance(uint32_t m) { cons += m; }
  // fw_shuffle_lds's interface
  __device__ __forceinline__ bool room() const { return false; }
  __device__ __forceinline__ void request() {}
  __device__ __forceinline__ void produce() {}
  __device__ __forceinline__ void prefetch() {}
  __device__ __forceinline__ uint32_t next_sync() { return next(); }
  __device__ __forceinline__ void finish(int& mti, int& mtw) const { mti = (int)cons; mtw = (int)tw; }
};

__device__ __forceinline__ uint64_t fws_shfl64(uint64_t x, int src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)x, src), hi = (uint32_t)__shfl((int)(uint32_t)(x >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t fws_shflx64(uint64_t x, int m) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), m);
  return ((uint64_t)hi << 32) | lo;
}
// inclusive sum over the lanes 0..w of a row of 16
__device__ __forceinline__ uint32_t fws_rowscan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  return v;
}

// ---------------------------------------------------------------------------------------------
// fw_wave_kernel: the whole re-slice of a replica in one wavefront.  The headers are read once (J nodes per lane,
// all loads in flight together with the old slices and the generator's position) and STAY IN REGISTERS; the node
// table (lo / hi, 8 bytes per node in LDS) is built from them twice -- for the ordering of the too-wide tensors,
// and again for the re-pricing -- because in between get_slices wants the LDS for the legs of the too-wide tensors:
// on the Sycamore-53 supremacy network a replica has 30-60 of them, the scan of the greedy pass visits each several
// times, and a visit that goes to memory is a cold round trip (5-10 us).
// LDS per replica: [list of the too-wide tensors 512 B][node table][one more region]; get_slices overlays the last
// two with [shuffle tables, generator ring, candidate list][legs of the first `cap` too-wide tensors]; the
// re-pricing uses the region for path masks / partial sums, the ordering for its keys.
// A replica one of the steps cannot do leaves with nwide = -2 (fw_reslice_a_kernel traverses it) or with the
// proposal written and fastflag = 0 (fw_reslice_b_kernel rebuilds it in full).
// ---------------------------------------------------------------------------------------------
constexpr int FWH_MAXH = 7;      // tensors holding one index at most, for the re-pricing (more: the full rebuild)
// re-pricing, per changed index: the two path starts (4 B) and joined / left (1 B); HYPER: 8 slots of (path start, leaf) + the holder count
__host__ __device__ inline size_t fww_chg_bytes(bool hyper, bool big) { return (size_t)(big ? FWT_MAXD<true> : FWT_MAXD<false>) * (4 + 1 + (hyper ? 8 * 2 * 2 + 1 : 0)) + 32; }
// get_slices' fixed part: shuffle steps per position [64 + 128], ring [256], candidates + swap targets (+ padding to 16)
__host__ __device__ inline size_t fww_gs_fixed(bool big) { return 192 * 8 + 1024 + (size_t)(big ? 512 : 128) * 3 + 64; }
__host__ __device__ inline size_t fww_lds_bytes(int n, int T, bool hyper = false, bool big = false) {  // T: lanes per mask (16, 32 or 64)
  const size_t nip = (size_t)((n - 1 + 63) & ~63), mw = big ? 1024 : 256;
  const size_t u1 = mw * (8 + 2 + 2);                                       // keys, nodes, depths
  const size_t u3 = nip * 8 + fww_chg_bytes(hyper, big);                    // masks / partial sums, change list, flags
  size_t body = nip * 8 + (u1 > u3 ? u1 : u3);
  const size_t gs = fww_gs_fixed(big) + (size_t)FWS_MINCAP * T * 8;         // get_slices at least
  body = body > gs ? body : gs;
  return (2 * mw /* list */ + body + 15) & ~(size_t)15;
}
// too-wide tensors whose legs get_slices keeps in LDS
__host__ __device__ inline int fww_cap(int n, int T, bool hyper = false, bool big = false) {
  const size_t mw = big ? 1024 : 256;
  size_t c = (fww_lds_bytes(n, T, hyper, big) - 2 * mw - fww_gs_fixed(big)) / ((size_t)T * 8);
  c = c < mw ? c : mw;
  return (int)(c & ~(size_t)7);  // (whole instructions of the LDS-direct loads: up to 8 tensors each)
}

#ifdef TNCO_FWW_PROF  // (diagnostic build: shader cycles per replica between the steps of fw_wave_kernel)
static __device__ unsigned long long g_fww_prof[24];
#define FWW_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define FWW_ACC(i, a, b) gacc_[i] += (b) - (a)
#define FWW_CNT(i, x) gacc_[i] += (x)
#else
#define FWW_T(v)
#define FWW_ACC(i, a, b)
#define FWW_CNT(i, x)
#endif
// inclusive sum over the lanes 0..w of a mask's 2^LOGT lanes
template <int LOGT>
__device__ __forceinline__ uint32_t fws_scan(uint32_t v, int lane) {
  v = fws_rowscan(v);
  if constexpr (LOGT == 5) {
    const uint32_t r0 = (uint32_t)__shfl((int)v, (lane & 32) + 15);
    v += (lane & 16) ? r0 : 0u;
  } else if constexpr (LOGT == 6) {
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const int row = lane >> 4;
    v += row == 1 ? r0 : (row == 2 ? r0 + r1 : (row == 3 ? r0 + r1 + r2 : 0u));
  }
  return v;
}

// The root sum of the re-priced costs (exponents in the node table), when it is exact AND does not beat `cur`: out of line,
// so that its few registers are not the ones that take fw_wave_kernel from four wavefronts per SIMD to three.
__device__ __attribute__((noinline)) bool fww_rejected_exactly(TNCO_LDS volatile uint32_t* hiv, int ni, int lane, double cur) {
  double s = 0.0;
  int emin = 2047;
  for (int i = lane; i < ni; i += 64) {
    const int e = (int)((hiv[i] >> 16) & 0x7FFu);
    emin = min(emin, e);
    s += __longlong_as_double((long long)((uint64_t)e << 52));
  }
  for (int st = 1; st < 64; st <<= 1) {
    emin = min(emin, __shfl_xor(emin, st));
    s += __longlong_as_double((long long)fws_shflx64((uint64_t)__double_as_longlong(s), st));
  }
  const int es = (__double2hiint(s) >> 20) & 0x7FF;
  return es - emin <= 51 && !(s < cur);  // (every partial sum is exact and the total does not beat the current one)
}

// LOGT: lanes per leg mask (4, 5, 6: networks of at most 16, 32, 64 mask words); 64 >> LOGT tensors per load instruction
// HYPER: indices held by more than two tensors (FwParams::holdern, up to FWH_MAXH each): the marks of the re-pricing below
template <int J, int LOGT, bool HYPER, bool BIG>
static __global__ __launch_bounds__(64) void fw_wave_kernel(const Params P, const FwParams F, const int cap, const int maxnp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fww_smem[];
  FWW_T(w0_);
  constexpr int GW = 64, IPP = 32, T = 1 << LOGT, TPL = 64 >> LOGT;
  const int lane = threadIdx.x, w = lane & (T - 1), g = lane >> LOGT;
  const int64_t r = blockIdx.x;
  const int n = P.n, N = P.N, ni = N - n, W = P.W, LK = F.I64 / 64;
  const int nip = (ni + 63) & ~63;
  constexpr int MW = FWO_MAXW<BIG>, MAXNP = FWS_MAXNP<BIG>, NPL = FWS_NPL<BIG>;
  TNCO_LDS volatile uint16_t* wls = (TNCO_LDS volatile uint16_t*)fww_smem;                 // [MW] too-wide tensors, post-order
  // node i of the table: lo = left | right << 16; hi = parent | cost exponent << 16 | internal children (later: still to arrive) << 27
  TNCO_LDS volatile uint32_t* lo = (TNCO_LDS volatile uint32_t*)(fww_smem + 2 * MW);       // [nip]
  TNCO_LDS uint32_t* hi = (TNCO_LDS uint32_t*)(lo + nip);                                  // [nip] (atomic arrivals)
  TNCO_LDS volatile uint32_t* hiv = (TNCO_LDS volatile uint32_t*)hi;
  uint8_t* U = fww_smem + 2 * MW + (size_t)nip * 8;                                        // the region behind the table
  // ordering
  TNCO_LDS volatile uint64_t* key = (TNCO_LDS volatile uint64_t*)U;                        // [MW]
  TNCO_LDS volatile uint16_t* wnode = (TNCO_LDS volatile uint16_t*)(key + MW);             // [MW]
  TNCO_LDS volatile uint16_t* dep = wnode + MW;                                            // [MW]
  // get_slices: over the node table and the region (the table is rebuilt from the registers afterwards)
  TNCO_LDS uint64_t* Mlo = (TNCO_LDS uint64_t*)(fww_smem + 2 * MW);                         // [64]  shuffle steps < 64 that target a position
  TNCO_LDS uint64_t* Mhi = Mlo + 64;                                                       // [128] ... steps 64..127 (the first T words: the picks, afterwards)
  lds_vu32* ring = (lds_vu32*)(Mhi + 128);                                                 // [256]
  lds_vu16* pos = (lds_vu16*)(ring + 256);                                                 // [MAXNP] candidate legs, ascending
  TNCO_LDS volatile uint8_t* jL = (TNCO_LDS volatile uint8_t*)(pos + MAXNP);                // [MAXNP] the position step i swaps with (np <= 128)
  TNCO_LDS volatile uint64_t* cache = (TNCO_LDS volatile uint64_t*)(fww_smem + 2 * MW + fww_gs_fixed(BIG));  // [cap][T] legs of the too-wide tensors
  // re-pricing
  TNCO_LDS volatile double* Pn = (TNCO_LDS volatile double*)U;                             // [nip]
  TNCO_LDS uint32_t* on = (TNCO_LDS uint32_t*)U;                                           // [nip][2] (the same memory)
  TNCO_LDS volatile uint32_t* onv = (TNCO_LDS volatile uint32_t*)U;
  constexpr int MAXD = FWT_MAXD<BIG>;
  TNCO_LDS volatile uint32_t* misc = (TNCO_LDS volatile uint32_t*)(Pn + nip);              // [8]
  TNCO_LDS volatile uint32_t* chgl = misc + 8;                                             // [MAXD] starts of the paths
  TNCO_LDS volatile uint8_t* pm = (TNCO_LDS volatile uint8_t*)(chgl + MAXD);               // [MAXD] 1: the index joins the slices, 0: it leaves
  // (HYPER) changed index o: its holders' path starts / leaves in slots 8 o .. 8 o + 7, their number | open << 7
  TNCO_LDS volatile uint16_t* pstart = (TNCO_LDS volatile uint16_t*)(pm + MAXD);           // [MAXD][8]
  TNCO_LDS volatile uint16_t* pleaf = pstart + MAXD * 8;                                   // [MAXD][8]
  TNCO_LDS volatile uint8_t* pcnt = (TNCO_LDS volatile uint8_t*)(pleaf + MAXD * 8);        // [MAXD]

  // ---- everything that depends on nothing, in flight at once
  uint8_t* hb = P.blocks + r * P.RB;
  uint64_t* sl = F.slices + r * 2 * (int64_t)LK;
  const FwScratch sc(F, r, N);
  ReplicaState* rs = P.rs + r;
  const bool has = w < W;
  const uint64_t old = has ? sl[w] : 0ull;                       // word w, in each of the 64 / T rows of T lanes
  const uint64_t skip = (F.skip != nullptr && has) ? F.skip[w] : 0ull;
  const int mti0 = rs->mti, mtw0 = rs->mtw;
  const double cur = reinterpret_cast<const NodeRec*>(hb + (int64_t)(ni - 1) * P.BS)->partial;
  const double* w64 = F.width64 ? F.width64 + r * (int64_t)N : nullptr;
  // node i = j * 64 + lane of the table, as it stays in registers: tl = left | right << 16; th = parent | cost exponent
  // << 16 | internal children << 27 (the root: parent 0xFFFF); iw = the spare header word (the cached float32 width)
  uint32_t tl[J], th[J];
  int32_t iw[J];
  bool widej[J];
  double wmax = 0.0;  // the widest too-wide tensor of this lane's nodes: legs = width / log2(d) bounds its candidate legs
  {
    int4 hd[J];
    uint32_t ce[J];
    double wd[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int i = j * GW + lane;
      hd[j] = make_int4(0, 0, 0, 0);
      ce[j] = 0;
      wd[j] = 0.0;
      if (i < ni) {
        hd[j] = *reinterpret_cast<const int4*>(hb + (int64_t)i * P.BS);
        ce[j] = *reinterpret_cast<const uint32_t*>(hb + (int64_t)i * P.BS + 20);  // high word of the cached cost
        if (!F.width_f32) wd[j] = w64[n + i];
      }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const uint32_t c = (hd[j].x >= n ? 1u : 0u) + (hd[j].y >= n ? 1u : 0u);
      tl[j] = (uint32_t)hd[j].x | ((uint32_t)hd[j].y << 16);
      th[j] = ((uint32_t)hd[j].z & 0xFFFFu) | (((ce[j] >> 20) & 0x7FFu) << 16) | (c << 27);
      iw[j] = hd[j].w;
      const double wv = F.width_f32 ? (double)__int_as_float(hd[j].w) : wd[j];
      widej[j] = (j * GW + lane < ni) && wv > F.max_width;
      if (widej[j] && wv > wmax) wmax = wv;
    }
  }
  if (lane == 0) F.fastflag[r] = 0;
  if (!__any(old != 0ull)) {  // greedy/optimizer.hpp:359: nothing to do without slices
    if (lane == 0) F.nwide[r] = -1;
    return;
  }
  auto leave_to_a = [&]() {  // fw_reslice_a_kernel traverses this replica, fw_reslice_b_kernel rebuilds it
    if (lane == 0) {
      F.nwide[r] = -2;
      atomicAdd(F.slowstat, 1ull);
      atomicAdd(F.slowstat + 1, 1ull);
    }
  };
  FWW_T(w1_);
#if defined(TNCO_FWW_STOP) && TNCO_FWW_STOP == 1  // (diagnostic builds, tools/pmc_fw_insts.sh: the phases' instruction counts by difference)
  leave_to_a();
  return;
#endif
  // ---- the node table; the too-wide tensors
  auto build_table = [&]() {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int i = j * GW + lane;
      if (i < ni) {
        lo[i] = tl[j];
        hiv[i] = th[j];
      }
    }
  };
  build_table();
  int nw = 0;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * GW + lane;
    const bool wide = widej[j];
    const unsigned long long b = __ballot(wide);
    if (wide) {
      const int k = nw + __popcll(b & ((1ull << lane) - 1ull));
      if (k < MW) wnode[k] = (uint16_t)(n + i);
    }
    nw += __popcll(b);
  }
  // (more too-wide tensors than the counts' planes hold, or a tensor with more legs -- hence possibly more candidate
  //  legs -- than the shuffle takes: nothing has been drawn yet)
  if (nw > MW - 1 || __any(wmax > F.log2d * (double)maxnp + 0.5)) {
    leave_to_a();
    return;
  }
  {  // root-path keys, ranks: ascending key, deeper first on equal keys (a node and its all-right ancestors)
    bool deep = false;
    for (int k0 = 0; k0 < nw; k0 += 64) {
      const int k = k0 + lane;
      if (k < nw) {
        int x = wnode[k], d = 0;
        uint64_t rev = 0;
        while (x != N - 1 && d <= 64) {
          const int p = (int)(hiv[x - n] & 0xFFFFu);
          rev = (rev << 1) | (uint64_t)((int)(lo[p - n] >> 16) == x);
          x = p;
          ++d;
        }
        if (d > 64) deep = true;
        uint64_t ky = d ? (__brevll((unsigned long long)rev)) : 0ull;  // level 0 (below the root) in bit 63
        if (d < 64) ky |= ~0ull >> d;
        key[k] = ky;
        dep[k] = (uint16_t)d;
      }
    }
    if (__any(deep)) {
      leave_to_a();
      return;
    }
    for (int k = lane; k < MW; k += 64) wls[k] = (uint16_t)n;
    for (int k0 = 0; k0 < nw; k0 += 64) {
      const int k = k0 + lane;
      if (k < nw) {
        const uint64_t ky = key[k];
        const int d = dep[k];
        int rank = 0;
        for (int m = 0; m < nw; ++m) {
          const uint64_t km = key[m];
          const int dm = dep[m];
          rank += (km < ky || (km == ky && dm > d)) ? 1 : 0;
        }
        wls[rank] = wnode[k];
      }
    }
  }
  FWW_T(w2_);
#if defined(TNCO_FWW_STOP) && TNCO_FWW_STOP == 2
  leave_to_a();
  return;
#endif
  // ---- get_slices (the ordering's keys and the node table are dead: their memory is the legs' now)
  __builtin_amdgcn_wave_barrier();
  const uint8_t* legs = P.blocks + r * P.RB + P.WOFF;
  const int WS = P.WS;
  uint64_t pl[NPL];
#pragma unroll
  for (int p = 0; p < NPL; ++p) pl[p] = 0ull;
  // The legs of the first `cap` too-wide tensors go STRAIGHT into LDS (global_load_lds, 16 bytes per lane: 1 KB = TPI
  // tensors per instruction), all requests in one flight together with the generator's first batches -- through
  // registers it took a round trip per sixteen tensors (three on the Sycamore network) before the counting could start.
  constexpr int LPT = T / 2, TPI = 64 / LPT;  // lanes per tensor at 16 bytes each; tensors per instruction
  const int nres = nw < cap ? nw : cap;
  for (int t0 = 0; t0 < nres; t0 += TPI) {  // (cap is a multiple of TPI: the last instruction stays inside the cache)
    const int t = t0 + lane / LPT;
    const int node = wls[t < nw ? t : nw - 1];
    const int off = 16 * (lane % LPT);  // (a record is WS <= 8 T bytes: the lanes beyond it re-read its start -- words >= W are never used)
    const uint8_t* src = legs + (int64_t)(node - n) * WS + (off < WS ? off : 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(cache + (size_t)t0 * T), 16, 0, 0);
  }
  uint64_t m[4];
  auto load16 = [&](int t0) {  // four rows of tensors from LDS, or (beyond the cache) from memory
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + TPL * u + g;
      m[u] = 0ull;
      if (t < nw && has) {
        if (t < nres) m[u] = cache[t * T + w];
        else m[u] = *reinterpret_cast<const uint64_t*>(legs + (int64_t)((int)wls[t] - n) * WS + 8 * w);
      }
    }
  };
  RngWave rng;
  rng.init(P.mt + r * 624, ring, mti0, mtw0, lane);
  if (nw > 0) rng.fill();
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the legs have landed in LDS)
  __builtin_amdgcn_wave_barrier();
  // Every index belongs to one lane (bit b of its word), so the too-wide counts are kept there, bit-sliced: plane p
  // holds bit p of the lane's 64 counters.  Four tensors are added per step with carry-save adders (three 3:2
  // compressors: ones, twos, one carry of weight four), the carry then ripples through the planes from 2 up --
  // 16 operations per tensor instead of the 24 of a ripple-carry per tensor.
  auto csa = [](uint64_t a, uint64_t b, uint64_t c, uint64_t& sum, uint64_t& carry) {
    const uint64_t x = a ^ b;
    sum = x ^ c;
    carry = (a & b) | (x & c);
  };
  for (int t0 = 0; t0 < nw; t0 += 4 * TPL) {
    load16(t0);
    uint64_t s1, c1, s2, c2, s4, c4;
    csa(pl[0], m[0], m[1], s1, c1);
    csa(s1, m[2], m[3], s2, c2);
    pl[0] = s2;
    csa(pl[1], c1, c2, s4, c4);
    pl[1] = s4;
    uint64_t carry = c4;
#pragma unroll
    for (int p = 2; p < NPL; ++p) {
      const uint64_t tt = pl[p] & carry;
      pl[p] ^= carry;
      carry = tt;
    }
  }
  FWW_T(w3_);
#if defined(TNCO_FWW_STOP) && TNCO_FWW_STOP == 3
  if (lane == 0) rs->mtw = (int)rng.tw;  // (the fill above twisted words of the state array: nothing was consumed)
  leave_to_a();
  return;
#endif
#pragma unroll
  for (int step = T; step <= 32; step <<= 1) {
    uint64_t o[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) o[p] = fws_shflx64(pl[p], step);
    uint64_t c = 0ull;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      const uint64_t a = pl[p], b = o[p];
      pl[p] = a ^ b ^ c;
      c = (a & b) | (c & (a ^ b));
    }
  }
  // The greedy pass (greedy/utils.hpp:62-101).  A tensor that is still too wide after the slices chosen so far has
  // its candidate legs shuffled, sorted (stable) by their too-wide counts and sliced in that order until it fits.
  // With uniform dims every slice narrows it by log2(d), so the NUMBER of picks is known up front (cnt - capw) and
  // only the SET matters: the `need` legs with the largest (count, -shuffled position).
  //   * by count: a radix select over the bit-sliced counters pl[] (the planes this lane holds for its mask word),
  //     most significant plane first -- a few popcounts over the row per plane, whatever the number of picks;
  //   * ties at the threshold count go by shuffled position, and only then is the permutation needed: std::shuffle
  //     swaps a[i] with a[j_i], j_i <= i, for i = 1 .. np - 1 (two per variate, stl_algo.h:3706-3792).  Position x
  //     is filled at step x and refilled by every later step i with j_i = x, which brings element i: the FINAL
  //     occupant of x is the element of the LAST such step -- the highest bit of M[x], the set of those steps -- or,
  //     if no later step touched x, what step x put there: x itself if j_x = x, else the occupant position j_x had
  //     before step x (the same question with steps below x only).  Every lane resolves its own final position
  //     that way: a chain of two or three LDS reads instead of one pass over all the swaps per tensor.
  // The variates are always drawn (the generator's position is part of the state); a re-draw of
  // uniform_int_distribution or the end of the generator's 624 words inside a shuffle runs the sequential shuffle.
  uint64_t ns = 0ull;  // the new slices, word w (the same in the rows of lanes)
  {
#ifdef TNCO_FWW_PROF
    unsigned long long gacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int capw = (int)floor(F.max_width / F.log2d);  // legs a tensor may keep
    int nplanes = 0;  // planes in use: the largest count over all indices is usually far below the number of too-wide tensors
#pragma unroll
    for (int p = 0; p < NPL; ++p)
      if (__any(pl[p] != 0ull)) nplanes = p + 1;
    // The list is scanned in groups of TPL tensors (one per row of lanes).  A group's legs are read ONCE -- from
    // LDS, or, beyond `cap` tensors, from memory with the next group's request in flight -- and after a tensor of
    // the group was sliced the rest of the group is re-tested from the registers.
    auto load_group = [&](int t0) -> uint64_t {
      const int t = t0 + g;
      uint64_t x = 0ull;
      if (t < nw) {
        if (!has) x = 0ull;  // (LDS holds whole records: the words beyond W are padding)
        else if (t < cap) x = cache[t * T + w];
        else x = *reinterpret_cast<const uint64_t*>(legs + (int64_t)((int)wls[t] - n) * WS + 8 * w);
      }
      return x;
    };
    uint64_t mnext = load_group(0);
    int j = 0, gdone = -1;  // the group [j, j + TPL); its rows <= gdone are dealt with
    uint64_t mm = 0ull;
    bool fresh = true;
    while (j < nw) {
      FWW_T(g0_);
      if (fresh) {
        mm = mnext;
        if (j + TPL < nw) mnext = load_group(j + TPL);
        gdone = -1;
        fresh = false;
      }
      const int t = j + g;
      const uint64_t sx = mm & ~ns;
      const uint32_t cnt = gsum<LOGT>((uint32_t)__popcll(sx));
      const bool wide = t < nw && g > gdone && (int)cnt > capw;  // (log2(d) * cnt > max_width: whole numbers, exact in either width type)
      const unsigned long long bal = __ballot(wide);
      if (bal == 0ull) {
        j += TPL;
        fresh = true;
        FWW_T(g1a_);
        FWW_ACC(0, g0_, g1a_);
        continue;
      }
      const int gs = (__ffsll(bal) - 1) >> LOGT;
      gdone = gs;
      FWW_T(g1_);
      FWW_ACC(0, g0_, g1_);
      FWW_CNT(6, 1);
      const uint64_t cand = fws_shfl64(sx, T * gs + w) & ~skip;
      const int cnts = __builtin_amdgcn_readlane((int)cnt, T * gs);
      const uint32_t mine = (uint32_t)__popcll(cand);
      const uint32_t incl = fws_scan<LOGT>(mine, lane);
      const int np = __builtin_amdgcn_readlane((int)incl, T - 1);
      int need = cnts - capw;
      need = need < np ? need : np;  // (fewer candidates than that: all of them, greedy/utils.hpp:86-100 runs out)
      // -- std::shuffle's variates: lane k draws the pair of swaps k
      const int nd = np >> 1, base = np & 1;
      bool fast = true;
      uint32_t p0 = 0, p1 = 0;
      if (np >= 2) {
        fast = np <= 128 && rng.ensure((uint32_t)nd);  // (more candidates, BIG only: the sequential shuffle below)
        if (fast) {
          const uint32_t raw = rng.peek((uint32_t)lane);
          const uint32_t i0 = (uint32_t)(base + 2 * lane);
          const uint32_t range = (i0 + 1u) * (i0 + 2u);  // (lane 0 of an even count: 2 -- d(0, 1), the swap of position 1)
          const uint64_t product = (uint64_t)raw * (uint64_t)range;
          const uint32_t low = (uint32_t)product;
          bool rej = false;
          if (lane < nd && low < range) rej = low < (0u - range) % range;
          if (__any(rej)) {
            fast = false;
          } else {
            // x < (i0 + 1) (i0 + 2) <= 128 * 129: the quotient through a float reciprocal, one step of correction either way
            const uint32_t x = (uint32_t)(product >> 32), dv = i0 + 2u;
            uint32_t q = (uint32_t)((float)x * __frcp_rn((float)dv));
            int rem = (int)x - (int)(q * dv);
            if (rem < 0) { q -= 1u; rem += (int)dv; }
            else if (rem >= (int)dv) { q += 1u; rem -= (int)dv; }
            p0 = q;
            p1 = (uint32_t)rem;
            rng.advance((uint32_t)nd);
          }
        }
      }
      FWW_T(g2_);
      FWW_ACC(1, g1_, g2_);
      // -- by count
      uint64_t alive = cand, taken = 0ull;
#pragma unroll
      for (int p = NPL - 1; p >= 0; --p) {
        if (p < nplanes) {
          const uint64_t hi = alive & pl[p];
          const int c = __builtin_amdgcn_readfirstlane((int)gsum<LOGT>((uint32_t)__popcll(hi)));
          if (c >= need) {
            alive = hi;
          } else {
            taken |= hi;
            need -= c;
            alive &= ~pl[p];
          }
        }
      }
      const int calive = __builtin_amdgcn_readfirstlane((int)gsum<LOGT>((uint32_t)__popcll(alive)));
      FWW_T(g3_);
      FWW_ACC(2, g2_, g3_);
      FWW_CNT(8, np);
      if (need >= calive && fast) {  // the whole tie group (need == calive; need == 0 == np: nothing)
        ns |= taken | alive;
        continue;
      }
      FWW_CNT(7, 1);
      // -- by shuffled position: the candidates in ascending order (row g lists bits T g .. T g + T - 1 of its word)
      {
        const uint64_t part = (T == 64) ? cand : ((cand >> (T * g)) & ((1ull << (T & 63)) - 1ull));
        uint32_t o = (incl - mine) + ((T == 64) ? 0u : (uint32_t)__popcll(cand & ((1ull << ((T * g) & 63)) - 1ull)));
        uint64_t x = part;
        while (x) {
          const int b = __ffsll((unsigned long long)x) - 1;
          pos[o++] = (uint16_t)(w * 64 + ((T == 64) ? 0 : T * g) + b);
          x &= x - 1;
        }
      }
      FWW_T(g4_);
      FWW_ACC(3, g3_, g4_);
      int e0 = lane, e1 = lane + 64;  // the element (candidate number) at final position lane, lane + 64
      if (fast) {
        const bool two = np > 64;
        Mlo[lane] = 0ull;
        Mhi[lane] = 0ull;
        if (two) Mhi[lane + 64] = 0ull;
        if (lane == 0) jL[0] = 0;
        __builtin_amdgcn_wave_barrier();
        if (lane < nd) {
          const uint32_t i0 = (uint32_t)(base + 2 * lane), i1 = i0 + 1u;
          jL[i0] = (uint8_t)p0;
          jL[i1] = (uint8_t)p1;
          if (p0 < i0) __hip_atomic_fetch_or(i0 < 64u ? &Mlo[p0] : &Mhi[p0], 1ull << (i0 & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (p1 < i1) __hip_atomic_fetch_or(i1 < 64u ? &Mlo[p1] : &Mhi[p1], 1ull << (i1 & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_wave_barrier();
        TNCO_LDS volatile uint64_t* MloV = (TNCO_LDS volatile uint64_t*)Mlo;
        TNCO_LDS volatile uint64_t* MhiV = (TNCO_LDS volatile uint64_t*)Mhi;
        auto occupant = [&](int f) -> int {
          int x = f, s = 128;
          for (;;) {  // (x falls with every round)
            uint64_t mh = two ? MhiV[x] : 0ull, ml = x < 64 ? MloV[x] : 0ull;
            if (s <= 64) {
              mh = 0ull;
              ml &= (s == 64) ? ~0ull : ((1ull << s) - 1ull);
            } else if (s < 128) {
              mh &= (1ull << (s - 64)) - 1ull;
            }
            if (mh) return 127 - __clzll((long long)mh);
            if (ml) return 63 - __clzll((long long)ml);
            const int jx = jL[x];
            if (jx == x) return x;
            s = x;
            x = jx;
          }
        };
        if (lane < np) e0 = occupant(lane);
        if (lane + 64 < np) e1 = occupant(lane + 64);
      } else if (np >= 2) {
        __builtin_amdgcn_wave_barrier();
        fw_shuffle_lds<6>(rng, pos, np, lane == 0);  // (in place: position f holds its leg)
      }
      __builtin_amdgcn_wave_barrier();
      FWW_T(g5_);
      FWW_ACC(4, g4_, g5_);
      // the first `need` members of the tie group in shuffled order, 64 final positions at a time
      const bool shuffled_in_place = !fast;
      if (lane < T) Mhi[lane] = 0ull;
      __builtin_amdgcn_wave_barrier();
      const unsigned long long below = (1ull << lane) - 1ull;
      int cum = 0;
      for (int q = 0; q * 64 < np && cum < need; ++q) {
        const int f = lane + 64 * q;
        const int leg = f < np ? (int)pos[shuffled_in_place ? f : (q == 0 ? e0 : e1)] : 0;
        const uint64_t aw = fws_shfl64(alive, leg >> 6);
        const bool in = f < np && ((aw >> (leg & 63)) & 1ull);
        const unsigned long long bq = __ballot(in);
        if (in && cum + (int)__popcll(bq & below) < need)
          __hip_atomic_fetch_or(&Mhi[leg >> 6], 1ull << (leg & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        cum += (int)__popcll(bq);
      }
      __builtin_amdgcn_wave_barrier();
      ns |= taken | ((TNCO_LDS volatile uint64_t*)Mhi)[w];
      __builtin_amdgcn_wave_barrier();
      FWW_T(g6_);
      FWW_ACC(5, g5_, g6_);
    }
#ifdef TNCO_FWW_PROF
    if (lane == 0) {
      for (int i = 0; i < 9; ++i) atomicAdd(&g_fww_prof[12 + i], gacc_[i]);
      atomicAdd(&g_fww_prof[21], (unsigned long long)nw);
    }
#endif
  }
  __builtin_amdgcn_wave_barrier();
  build_table();  // (get_slices used its memory)
  __builtin_amdgcn_wave_barrier();
  FWW_T(w4_);
  // ---- the indices that changed, with the starts of their paths (the region is the re-pricing's now).  Their
  // holders are requested first -- up to four per lane in one flight -- and only then the stores of this step are
  // issued: a load behind a store waits for the store's acknowledgement too.
  uint32_t* chg = reinterpret_cast<uint32_t*>(F.delta_scr + r * 64);  // (word 0: the count, for tnco_hip_diag_reslice_info)
  int nd;
  {
    uint64_t ch = g == 0 ? (ns ^ old) : 0ull;
    const uint32_t mine = (uint32_t)__popcll(ch);
    const uint32_t incl = fws_scan<LOGT>(mine, lane);
    nd = __builtin_amdgcn_readlane((int)incl, T - 1);
    bool unsup = nd > MAXD;
    int bits[4];
    int2 hold[4];
    uint4 holdn[HYPER ? 4 : 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bits[q] = ch ? __ffsll((unsigned long long)ch) - 1 : -1;
      ch &= ch - 1;  // (0 stays 0)
      hold[q] = make_int2(-1, -1);
      if constexpr (HYPER) {
        holdn[q] = make_uint4(0, 0, 0, 0);
        if (bits[q] >= 0 && !unsup) holdn[q] = *reinterpret_cast<const uint4*>(F.holdern + 8 * (w * 64 + bits[q]));
      } else {
        if (bits[q] >= 0 && !unsup) hold[q] = *reinterpret_cast<const int2*>(F.holder2 + 2 * (w * 64 + bits[q]));
      }
    }
    {  // get_slices is done: the generator's position, the proposal (fw_reslice_b_kernel reads it if the re-pricing gives up)
      int mti, mtw;
      rng.finish(mti, mtw);
      if (lane == 0) {
        rs->mti = mti;
        rs->mtw = mtw;
        F.nwide[r] = -3;  // fw_reslice_a_kernel skips this replica
      }
      uint64_t* prop = reinterpret_cast<uint64_t*>(const_cast<int16_t*>(sc.pos));
      if (g == 0 && w < LK) prop[w] = ns;
    }
#if defined(TNCO_FWW_STOP) && TNCO_FWW_STOP == 4
    unsup = true;  // (the full rebuild takes the proposal from here)
#endif
    // the parents of the leaves from the node table, not from the replica's (cold) parent array: one round trip less
    TNCO_LDS volatile uint16_t* lparL = (TNCO_LDS volatile uint16_t*)U;  // [n] (the path masks' memory: cleared below)
    for (int i = lane; i < ni; i += GW) {
      const uint32_t wq = lo[i];
      const int l = (int)(wq & 0xFFFFu), rr = (int)(wq >> 16);
      if (l < n) lparL[l] = (uint16_t)(n + i);
      if (rr < n) lparL[rr] = (uint16_t)(n + i);
    }
    if (!unsup) {
      uint32_t off = incl - mine;
      auto entry = [&](int bit, int2 t12) {
        if (t12.x < 0) { unsup = true; return; }
        const int s1 = lparL[t12.x], s2 = t12.y < 0 ? 0xFFFF : (int)lparL[t12.y];
        chgl[off] = (uint32_t)s1 | ((uint32_t)s2 << 16);
        pm[off] = (uint8_t)((ns >> bit) & 1ull);
        ++off;
      };
      // (HYPER) the index's entry of FwParams::holdern: count | open << 15, then the holders
      auto entry_n = [&](int bit, uint4 hv) {
        const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
        const int m = (int)(hw[0] & 0x7FFFu), open = (int)((hw[0] >> 15) & 1u);
        if (m < 1 || m > FWH_MAXH) { unsup = true; return; }
        pcnt[off] = (uint8_t)(m | (open << 7));
        for (int j = 0; j < m; ++j) {
          const int t = (int)((hw[(j + 1) >> 1] >> (((j + 1) & 1) * 16)) & 0xFFFFu);
          pleaf[off * 8 + j] = (uint16_t)t;
          pstart[off * 8 + j] = lparL[t];
        }
        pm[off] = (uint8_t)((ns >> bit) & 1ull);
        ++off;
      };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (bits[q] >= 0 && !unsup) {
          if constexpr (HYPER) entry_n(bits[q], holdn[q]); else entry(bits[q], hold[q]);
        }
      }
      while (ch && !unsup) {  // (more than four changed indices in one mask word)
        const int bit = __ffsll((unsigned long long)ch) - 1;
        ch &= ch - 1;
        if constexpr (HYPER) entry_n(bit, *reinterpret_cast<const uint4*>(F.holdern + 8 * (w * 64 + bit)));
        else entry(bit, *reinterpret_cast<const int2*>(F.holder2 + 2 * (w * 64 + bit)));
      }
    }
    unsup = __any(unsup);
    if (lane == 0) chg[0] = unsup ? 0xFFFFFFFFu : (uint32_t)nd;
    if (unsup) {  // more than FWT_MAXD indices, or an index held otherwise: the full rebuild
      if (lane == 0) {
        atomicAdd(F.slowstat, 1ull);
        atomicAdd(F.slowstat + 2, 1ull);
      }
      return;
    }
  }
  FWW_T(w5_);
#if defined(TNCO_FWW_STOP) && TNCO_FWW_STOP == 5
  return;
#endif
  // ---- the re-priced costs: path masks cleared, arrival counters = internal children
  uint32_t startmask = 0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * GW + lane;
    if (i < ni) {
      const uint32_t h = hiv[i];
      const int e = (int)((h >> 16) & 0x7FFu);
      bad = bad || e <= 0 || e >= 2047;
      onv[2 * i] = 0;
      onv[2 * i + 1] = 0;
      if ((h >> 27) == 0u) startmask |= 1u << j;
    }
  }
  const int log2d = P.log2d;
  for (int pass = 0; pass * IPP < nd || pass == 0; ++pass) {
    uint32_t plus, minus;  // which of this pass's 32 indices join / leave the slices
    {
      const int idx = IPP * pass + lane;
      const bool mine_ = lane < IPP && idx < nd;
      const bool joins = mine_ && pm[mine_ ? idx : 0] != 0;
      plus = (uint32_t)__ballot(joins);
      minus = (uint32_t)__ballot(mine_ && !joins);
    }
    const int dbase = __popc(plus) - __popc(minus);
    if (pass) {
      for (int i = lane; i < ni; i += GW) { onv[2 * i] = 0; onv[2 * i + 1] = 0; }
    }
    if constexpr (!HYPER) {
      const int k = lane >> 1, which = lane & 1;
      const int idx = IPP * pass + k;  // this lane's changed index
      const uint32_t e = idx < nd ? chgl[idx] : 0xFFFFFFFFu;
      const int st = which ? (int)(e >> 16) : (int)(e & 0xFFFFu);
      int x = (idx < nd && st != 0xFFFF) ? st : -1;  // the path of a holder starts at its parent
      for (int guard = 0; __any(x >= 0); ++guard) {
        if (guard > ni) { bad = true; break; }  // (cannot happen in a tree: never spin on corrupt links)
        if (x >= 0) {
          __hip_atomic_fetch_or(&on[2 * (x - n) + which], 1u << k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const int pp = (int)(hiv[x - n] & 0xFFFFu);
          x = pp == 0xFFFF ? -1 : pp;
        }
      }
    } else {
      // An index d held by m tensors is a leg of a subtree exactly when the subtree holds some but not all of them
      // (all of them: only if d is open -- an output, or held by one tensor).  So d is among the legs of node x's
      // children when x has a holder below and is not STRICTLY ABOVE the node where all the holders' paths have met.
      //   any[x] (on[2 x]):      a holder of index k below x -- every (index, holder) walks up from the holder's
      //                          parent; a walk that meets a node already marked for its index stops, the one
      //                          that marked it goes on to the root;
      //   above[x] (on[2 x + 1]): x strictly above the meeting point -- one lane per closed index walks DOWN from
      //                          the root while exactly one child has holders below.
      for (int rd = 0; rd < 4; ++rd) {
        const int sidx = rd * 64 + lane, k = sidx >> 3, jj = sidx & 7;
        const int idx = IPP * pass + k;
        int x = -1;
        if (idx < nd && jj < (int)(pcnt[idx] & 0x7Fu)) x = pstart[idx * 8 + jj];
        for (int guard = 0; __any(x >= 0); ++guard) {
          if (guard > ni) { bad = true; break; }
          if (x >= 0) {
            const uint32_t was = __hip_atomic_fetch_or(&on[2 * (x - n)], 1u << k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int pp = (int)(hiv[x - n] & 0xFFFFu);
            x = (((was >> k) & 1u) || pp == 0xFFFF) ? -1 : pp;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      {
        const int k = lane, idx = IPP * pass + k;
        const int c = (k < IPP && idx < nd) ? (int)pcnt[idx] : 0x80;
        const int m = c & 0x7F;
        int x = (c & 0x80) ? -1 : N - 1;  // (open indices: a leg all the way up)
        for (int guard = 0; __any(x >= 0); ++guard) {
          if (guard > ni) { bad = true; break; }
          if (x >= 0) {
            const uint32_t wq = lo[x - n];
            const int l = (int)(wq & 0xFFFFu), rr = (int)(wq >> 16);
            bool cl, cr;
            if (l >= n) {
              cl = (onv[2 * (l - n)] >> k) & 1u;
            } else {
              cl = false;
              for (int j = 0; j < m; ++j) cl = cl || (int)pleaf[idx * 8 + j] == l;
            }
            if (rr >= n) {
              cr = (onv[2 * (rr - n)] >> k) & 1u;
            } else {
              cr = false;
              for (int j = 0; j < m; ++j) cr = cr || (int)pleaf[idx * 8 + j] == rr;
            }
            if (cl == cr) {  // both: the paths meet here (neither: cannot happen)
              x = -1;
            } else {
              __hip_atomic_fetch_or(&on[2 * (x - n) + 1], 1u << k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              x = cl ? l : rr;
              if (x < n) x = -1;  // (cannot happen: a closed index has two holders at least)
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    for (int i = lane; i < ni; i += GW) {
      const uint32_t wq = lo[i], h = hiv[i];
      const int l = (int)(wq & 0xFFFFu), rr = (int)(wq >> 16);
      const uint32_t a = onv[2 * i], b = onv[2 * i + 1];
      uint32_t in_u;
      if constexpr (HYPER) {
        in_u = a & ~b;
      } else {
        const int il = l >= n ? l - n : i, ir = rr >= n ? rr - n : i;
        const uint32_t bl = onv[2 * il] & onv[2 * il + 1], br = onv[2 * ir] & onv[2 * ir + 1];
        const uint32_t both_below = (l >= n ? bl : 0u) | (rr >= n ? br : 0u);
        in_u = (a ^ b) | (a & b & ~both_below);
      }
      const int dex = dbase - __popc(in_u & plus) + __popc(in_u & minus);
      const int ne = (int)((h >> 16) & 0x7FFu) + log2d * dex;
      bad = bad || ne <= 0 || ne >= 2047;  // (also between the passes: the full rebuild decides then)
      hiv[i] = (h & 0xF800FFFFu) | ((uint32_t)(ne & 0x7FF) << 16);
    }
  }
  FWW_T(w6_);
#if defined(TNCO_FWW_STOP) && TNCO_FWW_STOP == 6
  return;
#endif
  // ---- is the proposal worth its partial sums?  The rebuilt cache is kept only if its root sum is below the current one
  // (greedy/optimizer.hpp:371-374) -- about half of the proposals are not.  The new costs are powers of two >= 2^emin,
  // so every sum over them is a multiple of 2^emin, and one below 2^(emin + 53) is a double whatever the order it was
  // added up in: then the plain sum over the wavefront IS the root sum the reference's bottom-up order gives, bit for
  // bit, and a proposal it rejects needs nothing more.  (A bit to spare in the test; a sum that might be inexact takes
  // the long way below.)
#ifndef TNCO_FWW_NO_REJECT  // (A/B builds)
  if (!__any(bad) && fww_rejected_exactly(hiv, ni, lane, cur)) {
    if (lane == 0) F.fastflag[r] = 1;  // fw_reslice_b_kernel only closes the sweep: slices and caches stay
    return;
  }
#endif
  // ---- children before parents: every lane starts at its nodes with two leaf children; the second child to
  // arrive at a parent goes on with it (the arrival returns the parent's record)
  int p = -1;
  uint32_t phi = 0, plo = 0;
  for (int guard = 0;; ++guard) {
    if (guard > 2 * ni + 64) { bad = true; break; }  // (cannot happen in a tree)
    if (p < 0 && startmask) {
      const int j = __ffs(startmask) - 1;
      startmask &= startmask - 1;
      p = j * GW + lane;
      phi = hiv[p];
      plo = lo[p];
    }
    if (!__any(p >= 0)) break;
    if (p >= 0) {
      const uint32_t wq = plo;
      const int l = (int)(wq & 0xFFFFu), rr = (int)(wq >> 16);
      const bool li = l >= n, ri = rr >= n;
      const double pl0 = Pn[li ? l - n : p], pr0 = Pn[ri ? rr - n : p];
      const double pL = li ? pl0 : 0.0, pR = ri ? pr0 : 0.0;
      const double c = __longlong_as_double((long long)((uint64_t)((phi >> 16) & 0x7FFu) << 52));
      Pn[p] = (c + pL) + pR;  // (the association order of finite_width/utils.hpp:36-47)
      if (p == ni - 1) {
        p = -1;  // the root
      } else {
        const int q = (int)(phi & 0xFFFFu) - n;
        const uint32_t oldc = __hip_atomic_fetch_add(&hi[q], 0u - (1u << 27), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t qlo = lo[q];
        if ((oldc >> 27) == 1u) { p = q; phi = oldc; plo = qlo; } else { p = -1; }
      }
    }
  }
  if (__any(bad)) {  // (a cost outside the powers of two of a double: the full rebuild decides)
    if (lane == 0) {
      atomicAdd(F.slowstat, 1ull);
      atomicAdd(F.slowstat + 3, 1ull);
    }
    return;
  }
  FWW_T(w7_);
  if (lane == ((ni - 1) & (GW - 1))) misc[0] = (Pn[ni - 1] < cur) ? 1u : 0u;  // greedy/optimizer.hpp:371-374
  if (misc[0]) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int i = j * GW + lane;
      if (i < ni) {
        const uint32_t wq = lo[i], h = hiv[i];
        const double c = __longlong_as_double((long long)((uint64_t)((h >> 16) & 0x7FFu) << 52)), pp = Pn[i];
        const int par = (int)(h & 0xFFFFu);
        int4* d = reinterpret_cast<int4*>(hb + (int64_t)i * P.BS);
        d[0] = make_int4((int)(wq & 0xFFFFu), (int)(wq >> 16), par == 0xFFFF ? -1 : par, iw[j]);
        d[1] = make_int4(__double2loint(c), __double2hiint(c), __double2loint(pp), __double2hiint(pp));
      }
    }
    if (lane < LK) sl[lane] = lane < W ? ns : 0ull;  // (lanes 0..T-1 hold word `lane` of the proposal)
  }
  if (lane == 0) F.fastflag[r] = 1;
#ifdef TNCO_FWW_PROF
  if (lane == 0) {
    const unsigned long long w8_ = __builtin_amdgcn_s_memtime();
    atomicAdd(&g_fww_prof[0], w1_ - w0_); atomicAdd(&g_fww_prof[1], w2_ - w1_); atomicAdd(&g_fww_prof[2], w3_ - w2_);
    atomicAdd(&g_fww_prof[3], w4_ - w3_); atomicAdd(&g_fww_prof[4], w5_ - w4_); atomicAdd(&g_fww_prof[5], w6_ - w5_);
    atomicAdd(&g_fww_prof[6], w7_ - w6_); atomicAdd(&g_fww_prof[7], w8_ - w7_); atomicAdd(&g_fww_prof[8], 1ull);
  }
#endif
}
