// greedy_set.h -- opt_einsum's greedy over index SETS (any network; round 2): greedy_kernel
// (part of greedy_device.hip, the only file that includes it: everything lives in its unnamed namespace)
#pragma once
#include "greedy_key.h"
#include "greedy_wave.h"

namespace tnco {
namespace {

// ---------------------------------------------------------------------------------------------
// the greedy path finder, one wavefront per tree
// ---------------------------------------------------------------------------------------------
struct GreedyParams {
  int32_t n, I, W, NW, SMAX, Q, TS;  // TS: table size (a power of two)
  int32_t M1;                         // (queue in registers) candidates a dim can start with: max holders - 1
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

// ROWS > 0: the candidate queue in REGISTERS, ROWS cells per lane (as in greedy_graph_kernel: 6 KB of LDS less at
// 768 candidates, 9 -> 15 trees per CU), cell (row, lane) = candidate `row % M1` of dim 64 (row / M1) + lane -- the
// initial candidates of a dim stay in its lane, a push goes into the cell just popped.  ROWS = 0: the queue in LDS
// (networks whose dims x candidates per dim need more than 16 rows: beyond that the registers cost more trees per CU
// than the LDS did).
// (three wavefronts per SIMD asked for: the 16-row instantiation sits at the edge of 168 VGPRs, and twelve trees per CU
//  against eight is a third of this latency-bound kernel's throughput)
template <int ROWS>
__global__ __launch_bounds__(64, ROWS == 24 ? 2 : 3) void greedy_kernel(const GreedyParams p) {
  extern __shared__ uint64_t lds_raw[];
  const int lane = threadIdx.x;
  const int n = p.n, I = p.I, W = p.W, NW = p.NW, SMAX = p.SMAX, Q = p.Q, TS = p.TS;
  const int QC = ROWS > 0 ? 0 : (Q + 63) & ~63;  // queue cells in LDS (a multiple of 64)
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
  uint64_t qc[ROWS > 0 ? ROWS : 1];
  for (int64_t r = g; r < p.R; r += gridDim.x) {
    const uint16_t* perm = p.perm + r * (int64_t)n;
    int status = 0;
    // ---- clear ----
    for (int c = lane; c < QC; c += 64) hk[c] = KMAX;
#pragma unroll
    for (int q = 0; q < (ROWS > 0 ? ROWS : 1); ++q) qc[q] = KMAX;
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
      int base = 0;
      if constexpr (ROWS == 0) {
        base = count + (int)wscan_excl((uint32_t)mine, lane);
        count += (int)wsum((uint32_t)mine);
        if (count > Q) {
          status = 5;
          break;
        }
      } else if (__any(mine > p.M1) || ((d0 >> 6) + 1) * p.M1 > ROWS) {
        status = 5;
        break;
      }
      uint64_t bk[GREEDY_MAXH - 1];
#pragma unroll
      for (int i = 0; i + 1 < GREEDY_MAXH; ++i) bk[i] = KMAX;
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
          const int seq = ROWS > 0 ? ((d0 >> 6) * p.M1 + i) * 64 + lane : base + i;
          uint64_t* row = arena + (size_t)seq * W;
          for (int x = 0; x < W; ++x) {
            const uint64_t a = keys[(size_t)sl[i] * W + x], b = keys[(size_t)sj * W + x];
            const uint64_t either = a | b, two = a & b, one = either & ~two;
            row[x] = (either & bc[W + x]) | (two & bc[3 * W + x]) | (one & bc[2 * W + x]);
          }
          if constexpr (ROWS == 0) hk[seq] = bestk;
          bk[i] = bestk;
        }
      }
      if constexpr (ROWS > 0) {  // this dim's candidates into this lane's cells, rows (d0 / 64) M1 + i
        const int row0 = (d0 >> 6) * p.M1;
#pragma unroll
        for (int i = 0; i + 1 < GREEDY_MAXH; ++i)
#pragma unroll
          for (int q = 0; q < ROWS; ++q)
            if (q == row0 + i && i + 1 < m) qc[q] = bk[i];
      }
    }
    __syncthreads();
    // every lane: the minimum of its cells (four reads in flight)
    uint64_t lkey = KMAX;
    int lrow = 0;
    auto rescan = [&]() {
      lkey = KMAX;
      lrow = 0;
      if constexpr (ROWS > 0) {
#pragma unroll
        for (int q = 0; q < ROWS; ++q)
          if (qc[q] < lkey) {
            lkey = qc[q];
            lrow = q;
          }
        return;
      }
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
    while (status == 0 && n_alive > 1) {  // (one tensor left: every candidate still queued is obsolete)
      const uint64_t best = wmin64(lkey);
      if (best == KMAX) break;
      const int wl = __ffsll((unsigned long long)__ballot(lkey == best)) - 1;
      const int qrow = uni(__shfl(lrow, wl));
      const int seq = qrow * 64 + wl;
      if constexpr (ROWS > 0) {
#pragma unroll
        for (int q = 0; q < ROWS; ++q)
          if (q == qrow && lane == wl) qc[q] = KMAX;
        rescan();
      } else if (lane == wl) {
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
      // the list of the result's neighbours (round 5: made BEFORE their rows are updated, so that the update is one
      // neighbour per lane -- one memory round trip -- instead of a lane's neighbours one after the other; the members of
      // `un`, whose rows change, come first, the neighbours a live equal set brought along behind them)
      int total = (int)wsum((uint32_t)__popcll(un));
      if (total > GREEDY_LCAP) {
        status = 7;
        break;
      }
      {
        int at = (int)wscan_excl((uint32_t)__popcll(un), lane);
        uint64_t u = un;
        while (u) {
          lst[at++] = (uint16_t)(lane * 64 + __ffsll((unsigned long long)u) - 1);
          u &= u - 1;
        }
      }
      __syncthreads();
      {
        const int w1 = s1 >> 6, w2 = s2 >> 6, w12 = s12 >> 6;
        const uint64_t m1 = 1ull << (s1 & 63), m2 = 1ull << (s2 & 63), m12 = 1ull << (s12 & 63);
        for (int j = lane; j < total; j += 64) {
          uint64_t* ry = nbr + (size_t)lst[j] * NW;
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
      if (merged) {  // (rare: the neighbours the live equal set brought along join the list)
        const uint64_t extra = n12m & ~un;
        const int at0 = total + (int)wscan_excl((uint32_t)__popcll(extra), lane);
        total += (int)wsum((uint32_t)__popcll(extra));
        if (total > GREEDY_LCAP) {
          status = 7;
          break;
        }
        int at = at0;
        uint64_t u = extra;
        while (u) {
          lst[at++] = (uint16_t)(lane * 64 + __ffsll((unsigned long long)u) - 1);
          u &= u - 1;
        }
        __syncthreads();
      }
      GP_T(7);
      // push the cheapest (k12, neighbour)
      if (total > 0) {
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
        const int wlane = __ffsll((unsigned long long)__ballot(bestk == wk)) - 1;
        const int sbest = uni(__shfl(bests, wlane));
        const uint64_t bb = inw ? keys[(size_t)sbest * W + lane] : 0;
        const uint64_t res = (bb & qmask) | (~bb & pmask);
        // into the cell (and the arena row) of the candidate popped in this iteration: the queue never
        // holds more than the initial candidates
        if (inw) arena[(size_t)seq * W + lane] = res;
        if constexpr (ROWS > 0) {
#pragma unroll
          for (int q = 0; q < ROWS; ++q)
            if (q == qrow && lane == wl) qc[q] = wk;
        }
        if (lane == wl) {
          if constexpr (ROWS == 0) hk[seq] = wk;
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


}  // namespace
}  // namespace tnco
