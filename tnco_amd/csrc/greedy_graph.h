// greedy_graph.h -- the same greedy over a MULTIGRAPH held in LDS (networks without hyper-indices; round 4): greedy_graph_kernel and its host tables
// (part of greedy_device.hip, the only file that includes it: everything lives in its unnamed namespace)
#pragma once
#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

#include "greedy_key.h"
#include "greedy_wave.h"

namespace tnco {
namespace {

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
//     and rec[] (for a dead id: its parent) leads from there to the live tensor that holds it now -- a
//     union-find whose paths the look-ups shorten; the entries of u and v resolve in parallel, one per lane;
//   * equal neighbours are joined through a byte per id (mark: the lane that speaks for the id);
//   * lists live in ONE arena of L + 256 entries (L = entries of the inputs; the live entries only ever
//     get fewer): when it is full the live lists move to its front, in place;
//   * the queue is greedy_kernel's (64-bit keys, cell c owned by lane c % 64, a push goes into the cell
//     just popped) but IN REGISTERS, ROWS cells per lane; the cost keys of the initial candidates do not
//     depend on the shuffle and come from the host.
// 512 tensors / 768 indices: 12 KB of LDS per tree (12 trees per CU: VALU-bound there), no memory traffic but the shuffled
// order read and the links written out.
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
  if (L + 256 > 0x7FFF) return false;  // (a list's offset is 15 bits of its tensor's record)
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


}  // namespace
}  // namespace tnco
