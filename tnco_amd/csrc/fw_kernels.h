// fw_kernels.h -- memory-constrained twin of the sweep: finite_width::greedy::Optimizer
// (include/tnco/optimize/finite_width/greedy/optimizer.hpp:43-460 of the reference) on gfx950.
//
// Same replica-per-lane-group mapping and node-block layout as sa_sweep.h.  On top of it, per
// replica: the sliced-index mask `slices` / `min_slices` (W words each), the un-sliced width of
// every internal node (WidthCache, finite_width/utils.hpp:49-76: float32 in the `pad` word of the
// node header, or a float64 array of its own), and scratch for the greedy re-slice.
//
// The moves of a run of sweeps are the staged sweep kernel (sa_sweep.h, FW = true); fw_move_kernel here is the
// unstaged version kept for max_number_new_slices > 0.  The end of a re-slicing sweep (greedy/utils.hpp:21-125
// + the CostCache rebuild of greedy/optimizer.hpp:359-376) comes in two forms:
//   * the general one: fw_walk2_kernel (two lanes per replica, from both ends of the post-order) leaves
//     sequential lists (internal nodes in post-order with their links; the too-wide tensors in post-order), then
//     fw_reslice_kernel iterates over them -- the k-th entry of all 16 replicas of a wavefront together, the next
//     entries' loads in flight; counters, candidate lists, shuffles and picks in registers / LDS; the cache
//     rebuilt from the leg masks;
//   * when every cost is a power of two (FwParams::fast_ok): fw_wave_kernel, one wavefront per replica, no walk --
//     the too-wide tensors ordered by their root paths | get_slices | the cache RE-PRICED from the old costs; then
//     fw_reslice_a_kernel (get_slices of its stragglers) | fw_reslice_b_kernel (their full rebuild, end of sweep).
// Covered: SimpleCostModel and SimpleSparseIndsCostModel (finite_width/cost_model/simple.hpp,
// simple_sparse_inds.hpp), uniform and per-index dims, width_type float32 / float64, and the
// max_number_new_slices > 0 branch (greedy/optimizer.hpp:226-321).
#pragma once
#include "sa_sweep.h"

namespace tnco {

template <int LOG2L, int K, bool HYPER>
__device__ __forceinline__ bool fw_test(const View<LOG2L, K, HYPER>& v, const Mask<K>& m, int pos) {
  bool mine = false;
#pragma unroll
  for (int k = 0; k < K; ++k) mine |= ((pos >> 6) == v.widx(k)) && ((m.w[k] >> (pos & 63)) & 1ull);
  return gany<LOG2L>(mine);
}
template <int LOG2L, int K, bool HYPER>
__device__ __forceinline__ void fw_flip(const View<LOG2L, K, HYPER>& v, Mask<K>& m, int pos) {
#pragma unroll
  for (int k = 0; k < K; ++k)
    if ((pos >> 6) == v.widx(k)) m.w[k] ^= 1ull << (pos & 63);
}

// get_delta_width: change of the width when position `pos` is toggled in `m`
// (simple.hpp:59-76, simple_sparse_inds.hpp:54-82)
template <int LOG2L, int K, bool HYPER>
__device__ __forceinline__ double fw_delta_width(const Params& P, const FwParams& F, const View<LOG2L, K, HYPER>& v,
                                                 const Mask<K>& m, int pos, int gbase) {
  if (P.sparse != nullptr) {
    const Mask<K> s = fw_sparse_mask<LOG2L, K>(P, v.lig);
    if (fw_test<LOG2L, K, HYPER>(v, s, pos)) {
      const Mask<K> cur = mand<K>(m, s);
      Mask<K> nxt = cur;
      fw_flip<LOG2L, K, HYPER>(v, nxt, pos);
      const double a = fw_width_simple<LOG2L, K>(F, nxt, gbase), b = fw_width_simple<LOG2L, K>(F, cur, gbase);
      const double ma = (a < F.log2np) ? a : fw_wr(F, F.log2np);
      const double mb = (b < F.log2np) ? b : fw_wr(F, F.log2np);
      return fw_wr(F, ma - mb);
    }
  }
  const int test = fw_test<LOG2L, K, HYPER>(v, m, pos) ? 1 : 0;
  const double l2 = F.log2dims ? F.log2dims[pos] : F.log2d;
  return fw_wr(F, (double)(1 - 2 * test) * l2);
}

// DimsCache::log2_dims, finite_width/utils.hpp:91-108
__device__ __forceinline__ double fw_log2dim(const FwParams& F, int pos) {
  return fw_wr(F, F.log2dims ? F.log2dims[pos] : F.log2d);
}

// /usr/include/c++/11/bits/uniform_int_dist.h:246-321, 32-bit generator, range < 2^32
template <int LOG2L, typename RNG>
__device__ __forceinline__ uint32_t fw_uniform_int(RNG& rng, uint32_t hi) {
  const uint32_t uerange = hi + 1u;
  uint64_t product = (uint64_t)rng.next_sync() * (uint64_t)uerange;
  uint32_t low = (uint32_t)product;
  if (low < uerange) {
    const uint32_t threshold = (0u - uerange) % uerange;
    while (low < threshold) {
      product = (uint64_t)rng.next_sync() * (uint64_t)uerange;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32);
}

// std::shuffle of /usr/include/c++/11/bits/stl_algo.h:3706-3792 on a scratch array (n < 65536: two
// swap positions per variate).  Executed redundantly by every lane of the group; lane 0 writes.
template <int LOG2L, typename RNG, typename A>
__device__ __forceinline__ void fw_shuffle(RNG& rng, A a, int n, bool lane0) {
  if (n <= 1) return;
  int i = 1;
  auto swp = [&](int x, int y) {
    const int ax = a[x], ay = a[y];
    if (lane0) { a[x] = ay; a[y] = ax; }
  };
  if ((n % 2) == 0) {
    const uint32_t j = fw_uniform_int<LOG2L>(rng, 1u);
    swp(i, (int)j);
    ++i;
  }
  while (i != n) {
    const uint32_t swap_range = (uint32_t)i + 1u;
    const uint32_t x = fw_uniform_int<LOG2L>(rng, swap_range * (swap_range + 1u) - 1u);
    const uint32_t p0 = x / (swap_range + 1u), p1 = x % (swap_range + 1u);
    swp(i, (int)p0);
    ++i;
    swp(i, (int)p1);
    ++i;
  }
}

// uniform_int_distribution on a raw output already drawn (the rare re-draws come from the ring)
template <int LOG2L, typename RNG>
__device__ __forceinline__ uint32_t fw_uniform_int_from(RNG& rng, uint32_t raw, uint32_t hi) {
  const uint32_t uerange = hi + 1u;
  uint64_t product = (uint64_t)raw * (uint64_t)uerange;
  uint32_t low = (uint32_t)product;
  if (low < uerange) {
    const uint32_t threshold = (0u - uerange) % uerange;
    while (low < threshold) {
      product = (uint64_t)rng.next_sync() * (uint64_t)uerange;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32);
}

// The same std::shuffle on an LDS array, arranged for latency: the loads of the generator's next
// block are issued as soon as there is room for it; the raw output of the NEXT pair of swaps is
// read from the ring before this pair's LDS traffic; the four array reads of a pair (a[i], a[i + 1]
// early, a[p0], a[p1] together once the variate is known) are independent, and the four writes in
// program order give exactly the two swaps (the second swap sees the first through the selects).
template <int LOG2L, typename RNG, typename A = lds_vi32*>
__device__ __forceinline__ void fw_shuffle_lds(RNG& rng, A a, int n, bool lane0) {
  if (n <= 1) return;
  // The generator: the sixteen replicas of a wavefront are at sixteen different phases of their
  // 16-word blocks, so "refill when there is room" makes the wavefront run the (long) request and
  // produce code at nearly every draw.  Instead every replica fills its ring now and, the draws
  // being one per iteration for all of them, all request a block at the same iterations (every
  // 16th) and produce it eight iterations later, its loads long landed.  (next_sync() still fetches
  // on demand should a ring run dry: re-draws of uniform_int, rings smaller than 64.)
  while (rng.room()) {
    rng.request();
    rng.produce();
  }
  int i = 1;
  if ((n % 2) == 0) {
    const uint32_t j = fw_uniform_int_from<LOG2L>(rng, rng.next_sync(), 1u);
    const int a1 = a[1], aj = a[j];
    if (lane0) { a[1] = aj; a[j] = a1; }
    i = 2;
  }
  if (i == n) return;
  uint32_t raw = rng.next_sync();
  for (int it = 0; i != n; ++it) {
    const int ai = a[i], ai1 = a[i + 1];
    const uint32_t sr = (uint32_t)i + 1u;
    const uint32_t x = fw_uniform_int_from<LOG2L>(rng, raw, sr * (sr + 1u) - 1u);
    const uint32_t p0 = x / (sr + 1u), p1 = x - p0 * (sr + 1u);
    if ((it & 15) == 15) rng.prefetch();
    if ((it & 15) == 7 && rng.pend) rng.produce();
    if (i + 2 != n) raw = rng.next_sync();  // (only if another pair follows: no draw may be consumed in vain)
    const int ap0 = a[p0], ap1 = a[p1];
    const int v1 = (p1 == (uint32_t)i) ? ap0 : ((p1 == p0) ? ai : ((p1 == (uint32_t)i + 1u) ? ai1 : ap1));
    if (lane0) {
      a[i] = ap0;
      a[p0] = ai;
      a[i + 1] = v1;
      a[p1] = ai1;
    }
    i += 2;
  }
}

typedef TNCO_LDS volatile uint16_t lds_vu16;

// One internal node of the post-order list: node | left << 16 | right << 32 | e << 48 (finite width: at
// most 65 535 nodes).  e = the exponent field of the node's cached contraction cost (the walk kernels
// read it with the links), 0 where nobody recorded it.
__device__ __forceinline__ uint64_t fw_rec(int node, int l, int rr, int e = 0) {
  return (uint64_t)(uint32_t)node | ((uint64_t)(uint32_t)l << 16) | ((uint64_t)(uint32_t)rr << 32) | ((uint64_t)(uint32_t)e << 48);
}
__device__ __forceinline__ int fw_rec_node(uint64_t x) { return (int)(x & 0xFFFFu); }
__device__ __forceinline__ int fw_rec_left(uint64_t x) { return (int)((x >> 16) & 0xFFFFu); }
__device__ __forceinline__ int fw_rec_right(uint64_t x) { return (int)((x >> 32) & 0xFFFFu); }
__device__ __forceinline__ int fw_rec_exp(uint64_t x) { return (int)(x >> 48); }
// the exponent field of a double
__device__ __forceinline__ int fw_exp_field(double c) { return (int)((__double_as_longlong(c) >> 52) & 0x7FF); }

// LDS of one replica for the traversal: `cap` stack entries (int32) + their left children (uint16)
struct FwStack {
  lds_vi32* e;
  lds_vu16* l;
  int cap;
};

// Post-order of include/tnco/utils.hpp:34-51 (child 0's subtree, child 1's subtree, the node).
// The re-slice is bound by the number of memory transactions, so this walk is the only time the
// tree's node headers are read: it leaves
//   rec[N - n]   the internal nodes in post-order WITH their child links (fw_rec; sequential 8-byte
//                records: what the cache rebuild iterates over),
//   wlist[]      (when not NULL) the tensors wider than max_width, in post-order -- the cached width
//                sits in the header line that is fetched for the links anyway (WidthCache,
//                finite_width/utils.hpp:49-76); their number is returned.
//
// With a stack (entry: node | right child << 13 | right-visited << 26 | too-wide << 27, left child
// beside it; the first st.cap entries in LDS, deeper ones in the global scratch `gstk`) every header
// is fetched ONCE, on the way down (the replicas' trees differ: a loop nest that follows one tree's
// shape would cost every replica the longest descent / ascent among the sixteen -- see below).
// Trees of more than 8192 nodes (or no stack) walk the links instead: the successor of x is its
// parent if x is the right child, else the left-most leaf below its sibling.
template <int LOG2L, int K, bool HYPER>
__device__ __forceinline__ int fw_traverse(const Params& P, const FwParams& F, const View<LOG2L, K, HYPER>& v,
                                           const double* w64, uint64_t* rec, int32_t* wlist, bool lane0, int gbase,
                                           FwStack st, volatile int32_t* gstk) {
  const int n = v.n, N = P.N;
  int ni = 0, nw = 0;
  auto emit_leaf = [&](int x) {
    if (wlist != nullptr && F.leaf_wide && ((F.leaf_bits[x >> 5] >> (x & 31)) & 1u)) {
      if (lane0) wlist[nw] = x;
      ++nw;
    }
  };
  auto emit_node = [&](int x, int l, int rr, bool wide) {
    if (lane0) rec[ni] = fw_rec(x, l, rr);
    ++ni;
    if (wide) {
      if (lane0) wlist[nw] = x;
      ++nw;
    }
  };
  auto is_wide = [&](int x, int pad) -> bool {
    if (wlist == nullptr) return false;
    const double w = F.width_f32 ? (double)__int_as_float(pad) : w64[x];
    return w > F.max_width;
  };
  const bool walk = (st.e == nullptr) || (gstk == nullptr) || N > 8192;
  if (!walk) {
    // One step per replica and iteration, whatever the shapes of the sixteen trees: a replica either
    // goes down (x: the node whose header to fetch; push it, on to its left child) or comes up
    // (x < 0: the top of the stack is entered to the right if that is still to do -- and that node's
    // header fetched in the same iteration -- else it is finished and popped).  Every replica needs
    // the same number of steps (two per internal node), so the wavefront is done when each of them is.  Measured alternatives, all slower: a
    // loop nest following the tree (descend-loop, pop-loop: every replica pays the longest run
    // among the sixteen), several pops per iteration (1 / 2 / 4 / 8: 4.2 / 4.4 / 5.0 / 5.9 M cycles),
    // header loads issued ahead of the pops (the loop is bound by its instructions, not the load).
    const int cap = st.cap;
    const int gh = (N + 1) / 2;  // gstk: [0, gh) entries, [gh, 2 gh) their left children
    int sp = 0, x = N - 1;
    bool done = false;
    if (x < n) {
      emit_leaf(x);
      done = true;
    }
    auto up = [&]() {
        int e, l;
        if (sp <= cap) {
          e = st.e[sp - 1];
          l = st.l[sp - 1];
        } else {
          __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (lane 0's store, every lane's load)
          e = gstk[sp - 1 - cap];
          l = gstk[gh + sp - 1 - cap];
        }
        const int node = e & 0x1FFF, rr = (e >> 13) & 0x1FFF;
        const bool fresh = ((e >> 26) & 1) == 0;
        if (fresh && rr >= n) {  // into the right subtree (its header is fetched in this same iteration)
          if (lane0) {
            if (sp <= cap) st.e[sp - 1] = e | (1 << 26); else gstk[sp - 1 - cap] = e | (1 << 26);
          }
          x = rr;
        } else {  // the node is finished (after its right child, if that is a leaf)
          if (fresh) emit_leaf(rr);
          --sp;
          emit_node(node, l, rr, ((e >> 27) & 1) != 0);
          if (sp == 0) done = true;
        }
    };
    while (!done) {
      if (x < 0) up();
      if (x >= n) {  // down
        const int4 h = *reinterpret_cast<const int4*>(v.hdr(x));  // (left, right, parent, width)
        const bool wide = is_wide(x, h.w);
        ++sp;
        if (lane0) {
          const int e = x | (h.y << 13) | (wide ? (1 << 27) : 0);
          if (sp <= cap) {
            st.e[sp - 1] = e;
            st.l[sp - 1] = (uint16_t)h.x;
          } else {
            gstk[sp - 1 - cap] = e;
            gstk[gh + sp - 1 - cap] = h.x;
          }
        }
        x = h.x;
        if (x < n) {
          emit_leaf(x);  // the left child is a leaf: up from here
          x = -1;
        }
      }
      if (!done && x < 0) up();  // (a second chance per iteration: an up step before and after the fetch)
    }
  }
  if (walk) {
    int x = N - 1;
    for (;;) {
      const int l = v.left(x);
      if (l < 0) break;
      x = l;
    }
    for (;;) {
      if (x < n) {
        emit_leaf(x);
      } else {
        const int4 h = *reinterpret_cast<const int4*>(v.hdr(x));
        emit_node(x, h.x, h.y, is_wide(x, h.w));
      }
      const int p = v.parent(x);
      if (p < 0) break;
      const int rr = v.right(p);
      if (rr == x) {
        x = p;
        continue;
      }
      x = rr;
      for (;;) {
        const int l = v.left(x);
        if (l < 0) break;
        x = l;
      }
    }
  }
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return nw;
}

// CostCache(ctree, ccost, slices) (finite_width/utils.hpp:36-47) into scratch; returns
// partial[root]; *sum = get_cost (finite_width/utils.hpp:24-33).  The contraction cost is taken
// over in1 | in2 | slices (finite_width/cost_model/simple.hpp:139-144).  Iterates over the internal
// nodes in post-order (`rec`): every iteration is work for every replica of the wavefront.
//
// B (four) nodes per iteration: their eight child masks are requested together (one memory latency
// per B nodes) with the records of the next B.  The (cost, partial sum) of node number j of the list
// goes to cp[j]: sequential 16-byte stores of lane 0, full lines by the time they leave the L2 (kept
// by node number they were two 8-byte read-modify-writes per node, and a third of the time of this
// function).  The partial sums are a stack machine of lane 0's: in post-order an internal right
// child is the node just before its parent and its sum still in a register (`part`); an internal
// left child's sum is the top of the stack of finished subtrees; a node with two leaf children
// pushes (pstk[]: consecutive doubles, the same few lines again and again).
typedef TNCO_LDS volatile double lds_vdouble;
typedef TNCO_LDS volatile uint64_t lds_vu64;

template <int LOG2L, int K, bool HYPER, int B = 4>
__device__ __forceinline__ double fw_rebuild(const Params& P, const View<LOG2L, K, HYPER>& v, const uint64_t* rec,
                                             const Mask<K>& slices, double2* cp, double* pstk, bool lane0,
                                             int gbase, double* sum, lds_vdouble* lstk = nullptr, int lcap = 0,
                                             lds_vu64* lmask = nullptr, int mcap = 0) {
  constexpr int L = 1 << LOG2L;
  constexpr int LK = L * K;
  const int n = P.n, ni = P.N - P.n;
  double s = 0.0, part = 0.0;
  if (ni <= 0) {
    *sum = 0.0;
    return 0.0;
  }
  auto ld_rec = [&](int j) -> uint64_t { return rec[j < ni ? j : ni - 1]; };
  // Without hyper-indices the legs of an internal node are left ^ right (tnco/ctree.py:163-189), so
  // the rebuild needs no leg mask of an internal node from memory.  In post-order the subtree
  // finished last is in registers (`prev`: the legs of its root, `part`: its partial sum): it is the
  // right child of the next node if that one's right child is internal, else its left child if that
  // is internal.  A node with two internal children takes its left child from the stack of finished
  // subtrees, where a subtree goes when the next node starts a new one (two leaf children).  The
  // stack's first entries are in LDS (partial sums: lcap, legs: mcap); deeper partial sums go to
  // global scratch, deeper legs are not kept and read from the node block when needed (that read is
  // predicted when the batch's loads are issued: the stack depth follows from the records alone).
  // Only leaf legs (the shared table, L2) are read otherwise.  (A skipped read fetches leaf 0
  // instead: one unconditional load per slot, "a loaded value has one definition".)
  uint64_t rc[B];
#pragma unroll
  for (int i = 0; i < B; ++i) rc[i] = ld_rec(i);
  Mask<K> prev = mzero<K>();
  int sp = 0;  // finished subtrees not yet consumed: the newest in registers, the others on the stack [0, sp - 1)
  auto stk_get = [&](int i) -> double { return i < lcap ? lstk[i] : pstk[i]; };
  auto stk_put = [&](int i, double x) {
    if (i < lcap) lstk[i] = x; else pstk[i] = x;
  };
  for (int j0 = 0; j0 < ni; j0 += B) {
    Mask<K> ml[B], mr[B];
    uint64_t rn[B];
    {
      int spp = sp;  // the stack depth when each record of the batch is reached
#pragma unroll
      for (int i = 0; i < B; ++i) {
        const int l = fw_rec_left(rc[i]), rr = fw_rec_right(rc[i]);
        const bool li = l >= n, ri = rr >= n;
        int lsrc = l, rsrc = rr;
        if constexpr (!HYPER) {
          if (ri) rsrc = 0;                                   // right child: the subtree finished last
          if (li && (!ri || spp - 2 < mcap)) lsrc = 0;        // left child: the subtree finished last / an LDS entry
        }
        ml[i] = v.mask(lsrc);
        mr[i] = v.mask(rsrc);
        spp += (li && ri) ? -1 : ((!li && !ri) ? 1 : 0);
      }
    }
#pragma unroll
    for (int i = 0; i < B; ++i) rn[i] = ld_rec(j0 + B + i);
#pragma unroll
    for (int i = 0; i < B; ++i) {
      if (j0 + i < ni) {
        const int l = fw_rec_left(rc[i]), rr = fw_rec_right(rc[i]);
        const bool li = l >= n, ri = rr >= n;
        Mask<K> mleft = ml[i], mright = mr[i];
        if constexpr (!HYPER) {
          if (ri) mright = prev;
          if (li && !ri) mleft = prev;
          if (li && ri && sp - 2 < mcap) {
#pragma unroll
            for (int k = 0; k < K; ++k) mleft.w[k] = lmask[(sp - 2) * LK + k * L + v.lig];
          }
          if (!li && !ri && sp >= 1 && sp - 1 < mcap) {  // a new subtree starts: the finished one waits on the stack
#pragma unroll
            for (int k = 0; k < K; ++k) lmask[(sp - 1) * LK + k * L + v.lig] = prev.w[k];
          }
        }
        const Mask<K> u = mor<K>(mor<K>(mleft, mright), slices);
        if constexpr (!HYPER) prev = mxor<K>(mleft, mright);
        const double c = generic_cost<LOG2L, K>(P, u, v.lig, gbase);
        s = rnd_cost(s + c, P.f32);
        if (lane0) {
          double pl = 0.0, pr = 0.0;
          if (li && ri) {
            pr = part;
            pl = stk_get(sp - 2);
          } else if (ri) {
            pr = part;
          } else if (li) {
            pl = part;
          } else if (sp >= 1) {
            stk_put(sp - 1, part);
          }
          part = rnd_cost(rnd_cost(c + pl, P.f32) + pr, P.f32);
          cp[j0 + i] = make_double2(c, part);
        }
        sp += (li && ri) ? -1 : ((!li && !ri) ? 1 : 0);
      }
    }
#pragma unroll
    for (int i = 0; i < B; ++i) rc[i] = rn[i];
  }
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  *sum = s;
  const int lo = __shfl(__double2loint(part), gbase), hi = __shfl(__double2hiint(part), gbase);
  return __hiloint2double(hi, lo);
}

// the rebuilt (cost, partial sum) of every internal node into its header
template <int LOG2L, int K, bool HYPER>
__device__ __forceinline__ void fw_commit(const Params& P, const View<LOG2L, K, HYPER>& v, const uint64_t* rec,
                                          const double2* cp) {
  constexpr int L = 1 << LOG2L;
  constexpr int B = 4;  // loads in flight per lane: 2 * B
  const int ni = P.N - P.n;
  for (int j0 = v.lig; j0 < ni; j0 += B * L) {
    uint64_t x[B];
    double2 c[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int j = j0 + i * L;
      x[i] = 0;
      c[i] = make_double2(0.0, 0.0);
      if (j < ni) {
        x[i] = rec[j];
        c[i] = cp[j];
      }
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int j = j0 + i * L;
      if (j < ni) *reinterpret_cast<double2*>(&v.hdr(fw_rec_node(x[i]))->ccost) = c[i];
    }
  }
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// WidthCache: width of internal node p as stored / of a leaf as computed
template <int LOG2L, int K, bool HYPER>
__device__ __forceinline__ double fw_node_width(const Params& P, const FwParams& F, const View<LOG2L, K, HYPER>& v,
                                                const double* w64, int t, int gbase) {
  if (t >= P.n) return F.width_f32 ? (double)__int_as_float(v.hdr(t)->pad) : w64[t];
  return fw_width<LOG2L, K>(P, F, v.mask(t), v.lig, gbase);
}
template <int LOG2L, int K, bool HYPER>
__device__ __forceinline__ void fw_set_node_width(const FwParams& F, const View<LOG2L, K, HYPER>& v, double* w64,
                                                  int p, double w, bool lane0) {
  if (!lane0) return;
  if (F.width_f32) v.hdr(p)->pad = __float_as_int((float)w);
  else w64[p] = w;
}

// number of set positions of `cand` over the group
template <int LOG2L, int K>
__device__ __forceinline__ uint32_t fw_count(const Mask<K>& cand) {
  return gsum<LOG2L>(mpopc<K>(cand));
}

// positions of `cand`, ascending (Bitset::positions), into pos[] (at most `cap`; global scratch or
// LDS); returns their number
template <int LOG2L, int K, bool HYPER, typename A>
__device__ __forceinline__ uint32_t fw_positions(const View<LOG2L, K, HYPER>& v, const Mask<K>& cand, A pos,
                                                 uint32_t cap, int gbase, bool lane0, int32_t* status) {
  constexpr int L = 1 << LOG2L;
  uint32_t np = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t mine = (uint32_t)__popcll(cand.w[k]);
    uint32_t before = 0, tot = 0;
    for (int j = 0; j < L; ++j) {
      const uint32_t c = (uint32_t)__shfl((int)mine, gbase + j);
      if (j < v.lig) before += c;
      tot += c;
    }
    uint32_t o = np + before;
    uint64_t x = cand.w[k];
    while (x) {
      const int b = __ffsll((unsigned long long)x) - 1;
      if (o < cap) pos[o] = (int16_t)(v.widx(k) * 64 + b);
      ++o;
      x &= x - 1;
    }
    np += tot;
  }
  if (np > cap) {
    if (lane0) *status = 1;
    np = cap;
  }
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return np;
}

#ifndef TNCO_FW_LDSPOS
#define TNCO_FW_LDSPOS 128
#endif
constexpr int FW_LDSPOS = TNCO_FW_LDSPOS;  // candidate legs per tensor that fit the LDS fast path

// too-wide counts per index (sc.n_big): bytes when one flush of the bit-sliced counters holds them
template <int K>
constexpr int FW_NBIG_PLANES = K <= 4 ? 8 : 5;
template <int K>
__device__ __forceinline__ bool fw_nbig8(int nw) {
  return FW_NBIG_PLANES<K> == 8 && nw <= 255;
}
__device__ __forceinline__ int fw_nbig_at(const int32_t* n_big, bool nb8, int x) {
  return nb8 ? (int)reinterpret_cast<const uint8_t*>(n_big)[x] : n_big[x];
}

#if defined(TNCO_PROFILE) && TNCO_PROFILE == 4  // cycles inside the greedy pass: [scan, positions, shuffle, keys + picks]
#define FW_GP_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define FW_GP_ADD(i, a, b) do { if (cnt) cnt[i] += (b) - (a); } while (0)
#define FW_GP_COUNT(i, x)
#else
#define FW_GP_T(var)
#define FW_GP_ADD(i, a, b)
#define FW_GP_COUNT(i, x) do { if (cnt) cnt[i] += (x); } while (0)
#endif

// A tensor that is still too wide after the slices chosen so far (sliced_xs = its legs - slices, of
// width sliced_width): shuffle its candidate legs and slice them in the order of `greater` until it
// fits (finite_width/greedy/utils.hpp:72-101).
template <int LOG2L, int K, bool HYPER, typename RNG>
__device__ __forceinline__ void fw_slice_wide(const Params& P, const FwParams& F, const View<LOG2L, K, HYPER>& v,
                                              RNG& rng, const int32_t* n_big, const bool nb8, volatile int16_t* pos,
                                              lds_vi32* lpos, bool lane0, int gbase, int32_t* status,
                                              const Mask<K>& skip, Mask<K> sliced_xs, double sliced_width,
                                              Mask<K>& slices, unsigned long long* cnt) {
  constexpr int L = 1 << LOG2L;
  const int lig = v.lig;
  FW_GP_COUNT(1, 1);  // ... still too wide after the slices so far: shuffled and picked from
  // candidate positions, ascending
  const Mask<K> cand = mandn<K>(sliced_xs, skip);
  if (F.log2dims == nullptr && fw_count<LOG2L, K>(cand) <= (uint32_t)FW_LDSPOS) {
    // Fast path (uniform dims, the usual number of candidates): the candidate list lives in LDS.
    // Same draws, same order: shuffle, then the keys are attached and every pick is a scan shared
    // by the lanes of the group + one DPP max (a candidate's key is >= 1: the tensor itself is too
    // wide, so 0 can stand for "taken").
    FW_GP_T(t0_);
    rng.prefetch();  // (the generator's next block travels while the positions are listed)
    const uint32_t np = fw_positions<LOG2L, K, HYPER>(v, cand, lpos, (uint32_t)FW_LDSPOS, gbase, lane0, status);
    FW_GP_T(t1_);
    fw_shuffle_lds<LOG2L>(rng, lpos, (int)np, lane0);
    FW_GP_T(t2_);
    FW_GP_ADD(1, t0_, t1_);
    FW_GP_ADD(2, t1_, t2_);
    FW_GP_COUNT(2, np);  // candidate legs
    // Keys ((too-wide count << 16) | 0xFFFF - shuffled rank: the stable order of :83) stay in
    // registers, FW_LDSPOS / L per lane, gathered in one flight; a pick is a register scan + one DPP
    // max over the group, a taken candidate's key becomes 0.  LDS keeps the shuffled positions.
    constexpr int G = FW_LDSPOS / L;
    uint32_t ck[G];
    {
      int xp[G];
#pragma unroll
      for (int i = 0; i < G; ++i) xp[i] = (uint32_t)(lig + i * L) < np ? (int)lpos[lig + i * L] : 0;
#pragma unroll
      for (int i = 0; i < G; ++i) ck[i] = ((uint32_t)(lig + i * L) < np) ? (uint32_t)fw_nbig_at(n_big, nb8, xp[i]) : 0u;
#pragma unroll
      for (int i = 0; i < G; ++i) ck[i] = ck[i] ? ((ck[i] << 16) | (0xFFFFu - (uint32_t)(lig + i * L))) : 0u;
    }
    for (uint32_t taken = 0; taken < np; ++taken) {
      uint32_t best = 0;
#pragma unroll
      for (int i = 0; i < G; ++i) best = ck[i] > best ? ck[i] : best;
      best = gmax<LOG2L>(best);
#pragma unroll
      for (int i = 0; i < G; ++i) ck[i] = (ck[i] == best) ? 0u : ck[i];
      const uint32_t qb = 0xFFFFu - (best & 0xFFFFu);
      const int xpos = lpos[qb];
      fw_flip<LOG2L, K, HYPER>(v, slices, xpos);
      sliced_width = fw_wr(F, sliced_width + fw_delta_width<LOG2L, K, HYPER>(P, F, v, sliced_xs, xpos, gbase));
      fw_flip<LOG2L, K, HYPER>(v, sliced_xs, xpos);
      FW_GP_COUNT(3, 1);  // picks
      if (sliced_width <= F.max_width) break;
    }
    FW_GP_T(t3_);
    FW_GP_ADD(3, t2_, t3_);
    return;
  }
  uint32_t np = fw_positions<LOG2L, K, HYPER>(v, cand, pos, (uint32_t)F.I64, gbase, lane0, status);
  // :80  std::shuffle(positions, prng)
  fw_shuffle<LOG2L>(rng, pos, (int)np, lane0);
  // :83-101  stable_sort with `greater` (:50-60: more too-wide tensors first; with per-index dims
  // ties go to the larger log2(dims)), then slice until the tensor fits: equivalent to repeatedly
  // taking the FIRST remaining position with the largest key.
  for (uint32_t taken = 0; taken < np; ++taken) {
    int best = -1, best_key = -1;
    double best_l2 = 0.0;
    for (uint32_t q = 0; q < np; ++q) {
      const int xp = pos[q];
      if (xp < 0) continue;
      const int key = fw_nbig_at(n_big, nb8, xp);
      if (F.log2dims == nullptr) {
        if (key > best_key) { best_key = key; best = (int)q; }
      } else {
        const double l2 = fw_log2dim(F, xp);
        if (best < 0 || key > best_key || (key == best_key && l2 > best_l2)) {
          best_key = key; best_l2 = l2; best = (int)q;
        }
      }
    }
    const int xpos = pos[best];
    if (lane0) pos[best] = -1;
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // slices.set(xpos); sliced_width += delta_width(sliced_xs, xpos); sliced_xs.reset(xpos)
    fw_flip<LOG2L, K, HYPER>(v, slices, xpos);  // xpos is not in slices: sliced_xs = inds - slices
    sliced_width = fw_wr(F, sliced_width + fw_delta_width<LOG2L, K, HYPER>(P, F, v, sliced_xs, xpos, gbase));
    fw_flip<LOG2L, K, HYPER>(v, sliced_xs, xpos);
    if (sliced_width <= F.max_width) break;
  }
}

// get_slices_impl, finite_width/greedy/utils.hpp:21-125.  The 16 replicas of a wavefront find their
// too-wide tensors at different places of their trees, and the work on one such tensor is long: run
// per replica "as it comes", that work would execute once per replica and tensor with 1/16 of the
// lanes.  So the too-wide tensors are listed first (by the walk over the tree), then all replicas of
// the wavefront work on their k-th one together.
//
// Part 1, :41-48: the walk (sc.rec, sc.wlist; see fw_traverse), then for every index the number of
// too-wide tensors it appears in (sc.n_big[]).  Returns the number of too-wide tensors.
template <int LOG2L, int K, bool HYPER>
__device__ __forceinline__ int fw_gs_mark(const Params& P, const FwParams& F, const View<LOG2L, K, HYPER>& v,
                                          const double* w64, const FwScratch& sc, FwStack st, bool lane0, int gbase,
                                          unsigned long long* prof = nullptr, int nw_pre = -1) {
  // (nw_pre >= 0: fw_walk2_kernel has left sc.rec / sc.wlist and this count)
  const int nw = nw_pre >= 0 ? nw_pre
                             : fw_traverse<LOG2L, K, HYPER>(P, F, v, w64, sc.rec, sc.wlist, lane0, gbase, st, sc.gstk);
#ifdef TNCO_PROFILE
  if (prof) prof[0] = __builtin_amdgcn_s_memtime();
#endif
  // The k-th too-wide tensor of every replica, together.  Every index belongs to one lane (bit b of
  // its word k), so the counts are kept there, bit-sliced: plane p holds bit p of the 64 * K
  // counters of the lane, adding a tensor's mask is a ripple-carry over the planes (no memory
  // traffic at all; the first version issued one atomic per leg: 5 000 per replica on config 5,
  // bound by the L2's atomic rate).  The planes are written out as int32 counts once per
  // 2^NP - 1 tensors.
  // Up to 255 too-wide tensors (the usual case: one flush) the counts are written as BYTES: the
  // greedy pass gathers them one candidate leg at a time, and a replica's counts then span a
  // quarter of the lines (fw_nbig8 / fw_nbig_at).
  int32_t* n_big = sc.n_big;
  constexpr int NP = FW_NBIG_PLANES<K>;  // (registers: 2 * K * NP)
  const bool nb8 = fw_nbig8<K>(nw);
  Mask<K> pl[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) pl[p] = mzero<K>();
  bool first = true;
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t w[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) w[p] = (uint32_t)(pl[p].w[k] >> (32 * h));
        if (nb8) {  // 32 counters -> 32 bytes; a nibble of plane bits is spread over the bytes of a word
          uint32_t o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            uint32_t c = 0;
#pragma unroll
            for (int p = 0; p < NP; ++p) c |= ((((w[p] >> (4 * j)) & 0xFu) * 0x00204081u) & 0x01010101u) << p;
            o[j] = c;
          }
          uint4* d8 = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(n_big) + v.widx(k) * 64 + 32 * h);
          d8[0] = make_uint4(o[0], o[1], o[2], o[3]);
          d8[1] = make_uint4(o[4], o[5], o[6], o[7]);
          continue;
        }
        int4* dst = reinterpret_cast<int4*>(n_big + v.widx(k) * 64 + 32 * h);
        for (int j = 0; j < 8; ++j) {
          int4 c = {0, 0, 0, 0};
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            c.x |= (int)((w[p] & 1u) << p);
            c.y |= (int)(((w[p] >> 1) & 1u) << p);
            c.z |= (int)(((w[p] >> 2) & 1u) << p);
            c.w |= (int)(((w[p] >> 3) & 1u) << p);
            w[p] >>= 4;
          }
          if (!first) {
            const int4 o = dst[j];
            c.x += o.x; c.y += o.y; c.z += o.z; c.w += o.w;
          }
          dst[j] = c;
        }
      }
    }
    first = false;
#pragma unroll
    for (int p = 0; p < NP; ++p) pl[p] = mzero<K>();
  };
  if (nw > 0) {
    int inchunk = 0;
    int ta = sc.wl(0, nw), tb = nw > 1 ? sc.wl(1, nw) : 0;
    Mask<K> m0 = v.mask(ta);
    for (int j = 0; j < nw; ++j) {
      // (the next mask is on its way while this one is added; past the end of the list: any tensor --
      // unconditional loads, one definition each)
      ta = tb;
      const Mask<K> m1 = v.mask(ta);
      tb = sc.wl(j + 2 < nw ? j + 2 : nw - 1, nw);
      Mask<K> carry = m0;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const uint64_t tt = pl[p].w[k] & carry.w[k];
          pl[p].w[k] ^= carry.w[k];
          carry.w[k] = tt;
        }
      }
      if (++inchunk == (1 << NP) - 1) {
        flush();
        inchunk = 0;
      }
      m0 = m1;
    }
    if (inchunk > 0 || first) flush();
  }
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TNCO_PROFILE
  if (prof) prof[1] = __builtin_amdgcn_s_memtime();
#endif
  return nw;
}

// Part 2, :62-101: the greedy pass over the nw too-wide tensors in post-order; returns the new slices.
template <int LOG2L, int K, bool HYPER, typename RNG>
__device__ __forceinline__ Mask<K> fw_gs_pick(const Params& P, const FwParams& F, const View<LOG2L, K, HYPER>& v,
                                              RNG& rng, const FwScratch& sc, int nw, lds_vi32* lpos,
                                              bool lane0, int gbase, int32_t* status,
                                              unsigned long long* cnt = nullptr) {
  const int lig = v.lig;
  const int32_t* n_big = sc.n_big;
  volatile int16_t* pos = sc.pos;
  Mask<K> slices = mzero<K>();
  Mask<K> skip = mzero<K>();
  if (F.skip) {
#pragma unroll
    for (int k = 0; k < K; ++k) skip.w[k] = F.skip[v.widx(k)];
  }
  // :62-101  post-order over the too-wide tensors.  Most of them fit once earlier ones have been
  // sliced (config 5: 110 marked, 24 still too wide when their turn comes), WHICH ones differs from
  // replica to replica, and the work on one that does not fit is long.  So every replica first runs
  // ahead to its next tensor that does not fit (a cheap scan: masks requested one tensor ahead), then
  // the replicas of the wavefront do the long part together.
  {
    int j = 0;
    int ta = nw > 0 ? sc.wl(0, nw) : 0, tb = nw > 1 ? sc.wl(1, nw) : 0;
    Mask<K> ma = mzero<K>();
    if (nw > 0) ma = v.mask(ta);
    for (;;) {
      bool have = false;
      Mask<K> sx = mzero<K>();
      double sw = 0.0;
      FW_GP_T(ts0_);
      while (j < nw) {
        const Mask<K> m = ma;
        ta = tb;
        ++j;
        ma = v.mask(ta);  // (past the end of the list: any tensor)
        tb = sc.wl(j + 1 < nw ? j + 1 : nw - 1, nw);
        FW_GP_COUNT(0, 1);  // too-wide tensors
        sx = mandn<K>(m, slices);
        sw = fw_width<LOG2L, K>(P, F, sx, lig, gbase);
        if (sw > F.max_width) {
          have = true;
          break;
        }
      }
      FW_GP_T(ts1_);
      FW_GP_ADD(0, ts0_, ts1_);
      if (!have) break;
      fw_slice_wide<LOG2L, K, HYPER>(P, F, v, rng, n_big, fw_nbig8<K>(nw), pos, lpos, lane0, gbase, status, skip, sx, sw,
                                     slices, cnt);
    }
  }
  return slices;
}

template <int LOG2L, int K, bool HYPER, typename RNG>
__device__ __forceinline__ Mask<K> fw_get_slices(const Params& P, const FwParams& F, const View<LOG2L, K, HYPER>& v,
                                                 const double* w64, RNG& rng, const FwScratch& sc, FwStack st,
                                                 lds_vi32* lpos, bool lane0, int gbase, int32_t* status,
                                                 unsigned long long* prof = nullptr,
                                                 unsigned long long* cnt = nullptr, int nw_pre = -1) {
  const int nw = fw_gs_mark<LOG2L, K, HYPER>(P, F, v, w64, sc, st, lane0, gbase, prof, nw_pre);
  return fw_gs_pick<LOG2L, K, HYPER>(P, F, v, rng, sc, nw, lpos, lane0, gbase, status, cnt);
}

// Which leaf tensors are wider than max_width (the leaves are the same in every replica and never
// change): bits[t >> 5] |= 1 << (t & 31), *any = 1 if there is one.  One workgroup.
template <int LOG2L, int K>
__global__ __launch_bounds__(256) void fw_leaf_bits_kernel(const Params P, const FwParams F, uint32_t* bits, int32_t* any) {
  constexpr int L = 1 << LOG2L;
  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gbase = (tid & 63) & ~(L - 1);
  View<LOG2L, K, false> v;
  v.init(P, P.blocks, P.lpar, lig);
  for (int t = tid >> LOG2L; t < P.n; t += 256 >> LOG2L) {
    if (fw_width<LOG2L, K>(P, F, v.mask(t), lig, gbase) > F.max_width && lig == 0) {
      atomicOr(&bits[t >> 5], 1u << (t & 31));
      *any = 1;
    }
  }
}

// The walk of the re-slice as a kernel of its own (fw_reslice_kernel then starts from the lists): scalar
// work per replica -- links and cached widths, no leg masks -- and a chain of dependent header reads.  Same
// step structure as fw_traverse: one fetch per iteration, an up-step before and after it; stack entries
// lane-interleaved in LDS (no bank conflicts whatever the depths), the deep end in global scratch.  Trees of
// at most 8192 nodes (13-bit stack fields).
#ifndef TNCO_FW_WALK_CAP
#define TNCO_FW_WALK_CAP 40
#endif
#ifndef TNCO_FW_WALK_POPS
#define TNCO_FW_WALK_POPS 1
#endif
constexpr int FW_WALK_CAP = TNCO_FW_WALK_CAP;

// The walk from BOTH ends of the post-order, two lanes per replica (lanes 0-31 of a wavefront walk
// forward, lane 32 + i walks backward for the replica of lane i).  A one-ended walk keeps one header
// read in flight per replica and sits at half of the chip's request rate; the reverse of a post-order
// is the pre-order that takes the RIGHT child first, so a second walker can emit the list from its end:
// node x at the moment its header arrives (position n - 2 - b for the b-th node), right subtree next,
// the left child waiting on a small stack.  Both fill the same rec[]; the too-wide tensors of the
// backward walker go to the END of wlist (FwScratch::wl).  The two lanes run in lock-step and exchange
// their counts every iteration: with `left` = nodes not yet listed, the backward walker lists one more
// only while left >= 2, the forward walker gets the rest -- every node exactly once.  A backward
// walker whose stack outgrows its LDS entries stops for good (the forward walker finishes alone).
// Not for trees with too-wide LEAVES (F.leaf_wide: their place in the list is the forward walker's
// business): fw_reslice_kernel traverses those itself.
constexpr int FW_WALK2_CAPB = 32;
static __global__ __launch_bounds__(256) void fw_walk2_kernel(const Params P, const FwParams F) {
  constexpr int NT = 128;  // replicas per block: the stride of the LDS arrays
  __shared__ int32_t se[FW_WALK_CAP * NT];
  __shared__ uint32_t sl[FW_WALK_CAP * NT];    // left child | exponent field of the cached cost << 16
  __shared__ uint16_t sb[FW_WALK2_CAPB * NT];  // backward walker: left children waiting
  __shared__ uint32_t rbuf[2 * 16 * NT];       // 8 records per walker
  // (the too-wide tensors go to their list one by one: a handful per replica)
  const int tid = threadIdx.x;
  const bool fwd = (tid & 32) == 0;
  const int slot = (tid >> 6) * 32 + (tid & 31);
  const int64_t r = (int64_t)blockIdx.x * NT + slot;
  const int n = P.n, N = P.N, LK = F.I64 / 64, ni_all = N - n;
  bool active = r < P.R;
  if (active) {  // greedy/optimizer.hpp:359: nothing to do without slices
    const uint64_t* sl0 = F.slices + r * 2 * (int64_t)LK;
    uint64_t any = 0;
    for (int w = 0; w < P.W; ++w) any |= sl0[w];
    if (!any) {
      if (fwd) F.nwide[r] = -1;
      active = false;
    }
  }
  const int64_t rr_ = active ? r : 0;
  const FwScratch sc(F, rr_, N);
  const uint8_t* blk = P.blocks + rr_ * P.RB;
  const double* w64 = F.width64 ? F.width64 + rr_ * (int64_t)N : nullptr;
  uint64_t* rec = sc.rec;
  int32_t* wlist = sc.wlist;
  volatile int32_t* gstk = sc.gstk;
  const int wcap = sc.wcap;
  TNCO_LDS volatile int32_t* e_ = (TNCO_LDS volatile int32_t*)se + slot;
  TNCO_LDS volatile uint32_t* l_ = (TNCO_LDS volatile uint32_t*)sl + slot;
  TNCO_LDS volatile uint16_t* b_ = (TNCO_LDS volatile uint16_t*)sb + slot;
  TNCO_LDS volatile uint32_t* rb = (TNCO_LDS volatile uint32_t*)rbuf + (fwd ? 0 : 16 * NT) + slot;
  const int gh = (N + 1) / 2;
  int cnt = 0, nw = 0;  // nodes / too-wide tensors this walker has listed
  // ---- forward walker ----
  int sp = 0, x = N - 1;
  auto f_put_wide = [&](int t) { wlist[nw++] = t; };
  auto f_put_rec = [&](uint64_t v) {
    rb[(2 * (cnt & 7)) * NT] = (uint32_t)v;
    rb[(2 * (cnt & 7) + 1) * NT] = (uint32_t)(v >> 32);
    ++cnt;
    if ((cnt & 7) == 0) {
      uint4* d = reinterpret_cast<uint4*>(rec + cnt - 8);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        d[q] = make_uint4(rb[(4 * q) * NT], rb[(4 * q + 1) * NT], rb[(4 * q + 2) * NT], rb[(4 * q + 3) * NT]);
    }
  };
  // one step up; lists a node only if `may` (returns 1 if it did)
  auto f_up = [&](bool may) -> int {
    int e, l;
    if (sp <= FW_WALK_CAP) {
      e = e_[(sp - 1) * NT];
      l = l_[(sp - 1) * NT];
    } else {
      e = gstk[sp - 1 - FW_WALK_CAP];
      l = gstk[gh + sp - 1 - FW_WALK_CAP];
    }
    const int node = e & 0x1FFF, rr = (e >> 13) & 0x1FFF;
    const bool fresh = ((e >> 26) & 1) == 0;
    if (fresh && rr >= n) {  // into the right subtree
      if (sp <= FW_WALK_CAP) e_[(sp - 1) * NT] = e | (1 << 26); else gstk[sp - 1 - FW_WALK_CAP] = e | (1 << 26);
      x = rr;
      return 0;
    }
    if (!may) return 0;
    --sp;
    f_put_rec(fw_rec(node, l & 0xFFFF, rr, (int)((uint32_t)l >> 16)));
    if ((e >> 27) & 1) f_put_wide(node);
    return 1;
  };
  // ---- backward walker ----
  int bx = N - 1, bsp = 0;
  bool balive = true;  // (false: stack overflow, this walker has stopped)
  auto b_put_wide = [&](int t) { wlist[wcap - 1 - (nw++)] = t; };  // the k-th goes to wlist[wcap - 1 - k]
  auto b_put_rec = [&](uint64_t v) {  // the k-th goes to rec[ni_all - 1 - k]
    const int pos = ni_all - 1 - cnt;
    rb[(2 * (pos & 7)) * NT] = (uint32_t)v;
    rb[(2 * (pos & 7) + 1) * NT] = (uint32_t)(v >> 32);
    ++cnt;
    if ((pos & 7) == 0) {
      if (pos + 8 <= ni_all) {  // a whole 8-record piece of this walker's
        uint4* d = reinterpret_cast<uint4*>(rec + pos);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          d[q] = make_uint4(rb[(4 * q) * NT], rb[(4 * q + 1) * NT], rb[(4 * q + 2) * NT], rb[(4 * q + 3) * NT]);
      } else {  // the piece holding the last record of the list, cut by its end
        for (int k = pos; k < ni_all; ++k)
          rec[k] = (uint64_t)rb[(2 * (k & 7)) * NT] | ((uint64_t)rb[(2 * (k & 7) + 1) * NT] << 32);
      }
    }
  };
  if (N - 1 < n) active = false;  // (a single tensor: no internal node)
  for (;;) {
    // counts of the pair, and whether the backward walker is still going
    const int word = active ? (cnt | ((fwd || balive) ? (1 << 20) : 0)) : (1 << 21);
    const int oth = __shfl_xor(word, 32);
    const bool pair_off = !active || ((oth >> 21) & 1);
    const int left = pair_off ? 0 : ni_all - cnt - (oth & 0xFFFFF);
    if (!__any(left > 0)) break;
    if (left <= 0) continue;
    if (fwd) {
      const bool b_on = ((oth >> 20) & 1) != 0;
      int may = left - ((b_on && left >= 2) ? 1 : 0);
      if (x < 0 && sp > 0) may -= f_up(may > 0);
      if (x >= n) {  // down: the only read of this node's header (links + cached width, one line)
        const int4 h = *reinterpret_cast<const int4*>(blk + (int64_t)(x - n) * P.BS);
        const int ce = fw_exp_field(*reinterpret_cast<const double*>(blk + (int64_t)(x - n) * P.BS + 16));  // (same sector)
        const double w = F.width_f32 ? (double)__int_as_float(h.w) : w64[x];
        const bool wide = w > F.max_width;
        ++sp;
        const int e = x | (h.y << 13) | (wide ? (1 << 27) : 0);
        const uint32_t le = (uint32_t)h.x | ((uint32_t)ce << 16);
        if (sp <= FW_WALK_CAP) {
          e_[(sp - 1) * NT] = e;
          l_[(sp - 1) * NT] = le;
        } else {
          gstk[sp - 1 - FW_WALK_CAP] = e;
          gstk[gh + sp - 1 - FW_WALK_CAP] = (int32_t)le;
        }
        x = h.x;
        if (x < n) x = -1;
      }
      if (x < 0 && sp > 0) may -= f_up(may > 0);
    } else if (balive && left >= 2 && bx >= n) {
      const int4 h = *reinterpret_cast<const int4*>(blk + (int64_t)(bx - n) * P.BS);
      const int ce = fw_exp_field(*reinterpret_cast<const double*>(blk + (int64_t)(bx - n) * P.BS + 16));
      const double w = F.width_f32 ? (double)__int_as_float(h.w) : w64[bx];
      b_put_rec(fw_rec(bx, h.x, h.y, ce));
      if (w > F.max_width) b_put_wide(bx);
      const bool li = h.x >= n, ri = h.y >= n;
      if (ri) {
        if (li) {
          if (bsp < FW_WALK2_CAPB) b_[(bsp++) * NT] = (uint16_t)h.x; else balive = false;
        }
        bx = h.y;
      } else if (li) {
        bx = h.x;
      } else if (bsp > 0) {
        bx = b_[(--bsp) * NT];
      } else {
        bx = -1;  // (only after the root's last descendant: nothing is left then)
      }
    }
  }
  if (!active) return;
  // the unfinished pieces of the lists, the counts
  if (fwd) {
    for (int k = cnt & ~7; k < cnt; ++k)
      rec[k] = (uint64_t)rb[(2 * (k & 7)) * NT] | ((uint64_t)rb[(2 * (k & 7) + 1) * NT] << 32);
  } else {
    // records [ni_all - cnt, ni_all): pieces went out when the walker reached their first record; the
    // piece cut by the lowest record is left
    const int lo = ni_all - cnt;
    for (int k = lo; k < ni_all && (k & ~7) < lo; ++k)
      rec[k] = (uint64_t)rb[(2 * (k & 7)) * NT] | ((uint64_t)rb[(2 * (k & 7) + 1) * NT] << 32);
  }
  const int onw = __shfl_xor(nw, 32);
  if (fwd) {
    F.nwide[r] = nw + onw;
    F.nwfront[r] = nw;
  }
}

struct FwInitArgs {
  const uint64_t* slices_in;  // [LK] per row or NULL: use instead of the initial get_slices
  int64_t slices_in_stride;   // uint64 elements between the replicas' rows; 0: one row for all
  double* out_total;          // [R]
  double* out_sum;            // [R]
};

// finite_width/greedy/optimizer.hpp:72-115: WidthCache, slices (greedy, draws from the PRNG),
// min_slices, CostCache(slices), min_total_cost = get_cost(min_ctree, min_slices).
template <int LOG2L, int K, bool HYPER>
__global__ __launch_bounds__(256, 2) void fw_init_kernel(const Params P, const FwParams F, const FwInitArgs a) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;
  constexpr int LK = L * K;
  using R = Rng<LOG2L>;
  __shared__ uint32_t rngbuf[GPB * R::RING];
  __shared__ int32_t posbuf[GPB * FW_LDSPOS];   // candidate legs of a tensor / traversal stack
  __shared__ uint16_t leftbuf[GPB * FW_LDSPOS];  // traversal stack: left children
  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gib = tid >> LOG2L;
  const int gbase = (tid & 63) & ~(L - 1);
  const int64_t r = (int64_t)blockIdx.x * GPB + gib;
  if (r >= P.R) return;
  lds_vi32* lpos = (lds_vi32*)posbuf + gib * FW_LDSPOS;
  const FwStack st{F.stack_cap > 0 ? lpos : nullptr, (lds_vu16*)leftbuf + gib * FW_LDSPOS, F.stack_cap};
  const bool lane0 = lig == 0;
  const int n = P.n, N = P.N;
  View<LOG2L, K, HYPER> v;
  v.init(P, P.blocks + r * P.RB, P.lpar + r * (int64_t)n * LPS, lig);
  ReplicaState* rs = P.rs + r;
  R rng;
  rng.init(P, r, (lds_vu32*)rngbuf + gib * R::RING, rs->mti, rs->mtw, lig);
  const FwScratch sc(F, r, N);
  double* w64 = F.width64 ? F.width64 + r * (int64_t)N : nullptr;
  // widths of the internal nodes
  for (int p = n; p < N; ++p)
    fw_set_node_width<LOG2L, K, HYPER>(F, v, w64, p, fw_width<LOG2L, K>(P, F, v.mask(p), lig, gbase), lane0);
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  Mask<K> slices;
  if (a.slices_in) {
    fw_traverse<LOG2L, K, HYPER>(P, F, v, w64, sc.rec, nullptr, lane0, gbase, st, sc.gstk);
#pragma unroll
    for (int k = 0; k < K; ++k) slices.w[k] = a.slices_in[r * a.slices_in_stride + v.widx(k)];
  } else {
    slices = fw_get_slices<LOG2L, K, HYPER>(P, F, v, w64, rng, sc, st, lpos, lane0, gbase, F.status + r);
  }
  double sum = 0;
  const double tot = fw_rebuild<LOG2L, K, HYPER>(P, v, sc.rec, slices, sc.cp, sc.pstk, lane0, gbase, &sum);
  fw_commit<LOG2L, K, HYPER>(P, v, sc.rec, sc.cp);
  uint64_t* sl = F.slices + r * 2 * (int64_t)LK;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    sl[v.widx(k)] = slices.w[k];
    sl[LK + v.widx(k)] = slices.w[k];
  }
  int mti, mtw;
  rng.finish(mti, mtw);
  if (lane0) {
    rs->mti = mti; rs->mtw = mtw;
    rs->min_cost = sum;
    rs->init_total = tot;
    a.out_total[r] = tot;
    a.out_sum[r] = sum;
  }
}

// Stage timing of a re-slicing sweep (diagnostic builds, -DTNCO_PROFILE): shader cycles of
// [walk, too-wide counts, greedy pass, rebuild + commit] and the number of re-slices, per replica, in
// ReplicaState::pad1 (tnco_hip_diag_stage_cycles; tools/stage_cycles.py --fw).
#ifdef TNCO_PROFILE
#define FW_PROF_DECL \
  unsigned long long ft_[5] = {0, 0, 0, 0, 0}, fa_[5] = {0, 0, 0, 0, 0}; \
  [[maybe_unused]] unsigned long long fc_[5] = {0, 0, 0, 0, 0}
#define FW_PROF_T(i) ft_[i] = __builtin_amdgcn_s_memtime()
#define FW_PROF_ACC                                              \
  do {                                                           \
    fa_[0] += ft_[1] - ft_[0]; fa_[1] += ft_[2] - ft_[1];        \
    fa_[2] += ft_[3] - ft_[2]; fa_[3] += ft_[4] - ft_[3];        \
    fa_[4] += 1;                                                 \
  } while (0)
#if TNCO_PROFILE == 3 || TNCO_PROFILE == 4
#define FW_PROF_OUT(rs)                                          \
  for (int k_ = 0; k_ < 4; ++k_) (rs)->pad1[k_] += fc_[k_];      \
  (rs)->pad1[4] += fa_[4]
#else
#define FW_PROF_OUT(rs) \
  for (int k_ = 0; k_ < 5; ++k_) (rs)->pad1[k_] += fa_[k_]
#endif
#else
#define FW_PROF_DECL
#define FW_PROF_T(i)
#define FW_PROF_ACC
#define FW_PROF_OUT(rs)
#endif

// update(prob, update_slices), finite_width/greedy/optimizer.hpp:117-390, is two kernels here: the
// moves of n_steps sweeps (:130-331, this one) and the re-slice at the end of a sweep (:360-376,
// fw_reslice_kernel) -- the host launches [moves up to and including a re-slicing sweep][re-slice]...
// (sweep k re-slices when (step_offset + k) % update_every == 0, tnco/app/finite_width/sa.py:228).
// Two kernels because the two phases want different register budgets: as one kernel the moves ran
// at the occupancy of the greedy pass and the greedy pass spilled.
//
// :385-389 (the best-so-far bookkeeping that ends every sweep) is done here for every sweep but,
// when `tail_last` is 0, the last: that one is followed by a re-slice, which does it afterwards.
// MAXNEW: with the max_number_new_slices > 0 branch (:226-321).
#ifndef TNCO_FW_MOVE_WAVES
#define TNCO_FW_MOVE_WAVES 2
#endif
#ifndef TNCO_FW_MAXNEW_WAVES
#define TNCO_FW_MAXNEW_WAVES 2
#endif
template <int LOG2L, int K, bool HYPER, bool MAXNEW>
__global__ __launch_bounds__(256, MAXNEW ? TNCO_FW_MAXNEW_WAVES : TNCO_FW_MOVE_WAVES) void fw_move_kernel(const Params P, const FwParams F, const double* __restrict__ betas,
                                                      const int64_t n_steps, const int prob_kind, const int tail_last) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;
  constexpr int LK = L * K;
  using M = Mask<K>;
  using R = Rng<LOG2L>;
  __shared__ uint32_t rngbuf[GPB * R::RING];
  __shared__ int32_t posbuf[MAXNEW ? GPB * FW_LDSPOS : 1];
  __shared__ uint16_t leftbuf[MAXNEW ? GPB * FW_LDSPOS : 1];
  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gib = tid >> LOG2L;
  const int gbase = (tid & 63) & ~(L - 1);
  const int64_t r = (int64_t)blockIdx.x * GPB + gib;
  if (r >= P.R || n_steps <= 0) return;
  [[maybe_unused]] const FwStack st{F.stack_cap > 0 ? (lds_vi32*)posbuf + (MAXNEW ? gib * FW_LDSPOS : 0) : nullptr,
                                    (lds_vu16*)leftbuf + (MAXNEW ? gib * FW_LDSPOS : 0), F.stack_cap};
  const bool lane0 = lig == 0;
  const int n = P.n, N = P.N;
  const int f32 = P.f32;
  View<LOG2L, K, HYPER> v;
  v.init(P, P.blocks + r * P.RB, P.lpar + r * (int64_t)n * LPS, lig);
  ReplicaState* rs = P.rs + r;
  R rng;
  rng.init(P, r, (lds_vu32*)rngbuf + gib * R::RING, rs->mti, rs->mtw, lig);
  [[maybe_unused]] const FwScratch sc(F, r, N);
  [[maybe_unused]] volatile int16_t* pos = sc.pos;
  double* w64 = F.width64 ? F.width64 + r * (int64_t)N : nullptr;
  uint64_t* sl = F.slices + r * 2 * (int64_t)LK;
  M slices;
  M skip = mzero<K>();
#pragma unroll
  for (int k = 0; k < K; ++k) slices.w[k] = sl[v.widx(k)];
  if (F.skip) {
#pragma unroll
    for (int k = 0; k < K; ++k) skip.w[k] = F.skip[v.widx(k)];
  }
  double min_cost = rs->min_cost;
  uint32_t n_moves = 0, n_acc = 0, n_impr = 0, n_rpick = 0, n_full = 0;
  int32_t* jlog = P.jlog + r * (int64_t)P.jcap;
  uint32_t jmin = rs->jmin, jtail = rs->jtail;
  bool jinvalid = rs->jinvalid != 0;
  auto log_rotation = [&](int E) {
    if (!jinvalid) {
      if (jtail == (uint32_t)P.jcap) jinvalid = true;
      else { if (lane0) jlog[jtail] = E; ++jtail; }
    }
  };
  auto uniform01 = [&]() {
    const uint32_t x1 = rng.next_sync(), x2 = rng.next_sync();
    const double s = (double)x1 + (double)x2 * 4294967296.0;
    double u = s * 5.421010862427522170037e-20;
    if (u >= 1.0) u = 0.99999999999999988897769753748;
    return u;
  };

  // The replicas of a wavefront walk leaf-to-root paths of different lengths: were the sweeps kept in
  // step, every sweep would cost each of them the longest of sixteen paths.  So every replica runs
  // its n_steps sweeps at its own pace -- one loop over moves, a replica that reaches the root closes
  // its sweep (:385-389) and starts its next one in the same iteration.
  //
  // Two dependent rounds of loads per move: [A's header, the masks and partial sums of B's children,
  // A's legs, both hyper masks] once B's header is known, [C's mask and partial sum] once A's is.
  // B's header itself is carried over from the move before (the next B is this move's A, whose new
  // header has just been written from registers).
  int64_t step = 0;
  double beta = 0.0, total = 0.0;
  int B = 0;
  NodeRec hb;
  // (a macro, not a lambda: as a lambda capturing `hb` by reference, the finite-width kernel with the
  // max_number_new_slices branch kept the struct in private memory behind a generic pointer and
  // faulted at address 0 on large networks -- tools/fuzz_gpu.py --nmin 150 --nmax 700 --new-slices 2)
#define TNCO_BEGIN_SWEEP(root_partial)                                  \
  do {                                                                  \
    beta = betas[step];                                                 \
    const int leaf_ = (int)(rng.next_sync() % (uint32_t)n); /* :130-139 */ \
    B = v.parent(leaf_);                                                \
    total = (root_partial);                                             \
    hb = *v.hdr(B);                                                     \
  } while (0)
  TNCO_BEGIN_SWEEP(v.hdr(N - 1)->partial);
  {
    for (;;) {
      // get_ctree_nn, optimize/optimizer.hpp:112-144
      const int A = hb.parent;
      if (A < 0) {  // B is the root: the sweep is over
        const double tc = hb.partial;
        if (!(step == n_steps - 1 && !tail_last)) {  // (else fw_reslice_kernel goes on from here)
          // :385-389
          if (tc < min_cost) {
            min_cost = tc;
            ++n_impr;
            if (jinvalid) {
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
#pragma unroll
            for (int k = 0; k < K; ++k) sl[LK + v.widx(k)] = slices.w[k];
          }
        }
        if (++step >= n_steps) break;
        TNCO_BEGIN_SWEEP(tc);
        continue;
      }
      const NodeRec ha = *v.hdr(A);
      const M m0 = v.mask(hb.left), m1 = v.mask(hb.right);
      const double p0 = v.partial(hb.left), p1 = v.partial(hb.right);
      const bool c_is_right = (ha.left == B);
      int C = c_is_right ? ha.right : ha.left;
      const M mC = v.mask(C);
      // hyper[p] = legs(p) & legs(c0) & legs(c1) (infinite_memory/utils.hpp:82-91): derived, not stored (sa_kernels.h, Params)
      const M iA = v.mask(A), iB = HYPER ? v.mask(B) : mzero<K>();
      const M hB = HYPER ? mand<K>(iB, mand<K>(m0, m1)) : mzero<K>(), hA = HYPER ? mand<K>(iA, mand<K>(iB, mC)) : mzero<K>();
      const double pC = v.partial(C);
      const uint32_t w = gsum<LOG2L>((mnonzero<K>(mand<K>(m0, mC)) ? 1u : 0u) |
                                     ((mnonzero<K>(mand<K>(m1, mC)) ? 1u : 0u) << 8));
      const bool inter0 = (w & 0xffu) != 0, inter1 = (w >> 8) != 0;
      bool pick0;
      if (P.disable_shared || (inter0 && inter1)) {
        pick0 = (rng.next_sync() & 1u) != 0;
        ++n_rpick;
      } else {
        pick0 = inter0;
      }
      int E = pick0 ? hb.right : hb.left;
      const M mD = msel<K>(pick0, m0, m1), mE = msel<K>(pick0, m1, m0);
      const double pD = pick0 ? p0 : p1, pE = pick0 ? p1 : p0;
      // :174-179
      const M newB = mor<K>(mor<K>(mxor<K>(mD, mC), hA), hB);
      const double new_width_B = fw_width<LOG2L, K>(P, F, newB, lig, gbase);
      double new_sliced_width_B = fw_width<LOG2L, K>(P, F, mandn<K>(newB, slices), lig, gbase);
      double ccB = hb.ccost, ccA = ha.ccost;
      int bl = hb.left, br = hb.right, al = ha.left, ar = ha.right;
      bool acc = false, skip_cost_propagation = false;
      ++n_moves;
      if (new_sliced_width_B <= F.max_width) {
        // :190-201
        const double nA = generic_cost<LOG2L, K>(P, mor<K>(mor<K>(newB, mE), slices), lig, gbase);
        const double nB = generic_cost<LOG2L, K>(P, mor<K>(mor<K>(mD, mC), slices), lig, gbase);
        const double delta = rnd_cost(rnd_cost(nB - ccB, f32) + rnd_cost(nA - ccA, f32), f32);
        const double u = uniform01();
        acc = accept_move(prob_kind, beta, delta, total, u, f32);
        if (acc) {
          ++n_acc;
          // :203-219
          if (pick0) br = C; else bl = C;
          if (c_is_right) ar = E; else al = E;
          log_rotation(E);
          if (lane0) {
            v.set_parent(C, B);
            v.set_parent(E, A);
          }
          v.set_mask(B, newB);
          ccB = nB;
          ccA = nA;
          total = rnd_cost(total + delta, f32);
          const int t = C; C = E; E = t;
        }
      } else if constexpr (MAXNEW) {
       if (F.max_new_slices > 0) {
        // :226-321  slice up to max_number_new_slices random further legs of the new B; if it then
        // fits, try the rotation against a FULL rebuild of the cost cache with the new slices
        M new_slices = slices;
        uint32_t n_pos = fw_positions<LOG2L, K, HYPER>(v, mandn<K>(mandn<K>(newB, slices), skip), pos,
                                                       (uint32_t)F.I64, gbase, lane0, F.status + r);
        int64_t n_new = 0;
        while (n_new < F.max_new_slices && new_sliced_width_B > F.max_width && n_pos > 0) {
          const uint32_t j = rng.next_sync() % n_pos;  // :245
          const int16_t pj = pos[j], pl = pos[n_pos - 1];
          if (lane0) { pos[j] = pl; pos[n_pos - 1] = pj; }
          __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const int xp = pj;
          fw_flip<LOG2L, K, HYPER>(v, new_slices, xp);
          new_sliced_width_B = fw_wr(F, new_sliced_width_B - fw_log2dim(F, xp));
          --n_pos;
          ++n_new;
        }
        if (new_sliced_width_B <= F.max_width) {
          // :287-290  B takes its new legs, rotate, rebuild
          const M oldB = v.mask(B);
          if (lane0) {
            NodeRec o = hb, oa = ha;
            if (pick0) o.right = C; else o.left = C;
            if (c_is_right) oa.right = E; else oa.left = E;
            *v.hdr(B) = o;
            *v.hdr(A) = oa;
            v.set_parent(C, B);
            v.set_parent(E, A);
          }
          v.set_mask(B, newB);
          __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
          fw_traverse<LOG2L, K, HYPER>(P, F, v, w64, sc.rec, nullptr, lane0, gbase, st, sc.gstk);
          double sum;
          const double tot = fw_rebuild<LOG2L, K, HYPER>(P, v, sc.rec, new_slices, sc.cp, sc.pstk, lane0, gbase, &sum);
          const double delta = rnd_cost(tot - total, f32);
          const double u = uniform01();
          if (accept_move(prob_kind, beta, delta, total, u, f32)) {
            // :296-312
            fw_commit<LOG2L, K, HYPER>(P, v, sc.rec, sc.cp);
            fw_set_node_width<LOG2L, K, HYPER>(F, v, w64, B, new_width_B, lane0);
            total = tot;
            slices = new_slices;
            skip_cost_propagation = true;
            log_rotation(E);
            ++n_acc;
          } else {
            // :317-318  undo
            if (lane0) {
              *v.hdr(B) = hb;
              *v.hdr(A) = ha;
              v.set_parent(C, A);
              v.set_parent(E, B);
            }
            v.set_mask(B, oldB);
          }
          __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
       }
      }
      // :324-331
      if (!skip_cost_propagation) {
        // (an accepted move has swapped C and E: B holds D and the old C, A holds B and the old E)
        const double partB = rnd_cost(rnd_cost(pD + (acc ? pC : pE), f32) + ccB, f32);
        const double partA = rnd_cost(rnd_cost(partB + (acc ? pE : pC), f32) + ccA, f32);
        NodeRec oa;
        oa.left = al; oa.right = ar; oa.parent = ha.parent; oa.pad = ha.pad; oa.ccost = ccA; oa.partial = partA;
        if (lane0) {
          NodeRec o;
          o.left = bl; o.right = br; o.parent = A; o.ccost = ccB; o.partial = partB;
          o.pad = (acc && F.width_f32) ? __float_as_int((float)new_width_B) : hb.pad;
          *v.hdr(B) = o;
          *v.hdr(A) = oa;
          if (acc && !F.width_f32) w64[B] = new_width_B;
        }
        hb = oa;
        // (no wait for the stores: the next move's loads come after them in this wavefront's
        // instruction stream, and memory operations of one wavefront to the same address stay in order)
      } else {
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        hb = *v.hdr(A);  // (rebuilt by fw_commit)
      }
      B = A;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) sl[v.widx(k)] = slices.w[k];
  int mti, mtw;
  rng.finish(mti, mtw);
  if (lane0) {
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
  }
}

// finite_width/greedy/optimizer.hpp:385-389, the end of a sweep, from the replica state in memory
template <int LOG2L, int K, bool HYPER>
__device__ __forceinline__ void fw_sweep_tail(const Params& P, const View<LOG2L, K, HYPER>& v, ReplicaState* rs,
                                              int64_t r, uint64_t* sl, const Mask<K>& slices, bool lane0) {
  constexpr int L = 1 << LOG2L;
  constexpr int LK = L * K;
  const int N = P.N, lig = v.lig;
  // :385-389
  const double tc = v.hdr(N - 1)->partial;
  if (tc < rs->min_cost) {
    uint32_t jtail = rs->jtail;
    if (rs->jinvalid != 0) {
      Links* __restrict__ ml = P.minlinks + r * (int64_t)N;
      for (int i = lig; i < N; i += L) {
        Links o;
        o.left = v.left(i); o.right = v.right(i); o.parent = v.parent(i); o.pad = 0;
        ml[i] = o;
      }
      jtail = 0;
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every lane has read rs->jinvalid)
      if (lane0) {
        rs->jtail = 0;
        rs->jinvalid = 0;
        rs->n_fullcopy += 1;
      }
    }
    if (lane0) {
      rs->min_cost = tc;
      rs->n_improved += 1;
      rs->jmin = jtail;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) sl[LK + v.widx(k)] = slices.w[k];
  }
}

// The end of a re-slicing sweep, finite_width/greedy/optimizer.hpp:360-389: new slices for the
// current tree (get_slices), the cost cache rebuilt with them, kept if the total improves; then the
// best-so-far bookkeeping of the sweep (:385-389).
#ifndef TNCO_FW_RESLICE_WAVES
#define TNCO_FW_RESLICE_WAVES 2
#endif
template <int LOG2L, int K, bool HYPER>
__global__ __launch_bounds__(256, TNCO_FW_RESLICE_WAVES) void fw_reslice_kernel(const Params P, const FwParams F, const int prewalked) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;
  constexpr int LK = L * K;
  using M = Mask<K>;
  using R = Rng<LOG2L, 64>;  // (a ring of 64 outputs: the lock-step refills of fw_shuffle_lds)
  __shared__ uint32_t rngbuf[GPB * R::RING];
  __shared__ __attribute__((aligned(8))) int32_t posbuf[GPB * FW_LDSPOS];  // candidate legs / partial sums
  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gib = tid >> LOG2L;
  const int gbase = (tid & 63) & ~(L - 1);
  const int64_t r = (int64_t)blockIdx.x * GPB + gib;
  if (r >= P.R) return;
  lds_vi32* lpos = (lds_vi32*)posbuf + gib * FW_LDSPOS;
  // (the walk is fw_walk2_kernel's; without it -- trees of more than 8192 nodes, test knobs -- the
  // links are walked in place, no LDS stack)
  const FwStack st{nullptr, nullptr, 0};
  const bool lane0 = lig == 0;
  const int n = P.n, N = P.N;
  View<LOG2L, K, HYPER> v;
  v.init(P, P.blocks + r * P.RB, P.lpar + r * (int64_t)n * LPS, lig);
  ReplicaState* rs = P.rs + r;
  FwScratch sc(F, r, N);
  if (prewalked == 2) sc.nwf = F.nwfront[r];  // (fw_walk2_kernel: the list in two parts)
  double* w64 = F.width64 ? F.width64 + r * (int64_t)N : nullptr;
  uint64_t* sl = F.slices + r * 2 * (int64_t)LK;
  M slices;
#pragma unroll
  for (int k = 0; k < K; ++k) slices.w[k] = sl[v.widx(k)];
  const int nw_pre = prewalked ? F.nwide[r] : -1;
  FW_PROF_DECL;
  FW_PROF_T(1);
  if (gany<LOG2L>(mnonzero<K>(slices))) {
    R rng;
    rng.init(P, r, (lds_vu32*)rngbuf + gib * R::RING, rs->mti, rs->mtw, lig);
#if defined(TNCO_PROFILE) && (TNCO_PROFILE == 3 || TNCO_PROFILE == 4)  // event counts / greedy-pass cycles instead
    const M ns = fw_get_slices<LOG2L, K, HYPER>(P, F, v, w64, rng, sc, st, lpos, lane0, gbase, F.status + r, nullptr, fc_, nw_pre);
#elif defined(TNCO_PROFILE)
    unsigned long long fp_[2] = {0, 0};  // end of the walk, end of the counts
    const M ns = fw_get_slices<LOG2L, K, HYPER>(P, F, v, w64, rng, sc, st, lpos, lane0, gbase, F.status + r, fp_, nullptr, nw_pre);
    ft_[2] = fp_[0];
    ft_[0] = ft_[1];                  // slot 0: the walk (from the start of the re-slice)
    ft_[1] = ft_[2];                  // slot 1: too-wide counts
    ft_[2] = fp_[1];                  // slot 2: the greedy pass (up to T(3))
#else
    const M ns = fw_get_slices<LOG2L, K, HYPER>(P, F, v, w64, rng, sc, st, lpos, lane0, gbase, F.status + r, nullptr, nullptr, nw_pre);
#endif
    FW_PROF_T(3);
    double sum;
    // (LDS of the candidate list, free now: 16 partial sums + as many leg masks as fit behind them)
    constexpr int PCAP = 16, MCAP = (FW_LDSPOS / 2 - PCAP) / LK;
    const double tot = fw_rebuild<LOG2L, K, HYPER>(P, v, sc.rec, ns, sc.cp, sc.pstk, lane0, gbase, &sum,
                                                   (lds_vdouble*)lpos, PCAP, (lds_vu64*)lpos + PCAP, MCAP);
    if (tot < v.hdr(N - 1)->partial) {
      slices = ns;
      fw_commit<LOG2L, K, HYPER>(P, v, sc.rec, sc.cp);
#pragma unroll
      for (int k = 0; k < K; ++k) sl[v.widx(k)] = slices.w[k];
    }
    int mti, mtw;
    rng.finish(mti, mtw);
    if (lane0) {
      rs->mti = mti;
      rs->mtw = mtw;
    }
    FW_PROF_T(4);
    FW_PROF_ACC;
  }
  fw_sweep_tail<LOG2L, K, HYPER>(P, v, rs, r, sl, slices, lane0);
  if (lane0) {
    FW_PROF_OUT(rs);
  }
}

// ---------------------------------------------------------------------------------------------
// The re-slice by RE-PRICING when the costs are powers of two (uniform dims 2^k, no sparse legs, float64):
// fw_wave_kernel below -- one wavefront per replica: get_slices, then the cost cache re-priced from the old
// costs (the new slices S' differ from the old S in a handful of indices; a node's contraction cost is
// 2^(k |u | S|), u = legs(left) | legs(right), and |u | S'| - |u | S| = sum over the changed indices d of
// (+1 if d joined, -1 if d left) * [d not in u]).  The replicas it leaves alone: get_slices by
// fw_reslice_a_kernel (lock-step, its own traverse), the full rebuild + the end of the sweep by
// fw_reslice_b_kernel.
// ---------------------------------------------------------------------------------------------
// changed indices the re-pricing handles, 32 per pass over the paths (more: the full rebuild).  96 instead of 64 in
// the lean configuration since round 4: on config 5 at max_width 32 one re-slice in 10^4 changes more than 64
// indices (99.99 % at most 62, none above 85 in 7.8 M: tools/fw_changed_hist.py), and each of those few held
// the whole batch up for the 0.5-1.7 ms of a lock-step full rebuild: +4 ... +9 % on the bench's schedule (A/B on
// one box, three times each).  96 keeps the 536-tensor network at 10 240 B of LDS, 16 wavefronts per CU; 128
// (10 400 B, 15 per CU) gained nothing more.  Late in a schedule, where no re-slice changes that much, the
// limits do not differ beyond the +-3 % between two runs.
#ifndef TNCO_FWT_MAXD
#define TNCO_FWT_MAXD 96  // (-DTNCO_FWT_MAXD=64: the earlier limit, for A/B runs)
#endif
#ifndef TNCO_FWT_MAXD_BIG
#define TNCO_FWT_MAXD_BIG 256  // (the roomier configuration: networks sliced far below their width change more at once)
#endif
template <bool BIG> constexpr int FWT_MAXD = BIG ? TNCO_FWT_MAXD_BIG : TNCO_FWT_MAXD;

#ifndef TNCO_FW_RESLICE_A_WAVES
#define TNCO_FW_RESLICE_A_WAVES TNCO_FW_RESLICE_WAVES
#endif
template <int LOG2L, int K, bool HYPER>
__global__ __launch_bounds__(256, TNCO_FW_RESLICE_A_WAVES) void fw_reslice_a_kernel(const Params P, const FwParams F, const int prewalked) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;
  constexpr int LK = L * K;
  using M = Mask<K>;
  using R = Rng<LOG2L, 64>;
  __shared__ uint32_t rngbuf[GPB * R::RING];
  __shared__ __attribute__((aligned(8))) int32_t posbuf[GPB * FW_LDSPOS];
  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gib = tid >> LOG2L;
  const int gbase = (tid & 63) & ~(L - 1);
  const int64_t r = (int64_t)blockIdx.x * GPB + gib;
  if (r >= P.R) return;
  lds_vi32* lpos = (lds_vi32*)posbuf + gib * FW_LDSPOS;
  const FwStack st{nullptr, nullptr, 0};
  const bool lane0 = lig == 0;
  const int n = P.n, N = P.N;
  View<LOG2L, K, HYPER> v;
  v.init(P, P.blocks + r * P.RB, P.lpar + r * (int64_t)n * LPS, lig);
  ReplicaState* rs = P.rs + r;
  FwScratch sc(F, r, N);
  if (prewalked && F.nwide[r] == -3) return;  // (fw_wave_kernel has done this replica)
  // (-2: fw_wave_kernel has left this replica's too-wide tensors to the traverse in here)
  const int nw_pre = prewalked ? (F.nwide[r] == -2 ? -1 : F.nwide[r]) : -1;
  if (prewalked == 2 && nw_pre >= 0) sc.nwf = F.nwfront[r];
  double* w64 = F.width64 ? F.width64 + r * (int64_t)N : nullptr;
  const uint64_t* sl = F.slices + r * 2 * (int64_t)LK;
  M slices;
#pragma unroll
  for (int k = 0; k < K; ++k) slices.w[k] = sl[v.widx(k)];
  if (!gany<LOG2L>(mnonzero<K>(slices))) return;  // greedy/optimizer.hpp:359
  R rng;
  rng.init(P, r, (lds_vu32*)rngbuf + gib * R::RING, rs->mti, rs->mtw, lig);
  const M ns = fw_get_slices<LOG2L, K, HYPER>(P, F, v, w64, rng, sc, st, lpos, lane0, gbase, F.status + r, nullptr, nullptr,
                                              nw_pre);
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last reads of the position scratch)
  // the proposed slices travel in the candidate-position scratch, free now
  uint64_t* prop = reinterpret_cast<uint64_t*>(const_cast<int16_t*>(sc.pos));
#pragma unroll
  for (int k = 0; k < K; ++k) prop[v.widx(k)] = ns.w[k];
  if (lane0) reinterpret_cast<uint32_t*>(F.delta_scr + r * 64)[0] = 0xFFFFFFFFu;  // (tnco_hip_diag_reslice_info: not re-priced)
  int mti, mtw;
  rng.finish(mti, mtw);
  if (lane0) {
    rs->mti = mti;
    rs->mtw = mtw;
  }
}

#include "fw_wave.h"  // fw_wave_kernel: the whole re-slice of a replica in one wavefront

#ifndef TNCO_FW_RESLICE_B_WAVES
#define TNCO_FW_RESLICE_B_WAVES TNCO_FW_RESLICE_WAVES
#endif
template <int LOG2L, int K, bool HYPER>
__global__ __launch_bounds__(256, TNCO_FW_RESLICE_B_WAVES) void fw_reslice_b_kernel(const Params P, const FwParams F, const int need_rec) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;
  constexpr int LK = L * K;
  using M = Mask<K>;
  __shared__ __attribute__((aligned(8))) int32_t posbuf[GPB * FW_LDSPOS];
  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gib = tid >> LOG2L;
  const int gbase = (tid & 63) & ~(L - 1);
  const int64_t r = (int64_t)blockIdx.x * GPB + gib;
  if (r >= P.R) return;
  lds_vi32* lpos = (lds_vi32*)posbuf + gib * FW_LDSPOS;
  const bool lane0 = lig == 0;
  const int n = P.n, N = P.N;
  View<LOG2L, K, HYPER> v;
  v.init(P, P.blocks + r * P.RB, P.lpar + r * (int64_t)n * LPS, lig);
  ReplicaState* rs = P.rs + r;
  const FwScratch sc(F, r, N);
  uint64_t* sl = F.slices + r * 2 * (int64_t)LK;
  M slices;
#pragma unroll
  for (int k = 0; k < K; ++k) slices.w[k] = sl[v.widx(k)];
  if (F.fastflag[r] == 0 && gany<LOG2L>(mnonzero<K>(slices))) {  // (not re-priced: rebuilt in full)
    const uint64_t* prop = reinterpret_cast<const uint64_t*>(const_cast<const int16_t*>(sc.pos));
    M ns;
#pragma unroll
    for (int k = 0; k < K; ++k) ns.w[k] = prop[v.widx(k)];
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (need_rec) {  // (no walk kernel ran: the post-order of this replica, here)
      const FwStack st{nullptr, nullptr, 0};
      fw_traverse<LOG2L, K, HYPER>(P, F, v, F.width64 ? F.width64 + r * (int64_t)N : nullptr, sc.rec, nullptr, lane0, gbase,
                                   st, sc.gstk);
    }
    double sum;
    constexpr int PCAP = 16, MCAP = (FW_LDSPOS / 2 - PCAP) / LK;
    const double tot = fw_rebuild<LOG2L, K, HYPER>(P, v, sc.rec, ns, sc.cp, sc.pstk, lane0, gbase, &sum,
                                                   (lds_vdouble*)lpos, PCAP, (lds_vu64*)lpos + PCAP, MCAP);
    if (tot < v.hdr(N - 1)->partial) {
      slices = ns;
      fw_commit<LOG2L, K, HYPER>(P, v, sc.rec, sc.cp);
#pragma unroll
      for (int k = 0; k < K; ++k) sl[v.widx(k)] = slices.w[k];
    }
  }
  fw_sweep_tail<LOG2L, K, HYPER>(P, v, rs, r, sl, slices, lane0);
}

// is_valid of the finite-width optimizer, the part on top of the infinite-memory checks
// (finite_width/greedy/optimizer.hpp:404-444): every tensor of the tree in `ref` (blocks rebuilt
// from scratch by build_kernel) fits max_width once the sliced indices are removed, and -- for the
// current tree -- the cached widths are the recomputed ones.
template <int LOG2L, int K, bool HYPER>
__global__ __launch_bounds__(256) void fw_check_kernel(const Params P, const FwParams F, const BuildArgs a,
                                                       const int which_min, const double atol, int32_t* out_bad) {
  constexpr int L = 1 << LOG2L;
  constexpr int GPB = 256 >> LOG2L;
  constexpr int LK = L * K;
  const int tid = threadIdx.x;
  const int lig = tid & (L - 1);
  const int gbase = (tid & 63) & ~(L - 1);
  const int64_t q = (int64_t)blockIdx.x * GPB + (tid >> LOG2L);
  if (q >= a.count) return;
  const int64_t r = a.r0 + q;
  const int n = P.n, N = P.N;
  View<LOG2L, K, HYPER> ref, cur;
  ref.init(P, a.out_blocks + q * P.RB, a.out_lpar + q * (int64_t)n * LPS, lig);
  cur.init(P, P.blocks + r * P.RB, P.lpar + r * (int64_t)n * LPS, lig);
  const double* w64 = F.width64 ? F.width64 + r * (int64_t)N : nullptr;
  Mask<K> sl;
#pragma unroll
  for (int k = 0; k < K; ++k) sl.w[k] = F.slices[r * 2 * (int64_t)LK + (which_min ? LK : 0) + ref.widx(k)];
  int bad = 0;
  for (int t = 0; t < N; ++t) {
    const Mask<K> m = ref.mask(t);
    if (fw_width<LOG2L, K>(P, F, mandn<K>(m, sl), lig, gbase) > F.max_width) bad = bad ? bad : 35;
    if (!which_min && t >= n) {
      const double w = fw_width<LOG2L, K>(P, F, m, lig, gbase);
      const double c = F.width_f32 ? (double)__int_as_float(cur.hdr(t)->pad) : w64[t];
      if (!(fabs(w - c) <= atol)) bad = bad ? bad : 36;
    }
  }
  if (lig == 0 && bad && out_bad[q] == 0) out_bad[q] = bad;
}

}  // namespace tnco
