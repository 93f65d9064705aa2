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
//   * the general one: a walk kernel (fw_walk2_kernel from both ends of the post-order, fw_walk_kernel) leaves
//     sequential lists (internal nodes in post-order with their links; the too-wide tensors in post-order), then
//     fw_reslice_kernel iterates over them -- the k-th entry of all 16 replicas of a wavefront together, the next
//     entries' loads in flight; counters, candidate lists, shuffles and picks in registers / LDS; the cache
//     rebuilt from the leg masks;
//   * when every cost is a power of two (fast_ok / tree_ok): no walk -- fw_order_kernel | get_slices
//     (fw_slices_kernel, one wavefront per replica; its stragglers and networks of more than 16 mask words:
//     fw_reslice_a_kernel) | fw_tree_kernel (the cache RE-PRICED from the old costs) | fw_reslice_b_kernel; see
//     the comment blocks at fw_order_kernel and fw_slices_kernel.  (fw_delta_kernel: the re-pricing over the walk's post-order records,
//     round 2, behind TNCO_HIP_FW_NO_TREE.)
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
template <int LOG2L, typename RNG>
__device__ __forceinline__ void fw_shuffle_lds(RNG& rng, lds_vi32* a, int n, bool lane0) {
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
// read it with the links: fw_delta_kernel re-prices the node from it), 0 where nobody recorded it.
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

#if (defined(TNCO_PROFILE) && TNCO_PROFILE == 4) || defined(TNCO_FWA_PROF)  // cycles inside the greedy pass: [scan, positions, shuffle, keys + picks]
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
  // (nw_pre >= 0: fw_walk_kernel has left sc.rec / sc.wlist and this count)
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

// The walk of the re-slice as a kernel of its own, ONE LANE per replica (fw_reslice_kernel then
// starts from the lists).  The walk is scalar work per replica -- links and cached widths, no leg
// masks -- and a chain of dependent header reads: with four lanes per replica it ran 16 chains per
// wavefront and every lane of a group repeated the same instructions.  Here a wavefront runs 64
// chains, so the same number of resident wavefronts keeps four times as many header reads in
// flight (the chip retires ~47 G random reads/s, tools/hbm_random.hip; the 4-lane walk reached 25 G/s)
// at a quarter of the instructions.  Same step structure as fw_traverse: one fetch per iteration, an
// up-step before and after it; stack entries lane-interleaved in LDS (no bank conflicts whatever
// the depths), the deep end in global scratch.  Trees of at most 8192 nodes (13-bit stack fields).
// Measured and rejected (config 5, 65536 replicas, ms per walk): this loop 1.41; up to 3 or 4 up-steps
// per iteration 1.65 / 1.68; both children's headers requested together, the right one's kept in
// the stack entry until its subtree is entered (half the dependent reads, six actions per
// iteration) 3.85 -- the loop is bound by its instructions under 32- / 64-way divergence, not by
// the reads; 64 busy lanes per wavefront instead of 32: 1.44.
#ifndef TNCO_FW_WALK_CAP
#define TNCO_FW_WALK_CAP 40
#endif
#ifndef TNCO_FW_WALK_POPS
#define TNCO_FW_WALK_POPS 1
#endif
#ifndef TNCO_FW_WALK_LANES
#define TNCO_FW_WALK_LANES 32
#endif
constexpr int FW_WALK_CAP = TNCO_FW_WALK_CAP;
// Only FW_WALK_LANES lanes of a wavefront carry a replica: with all 64 busy, 65536 replicas are one
// wavefront per SIMD, and nothing overlaps that wavefront's LDS / ALU latency with its memory waits;
// half-empty wavefronts give every SIMD two.
constexpr int FW_WALK_LANES = TNCO_FW_WALK_LANES;
constexpr int FW_WALK_PER_BLOCK = 4 * FW_WALK_LANES;  // replicas per 256-thread block

static __global__ __launch_bounds__(256) void fw_walk_kernel(const Params P, const FwParams F) {
  constexpr int NT = FW_WALK_PER_BLOCK;  // replicas (busy lanes) per block: the stride of the LDS arrays
  __shared__ int32_t se[FW_WALK_CAP * NT];
  __shared__ uint16_t sl[FW_WALK_CAP * NT];
  // The lists leave through LDS, eight entries at a time: a lane's 8-byte stores to its own list
  // reached the memory one by one, as partial writes (rocprofv3: 660 write requests per replica, 83 %
  // of them 32-byte ones -- the header reads turn the L2 over long before a line of records is full);
  // eight records are four back-to-back 16-byte stores of one 64-byte piece.
  __shared__ uint32_t rbuf[16 * NT];  // 8 records (low, high words) per replica
  __shared__ int32_t wbuf[8 * NT];    // 8 too-wide tensors per replica
  const int tid = threadIdx.x;
  if ((tid & 63) >= FW_WALK_LANES) return;
  const int slot = (tid >> 6) * FW_WALK_LANES + (tid & 63);
  const int64_t r = (int64_t)blockIdx.x * FW_WALK_PER_BLOCK + slot;
  if (r >= P.R) return;
  const int n = P.n, N = P.N, LK = F.I64 / 64;
  {  // greedy/optimizer.hpp:359: nothing to do without slices
    const uint64_t* sl0 = F.slices + r * 2 * (int64_t)LK;
    uint64_t any = 0;
    for (int w = 0; w < P.W; ++w) any |= sl0[w];
    if (!any) {
      F.nwide[r] = -1;
      return;
    }
  }
  const FwScratch sc(F, r, N);
  const uint8_t* blk = P.blocks + r * P.RB;
  const double* w64 = F.width64 ? F.width64 + r * (int64_t)N : nullptr;
  uint64_t* rec = sc.rec;  // (64-byte aligned: fw_np, fw_scratch_ints)
  int32_t* wlist = sc.wlist;
  volatile int32_t* gstk = sc.gstk;
  TNCO_LDS volatile int32_t* e_ = (TNCO_LDS volatile int32_t*)se + slot;
  TNCO_LDS volatile uint16_t* l_ = (TNCO_LDS volatile uint16_t*)sl + slot;
  TNCO_LDS volatile uint32_t* rb = (TNCO_LDS volatile uint32_t*)rbuf + slot;
  TNCO_LDS volatile int32_t* wb = (TNCO_LDS volatile int32_t*)wbuf + slot;
  const int gh = (N + 1) / 2;
  int ni = 0, nw = 0;
  auto put_wide = [&](int x) {
    wb[(nw & 7) * NT] = x;
    ++nw;
    if ((nw & 7) == 0) {
      int4* d = reinterpret_cast<int4*>(wlist + nw - 8);
      d[0] = make_int4(wb[0], wb[NT], wb[2 * NT], wb[3 * NT]);
      d[1] = make_int4(wb[4 * NT], wb[5 * NT], wb[6 * NT], wb[7 * NT]);
    }
  };
  auto put_rec = [&](uint64_t x) {
    rb[(2 * (ni & 7)) * NT] = (uint32_t)x;
    rb[(2 * (ni & 7) + 1) * NT] = (uint32_t)(x >> 32);
    ++ni;
    if ((ni & 7) == 0) {
      uint4* d = reinterpret_cast<uint4*>(rec + ni - 8);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        d[q] = make_uint4(rb[(4 * q) * NT], rb[(4 * q + 1) * NT], rb[(4 * q + 2) * NT], rb[(4 * q + 3) * NT]);
    }
  };
  auto emit_leaf = [&](int x) {
    if (F.leaf_wide && ((F.leaf_bits[x >> 5] >> (x & 31)) & 1u)) put_wide(x);
  };
  int sp = 0, x = N - 1;
  bool done = false;
  if (x < n) {
    emit_leaf(x);
    done = true;
  }
  auto up = [&]() {
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
    } else {
      if (fresh) emit_leaf(rr);
      --sp;
      put_rec(fw_rec(node, l, rr));
      if ((e >> 27) & 1) put_wide(node);
      if (sp == 0) done = true;
    }
  };
  while (!done) {
    if (x < 0) up();
    if (x >= n) {  // down: the only read of this node's header (links + cached width, one line)
      const int4 h = *reinterpret_cast<const int4*>(blk + (int64_t)(x - n) * P.BS);
      const double w = F.width_f32 ? (double)__int_as_float(h.w) : w64[x];
      const bool wide = w > F.max_width;
      ++sp;
      const int e = x | (h.y << 13) | (wide ? (1 << 27) : 0);
      if (sp <= FW_WALK_CAP) {
        e_[(sp - 1) * NT] = e;
        l_[(sp - 1) * NT] = (uint16_t)h.x;
      } else {
        gstk[sp - 1 - FW_WALK_CAP] = e;
        gstk[gh + sp - 1 - FW_WALK_CAP] = h.x;
      }
      x = h.x;
      if (x < n) {
        emit_leaf(x);
        x = -1;
      }
    }
    // (up-steps are LDS only: a few per iteration, so that nearly every iteration of the wavefront --
    // one memory latency each -- fetches a header for every lane)
#pragma unroll 1
    for (int q = 0; q < TNCO_FW_WALK_POPS && !done && x < 0; ++q) up();
  }
  // the unfinished pieces of the lists
  for (int k = ni & ~7; k < ni; ++k)
    rec[k] = (uint64_t)rb[(2 * (k & 7)) * NT] | ((uint64_t)rb[(2 * (k & 7) + 1) * NT] << 32);
  for (int k = nw & ~7; k < nw; ++k) wlist[k] = wb[(k & 7) * NT];
  F.nwide[r] = nw;
}

// The same walk from BOTH ends of the post-order, two lanes per replica (lanes 0-31 of a wavefront walk
// forward, lane 32 + i walks backward for the replica of lane i).  The one-ended walk keeps one header
// read in flight per replica and sits at half of the chip's request rate; the reverse of a post-order
// is the pre-order that takes the RIGHT child first, so a second walker can emit the list from its end:
// node x at the moment its header arrives (position n - 2 - b for the b-th node), right subtree next,
// the left child waiting on a small stack.  Both fill the same rec[]; the too-wide tensors of the
// backward walker go to the END of wlist (FwScratch::wl).  The two lanes run in lock-step and exchange
// their counts every iteration: with `left` = nodes not yet listed, the backward walker lists one more
// only while left >= 2, the forward walker gets the rest -- every node exactly once.  A backward
// walker whose stack outgrows its LDS entries stops for good (the forward walker finishes alone).
// Not for trees with too-wide LEAVES (F.leaf_wide: their place in the list is the forward walker's
// business): the host launches fw_walk_kernel for those.
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
  // ---- forward walker (as fw_walk_kernel) ----
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
// ReplicaState::pad1 (tnco_hip_get_stage_cycles; tools/stage_cycles.py --fw).
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
      const M iA = v.mask(A), hA = v.hyper(A), hB = v.hyper(B);
      const bool c_is_right = (ha.left == B);
      int C = c_is_right ? ha.right : ha.left;
      const M mC = v.mask(C);
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
          v.set_hyper(A, mand<K>(mand<K>(iA, newB), mE));
          v.set_hyper(B, mand<K>(mand<K>(newB, mD), mC));
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
            v.set_hyper(A, mand<K>(mand<K>(iA, newB), mE));
            v.set_hyper(B, mand<K>(mand<K>(newB, mD), mC));
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
  // (the walk is fw_walk_kernel's; without it -- trees of more than 8192 nodes, test knobs -- the
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
// The re-slice as three kernels when the costs are powers of two (uniform dims 2^k, no sparse legs, no
// hyper-indices, float64): get_slices (phase A, four lanes per replica as before) | the cache rebuild
// as a RE-PRICING of the old costs, one lane per replica (fw_delta_kernel) | the end of the sweep (phase
// B, which also does the full rebuild of the few replicas the re-pricing has left).
//
// Why: from the reference's greedy starts four fifths of a re-slice were the rebuild -- the legs of
// all n - 1 nodes re-derived (16 words each) only to count |legs(left) | legs(right) | slices| again,
// although the new slices S' differ from the old S in a handful of indices.  A node's contraction cost
// is 2^(k |u | S|), u = legs(left) | legs(right), and |u | S'| - |u | S| = sum over the changed
// indices d of (+1 if d joined, -1 if d left) * [d not in u].  An index held by two tensors is a leg
// of a subtree exactly when the subtree holds ONE of them, so per subtree a count vector suffices
// (two bits per changed index: holders inside), added up the post-order like the partial sums.  The
// walk kernels record the exponent field of every node's old cost next to its links (the same 32-byte
// header): the new cost is that exponent moved, a double built from bits -- exact, like the power of
// two the full rebuild computes.  No leg masks, no leaf table, no reductions: a stack machine of
// scalars, 64 replicas per wavefront; the total first, and only if it improves a second pass that
// writes (cost, partial sum) straight into the node headers (no scratch list, no scatter pass).
// ---------------------------------------------------------------------------------------------
constexpr int FWT_MAXD = 64;   // changed indices handled by fw_tree_kernel (below), 32 per pass over the paths (more: the full rebuild)
constexpr int FWD_MAXD = 64;   // changed indices per re-slice handled here (more: the full rebuild); beyond 32 a
                               // second count-vector word joins in (early in a schedule, one re-slice in 2 000)
#ifdef TNCO_FW_DELTA_STATS  // (diagnostic build: how many indices change, why replicas take the full rebuild)
__device__ unsigned long long g_fwd_stats[80];
#define FWD_STAT(i) atomicAdd(&g_fwd_stats[i], 1ull)
#else
#define FWD_STAT(i)
#endif
constexpr int FWD_STK = 24;    // stack entries in LDS per replica (deeper: global scratch)
constexpr int FWD_BITW = 32;   // words of the per-replica leaf bitmap: <= 1024 tensors
#ifndef TNCO_FWD_LANES
#define TNCO_FWD_LANES 32
#endif
constexpr int FWD_LANES = TNCO_FWD_LANES;  // replicas (busy lanes) per wavefront of fw_delta_kernel

#ifdef TNCO_FWA_PROF  // (diagnostic build: shader cycles per wavefront of [generator init, too-wide counts, greedy pass], wavefronts, too-wide tensors of lane 0's replica)
static __device__ unsigned long long g_fwa_prof[8];
#endif
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
  if (prewalked && F.nwide[r] == -3) return;  // (fw_slices_kernel has done this replica)
  // (-2: fw_order_kernel has left this replica's too-wide tensors to the traverse in here)
  const int nw_pre = prewalked ? (F.nwide[r] == -2 ? -1 : F.nwide[r]) : -1;
  if (prewalked == 2 && nw_pre >= 0) sc.nwf = F.nwfront[r];
  double* w64 = F.width64 ? F.width64 + r * (int64_t)N : nullptr;
  const uint64_t* sl = F.slices + r * 2 * (int64_t)LK;
  M slices;
#pragma unroll
  for (int k = 0; k < K; ++k) slices.w[k] = sl[v.widx(k)];
  if (!gany<LOG2L>(mnonzero<K>(slices))) return;  // greedy/optimizer.hpp:359
#ifdef TNCO_FWA_PROF
  const unsigned long long ta0 = __builtin_amdgcn_s_memtime();
#endif
  R rng;
  rng.init(P, r, (lds_vu32*)rngbuf + gib * R::RING, rs->mti, rs->mtw, lig);
#ifdef TNCO_FWA_PROF
  const unsigned long long ta1 = __builtin_amdgcn_s_memtime();
  const int nwm = fw_gs_mark<LOG2L, K, HYPER>(P, F, v, w64, sc, st, lane0, gbase, nullptr, nw_pre);
  const unsigned long long ta2 = __builtin_amdgcn_s_memtime();
  unsigned long long gp_[4] = {0, 0, 0, 0};
  const M ns = fw_gs_pick<LOG2L, K, HYPER>(P, F, v, rng, sc, nwm, lpos, lane0, gbase, F.status + r, gp_);
  const unsigned long long ta3 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&g_fwa_prof[0], ta1 - ta0); atomicAdd(&g_fwa_prof[1], ta2 - ta1); atomicAdd(&g_fwa_prof[2], ta3 - ta2);
    atomicAdd(&g_fwa_prof[3], 1ull);
    for (int q = 0; q < 4; ++q) atomicAdd(&g_fwa_prof[4 + q], gp_[q]);
  }
#else
  const M ns = fw_get_slices<LOG2L, K, HYPER>(P, F, v, w64, rng, sc, st, lpos, lane0, gbase, F.status + r, nullptr, nullptr,
                                              nw_pre);
#endif
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last reads of the position scratch)
  // the proposed slices travel in the candidate-position scratch, free now
  uint64_t* prop = reinterpret_cast<uint64_t*>(const_cast<int16_t*>(sc.pos));
#pragma unroll
  for (int k = 0; k < K; ++k) prop[v.widx(k)] = ns.w[k];
  if (F.tree_ok) {
    // ... and for fw_tree_kernel the indices that changed, each with the parents of the (one or two) leaves
    // holding it -- where its two paths to the root start: F.delta_scr, as 32-bit words
    //   [0] how many (0xFFFFFFFF: more than FWT_MAXD, or an index held otherwise: the full rebuild)
    //   [4..5] [6..7] which of them join / leave the slices (64 bits each)
    //   [8 + k] first start | second start << 16 (0xFFFF: none)
    uint32_t* chg = reinterpret_cast<uint32_t*>(F.delta_scr + r * 64);
    int mine = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) mine += __popcll(ns.w[k] ^ slices.w[k]);
    int off = 0, total = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const int c = __shfl(mine, gbase + j);
      off += j < lig ? c : 0;
      total += c;
    }
    bool unsup = total > FWT_MAXD;
    uint64_t plus = 0, minus = 0;
    if (!unsup) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        uint64_t ch = ns.w[k] ^ slices.w[k];
        while (ch) {
          const int bit = __ffsll((unsigned long long)ch) - 1;
          ch &= ch - 1;
          const int2 t12 = *reinterpret_cast<const int2*>(F.holder2 + 2 * (v.widx(k) * 64 + bit));
          if (t12.x < 0) { unsup = true; break; }
          const int s1 = v.lpar[(int64_t)t12.x * LPS], s2 = t12.y < 0 ? 0xFFFF : v.lpar[(int64_t)t12.y * LPS];
          chg[8 + off] = (uint32_t)s1 | ((uint32_t)s2 << 16);
          if ((ns.w[k] >> bit) & 1ull) plus |= 1ull << off; else minus |= 1ull << off;
          ++off;
        }
      }
    }
    unsup = gany<LOG2L>(unsup);
    // (disjoint bits: the sum over the group is the union)
    const uint32_t p0 = gsum<LOG2L>((uint32_t)plus), p1 = gsum<LOG2L>((uint32_t)(plus >> 32));
    const uint32_t m0 = gsum<LOG2L>((uint32_t)minus), m1 = gsum<LOG2L>((uint32_t)(minus >> 32));
    if (lane0) {
      chg[0] = unsup ? 0xFFFFFFFFu : (uint32_t)total;
      *reinterpret_cast<uint4*>(chg + 4) = make_uint4(p0, p1, m0, m1);
    }
  }
  int mti, mtw;
  rng.finish(mti, mtw);
  if (lane0) {
    rs->mti = mti;
    rs->mtw = mtw;
  }
}

static __global__ __launch_bounds__(64) void fw_delta_kernel(const Params P, const FwParams F) {
  // FWD_LANES busy lanes per wavefront: the work of a replica is a chain (LDS stack, record loads), so
  // more, emptier wavefronts hide more of its latency (as in fw_walk_kernel)
  constexpr int NL = FWD_LANES;
  __shared__ double sstk[FWD_STK * NL];
  __shared__ uint64_t scv[FWD_STK * NL];
  __shared__ uint32_t sbm[FWD_BITW * NL];
  __shared__ uint16_t sh[2 * 32 * NL];  // holders of the first 32 changed indices (the others: memory)
  const int lane = threadIdx.x;
  if (lane >= NL) return;
  const int64_t r = (int64_t)blockIdx.x * NL + lane;
  if (r >= P.R) return;
  const int n = P.n, N = P.N, LK = F.I64 / 64, ni = N - n;
  F.fastflag[r] = 0;
  const FwScratch sc(F, r, N);
  uint64_t* sl = F.slices + r * 2 * (int64_t)LK;
  const uint64_t* prop = reinterpret_cast<const uint64_t*>(const_cast<const int16_t*>(sc.pos));
  {
    uint64_t any = 0;
    for (int w = 0; w < P.W; ++w) any |= sl[w];
    if (!any) return;  // (nothing was proposed)
  }
  TNCO_LDS volatile double* stk = (TNCO_LDS volatile double*)sstk + lane;
  TNCO_LDS volatile uint64_t* cvs = (TNCO_LDS volatile uint64_t*)scv + lane;
  TNCO_LDS volatile uint32_t* bm = (TNCO_LDS volatile uint32_t*)sbm + lane;
  TNCO_LDS volatile uint16_t* hl = (TNCO_LDS volatile uint16_t*)sh + lane;
  // (rarely touched: in memory -- 20 KB of LDS per 32 replicas keep all 65 536 of a launch resident)
  volatile uint16_t* hl2 = reinterpret_cast<volatile uint16_t*>(F.delta_scr + r * 64);  // holders of the indices 32..63
  volatile uint64_t* cvs2 = F.delta_scr + r * 64 + 32;                                  // second words of the stack
  // the changed indices, their holders, which way they changed
  for (int w = 0; w < FWD_BITW; ++w) bm[w * NL] = 0;
  int nd = 0, base = 0;
  uint64_t plus = 0, minus = 0, plus2 = 0, minus2 = 0;
  bool slow = false;
  for (int w = 0; w < P.W && !slow; ++w) {
    const uint64_t o = sl[w], q = prop[w];
    uint64_t ch = o ^ q;
    while (ch) {
      const int bit = __ffsll((unsigned long long)ch) - 1;
      ch &= ch - 1;
      const int d = w * 64 + bit;
      const int t1 = F.holder2[2 * d], t2 = F.holder2[2 * d + 1];
      if (nd >= FWD_MAXD || t1 < 0) {
        slow = true;
        FWD_STAT(t1 < 0 ? 70 : 71);
        break;
      }
      const uint16_t u1 = (uint16_t)t1, u2 = (uint16_t)(t2 < 0 ? 0xFFFF : t2);
      if (nd < 32) {
        hl[(2 * nd) * NL] = u1;
        hl[(2 * nd + 1) * NL] = u2;
      } else {
        hl2[2 * (nd - 32)] = u1;
        hl2[2 * (nd - 32) + 1] = u2;
      }
      bm[(t1 >> 5) * NL] |= 1u << (t1 & 31);
      if (t2 >= 0) bm[(t2 >> 5) * NL] |= 1u << (t2 & 31);
      const uint64_t fld = 1ull << (2 * (nd & 31));
      if ((q >> bit) & 1ull) {
        if (nd < 32) plus |= fld; else plus2 |= fld;
        base += 1;
      } else {
        if (nd < 32) minus |= fld; else minus2 |= fld;
        base -= 1;
      }
      ++nd;
    }
  }
  if (slow) {  // (phase B rebuilds this replica in full; the host watches how often: tnco_hip_run_fw)
    atomicAdd(F.slowstat, 1ull);
    return;
  }
  FWD_STAT(nd < 40 ? nd : 40);
  // (nd == 0, the slices the replica has: the rebuild still runs -- its partial sums are those of
  //  finite_width/utils.hpp:36-47, (cost + left) + right at every node, which the moves' incremental
  //  updates do not always reproduce to the last bit: the reference compares and may commit)
  // holders of changed indices inside a leaf: two bits per index
  const bool wide = nd > 32;
  auto leafcv = [&](int t, uint64_t& hi) -> uint64_t {
    hi = 0;
    if (!((bm[(t >> 5) * NL] >> (t & 31)) & 1u)) return 0ull;
    uint64_t cv = 0;
    const int n1 = nd < 32 ? nd : 32;
    for (int k = 0; k < n1; ++k)
      if (hl[(2 * k) * NL] == t || hl[(2 * k + 1) * NL] == t) cv += 1ull << (2 * k);
    for (int k = 32; k < nd; ++k)
      if (hl2[2 * (k - 32)] == t || hl2[2 * (k - 32) + 1] == t) hi += 1ull << (2 * (k - 32));
    return cv;
  };
  const uint8_t* blk = P.blocks + r * P.RB;
  const int log2d = P.log2d;
  bool bad = false;
  const uint4* rec4 = reinterpret_cast<const uint4*>(sc.rec);  // (64-byte aligned; 8 records per 64 bytes)
  // ONE pass: the new (cost, partial sum) of every node go to the sequential scratch list (whole lines);
  // only a replica whose total improves scatters them into its node headers afterwards.  (Recomputing
  // in a second, storing pass cost as much as the first for EVERY wavefront: one improving replica
  // among its lanes is enough.)  Straight-line code with selects -- the lanes of a wavefront are at
  // different (left internal?, right internal?) cases at every node.
  // the new partial sums (8 B) and cost exponents (2 B) of the nodes, by post-order number: whole 64-byte
  // pieces per eight nodes (a (cost, partial) pair per node was 135 lines per replica instead of 85)
  double* plist = reinterpret_cast<double*>(sc.cp);
  uint16_t* elist = reinterpret_cast<uint16_t*>(plist + ((ni + 9) & ~1));
  double part = 0.0;
  uint64_t cvp = 0, cvp2 = 0;
  int sp = 0;
  {
    uint4 nb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) nb[q] = rec4[q];
    for (int j0 = 0; j0 < ni; j0 += 8) {
      uint64_t rc[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        rc[2 * q] = (uint64_t)nb[q].x | ((uint64_t)nb[q].y << 32);
        rc[2 * q + 1] = (uint64_t)nb[q].z | ((uint64_t)nb[q].w << 32);
      }
      const int jn = j0 + 8 < ni ? j0 + 8 : j0;  // (past the end: the same piece again)
#pragma unroll
      for (int q = 0; q < 4; ++q) nb[q] = rec4[(jn >> 1) + q];
      double pv[8];
      uint32_t ev[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        pv[i] = 0.0;
        ev[i] = 0;
        if (j0 + i < ni) {
          const uint64_t cur = rc[i];
          const int l = fw_rec_left(cur), rr = fw_rec_right(cur), e = fw_rec_exp(cur);
          const bool li = l >= n, ri = rr >= n, both = li && ri, none = !li && !ri;
          // the entry below the newest finished subtree (read whether needed or not)
          const int k2 = sp - 2 < 0 ? 0 : sp - 2;
          double top;
          uint64_t topcv, topcv2 = 0;
          if (k2 < FWD_STK) {
            top = stk[k2 * NL];
            topcv = cvs[k2 * NL];
            if (wide) topcv2 = cvs2[k2];
          } else {
            top = sc.pstk[k2];
            topcv = reinterpret_cast<const uint64_t*>(sc.gstk)[k2];
            if (wide) bad = true;  // (both at once never seen: the full rebuild takes it)
          }
          if (none && sp >= 1) {  // a new subtree starts: the finished one waits on the stack
            const int k1 = sp - 1;
            if (k1 < FWD_STK) {
              stk[k1 * NL] = part;
              cvs[k1 * NL] = cvp;
              if (wide) cvs2[k1] = cvp2;
            } else {
              sc.pstk[k1] = part;
              reinterpret_cast<uint64_t*>(sc.gstk)[k1] = cvp;
            }
          }
          uint64_t lcl2 = 0, lcr2 = 0;
          const uint64_t lcl = li ? 0ull : leafcv(l, lcl2), lcr = ri ? 0ull : leafcv(rr, lcr2);
          const double pl = both ? top : (li ? part : 0.0);
          const double pr = ri ? part : 0.0;
          const uint64_t cvl = both ? topcv : (li ? cvp : lcl);
          const uint64_t cvr = ri ? cvp : lcr;
          // a changed index is among the legs of a child that holds exactly one of its tensors
          const uint64_t in_u = ((cvl & ~(cvl >> 1)) | (cvr & ~(cvr >> 1))) & 0x5555555555555555ull;
          int dex = base - __popcll(in_u & plus) + __popcll(in_u & minus);
          if (wide) {  // (indices 32..63 of the list)
            const uint64_t cvl2 = both ? topcv2 : (li ? cvp2 : lcl2), cvr2 = ri ? cvp2 : lcr2;
            const uint64_t in_u2 = ((cvl2 & ~(cvl2 >> 1)) | (cvr2 & ~(cvr2 >> 1))) & 0x5555555555555555ull;
            dex += __popcll(in_u2 & minus2) - __popcll(in_u2 & plus2);
            cvp2 = cvl2 + cvr2;
          }
          const int ne = e + log2d * dex;
          bad = bad || e <= 0 || e >= 2047 || ne <= 0 || ne >= 2047;
          const double c = __longlong_as_double((long long)((uint64_t)(uint32_t)ne << 52));
          part = (c + pl) + pr;  // (the association order of finite_width/utils.hpp:36-47)
          cvp = cvl + cvr;
          pv[i] = part;
          ev[i] = (uint32_t)ne & 0xFFFFu;
          sp += both ? -1 : (none ? 1 : 0);
        }
      }
      // (the lists are padded by eight entries: whole pieces also at the end)
      double2* pd = reinterpret_cast<double2*>(plist + j0);
#pragma unroll
      for (int q = 0; q < 4; ++q) pd[q] = make_double2(pv[2 * q], pv[2 * q + 1]);
      *reinterpret_cast<uint4*>(elist + j0) = make_uint4(ev[0] | (ev[1] << 16), ev[2] | (ev[3] << 16), ev[4] | (ev[5] << 16), ev[6] | (ev[7] << 16));
    }
  }
  if (bad) {
    FWD_STAT(72);
    atomicAdd(F.slowstat, 1ull);
    return;  // (a cost outside the powers of two of a double: the full rebuild decides)
  }
  const double cur = reinterpret_cast<const NodeRec*>(blk + (int64_t)(N - 1 - n) * P.BS)->partial;
  if (part < cur) {  // greedy/optimizer.hpp:371-374
#ifndef TNCO_FWD_NO_STORE  // (measurement builds only)
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int j0 = 0; j0 < ni; j0 += 4) {
      uint64_t x[4];
      double pp[4];
      uint32_t ee[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = j0 + i < ni ? j0 + i : ni - 1;
        x[i] = sc.rec[j];
        pp[i] = plist[j];
        ee[i] = elist[j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (j0 + i < ni)
          *reinterpret_cast<double2*>(const_cast<uint8_t*>(blk) + (int64_t)(fw_rec_node(x[i]) - n) * P.BS + 16) =
              make_double2(__longlong_as_double((long long)((uint64_t)ee[i] << 52)), pp[i]);
    }
#endif
    for (int w = 0; w < LK; ++w) sl[w] = prop[w];
  }
  F.fastflag[r] = 1;
}

// ---------------------------------------------------------------------------------------------
// Round 3: the re-slice WITHOUT a walk over the tree (split layout, re-pricing form).
//
// fw_walk2_kernel + fw_delta_kernel were n - 1 dependent, random header reads per replica and re-slice
// (at the chip's random-request ceiling: 0.9 ms), 8-byte records written and read again, and -- for the
// replicas that keep their new slices -- n - 1 scattered 16-byte stores.  Neither needs a traversal:
//   * get_slices wants the TOO-WIDE tensors in post-order -- a handful.  fw_order_kernel reads the replica's
//     header array (one contiguous piece in the split layout), lists the too-wide nodes and gives each the
//     key of its root path (one bit per level, 0 = left; padded with ones; deeper first on ties):
//     ascending keys ARE the post-order of include/tnco/utils.hpp:34-51.
//   * the re-priced CostCache (finite_width/utils.hpp:36-47) only needs children before parents.
//     fw_tree_kernel: one wavefront per replica, the node table in LDS (16 bytes per node).  The new cost
//     of a node is its old one times 2^(k * (joined - left)) over the changed indices that are NOT among
//     its children's legs -- and those that are sit on the paths from the indices' holders up to where the
//     two paths meet: one lane per (index, holder) marks its path, then every node prices itself.  The
//     partial sums follow: every lane starts at its nodes with two leaf children, and the LAST of two
//     children to arrive at a parent (an LDS counter) goes on with the parent.  partial = (cost + left)
//     + right whatever the order of evaluation, so the sums are the reference's bit for bit.  A replica
//     that keeps the new slices rewrites its header array as whole lines.
// ---------------------------------------------------------------------------------------------
constexpr int FWO_MAXW = 256;   // too-wide tensors fw_order_kernel orders (more: the traverse inside fw_reslice_a_kernel)
constexpr int FWT_JMAX = 16;    // internal nodes per lane of fw_tree_kernel at most: n - 1 <= 1024
__host__ __device__ inline size_t fwo_lds_bytes(int n) {  // per replica (wavefront)
  const size_t ni = (size_t)((n - 1 + 63) & ~63);
  return (ni * 4 + ni * 2 + (size_t)FWO_MAXW * (8 + 2 + 2) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t fwt_lds_bytes(int n) {
  const size_t ni = (size_t)((n - 1 + 63) & ~63), nl = (size_t)((n + 31) / 32);
  (void)nl;
  return (ni * (8 + 4 + 4) + 32 + 15) & ~(size_t)15;
}

static __global__ __launch_bounds__(256) void fw_order_kernel(const Params P, const FwParams F) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fwo_smem[];
  const int n = P.n, N = P.N, ni = N - n, LK = F.I64 / 64;
  const int nip = (ni + 63) & ~63;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wv;
  if (r >= P.R) return;
  uint8_t* base = fwo_smem + (size_t)wv * fwo_lds_bytes(n);
  TNCO_LDS volatile uint64_t* key = (TNCO_LDS volatile uint64_t*)base;                 // [FWO_MAXW]
  TNCO_LDS volatile uint32_t* lr = (TNCO_LDS volatile uint32_t*)(key + FWO_MAXW);       // [nip] left | right << 16
  TNCO_LDS volatile uint16_t* par = (TNCO_LDS volatile uint16_t*)(lr + nip);            // [nip]
  TNCO_LDS volatile uint16_t* wnode = par + nip;                                        // [FWO_MAXW]
  TNCO_LDS volatile uint16_t* dep = wnode + FWO_MAXW;                                   // [FWO_MAXW]
  {  // greedy/optimizer.hpp:359: nothing to do without slices
    const uint64_t* sl0 = F.slices + r * 2 * (int64_t)LK;
    uint64_t any = 0;
    for (int w = lane; w < P.W; w += 64) any |= sl0[w];
    if (!__any(any != 0)) {
      if (lane == 0) F.nwide[r] = -1;
      return;
    }
  }
  const uint8_t* hb = P.blocks + r * P.RB;
  const double* w64 = F.width64 ? F.width64 + r * (int64_t)N : nullptr;
  const FwScratch sc(F, r, N);
  uint2* img = reinterpret_cast<uint2*>(sc.rec);
  int32_t* imgw = sc.gstk;  // [n - 1] (the deep end of the traversal stacks: unused without a walk)
  int nw = 0;
  for (int j0 = 0; j0 < ni; j0 += 64) {
    const int i = j0 + lane;
    bool wide = false;
    if (i < ni) {
      const int4 h = *reinterpret_cast<const int4*>(hb + (int64_t)i * P.BS);
      const uint32_t ce = *reinterpret_cast<const uint32_t*>(hb + (int64_t)i * P.BS + 20);  // high word of the cached cost
      lr[i] = (uint32_t)h.x | ((uint32_t)h.y << 16);
      par[i] = (uint16_t)h.z;  // (the root: 0xFFFF)
      const double w = F.width_f32 ? (double)__int_as_float(h.w) : w64[n + i];
      wide = w > F.max_width;
      // the node table of fw_tree_kernel, 8 bytes per node, in the (otherwise unused) record scratch:
      // left | right << 16;  parent | cost exponent << 16 | internal children << 27
      const uint32_t c = (h.x >= n ? 1u : 0u) + (h.y >= n ? 1u : 0u);
      img[i] = make_uint2((uint32_t)h.x | ((uint32_t)h.y << 16), ((uint32_t)h.z & 0xFFFFu) | (((ce >> 20) & 0x7FFu) << 16) | (c << 27));
      imgw[i] = h.w;  // (the spare header word -- the cached float32 width: a kept re-slice rewrites whole headers)
    }
    const unsigned long long b = __ballot(wide);
    if (wide) {
      const int k = nw + __popcll(b & ((1ull << lane) - 1ull));
      if (k < FWO_MAXW) wnode[k] = (uint16_t)(n + i);
    }
    nw += __popcll(b);
  }
  if (nw > FWO_MAXW) {  // (fw_reslice_a_kernel traverses this replica itself)
    if (lane == 0) F.nwide[r] = -2;
    return;
  }
  // root-path keys
  bool deep = false;
  for (int k0 = 0; k0 < nw; k0 += 64) {
    const int k = k0 + lane;
    if (k < nw) {
      int x = wnode[k], d = 0;
      uint64_t rev = 0;
      while (x != N - 1 && d <= 64) {
        const int p = par[x - n];
        rev = (rev << 1) | (uint64_t)((int)(lr[p - n] >> 16) == x);
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
    if (lane == 0) F.nwide[r] = -2;
    return;
  }
  // ranks: ascending key, deeper first on equal keys (a node and its all-right ancestors)
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
      sc.wlist[rank] = wnode[k];
    }
  }
  if (lane == 0) {
    F.nwide[r] = nw;
    F.nwfront[r] = nw;
  }
}

// ---------------------------------------------------------------------------------------------
// get_slices (finite_width/greedy/utils.hpp:21-125) with one WAVEFRONT per replica: fw_slices_kernel, between
// fw_order_kernel and fw_tree_kernel, for the networks of the tree path (uniform power-of-two dims, no sparse
// legs) with at most 16 mask words.  fw_reslice_a_kernel (16 replicas per wavefront, in lock step) read the legs
// of every too-wide tensor twice -- once for the counts, once in the greedy pass: ~2 x 110 lines per replica on
// config 5, 2/3 of the chip's random-request rate for the length of the kernel -- and then gathered the counts
// one byte per candidate leg from memory.  Here
//   * lane 16 g + w holds word w of a mask, four tensors per load instruction; the legs read for the counts stay
//     in LDS (the first `cap` tensors of the list) for the greedy pass; the counts become a byte table in LDS;
//   * the scan for the next tensor that does not fit tests four tensors per step;
//   * std::shuffle's variates are drawn for all pairs of swaps at once (lane k: pair k) and the permutation is
//     applied by every lane tracing ITS final position back through the swaps -- registers only; the rare cases
//     (a re-draw of uniform_int_distribution, the generator's 624 words ending inside the shuffle) run the
//     sequential fw_shuffle_lds from the same generator position;
//   * a pick is one maximum over the wavefront.
// The generator is RngWave: 64 outputs per memory round trip, never ahead of the generation being consumed.
// Replicas it leaves alone (nwide -2, > 255 too-wide tensors, > FWS_MAXNP candidate legs in a tensor) are done by
// fw_reslice_a_kernel, which skips those marked done (nwide = -3).  Same outputs as that kernel: the proposed
// slices, the change list for fw_tree_kernel, the generator's position.
// ---------------------------------------------------------------------------------------------
#ifndef TNCO_FWS_CAP
#define TNCO_FWS_CAP 16
#endif
constexpr int FWS_CAP = TNCO_FWS_CAP;  // too-wide tensors whose legs stay in LDS between the two passes (default)
constexpr int FWS_MAXNP = 128;          // candidate legs of one tensor: two per lane
__host__ __device__ inline size_t fws_lds_bytes(int cap) {
  return (size_t)cap * 128 + 1024 /* counts */ + 1024 /* generator ring */ + 512 /* positions */ + 512 /* list */;
}

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

#ifdef TNCO_FWS_PROF  // (diagnostic build: shader cycles per replica of [list + counts, count table, greedy pass; of it scan, positions, generator, shuffle, keys + picks], replicas, slicings)
static __device__ unsigned long long g_fws_prof[12];
#define FWS_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define FWS_ACC(i, a, b) acc_[i] += (b) - (a)
#else
#define FWS_T(v)
#define FWS_ACC(i, a, b)
#endif
static __global__ __launch_bounds__(64) void fw_slices_kernel(const Params P, const FwParams F, const int cap, const int maxnp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fws_smem[];
  const int lane = threadIdx.x, w = lane & 15, g = lane >> 4;
  const int64_t r = blockIdx.x;
  // (everything that does not depend on anything else is requested first: a dependent round trip costs 5-10 us here)
  const int nw = F.nwide[r];
  const int n = P.n, N = P.N, W = P.W, LK = F.I64 / 64;
  TNCO_LDS volatile uint64_t* cache = (TNCO_LDS volatile uint64_t*)fws_smem;          // [cap][16] legs of the listed tensors
  TNCO_LDS volatile uint8_t* nbig = (TNCO_LDS volatile uint8_t*)(cache + (size_t)cap * 16);  // [1024] too-wide counts
  lds_vu32* ring = (lds_vu32*)(nbig + 1024);                                          // [256]
  lds_vi32* pos = (lds_vi32*)(ring + 256);                                            // [FWS_MAXNP]
  TNCO_LDS volatile uint16_t* wls = (TNCO_LDS volatile uint16_t*)(pos + FWS_MAXNP);    // [256] the list
  const FwScratch sc(F, r, N);
  const uint8_t* legs = P.blocks + r * P.RB + P.WOFF;
  const int WS = P.WS;
  const bool has = w < W;
  ReplicaState* rs = P.rs + r;
  const int mti0 = rs->mti, mtw0 = rs->mtw;
  const uint64_t old = has ? F.slices[r * 2 * (int64_t)LK + w] : 0ull;
  const uint64_t skip = (F.skip != nullptr && has) ? F.skip[w] : 0ull;
  int wlq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wlq[q] = (64 * q + lane < sc.wcap) ? sc.wlist[64 * q + lane] : n;
  if (nw < 0 || nw > 255) return;  // (-1: no slices, nothing to do; the others: fw_reslice_a_kernel)
#ifdef TNCO_FWS_PROF
  unsigned long long acc_[6] = {0, 0, 0, 0, 0, 0};
#endif
  FWS_T(q0_);
#pragma unroll
  for (int q = 0; q < 4; ++q) wls[64 * q + lane] = (uint16_t)(64 * q + lane < nw ? wlq[q] : n);
  // ---- :41-48: for every index the number of too-wide tensors it appears in (bit-sliced, four tensors side by side)
  uint64_t pl[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) pl[p] = 0ull;
  uint32_t maxc = 0;
  uint64_t m[4];
  auto load16 = [&](int t0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + 4 * u + g;
      const int node = wls[t < nw ? t : 0];
      m[u] = 0ull;
      if (t < nw && has) m[u] = *reinterpret_cast<const uint64_t*>(legs + (int64_t)(node - n) * WS + 8 * w);
    }
  };
  load16(0);
  // the generator's first outputs travel with the first legs
  RngWave rng;
  rng.init(P.mt + r * 624, ring, mti0, mtw0, lane);
  if (nw > 0) rng.fill();
  for (int t0 = 0; t0 < nw; t0 += 16) {
    if (t0) load16(t0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + 4 * u + g;
      if (t < nw && t < cap) cache[t * 16 + w] = m[u];
      const uint32_t c = gsum<4>((uint32_t)__popcll(m[u] & ~skip));
      maxc = c > maxc ? c : maxc;
      uint64_t carry = m[u];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const uint64_t tt = pl[p] & carry;
        pl[p] ^= carry;
        carry = tt;
      }
    }
  }
  if (gmax<6>(maxc) > (uint32_t)maxnp) {  // (nothing drawn yet; the generator's words twisted ahead stay)
    if (lane == 0) rs->mtw = (int)rng.tw;
    return;
  }
  FWS_T(q1_);
  // the four partial counts of a lane's 64 indices, added
#pragma unroll
  for (int step = 16; step <= 32; step <<= 1) {
    uint64_t o[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) o[p] = fws_shflx64(pl[p], step);
    uint64_t c = 0ull;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint64_t a = pl[p], b = o[p];
      pl[p] = a ^ b ^ c;
      c = (a & b) | (c & (a ^ b));
    }
  }
  {  // counters 16 g .. 16 g + 15 of word w -> sixteen bytes (a nibble of plane bits is spread over the bytes of a word)
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t c = 0;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const uint32_t nib = (uint32_t)(pl[p] >> (16 * g + 4 * j)) & 0xFu;
        c |= ((nib * 0x00204081u) & 0x01010101u) << p;
      }
      o[j] = c;
    }
    TNCO_LDS volatile uint32_t* d = (TNCO_LDS volatile uint32_t*)(nbig + w * 64 + 16 * g);
    d[0] = o[0]; d[1] = o[1]; d[2] = o[2]; d[3] = o[3];
  }
  // ---- :62-101: the greedy pass over the list, in post-order
  FWS_T(q2_);
  uint64_t ns = 0ull;  // the new slices, word w (the same in the four rows of lanes)
  const double mdl = fw_wr(F, -F.log2d);  // get_delta_width of a set position (simple.hpp:59-76)
  int j = 0;
  while (j < nw) {
    FWS_T(s0_);
    const int t = j + g;
    uint64_t m = 0ull;
    if (t < nw) {
      if (t < cap) m = cache[t * 16 + w];
      else if (has) m = *reinterpret_cast<const uint64_t*>(legs + (int64_t)((int)wls[t] - n) * WS + 8 * w);
    }
    const uint64_t sx = m & ~ns;
    const uint32_t cnt = gsum<4>((uint32_t)__popcll(sx));
    const bool wide = t < nw && fw_wr(F, F.log2d * (double)cnt) > F.max_width;
    const unsigned long long bal = __ballot(wide);
    if (bal == 0ull) {
      j += 4;
      FWS_T(s1_);
      FWS_ACC(0, s0_, s1_);
      continue;
    }
    FWS_T(s1_);
    FWS_ACC(0, s0_, s1_);
    const int gs = (__ffsll(bal) - 1) >> 4;  // the first of the four that does not fit
    j += gs + 1;
    const uint64_t sxw = fws_shfl64(sx, 16 * gs + w);
    double sw = fw_wr(F, F.log2d * (double)(uint32_t)__shfl((int)cnt, 16 * gs));
    // candidate positions, ascending
    const uint64_t cand = sxw & ~skip;
    const uint32_t mine = (uint32_t)__popcll(cand);
    const uint32_t incl = fws_rowscan(mine);
    const int np = __builtin_amdgcn_readlane((int)incl, 15);
    if (g == 0) {
      uint32_t o = incl - mine;
      uint64_t x = cand;
      while (x) {
        const int b = __ffsll((unsigned long long)x) - 1;
        pos[o++] = w * 64 + b;
        x &= x - 1;
      }
    }
    // :80 std::shuffle -- src0 / src1: where the candidates that end at ranks lane / lane + 64 stand before it
    int src0 = lane, src1 = lane + 64;
    FWS_T(s2_);
    FWS_ACC(1, s1_, s2_);
    if (np >= 2) {
      const uint32_t nd = (uint32_t)np >> 1;  // variates: one per pair of swaps (+ the single swap of an even count)
      bool fast = rng.ensure(nd);
      FWS_T(s3_);
      FWS_ACC(2, s2_, s3_);
      uint32_t p0 = 0, p1 = 0;
      const int base = (np & 1) ? 1 : 0;  // lane k swaps positions base + 2 k, base + 2 k + 1 (position 0 with itself)
      if (fast) {
        const uint32_t raw = rng.peek((uint32_t)lane);
        const uint32_t i0 = (uint32_t)(base + 2 * lane);
        const bool single = base == 0 && lane == 0;  // (stl_algo.h:3760-3765: d(0, 1), the swap of position 1)
        const uint32_t range = single ? 2u : (i0 + 1u) * (i0 + 2u);
        const uint64_t product = (uint64_t)raw * (uint64_t)range;
        const uint32_t low = (uint32_t)product;
        bool rej = false;
        if ((uint32_t)lane < nd && low < range) rej = low < (0u - range) % range;
        if (__any(rej)) {
          fast = false;
        } else {
          const uint32_t x = (uint32_t)(product >> 32);
          if (single) {
            p0 = 0;
            p1 = x;
          } else {
            p0 = x / (i0 + 2u);
            p1 = x - p0 * (i0 + 2u);
          }
          rng.advance(nd);
        }
      }
      if (fast) {
        const bool two = np > 64;
        for (int k = (int)nd - 1; k >= 0; --k) {
          const int j1 = __builtin_amdgcn_readlane((int)p1, k), j0 = __builtin_amdgcn_readlane((int)p0, k);
          const int i = base + 2 * k;
          src0 = src0 == i + 1 ? j1 : (src0 == j1 ? i + 1 : src0);
          src0 = src0 == i ? j0 : (src0 == j0 ? i : src0);
          if (two) {
            src1 = src1 == i + 1 ? j1 : (src1 == j1 ? i + 1 : src1);
            src1 = src1 == i ? j0 : (src1 == j0 ? i : src1);
          }
        }
      } else {
        fw_shuffle_lds<6>(rng, pos, np, lane == 0);
      }
      FWS_T(s4_);
      FWS_ACC(3, s3_, s4_);
    }
    FWS_T(s5_);
    // :83-101 the keys ((too-wide count << 16) | 0xFFFF - shuffled rank: the stable order of :83); picks until it fits
    const int xp0 = lane < np ? (int)pos[src0] : 0, xp1 = lane + 64 < np ? (int)pos[src1] : 0;
    uint32_t k0 = lane < np ? (((uint32_t)nbig[xp0] << 16) | (0xFFFFu - (uint32_t)lane)) : 0u;
    uint32_t k1 = lane + 64 < np ? (((uint32_t)nbig[xp1] << 16) | (0xFFFFu - (uint32_t)(lane + 64))) : 0u;
    for (int taken = 0; taken < np; ++taken) {
      const uint32_t best = gmax<6>(k0 > k1 ? k0 : k1);
      const int qb = (int)(0xFFFFu - (best & 0xFFFFu));
      const int xpos = __shfl(qb >= 64 ? xp1 : xp0, qb & 63);
      if (lane == (qb & 63)) {
        if (qb >= 64) k1 = 0u; else k0 = 0u;
      }
      if (w == (xpos >> 6)) ns |= 1ull << (xpos & 63);
      sw = fw_wr(F, sw + mdl);
      if (sw <= F.max_width) break;
    }
    FWS_T(s6_);
    FWS_ACC(4, s5_, s6_);
#ifdef TNCO_FWS_PROF
    acc_[5] += 1;
#endif
  }
  FWS_T(q3_);
  // ---- the proposed slices, and for fw_tree_kernel the indices that changed (as fw_reslice_a_kernel leaves them)
  uint64_t* prop = reinterpret_cast<uint64_t*>(const_cast<int16_t*>(sc.pos));
  if (g == 0 && w < LK) prop[w] = ns;
  {
    uint32_t* chg = reinterpret_cast<uint32_t*>(F.delta_scr + r * 64);
    const int32_t* lpar = P.lpar + r * (int64_t)n * LPS;
    uint64_t ch = g == 0 ? (ns ^ old) : 0ull;
    const uint32_t mine = (uint32_t)__popcll(ch);
    const uint32_t incl = fws_rowscan(mine);
    const int total = __builtin_amdgcn_readlane((int)incl, 15);
    bool unsup = total > FWT_MAXD;
    uint64_t plus = 0ull, minus = 0ull;
    if (!unsup) {
      uint32_t off = incl - mine;
      while (ch) {
        const int bit = __ffsll((unsigned long long)ch) - 1;
        ch &= ch - 1;
        const int2 t12 = *reinterpret_cast<const int2*>(F.holder2 + 2 * (w * 64 + bit));
        if (t12.x < 0) { unsup = true; break; }
        const int s1 = lpar[(int64_t)t12.x * LPS], s2 = t12.y < 0 ? 0xFFFF : lpar[(int64_t)t12.y * LPS];
        chg[8 + off] = (uint32_t)s1 | ((uint32_t)s2 << 16);
        if ((ns >> bit) & 1ull) plus |= 1ull << off; else minus |= 1ull << off;
        ++off;
      }
    }
    unsup = __any(unsup);
    // (disjoint bits: the sum over the row is the union)
    const uint32_t a0 = gsum<4>((uint32_t)plus), a1 = gsum<4>((uint32_t)(plus >> 32));
    const uint32_t b0 = gsum<4>((uint32_t)minus), b1 = gsum<4>((uint32_t)(minus >> 32));
    if (lane == 0) {
      chg[0] = unsup ? 0xFFFFFFFFu : (uint32_t)total;
      *reinterpret_cast<uint4*>(chg + 4) = make_uint4(a0, a1, b0, b1);
    }
  }
  int mti, mtw;
  rng.finish(mti, mtw);
  if (lane == 0) {
    rs->mti = mti;
    rs->mtw = mtw;
    F.nwide[r] = -3;  // done: fw_reslice_a_kernel skips this replica
  }
#ifdef TNCO_FWS_PROF
  if (lane == 0) {
    atomicAdd(&g_fws_prof[0], q1_ - q0_); atomicAdd(&g_fws_prof[1], q2_ - q1_); atomicAdd(&g_fws_prof[2], q3_ - q2_);
    for (int q = 0; q < 5; ++q) atomicAdd(&g_fws_prof[3 + q], acc_[q]);
    atomicAdd(&g_fws_prof[8], 1ull); atomicAdd(&g_fws_prof[9], acc_[5]); atomicAdd(&g_fws_prof[10], (unsigned long long)nw);
    atomicAdd(&g_fws_prof[11], __builtin_amdgcn_s_memtime() - q0_);
  }
#endif
}

#ifdef TNCO_FWT_PROF  // (diagnostic build: shader cycles of [setup, header load, the loop, commit], loop iterations, replicas, commits)
static __device__ unsigned long long g_fwt_prof[8];
#define FWT_T(i) const unsigned long long tt##i = __builtin_amdgcn_s_memtime()
#else
#define FWT_T(i)
#endif
// GW lanes per replica (64: one per wavefront; 32: two -- half the instructions per replica, for networks of at most
// 2048 indices), J = ceil((n - 1) / GW) nodes per lane.
template <int J, int GW>
static __global__ __launch_bounds__(256, (J * GW <= 576 ? 4 : 2)) void fw_tree_kernel(const Params P, const FwParams F) {
  FWT_T(0);
  extern __shared__ __attribute__((aligned(16))) uint8_t fwt_smem[];
  constexpr int GPW = 64 / GW;   // replicas per wavefront
  constexpr int IPP = GW / 2;    // changed indices per pass over the paths: one lane per (index, holder)
  const int n = P.n, N = P.N, ni = N - n, LK = F.I64 / 64;
  const int nip = (ni + 63) & ~63;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gl = lane & (GW - 1), g = lane / GW;
  const int64_t r = ((int64_t)blockIdx.x * 4 + wv) * GPW + g;
  if (r >= P.R) return;
  // (every exchange below stays inside the GW lanes of a replica: a replica that leaves early takes nobody with it)
  const unsigned long long gmask = GW == 64 ? ~0ull : (0xFFFFFFFFull << (32 * g));
  auto gany = [&](bool x) -> bool { return (__ballot(x) & gmask) != 0ull; };
  uint8_t* base = fwt_smem + (size_t)(wv * GPW + g) * fwt_lds_bytes(n);
  // node i of the table: lo = left | right << 16; hi = parent | cost exponent << 16 | children still to arrive << 27;
  // Pn = the new partial sum -- before that the two path masks of the node (see below)
  TNCO_LDS volatile double* Pn = (TNCO_LDS volatile double*)base;                        // [nip]
  TNCO_LDS uint32_t* on = (TNCO_LDS uint32_t*)base;                                      // [nip][2] (the same memory)
  TNCO_LDS volatile uint32_t* onv = (TNCO_LDS volatile uint32_t*)base;
  TNCO_LDS volatile uint32_t* lo = (TNCO_LDS volatile uint32_t*)(Pn + nip);              // [nip]
  TNCO_LDS uint32_t* hi = (TNCO_LDS uint32_t*)(lo + nip);                                // [nip] (atomic arrivals)
  TNCO_LDS volatile uint32_t* hiv = (TNCO_LDS volatile uint32_t*)hi;                     // (plain accesses)
  TNCO_LDS volatile uint32_t* misc = (TNCO_LDS volatile uint32_t*)(hi + nip);            // [8]
  if (gl == 0) F.fastflag[r] = 0;
  uint64_t* sl = F.slices + r * 2 * (int64_t)LK;
  const FwScratch sc(F, r, N);
  // ---- everything this replica needs, in flight at once: the node table fw_order_kernel has left (8 bytes per
  // node, one contiguous piece), the changed indices with the starts of their paths (fw_reslice_a_kernel), the
  // current total, the old slices
  uint8_t* hb = P.blocks + r * P.RB;
  const uint32_t* chg = reinterpret_cast<const uint32_t*>(F.delta_scr + r * 64);
  const uint2 c0 = *reinterpret_cast<const uint2*>(chg);
  const uint4 c1 = *reinterpret_cast<const uint4*>(chg + 4);
  uint32_t myent[GPW];  // (FWT_MAXD = 64 entries: 64 / GW per lane)
#pragma unroll
  for (int q = 0; q < GPW; ++q) myent[q] = chg[8 + q * GW + gl];
  uint64_t myold = 0;
  if (gl < P.W) myold = sl[gl];
  const double cur = reinterpret_cast<const NodeRec*>(hb + (int64_t)(ni - 1) * P.BS)->partial;
  const int nwide_r = F.nwide[r];
  const uint2* img = reinterpret_cast<const uint2*>(sc.rec);
  const int32_t* imgw = sc.gstk;  // (the spare header words: a kept re-slice rewrites whole headers)
  uint2 im[J];  // (only until the table is in LDS: the kernel must stay below 128 registers, four wavefronts per SIMD)
  int32_t iw[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * GW + gl;
    im[j] = make_uint2(0, 0);
    iw[j] = 0;
    if (i < ni) {
      im[j] = img[i];
      iw[j] = imgw[i];
    }
  }
  if (!gany(myold != 0)) return;  // (nothing was proposed: greedy/optimizer.hpp:359)
  // (-2: fw_reslice_a_kernel has traversed this replica itself -- over the node table; no list: the full rebuild)
  if (nwide_r == -2 || c0.x == 0xFFFFFFFFu) {
    if (gl == 0) atomicAdd(F.slowstat, 1ull);
    return;
  }
  const int nd = (int)c0.x;
  // bit k of the pair: changed index number k joins / leaves the slices
  const uint64_t plus64 = (uint64_t)c1.x | ((uint64_t)c1.y << 32), minus64 = (uint64_t)c1.z | ((uint64_t)c1.w << 32);
  const int dbase = __popcll(plus64) - __popcll(minus64);
  FWT_T(1);
  // ---- the node table (links, old exponents, arrival counters), path masks cleared ------------------
  uint32_t startmask = 0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * GW + gl;
    if (i < ni) {
      const int e = (int)((im[j].y >> 16) & 0x7FFu);
      bad = bad || e <= 0 || e >= 2047;
      lo[i] = im[j].x;
      hiv[i] = im[j].y;
      onv[2 * i] = 0;
      onv[2 * i + 1] = 0;
      if ((im[j].y >> 27) == 0u) startmask |= 1u << j;
    }
  }
  // ---- which nodes see a changed index among the legs of their children ------------------------------
  // An index held by two tensors is a leg of a subtree exactly when the subtree holds ONE of them: of the
  // nodes above the first holder (mask 0) and above the second (mask 1), those below their meeting point have
  // it among their children's legs, the meeting point too, the nodes above it not.  One lane per (index,
  // holder) walks its path to the root; GW / 2 indices per pass (more passes: one re-slice in some hundreds).
  const int log2d = P.log2d;
  for (int pass = 0; pass * IPP < nd || pass == 0; ++pass) {
    const uint32_t plus = (uint32_t)(plus64 >> (IPP * pass)) & (uint32_t)((1ull << IPP) - 1ull);
    const uint32_t minus = (uint32_t)(minus64 >> (IPP * pass)) & (uint32_t)((1ull << IPP) - 1ull);
    if (pass) {
      for (int i = gl; i < ni; i += GW) { onv[2 * i] = 0; onv[2 * i + 1] = 0; }
    }
    {
      const int k = gl >> 1, which = gl & 1;
      const int idx = IPP * pass + k;  // this lane's changed index; its entry sits in lane idx % GW, slot idx / GW
      uint32_t esel = myent[0];
#pragma unroll
      for (int q = 1; q < GPW; ++q) esel = (idx / GW == q) ? myent[q] : esel;  // (idx / GW is the same for the whole pass)
      const uint32_t e = (uint32_t)__shfl((int)esel, idx & (GW - 1), GW);
      const int st = which ? (int)(e >> 16) : (int)(e & 0xFFFFu);
      int x = (idx < nd && st != 0xFFFF) ? st : -1;  // the path of a holder starts at its parent
      for (int guard = 0; gany(x >= 0); ++guard) {
        if (guard > ni) { bad = true; break; }  // (cannot happen in a tree: never spin on corrupt links)
        if (x >= 0) {
          __hip_atomic_fetch_or(&on[2 * (x - n) + which], 1u << k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const int pp = (int)(hiv[x - n] & 0xFFFFu);
          x = pp == 0xFFFF ? -1 : pp;
        }
      }
    }
    for (int i = gl; i < ni; i += GW) {
      const uint32_t w = lo[i], h = hiv[i];
      const int l = (int)(w & 0xFFFFu), rr = (int)(w >> 16);
      const uint32_t a = onv[2 * i], b = onv[2 * i + 1];
      const int il = l >= n ? l - n : i, ir = rr >= n ? rr - n : i;
      const uint32_t bl = onv[2 * il] & onv[2 * il + 1], br = onv[2 * ir] & onv[2 * ir + 1];
      const uint32_t both_below = (l >= n ? bl : 0u) | (rr >= n ? br : 0u);
      const uint32_t in_u = (a ^ b) | (a & b & ~both_below);
      const int dex = (pass ? 0 : dbase) - __popc(in_u & plus) + __popc(in_u & minus);
      const int ne = (int)((h >> 16) & 0x7FFu) + log2d * dex;
      bad = bad || ne <= 0 || ne >= 2047;  // (also between the passes: the full rebuild decides then)
      hiv[i] = (h & 0xF800FFFFu) | ((uint32_t)(ne & 0x7FF) << 16);  // (nobody reads another node's exponent in this pass)
    }
  }
  FWT_T(2);
  [[maybe_unused]] unsigned long long iters_ = 0;
  // ---- children before parents: every lane starts at its nodes with two leaf children; the second child to
  // arrive at a parent goes on with it (the arrival returns the parent's record)
  int p = -1;
  uint32_t phi = 0, plo = 0;
  for (int guard = 0;; ++guard) {
    if (guard > 2 * ni + 64) { bad = true; break; }  // (cannot happen in a tree)
    if (p < 0 && startmask) {
      const int j = __ffs(startmask) - 1;
      startmask &= startmask - 1;
      p = j * GW + gl;
      phi = hiv[p];
      plo = lo[p];
    }
    if (!gany(p >= 0)) break;
#ifdef TNCO_FWT_PROF
    ++iters_;
#endif
    if (p >= 0) {
      const uint32_t w = plo;
      const int l = (int)(w & 0xFFFFu), rr = (int)(w >> 16);
      const bool li = l >= n, ri = rr >= n;
      const double pl0 = Pn[li ? l - n : p], pr0 = Pn[ri ? rr - n : p];
      const double pl = li ? pl0 : 0.0, pr = ri ? pr0 : 0.0;
      const double c = __longlong_as_double((long long)((uint64_t)((phi >> 16) & 0x7FFu) << 52));
      Pn[p] = (c + pl) + pr;  // (the association order of finite_width/utils.hpp:36-47)
      if (p == ni - 1) {
        p = -1;  // the root
      } else {
        const int q = (int)(phi & 0xFFFFu) - n;
        const uint32_t old = __hip_atomic_fetch_add(&hi[q], 0u - (1u << 27), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t qlo = lo[q];  // (with the arrival: the parent's children, in case this lane goes on with it)
        if ((old >> 27) == 1u) { p = q; phi = old; plo = qlo; } else { p = -1; }
      }
    }
  }
  FWT_T(3);
  if (gany(bad)) {  // (a cost outside the powers of two of a double: the full rebuild decides)
    if (gl == 0) atomicAdd(F.slowstat, 1ull);
    return;
  }
  if (gl == ((ni - 1) & (GW - 1))) misc[0] = (Pn[ni - 1] < cur) ? 1u : 0u;  // greedy/optimizer.hpp:371-374
  if (misc[0]) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int i = j * GW + gl;
      if (i < ni) {
        // the whole header, links and width as they were: two 16-byte stores per node that leave the L2 as whole
        // lines (the second halves alone, 16 of every 32 bytes, were read-modify-writes in the memory: 2.5 x the time)
        const uint32_t w = lo[i], h = hiv[i];
        const double c = __longlong_as_double((long long)((uint64_t)((h >> 16) & 0x7FFu) << 52)), pp = Pn[i];
        const int par = (int)(h & 0xFFFFu);
        int4* d = reinterpret_cast<int4*>(hb + (int64_t)i * P.BS);
        d[0] = make_int4((int)(w & 0xFFFFu), (int)(w >> 16), par == 0xFFFF ? -1 : par, iw[j]);
        d[1] = make_int4(__double2loint(c), __double2hiint(c), __double2loint(pp), __double2hiint(pp));
      }
    }
    const uint64_t* prop = reinterpret_cast<const uint64_t*>(const_cast<const int16_t*>(sc.pos));
    if (gl < LK) sl[gl] = gl < P.W ? prop[gl] : 0ull;
  }
  if (gl == 0) F.fastflag[r] = 1;
#ifdef TNCO_FWT_PROF
  if (gl == 0) {
    const unsigned long long tt4 = __builtin_amdgcn_s_memtime();
    atomicAdd(&g_fwt_prof[0], tt1 - tt0); atomicAdd(&g_fwt_prof[1], tt2 - tt1); atomicAdd(&g_fwt_prof[2], tt3 - tt2);
    atomicAdd(&g_fwt_prof[3], tt4 - tt3); atomicAdd(&g_fwt_prof[4], iters_); atomicAdd(&g_fwt_prof[5], 1ull);
    atomicAdd(&g_fwt_prof[6], misc[0] ? 1ull : 0ull); atomicAdd(&g_fwt_prof[7], (unsigned long long)nd);
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// The three steps above as ONE kernel, one wavefront per replica: fw_wave_kernel = fw_order_kernel |
// fw_slices_kernel | fw_tree_kernel without the memory between them.  Separately, every step started cold (a
// dependent round trip costs 5-10 us here) and the tree kernel's first act was to read back the 12 bytes per node
// the order kernel had just written: per replica and re-slice 6.5 KB written + 6.5 KB read, the list, the change
// list and the proposal through memory, two launches.  Here the headers are read once (J nodes per lane, all loads
// in flight together with the old slices and the generator's position), the node table goes straight into the LDS
// layout of the tree step (lo / hi), the list of the too-wide tensors stays in LDS, the change list is built in LDS.
// LDS per replica: the node table (8 bytes per node) + one region used three times (keys of the ordering; legs,
// counts, generator ring and candidate positions of get_slices; path masks / partial sums of the re-pricing).
// A replica one of the steps cannot do leaves with nwide = -2 (fw_reslice_a_kernel traverses it) or with the
// proposal written and fastflag = 0 (fw_reslice_b_kernel rebuilds it in full), exactly as from the separate kernels.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline size_t fww_lds_bytes(int n, int cap, int T) {  // T: lanes per mask (16, 32 or 64)
  const size_t nip = (size_t)((n - 1 + 63) & ~63);
  const size_t u1 = (size_t)FWO_MAXW * (8 + 2 + 2);                         // keys, nodes, depths
  const size_t u2 = (size_t)cap * T * 8 + (size_t)64 * T + 1024 + 512;       // legs, counts, ring, positions
  const size_t u3 = nip * 8 + 256 + 32;                                     // masks / partial sums, change list, flags
  size_t u = u1 > u2 ? u1 : u2;
  u = u > u3 ? u : u3;
  return (nip * 8 + 512 /* list */ + u + 15) & ~(size_t)15;
}

#ifdef TNCO_FWW_PROF  // (diagnostic build: shader cycles per replica between the steps of fw_wave_kernel)
static __device__ unsigned long long g_fww_prof[12];
#define FWW_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define FWW_T(v)
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

// LOGT: lanes per leg mask (4, 5, 6: networks of at most 16, 32, 64 mask words); 64 >> LOGT tensors per load instruction
template <int J, int LOGT>
static __global__ __launch_bounds__(64) void fw_wave_kernel(const Params P, const FwParams F, const int cap, const int maxnp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fww_smem[];
  FWW_T(w0_);
  constexpr int GW = 64, IPP = 32, T = 1 << LOGT, TPL = 64 >> LOGT;
  const int lane = threadIdx.x, w = lane & (T - 1), g = lane >> LOGT;
  const int64_t r = blockIdx.x;
  const int n = P.n, N = P.N, ni = N - n, W = P.W, LK = F.I64 / 64;
  const int nip = (ni + 63) & ~63;
  // node i of the table: lo = left | right << 16; hi = parent | cost exponent << 16 | internal children (later: still to arrive) << 27
  TNCO_LDS volatile uint32_t* lo = (TNCO_LDS volatile uint32_t*)fww_smem;                 // [nip]
  TNCO_LDS uint32_t* hi = (TNCO_LDS uint32_t*)(lo + nip);                                  // [nip] (atomic arrivals)
  TNCO_LDS volatile uint32_t* hiv = (TNCO_LDS volatile uint32_t*)hi;
  TNCO_LDS volatile uint16_t* wls = (TNCO_LDS volatile uint16_t*)(hi + nip);               // [256] too-wide tensors, post-order
  uint8_t* U = fww_smem + (size_t)nip * 8 + 512;                                           // the region used three times
  // ordering
  TNCO_LDS volatile uint64_t* key = (TNCO_LDS volatile uint64_t*)U;                        // [FWO_MAXW]
  TNCO_LDS volatile uint16_t* wnode = (TNCO_LDS volatile uint16_t*)(key + FWO_MAXW);       // [FWO_MAXW]
  TNCO_LDS volatile uint16_t* dep = wnode + FWO_MAXW;                                      // [FWO_MAXW]
  // get_slices
  TNCO_LDS volatile uint64_t* cache = (TNCO_LDS volatile uint64_t*)U;                      // [cap][T]
  TNCO_LDS volatile uint8_t* nbig = (TNCO_LDS volatile uint8_t*)(cache + (size_t)cap * T);   // [64 T]
  lds_vu32* ring = (lds_vu32*)(nbig + 64 * T);                                             // [256]
  lds_vi32* pos = (lds_vi32*)(ring + 256);                                                 // [FWS_MAXNP]
  // re-pricing
  TNCO_LDS volatile double* Pn = (TNCO_LDS volatile double*)U;                             // [nip]
  TNCO_LDS uint32_t* on = (TNCO_LDS uint32_t*)U;                                           // [nip][2] (the same memory)
  TNCO_LDS volatile uint32_t* onv = (TNCO_LDS volatile uint32_t*)U;
  TNCO_LDS volatile uint32_t* chgl = (TNCO_LDS volatile uint32_t*)(Pn + nip);              // [64] starts of the paths
  TNCO_LDS volatile uint32_t* misc = chgl + 64;                                            // [8]

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
  // ---- the node table; the too-wide tensors (fw_order_kernel)
  int32_t iw[J];  // (the spare header words: a kept re-slice rewrites whole headers)
  int nw = 0;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * GW + lane;
    bool wide = false;
    iw[j] = hd[j].w;
    if (i < ni) {
      const uint32_t c = (hd[j].x >= n ? 1u : 0u) + (hd[j].y >= n ? 1u : 0u);
      lo[i] = (uint32_t)hd[j].x | ((uint32_t)hd[j].y << 16);
      hiv[i] = ((uint32_t)hd[j].z & 0xFFFFu) | (((ce[j] >> 20) & 0x7FFu) << 16) | (c << 27);  // (the root: parent 0xFFFF)
      const double wv = F.width_f32 ? (double)__int_as_float(hd[j].w) : wd[j];
      wide = wv > F.max_width;
    }
    const unsigned long long b = __ballot(wide);
    if (wide) {
      const int k = nw + __popcll(b & ((1ull << lane) - 1ull));
      if (k < FWO_MAXW) wnode[k] = (uint16_t)(n + i);
    }
    nw += __popcll(b);
  }
  if (nw > 255) {
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
    for (int k = lane; k < 256; k += 64) wls[k] = (uint16_t)n;
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
  // ---- get_slices (fw_slices_kernel; the ordering's keys are dead: the region is the legs' now)
  const uint8_t* legs = P.blocks + r * P.RB + P.WOFF;
  const int WS = P.WS;
  uint64_t pl[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) pl[p] = 0ull;
  uint32_t maxc = 0;
  uint64_t m[4];
  auto load16 = [&](int t0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + TPL * u + g;
      const int node = wls[t < nw ? t : 0];
      m[u] = 0ull;
      if (t < nw && has) m[u] = *reinterpret_cast<const uint64_t*>(legs + (int64_t)(node - n) * WS + 8 * w);
    }
  };
  load16(0);
  RngWave rng;
  rng.init(P.mt + r * 624, ring, mti0, mtw0, lane);
  if (nw > 0) rng.fill();
  for (int t0 = 0; t0 < nw; t0 += 4 * TPL) {
    if (t0) load16(t0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + TPL * u + g;
      if (t < nw && t < cap) cache[t * T + w] = m[u];
      const uint32_t c = gsum<LOGT>((uint32_t)__popcll(m[u] & ~skip));
      maxc = c > maxc ? c : maxc;
      uint64_t carry = m[u];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const uint64_t tt = pl[p] & carry;
        pl[p] ^= carry;
        carry = tt;
      }
    }
  }
  if (gmax<6>(maxc) > (uint32_t)maxnp) {  // (nothing drawn yet; the generator's words twisted ahead stay)
    if (lane == 0) rs->mtw = (int)rng.tw;
    leave_to_a();
    return;
  }
  FWW_T(w3_);
#pragma unroll
  for (int step = T; step <= 32; step <<= 1) {
    uint64_t o[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) o[p] = fws_shflx64(pl[p], step);
    uint64_t c = 0ull;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint64_t a = pl[p], b = o[p];
      pl[p] = a ^ b ^ c;
      c = (a & b) | (c & (a ^ b));
    }
  }
  {
    // counters T g .. T g + T - 1 of word w -> T bytes (a nibble of plane bits is spread over the bytes of a word)
    TNCO_LDS volatile uint32_t* d = (TNCO_LDS volatile uint32_t*)(nbig + w * 64 + T * g);
#pragma unroll
    for (int j = 0; j < T / 4; ++j) {
      uint32_t c = 0;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const uint32_t nib = (uint32_t)(pl[p] >> (T * g + 4 * j)) & 0xFu;
        c |= ((nib * 0x00204081u) & 0x01010101u) << p;
      }
      d[j] = c;
    }
  }
  uint64_t ns = 0ull;  // the new slices, word w (the same in the four rows of lanes)
  {
    const double mdl = fw_wr(F, -F.log2d);
    int j = 0;
    while (j < nw) {
      const int t = j + g;
      uint64_t mm = 0ull;
      if (t < nw) {
        if (t < cap) mm = cache[t * T + w];
        else if (has) mm = *reinterpret_cast<const uint64_t*>(legs + (int64_t)((int)wls[t] - n) * WS + 8 * w);
      }
      const uint64_t sx = mm & ~ns;
      const uint32_t cnt = gsum<LOGT>((uint32_t)__popcll(sx));
      const bool wide = t < nw && fw_wr(F, F.log2d * (double)cnt) > F.max_width;
      const unsigned long long bal = __ballot(wide);
      if (bal == 0ull) {
        j += TPL;
        continue;
      }
      const int gs = (__ffsll(bal) - 1) >> LOGT;
      j += gs + 1;
      const uint64_t sxw = fws_shfl64(sx, T * gs + w);
      double sw = fw_wr(F, F.log2d * (double)(uint32_t)__shfl((int)cnt, T * gs));
      const uint64_t cand = sxw & ~skip;
      const uint32_t mine = (uint32_t)__popcll(cand);
      const uint32_t incl = fws_scan<LOGT>(mine, lane);
      const int np = __builtin_amdgcn_readlane((int)incl, T - 1);
      if (g == 0) {
        uint32_t o = incl - mine;
        uint64_t x = cand;
        while (x) {
          const int b = __ffsll((unsigned long long)x) - 1;
          pos[o++] = w * 64 + b;
          x &= x - 1;
        }
      }
      int src0 = lane, src1 = lane + 64;
      if (np >= 2) {
        const uint32_t nd = (uint32_t)np >> 1;
        bool fast = rng.ensure(nd);
        uint32_t p0 = 0, p1 = 0;
        const int base = (np & 1) ? 1 : 0;
        if (fast) {
          const uint32_t raw = rng.peek((uint32_t)lane);
          const uint32_t i0 = (uint32_t)(base + 2 * lane);
          const uint32_t range = (i0 + 1u) * (i0 + 2u);  // (lane 0 of an even count: 2 -- d(0, 1), the swap of position 1)
          const uint64_t product = (uint64_t)raw * (uint64_t)range;
          const uint32_t low = (uint32_t)product;
          bool rej = false;
          if ((uint32_t)lane < nd && low < range) rej = low < (0u - range) % range;
          if (__any(rej)) {
            fast = false;
          } else {
            const uint32_t x = (uint32_t)(product >> 32);
            p0 = x / (i0 + 2u);
            p1 = x - p0 * (i0 + 2u);
            rng.advance(nd);
          }
        }
        if (fast) {
          const bool two = np > 64;
          for (int k = (int)nd - 1; k >= 0; --k) {
            const int j1 = __builtin_amdgcn_readlane((int)p1, k), j0 = __builtin_amdgcn_readlane((int)p0, k);
            const int i = base + 2 * k;
            src0 = src0 == i + 1 ? j1 : (src0 == j1 ? i + 1 : src0);
            src0 = src0 == i ? j0 : (src0 == j0 ? i : src0);
            if (two) {
              src1 = src1 == i + 1 ? j1 : (src1 == j1 ? i + 1 : src1);
              src1 = src1 == i ? j0 : (src1 == j0 ? i : src1);
            }
          }
        } else {
          fw_shuffle_lds<6>(rng, pos, np, lane == 0);
        }
      }
      const int xp0 = lane < np ? (int)pos[src0] : 0, xp1 = lane + 64 < np ? (int)pos[src1] : 0;
      uint32_t k0 = lane < np ? (((uint32_t)nbig[xp0] << 16) | (0xFFFFu - (uint32_t)lane)) : 0u;
      uint32_t k1 = lane + 64 < np ? (((uint32_t)nbig[xp1] << 16) | (0xFFFFu - (uint32_t)(lane + 64))) : 0u;
      for (int taken = 0; taken < np; ++taken) {
        const uint32_t best = gmax<6>(k0 > k1 ? k0 : k1);
        const int qb = (int)(0xFFFFu - (best & 0xFFFFu));
        const int xpos = __shfl(qb >= 64 ? xp1 : xp0, qb & 63);
        if (lane == (qb & 63)) {
          if (qb >= 64) k1 = 0u; else k0 = 0u;
        }
        if (w == (xpos >> 6)) ns |= 1ull << (xpos & 63);
        sw = fw_wr(F, sw + mdl);
        if (sw <= F.max_width) break;
      }
    }
  }
  FWW_T(w4_);
  // ---- the indices that changed, with the starts of their paths (the region is the re-pricing's now).  Their
  // holders are requested first -- up to four per lane in one flight -- and only then the stores of this step are
  // issued: a load behind a store waits for the store's acknowledgement too.
  uint32_t* chg = reinterpret_cast<uint32_t*>(F.delta_scr + r * 64);  // (word 0: the count, for tnco_hip_get_reslice_info)
  int nd;
  uint64_t plus64, minus64;
  {
    uint64_t ch = g == 0 ? (ns ^ old) : 0ull;
    const uint32_t mine = (uint32_t)__popcll(ch);
    const uint32_t incl = fws_scan<LOGT>(mine, lane);
    nd = __builtin_amdgcn_readlane((int)incl, T - 1);
    bool unsup = nd > FWT_MAXD;
    int bits[4];
    int2 hold[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bits[q] = ch ? __ffsll((unsigned long long)ch) - 1 : -1;
      ch &= ch - 1;  // (0 stays 0)
      hold[q] = make_int2(-1, -1);
      if (bits[q] >= 0 && !unsup) hold[q] = *reinterpret_cast<const int2*>(F.holder2 + 2 * (w * 64 + bits[q]));
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
    // the parents of the leaves from the node table, not from the replica's (cold) parent array: one round trip less
    TNCO_LDS volatile uint16_t* lparL = (TNCO_LDS volatile uint16_t*)U;  // [n] (the path masks' memory: cleared below)
    for (int i = lane; i < ni; i += GW) {
      const uint32_t wq = lo[i];
      const int l = (int)(wq & 0xFFFFu), rr = (int)(wq >> 16);
      if (l < n) lparL[l] = (uint16_t)(n + i);
      if (rr < n) lparL[rr] = (uint16_t)(n + i);
    }
    uint64_t plus = 0ull, minus = 0ull;
    if (!unsup) {
      uint32_t off = incl - mine;
      auto entry = [&](int bit, int2 t12) {
        if (t12.x < 0) { unsup = true; return; }
        const int s1 = lparL[t12.x], s2 = t12.y < 0 ? 0xFFFF : (int)lparL[t12.y];
        chgl[off] = (uint32_t)s1 | ((uint32_t)s2 << 16);
        if ((ns >> bit) & 1ull) plus |= 1ull << off; else minus |= 1ull << off;
        ++off;
      };
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (bits[q] >= 0 && !unsup) entry(bits[q], hold[q]);
      while (ch && !unsup) {  // (more than four changed indices in one mask word)
        const int bit = __ffsll((unsigned long long)ch) - 1;
        ch &= ch - 1;
        entry(bit, *reinterpret_cast<const int2*>(F.holder2 + 2 * (w * 64 + bit)));
      }
    }
    unsup = __any(unsup);
    const uint32_t a0 = gsum<LOGT>((uint32_t)plus), a1 = gsum<LOGT>((uint32_t)(plus >> 32));
    const uint32_t b0 = gsum<LOGT>((uint32_t)minus), b1 = gsum<LOGT>((uint32_t)(minus >> 32));
    plus64 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a0) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a1) << 32);
    minus64 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)b0) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)b1) << 32);
    if (lane == 0) chg[0] = unsup ? 0xFFFFFFFFu : (uint32_t)nd;
    if (unsup) {  // more than FWT_MAXD indices, or an index held otherwise: the full rebuild
      if (lane == 0) {
        atomicAdd(F.slowstat, 1ull);
        atomicAdd(F.slowstat + 2, 1ull);
      }
      return;
    }
  }
  const int dbase = __popcll(plus64) - __popcll(minus64);
  FWW_T(w5_);
  // ---- the re-priced costs (fw_tree_kernel): path masks cleared, arrival counters = internal children
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
    const uint32_t plus = (uint32_t)(plus64 >> (IPP * pass)), minus = (uint32_t)(minus64 >> (IPP * pass));
    if (pass) {
      for (int i = lane; i < ni; i += GW) { onv[2 * i] = 0; onv[2 * i + 1] = 0; }
    }
    {
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
    }
    for (int i = lane; i < ni; i += GW) {
      const uint32_t wq = lo[i], h = hiv[i];
      const int l = (int)(wq & 0xFFFFu), rr = (int)(wq >> 16);
      const uint32_t a = onv[2 * i], b = onv[2 * i + 1];
      const int il = l >= n ? l - n : i, ir = rr >= n ? rr - n : i;
      const uint32_t bl = onv[2 * il] & onv[2 * il + 1], br = onv[2 * ir] & onv[2 * ir + 1];
      const uint32_t both_below = (l >= n ? bl : 0u) | (rr >= n ? br : 0u);
      const uint32_t in_u = (a ^ b) | (a & b & ~both_below);
      const int dex = (pass ? 0 : dbase) - __popc(in_u & plus) + __popc(in_u & minus);
      const int ne = (int)((h >> 16) & 0x7FFu) + log2d * dex;
      bad = bad || ne <= 0 || ne >= 2047;  // (also between the passes: the full rebuild decides then)
      hiv[i] = (h & 0xF800FFFFu) | ((uint32_t)(ne & 0x7FF) << 16);
    }
  }
  // ---- children before parents: every lane starts at its nodes with two leaf children; the second child to
  // arrive at a parent goes on with it (the arrival returns the parent's record)
  FWW_T(w6_);
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
    if (need_rec) {  // (no walk kernel ran -- fw_order_kernel / fw_tree_kernel: the post-order of this replica, here)
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
