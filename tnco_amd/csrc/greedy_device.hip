// greedy_device.hip -- the reference's initial contraction trees drawn ON THE DEVICE: the batched
// twin of host_greedy.cpp (same trees, tree for tree; tests/test_gpu_greedy.py).
//
//   tnco/utils/tn.py:189-230: Random(seed).shuffle of the component's tensors (CPython's MT19937,
//   init_by_array, _randbelow), then opt_einsum's greedy path finder with every dimension 2
//   (paths.ssa_greedy_optimize, _simple_chooser, 'memory-removed'; restated, "parity unpinned" like
//   the host version: opt_einsum is not pinned by the reference and absent here).
//
// Why a kernel: for 65 536 runs of a 512-tensor network the host version takes 3 s on the GPU box's 16
// threads -- seven times the 1 000 SA sweeps that follow.
//
// Files: greedy_wave.h (wave helpers), greedy_shuffle.h (the shuffle), greedy_set.h / greedy_graph.h (the two forms of
// the greedy), this file (device memory, the C entry points).
// Two forms (round 4).  Networks without hyper-indices -- every circuit and graph network the benchmarks
// use -- take py_shuffle_lds_kernel + greedy_graph_kernel (below: the greedy over a multigraph, a tree's
// whole state in LDS and registers; 65 536 trees of 512 tensors in 2.6 + 35-41 ms).  The others take the
// set form of round 2 (11 + 177 ms on the same network):
// py_shuffle_kernel: ONE LANE per tree (seeding and shuffling are a serial chain per
// tree, 2 500 dependent steps; the generator's state is a column of a [624][R] array, so the lanes of
// a wavefront read and write whole lines).  greedy_kernel: ONE WAVEFRONT per tree, a persistent grid:
//   * an index set = W 64-bit words, word x in lane x (W <= 64); |set| = a wave reduction;
//   * opt_einsum's candidate queue ordered by (cost, id2, id1) with cost = 2^|k12| - 2^|k1| - 2^|k2|
//     becomes ONE 64-bit key per candidate (greedy_key.h: the cost in non-adjacent form, exact), the
//     queue a flat array in LDS, cell c owned by lane c % 64: every lane keeps the minimum of its
//     cells, a pop is a wave-min over the lanes + a rescan of the winner's cells, a push one LDS
//     store into the cell just popped -- no sift-down chains, never more cells than initial candidates;
//   * the neighbours of the tensor just made (the keys sharing a contractible dim with it) are a
//     bitset over the slots, word x in lane x; the candidates of one push are evaluated one per lane;
//   * index sets are content-addressed slots (a stale queue entry revives when an equal set
//     reappears, as with the frozenset keys of the published code): an open-addressing table in LDS.
// Limits of this path (the host version takes the rest): n_inds <= 2040, n_leaves <= 2000, every
// index held by at most GREEDY_MAXH tensors, LDS need <= 64 KB.  A tree that ends with more than one
// tensor (outer products left) is flagged and redone on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/tnco_hip.h"
#include "dev_cache.h"
#include "greedy_key.h"
#include "greedy_wave.h"
#include "greedy_shuffle.h"
#include "greedy_set.h"
#include "greedy_graph.h"

namespace tnco {
namespace {

// Device memory of a call: ONE block, kept between calls (hipFree of the ~2.5 GB a 65 536-tree batch used
// took 0.1 s -- as long as half of the kernel).  tnco_hip_greedy_device_release() gives it back.
struct Pool {
  void* base = nullptr;
  size_t bytes = 0;
  int device = -1;
  std::mutex mu;
};
Pool g_pool;

struct DevBufs {
  bool dry = true;   // first pass: only add up the sizes
  size_t off = 0;
  char* base = nullptr;
  template <typename T>
  hipError_t alloc(T** p, size_t count) {
    const size_t nb = (std::max<size_t>(count * sizeof(T), 16) + 255) & ~(size_t)255;
    *p = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += nb;
    return hipSuccess;
  }
  // second pass: the block (from the pool if it is large enough and on this device)
  hipError_t commit(int device) {
    const size_t need = off;
    if (g_pool.base == nullptr || g_pool.bytes < need || g_pool.device != device) {
      if (g_pool.base) (void)hipFree(g_pool.base);
      g_pool.base = nullptr;
      g_pool.bytes = 0;
      const hipError_t e = tnco::dev_malloc(&g_pool.base, need);
      if (e != hipSuccess) {
        g_pool.base = nullptr;
        return e;
      }
      g_pool.bytes = need;
      g_pool.device = device;
    }
    base = static_cast<char*>(g_pool.base);
    off = 0;
    dry = false;
    return hipSuccess;
  }
};

void py_init_genrand(uint32_t* mt, uint32_t s) {
  mt[0] = s;
  for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

size_t lds_bytes(int W, int SMAX, int I, int QC, int TS) {
  return (size_t)QC * 8 + (size_t)4 * W * 8 + (size_t)SMAX * 2 + (size_t)GREEDY_LCAP * 2 + (size_t)(SMAX - 8) * 2 + (size_t)I * 2 + (size_t)TS * 2 + 16;
}

}  // namespace
}  // namespace tnco

using namespace tnco;

#define G_TRY(expr)                      \
  do {                                   \
    const hipError_t e_ = (expr);        \
    if (e_ != hipSuccess) return TNCO_HIP_ERUNTIME; \
  } while (0)

// gives the device memory kept between calls back to the driver
extern "C" void tnco_hip_greedy_device_release(void) {
  std::lock_guard<std::mutex> lock(g_pool.mu);
  if (g_pool.base) {
    (void)hipSetDevice(g_pool.device);
    (void)hipFree(g_pool.base);
  }
  g_pool.base = nullptr;
  g_pool.bytes = 0;
  g_pool.device = -1;
}

static int64_t g_last_redone = -1;
// trees of the last tnco_hip_greedy_trees_device call that the host version did (-1: the whole batch)
extern "C" int64_t tnco_hip_diag_greedy_device_redone(void) { return g_last_redone; }

// 1 when tnco_hip_greedy_trees_device takes this network itself (else it hands the batch to the host version)
extern "C" int tnco_hip_diag_greedy_device_supported(int32_t n_leaves, int32_t n_inds, const int32_t* holders_off) {
  if (n_leaves < 3 || n_leaves > 2000 || n_inds < 1 || n_inds > GREEDY_KEY_MAX_EXP || !holders_off) return 0;
  int64_t q = 0;  // candidates queued at most: the initial ones (a push takes the cell of the pop before it)
  for (int32_t i = 0; i < n_inds; ++i) {
    const int32_t m = holders_off[i + 1] - holders_off[i];
    if (m > GREEDY_MAXH) return 0;
    q += m > 1 ? m - 1 : 0;
  }
  const int W = (n_inds + 63) / 64, SMAX = 2 * n_leaves + 8;
  int TS = 64;
  while (4 * TS < 5 * SMAX) TS <<= 1;
  const int QC = (int)((q + 63) & ~63ll);
  return lds_bytes(W, SMAX, n_inds, QC, TS) <= 64 * 1024 ? 1 : 0;
}

extern "C" int tnco_hip_greedy_trees_device(int32_t device, int32_t n_leaves, int32_t n_inds, const int32_t* holders_off,
                                            const int32_t* holders, const uint64_t* output_mask, int64_t n_replicas,
                                            const uint32_t* seeds, uint64_t* draws, int32_t* links_out,
                                            int32_t** links_device, int32_t n_threads) {
  if (n_leaves < 1 || n_inds < 0 || !holders_off || !holders || !seeds || (!links_out && !links_device) || n_replicas < 0)
    return TNCO_HIP_EINVAL;
  if (links_device) *links_device = nullptr;
  if (n_replicas == 0) return TNCO_HIP_OK;
  g_last_redone = -1;
  if (!tnco_hip_diag_greedy_device_supported(n_leaves, n_inds, holders_off)) {
    // the host version; the trees go to the device afterwards if the caller wants them there
    const size_t cnt = (size_t)n_replicas * 3 * (2 * (size_t)n_leaves - 1);
    std::vector<int32_t> tmp;
    if (!links_out) tmp.resize(cnt);
    int32_t* host = links_out ? links_out : tmp.data();
    const int rc = tnco_hip_greedy_trees(n_leaves, n_inds, holders_off, holders, output_mask, n_replicas, seeds, draws,
                                         host, n_threads);
    if (rc || !links_device) return rc;
    G_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> pool_lock(g_pool.mu);
    DevBufs db;
    int32_t* dl = nullptr;
    for (int pass = 0; pass < 2; ++pass) {
      G_TRY(db.alloc(&dl, cnt));
      if (pass == 0) G_TRY(db.commit(device));
    }
    G_TRY(hipMemcpy(dl, host, cnt * 4, hipMemcpyHostToDevice));
    *links_device = dl;
    return TNCO_HIP_OK;
  }
  const auto t_start = std::chrono::steady_clock::now();
  G_TRY(hipSetDevice(device));
  const int n = n_leaves, I = n_inds, W = (I + 63) / 64, SMAX = 2 * n + 8, NW = (SMAX + 63) / 64;
  const int64_t R = n_replicas, N = 2 * (int64_t)n - 1;
  int64_t q = 0;
  for (int i = 0; i < I; ++i) q += std::max(0, holders_off[i + 1] - holders_off[i] - 1);
  const int Q = (int)q, QC = (Q + 63) & ~63;
  int TS = 64;
  while (4 * TS < 5 * SMAX) TS <<= 1;
  std::vector<uint64_t> leaf((size_t)n * W, 0), outm((size_t)W, 0);
  for (int i = 0; i < I; ++i)
    for (int k = holders_off[i]; k < holders_off[i + 1]; ++k) {
      const int t = holders[k];
      if (t < 0 || t >= n) return TNCO_HIP_EINVAL;
      leaf[(size_t)t * W + (i >> 6)] |= 1ull << (i & 63);
    }
  if (output_mask)
    for (int x = 0; x < W; ++x) outm[x] = output_mask[x];
  uint32_t mt0[624];
  py_init_genrand(mt0, 19650218u);

  int cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  GraphHost gh;
  const bool graph = build_graph(n, I, holders_off, holders, output_mask, gh);
  // the set form's queue in registers where the dims' candidates fit 16 rows of 64 cells (greedy_set.h)
  int m1 = 1, set_rows = 0;
  for (int i = 0; i < I; ++i) {
    const bool is_out = output_mask && ((output_mask[i >> 6] >> (i & 63)) & 1ull);
    if (!is_out) m1 = std::max(m1, holders_off[i + 1] - holders_off[i] - 1);
  }
  if (!graph && !(std::getenv("TNCO_HIP_GREEDY_LDS_QUEUE") && std::atoi(std::getenv("TNCO_HIP_GREEDY_LDS_QUEUE")) != 0)) {
    const int need = (I + 63) / 64 * m1;
    for (int rw : {4, 8, 12, 16})
      if (set_rows == 0 && need <= rw) set_rows = rw;
    // 24 rows: two wavefronts per SIMD, eight trees per CU -- taken where the queue in LDS would leave fewer than that
    // (round 5: 1 000 tensors / 689 indices of the raw CZ circuit: 28 KB of LDS per tree with the queue, five per CU)
    if (set_rows == 0 && need <= 24 && (size_t)(160 * 1024) / lds_bytes(W, SMAX, I, QC, TS) < 8) set_rows = 24;
  }
  const int Qa = set_rows ? set_rows * 64 : Q;  // rows of the stored results (one per queue cell)
  const size_t lds = graph ? graph_lds_bytes(n, gh.CAP) : lds_bytes(W, SMAX, I, set_rows ? 0 : QC, TS);
  int per_cu = (int)std::max<size_t>(1, std::min<size_t>(16, (size_t)(160 * 1024) / lds));
  if (per_cu > 12) per_cu &= ~3;  // (13 per CU measured a third slower than 12; 9, 10, 11 each faster than the one before)
  if (!graph) {
    // ... and no more wavefronts than the kernel's registers let a CU hold: every wavefront works through R / G trees, and
    // the ones that only start when the first have finished are a second round of almost the same length (round 5: the
    // 16-row instantiation, 166 VGPRs, was launched 16 per CU where 12 fit)
    const void* kp = set_rows == 4 ? (const void*)greedy_kernel<4> : set_rows == 8 ? (const void*)greedy_kernel<8>
                   : set_rows == 12 ? (const void*)greedy_kernel<12> : set_rows == 16 ? (const void*)greedy_kernel<16>
                   : set_rows == 24 ? (const void*)greedy_kernel<24> : (const void*)greedy_kernel<0>;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kp, 64, lds) == hipSuccess && nb > 0) per_cu = std::min(per_cu, nb);
    else (void)hipGetLastError();
  }
  const int G = (int)std::min<int64_t>(R, (int64_t)cus * per_cu);

  std::lock_guard<std::mutex> pool_lock(g_pool.mu);
  DevBufs db;
  ShuffleParams sp{};
  GreedyParams gp{};
  uint32_t *d_seeds, *d_mt0, *d_mt;
  uint64_t *d_draws = nullptr, *d_leaf, *d_out;
  int32_t *d_hoff, *d_hold;
  uint16_t* d_perm;
  uint16_t *d_toff = nullptr, *d_tnbr = nullptr;
  uint8_t *d_tmult = nullptr, *d_tfp = nullptr, *d_tkp = nullptr;
  uint32_t* d_eends = nullptr;
  uint64_t* d_ekey = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    G_TRY(db.alloc(&d_seeds, (size_t)R));
    G_TRY(db.alloc(&d_mt0, 624));
    G_TRY(db.alloc(&d_mt, (size_t)624 * R));
    G_TRY(db.alloc(&d_perm, (size_t)R * n));
    if (draws) G_TRY(db.alloc(&d_draws, (size_t)R));
    G_TRY(db.alloc(&d_leaf, leaf.size()));
    G_TRY(db.alloc(&d_out, outm.size()));
    G_TRY(db.alloc(&d_hoff, (size_t)I + 1));
    G_TRY(db.alloc(&d_hold, (size_t)std::max(1, holders_off[I])));
    if (!graph) {
      G_TRY(db.alloc(&gp.keys, (size_t)G * SMAX * W));
      G_TRY(db.alloc(&gp.nbr, (size_t)G * SMAX * NW));
      G_TRY(db.alloc(&gp.arena, (size_t)G * Qa * W));
      G_TRY(db.alloc(&gp.path, (size_t)G * n * 2));
      G_TRY(db.alloc(&gp.slot_of_leaf, (size_t)G * n));
    } else {
      G_TRY(db.alloc(&d_toff, gh.t_off.size()));
      G_TRY(db.alloc(&d_tnbr, gh.t_nbr.size()));
      G_TRY(db.alloc(&d_tmult, gh.t_mult.size()));
      G_TRY(db.alloc(&d_tfp, gh.t_fp.size()));
      G_TRY(db.alloc(&d_tkp, gh.t_kp.size()));
      G_TRY(db.alloc(&d_eends, gh.e_ends.size()));
      G_TRY(db.alloc(&d_ekey, gh.e_key.size()));
    }
    G_TRY(db.alloc(&gp.links, (size_t)R * 3 * N));
    G_TRY(db.alloc(&gp.status, (size_t)R));
    G_TRY(db.alloc(&gp.prof, (size_t)G * 16));
    if (pass == 0) G_TRY(db.commit(device));
  }
  G_TRY(hipMemcpy(d_seeds, seeds, (size_t)R * 4, hipMemcpyHostToDevice));
  G_TRY(hipMemcpy(d_mt0, mt0, sizeof(mt0), hipMemcpyHostToDevice));
  if (draws) G_TRY(hipMemcpy(d_draws, draws, (size_t)R * 8, hipMemcpyHostToDevice));
  G_TRY(hipMemcpy(d_leaf, leaf.data(), leaf.size() * 8, hipMemcpyHostToDevice));
  G_TRY(hipMemcpy(d_out, outm.data(), outm.size() * 8, hipMemcpyHostToDevice));
  G_TRY(hipMemcpy(d_hoff, holders_off, ((size_t)I + 1) * 4, hipMemcpyHostToDevice));
  if (holders_off[I] > 0) G_TRY(hipMemcpy(d_hold, holders, (size_t)holders_off[I] * 4, hipMemcpyHostToDevice));
  if (graph) {
    G_TRY(hipMemcpy(d_toff, gh.t_off.data(), gh.t_off.size() * 2, hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_tnbr, gh.t_nbr.data(), gh.t_nbr.size() * 2, hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_tmult, gh.t_mult.data(), gh.t_mult.size(), hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_tfp, gh.t_fp.data(), gh.t_fp.size(), hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_tkp, gh.t_kp.data(), gh.t_kp.size(), hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_eends, gh.e_ends.data(), gh.e_ends.size() * 4, hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_ekey, gh.e_key.data(), gh.e_key.size() * 8, hipMemcpyHostToDevice));
  }

  const bool dbg = std::getenv("TNCO_HIP_DEBUG") != nullptr;
  if (dbg) {
    G_TRY(hipDeviceSynchronize());
    std::fprintf(stderr, "greedy_device: set-up (allocations, inputs to the device) %.1f ms\n",
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  }
  hipEvent_t ev[4];
  if (dbg) {
    for (auto& e : ev) G_TRY(hipEventCreate(&e));
    G_TRY(hipEventRecord(ev[0], 0));
  }
  sp.n = n; sp.R = R; sp.seeds = d_seeds; sp.draws = d_draws; sp.mt0 = d_mt0; sp.mt = d_mt; sp.perm = d_perm;
  {
    // the state in LDS where 4+ trees fit a workgroup's 64 KB; TNCO_HIP_SHUFFLE_LDS=0: the state in memory (tests)
    const size_t per_tree = (size_t)624 * 4 + (size_t)n * 2;
    const char* e = std::getenv("TNCO_HIP_SHUFFLE_LDS");
    const bool in_lds = !(e && std::atoi(e) == 0) && 4 * per_tree <= 64 * 1024;
    if (!in_lds) {
      hipLaunchKernelGGL(py_shuffle_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, 0, sp);
    } else if (16 * per_tree <= 64 * 1024) {
      hipLaunchKernelGGL(py_shuffle_lds_kernel<16>, dim3((unsigned)((R + 15) / 16)), dim3(64), 16 * per_tree, 0, sp);
    } else if (8 * per_tree <= 64 * 1024) {
      hipLaunchKernelGGL(py_shuffle_lds_kernel<8>, dim3((unsigned)((R + 7) / 8)), dim3(64), 8 * per_tree, 0, sp);
    } else {
      hipLaunchKernelGGL(py_shuffle_lds_kernel<4>, dim3((unsigned)((R + 3) / 4)), dim3(64), 4 * per_tree, 0, sp);
    }
  }
  G_TRY(hipGetLastError());
  if (dbg) G_TRY(hipEventRecord(ev[1], 0));
  gp.n = n; gp.I = I; gp.W = W; gp.NW = NW; gp.SMAX = SMAX; gp.Q = Qa; gp.M1 = m1; gp.TS = TS; gp.R = R;
  gp.leaf = d_leaf; gp.output = d_out; gp.hoff = d_hoff; gp.holders = d_hold; gp.perm = d_perm;
  if (graph) {
    GraphParams qp{};
    qp.n = n; qp.E = gh.E; qp.L = gh.L; qp.CAP = gh.CAP; qp.R = R; qp.perm = d_perm;
    qp.dangling = 0;
    for (int t = 0; t < n; ++t)
      if (gh.t_kp[t] != gh.t_fp[t]) qp.dangling = 1;
    qp.t_off = d_toff; qp.t_nbr = d_tnbr; qp.t_mult = d_tmult; qp.t_fp = d_tfp; qp.t_kp = d_tkp;
    qp.e_ends = d_eends; qp.e_key = d_ekey; qp.links = gp.links; qp.status = gp.status; qp.prof = gp.prof;
    const int rows = (gh.E + 63) / 64;
#define TNCO_GRAPH_LAUNCH(RW) hipLaunchKernelGGL(greedy_graph_kernel<RW>, dim3((unsigned)G), dim3(64), lds, 0, qp)
    if (rows <= 4) TNCO_GRAPH_LAUNCH(4);
    else if (rows <= 8) TNCO_GRAPH_LAUNCH(8);
    else if (rows <= 12) TNCO_GRAPH_LAUNCH(12);
    else if (rows <= 16) TNCO_GRAPH_LAUNCH(16);
    else TNCO_GRAPH_LAUNCH(24);
#undef TNCO_GRAPH_LAUNCH
  } else {
#define TNCO_SET_LAUNCH(RW) hipLaunchKernelGGL(greedy_kernel<RW>, dim3((unsigned)G), dim3(64), lds, 0, gp)
    switch (set_rows) {
      case 4: TNCO_SET_LAUNCH(4); break;
      case 8: TNCO_SET_LAUNCH(8); break;
      case 12: TNCO_SET_LAUNCH(12); break;
      case 16: TNCO_SET_LAUNCH(16); break;
      case 24: TNCO_SET_LAUNCH(24); break;
      default: TNCO_SET_LAUNCH(0); break;
    }
#undef TNCO_SET_LAUNCH
  }
  G_TRY(hipGetLastError());
  if (dbg) G_TRY(hipEventRecord(ev[2], 0));
  G_TRY(hipDeviceSynchronize());
  if (links_out) G_TRY(hipMemcpy(links_out, gp.links, (size_t)R * 3 * N * 4, hipMemcpyDeviceToHost));
  if (dbg) {
    G_TRY(hipEventRecord(ev[3], 0));
    G_TRY(hipEventSynchronize(ev[3]));
    float a = 0, b = 0, c = 0;
    (void)hipEventElapsedTime(&a, ev[0], ev[1]);
    (void)hipEventElapsedTime(&b, ev[1], ev[2]);
    (void)hipEventElapsedTime(&c, ev[2], ev[3]);
    std::fprintf(stderr, "greedy_device: shuffle kernel %.1f ms, greedy kernel (%s form) %.1f ms (%d wavefronts, %zu B LDS each), "
                 "links to the host %.1f ms\n", a, graph ? "graph" : "set", b, G, lds, c);
    for (auto& e : ev) (void)hipEventDestroy(e);
#ifdef TNCO_GREEDY_PROF
    std::vector<unsigned long long> pr((size_t)G * 16);
    G_TRY(hipMemcpy(pr.data(), gp.prof, pr.size() * 8, hipMemcpyDeviceToHost));
    const char* nmg[12] = {"set-up", "pop", "bookkeeping, links", "lists joined", "candidates + push", "end", "-", "-", "-", "-", "-", "-"};
    const char* nms[12] = {"clear", "inputs", "dims: counts, neighbours", "dims: candidates", "pop", "slot of the result", "holders per dim", "neighbour rows", "list + broadcast", "evaluate", "winner + push", "links"};
    const char** nm = graph ? nmg : nms;
    double tot = 0;
    for (size_t i = 0; i < pr.size(); ++i) tot += (double)pr[i];
    for (int i = 0; i < 12; ++i) {
      double v = 0;
      for (int g2 = 0; g2 < G; ++g2) v += (double)pr[(size_t)g2 * 16 + i];
      std::fprintf(stderr, "  %-26s %5.1f %%  %9.0f ticks per tree\n", nm[i], 100 * v / tot, v / (double)R);
    }
#endif
  }
  std::vector<int32_t> status((size_t)R);
  G_TRY(hipMemcpy(status.data(), gp.status, (size_t)R * 4, hipMemcpyDeviceToHost));
  std::vector<uint64_t> draws_in;
  if (draws) {
    draws_in.assign(draws, draws + R);
    G_TRY(hipMemcpy(draws, d_draws, (size_t)R * 8, hipMemcpyDeviceToHost));
  }
  // the trees the kernel left to the host (outer products at the end, limits)
  std::vector<int64_t> redo;
  for (int64_t r = 0; r < R; ++r)
    if (status[r] != 0) redo.push_back(r);
  g_last_redone = (int64_t)redo.size();
  if (!redo.empty()) {
    if (std::getenv("TNCO_HIP_DEBUG"))
      for (size_t k = 0; k < std::min<size_t>(redo.size(), 8); ++k)
        std::fprintf(stderr, "greedy_device: tree %lld status %d\n", (long long)redo[k], status[redo[k]]);
    std::vector<uint32_t> s2(redo.size());
    std::vector<uint64_t> d2(redo.size());
    std::vector<int32_t> l2(redo.size() * 3 * (size_t)N);
    for (size_t k = 0; k < redo.size(); ++k) {
      s2[k] = seeds[redo[k]];
      if (draws) d2[k] = draws_in[redo[k]];
    }
    const int rc = tnco_hip_greedy_trees(n_leaves, n_inds, holders_off, holders, output_mask, (int64_t)redo.size(),
                                         s2.data(), draws ? d2.data() : nullptr, l2.data(), n_threads);
    if (rc) return rc;
    for (size_t k = 0; k < redo.size(); ++k) {
      if (links_out) std::memcpy(links_out + redo[k] * 3 * N, l2.data() + k * 3 * (size_t)N, (size_t)3 * N * 4);
      if (draws) draws[redo[k]] = d2[k];
    }
    // back to the device in runs of consecutive trees: a network whose every tree ends in outer products (all of them
    // redone here) is one copy, not one per tree (65 536 copies of 6 KB took 5.6 s)
    for (size_t k = 0; k < redo.size();) {
      size_t e = k + 1;
      while (e < redo.size() && redo[e] == redo[e - 1] + 1) ++e;
      G_TRY(hipMemcpy(gp.links + redo[k] * 3 * N, l2.data() + k * 3 * (size_t)N, (e - k) * (size_t)3 * N * 4, hipMemcpyHostToDevice));
      k = e;
    }
  }
  if (dbg)
    std::fprintf(stderr, "greedy_device: %.1f ms from entry to the return (before the buffers are freed)\n",
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  if (links_device) *links_device = gp.links;
  return TNCO_HIP_OK;
}

// device -> host copy of a buffer this library handed out (the trees of tnco_hip_greedy_trees_device)
extern "C" int tnco_hip_copy_to_host(void* dst, const void* device_src, uint64_t bytes) {
  if (!dst || !device_src) return TNCO_HIP_EINVAL;
  G_TRY(hipMemcpy(dst, device_src, (size_t)bytes, hipMemcpyDeviceToHost));
  return TNCO_HIP_OK;
}
