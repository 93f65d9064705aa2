// tnco_hip.hip -- C ABI (include/tnco_hip.h) over the gfx950 kernels of sa_kernels.h / sa_sweep.h.
// Host side: argument checking, device memory, launches, read-back.
#include "../../include/tnco_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "host_ctx.h"
#include "extract_kernels.h"

using namespace tnco;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(TNCO_HIP_ERUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

// frees a set of temporary device buffers on scope exit
// scratch of one call, from / back to the cache of device blocks (dev_cache.h)
struct TempBufs {
  std::vector<void*> ptrs;
  std::vector<size_t> sizes;
  int device = 0;
  TempBufs() { (void)hipGetDevice(&device); }
  template <typename T>
  hipError_t alloc(T** p, int64_t count) {
    void* q = nullptr;
    const size_t nb = (size_t)std::max<int64_t>(count, 1) * sizeof(T);
    hipError_t e = tnco::DevCache::get().take(&q, nb, device);
    if (e == hipSuccess) {
      ptrs.push_back(q);
      sizes.push_back(nb);
    }
    *p = (T*)q;
    return e;
  }
  ~TempBufs() {
    if (ptrs.empty()) return;
    (void)hipDeviceSynchronize();  // (hipFree used to wait for the kernels still reading these)
    for (size_t i = 0; i < ptrs.size(); ++i) tnco::DevCache::get().give(ptrs[i], sizes[i], device);
  }
};

}  // namespace

namespace {

// (LOG2L, K) for W mask words
void choose_lanes(int W, int* log2l, int* K) {
  if (W <= 16) { *log2l = 2; *K = (W + 3) / 4; }
  else if (W <= 32) { *log2l = 3; *K = W <= 24 ? 3 : 4; }
  else { *log2l = 4; *K = W <= 48 ? 3 : 4; }
}

// A replica's W mask words are spread over L = 2^LOG2L lanes, K words per lane.  Small groups
// keep the per-replica scalar work (done by every lane of the group) cheap and put more replicas
// in a wavefront:  W <= 16 -> 4 lanes;  W <= 32 -> 8 lanes;  W <= 64 -> 16 lanes.  The kernels of
// one (LOG2L, K) pair are compiled in a translation unit of their own (inst_<L>_<K>.hip).
#define DISPATCH_LK(h, CALL)                                  \
  switch ((h)->log2l * 8 + (h)->K) {                          \
    case 2 * 8 + 1: CALL(2, 1); break;                        \
    case 2 * 8 + 2: CALL(2, 2); break;                        \
    case 2 * 8 + 3: CALL(2, 3); break;                        \
    case 2 * 8 + 4: CALL(2, 4); break;                        \
    case 3 * 8 + 3: CALL(3, 3); break;                        \
    case 3 * 8 + 4: CALL(3, 4); break;                        \
    case 4 * 8 + 3: CALL(4, 3); break;                        \
    default: CALL(4, 4); break;                               \
  }

void launch_run(tnco_hip_ctx* h, const double* betas, int64_t n_steps, int prob_kind, hipStream_t s, int block0 = 0,
                int nblocks = -1) {
#define CALL_RUN(LL, KK) launch_run_lk<LL, KK>(h, betas, n_steps, prob_kind, s, block0, nblocks)
  DISPATCH_LK(h, CALL_RUN)
#undef CALL_RUN
}
int run_blocks_per_cu(tnco_hip_ctx* h) {
  int nb = 0;
#define CALL_OCC(LL, KK) nb = run_blocks_per_cu_lk<LL, KK>(h)
  DISPATCH_LK(h, CALL_OCC)
#undef CALL_OCC
  return nb;
}
int lds_kernel_prepare(tnco_hip_ctx* h, int device_lds_bytes) {
  int rc = 1;
#define CALL_LDSP(LL, KK) rc = lds_kernel_prepare_lk<LL, KK>(h, device_lds_bytes)
  DISPATCH_LK(h, CALL_LDSP)
#undef CALL_LDSP
  return rc;
}
// LDS a workgroup may use on this device (gfx950: 160 KiB per CU; a runtime that reports less for it is not believed).
static int device_lds_bytes(const hipDeviceProp_t& prop) {
  int v = (int)std::max<size_t>(prop.maxSharedMemoryPerMultiProcessor, prop.sharedMemPerBlock);
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) v = std::max(v, 160 * 1024);
  return v;
}
void launch_build(tnco_hip_ctx* h, const BuildArgs& a) {
#define CALL_BUILD(LL, KK) launch_build_lk<LL, KK>(h, a)
  DISPATCH_LK(h, CALL_BUILD)
#undef CALL_BUILD
}
void launch_compare(tnco_hip_ctx* h, const BuildArgs& a, double atol, int32_t* out_bad) {
#define CALL_CMP(LL, KK) launch_compare_lk<LL, KK>(h, a, atol, out_bad)
  DISPATCH_LK(h, CALL_CMP)
#undef CALL_CMP
}

void launch_fw_check(tnco_hip_ctx* h, const BuildArgs& a, int which_min, double atol, int32_t* out_bad) {
#define CALL_FWC(LL, KK) launch_fw_check_lk<LL, KK>(h, a, which_min, atol, out_bad)
  DISPATCH_LK(h, CALL_FWC)
#undef CALL_FWC
}

void launch_fw_init(tnco_hip_ctx* h, const FwInitArgs& a) {
#define CALL_FWI(LL, KK) launch_fw_init_lk<LL, KK>(h, a)
  DISPATCH_LK(h, CALL_FWI)
#undef CALL_FWI
}

// (two lanes per replica, from both ends of the post-order: the same kernel whatever the lane layout of the handle)
void launch_fw_walk(tnco_hip_ctx* h) {
  hipLaunchKernelGGL(fw_walk2_kernel, dim3((unsigned)((h->P.R + 127) / 128)), dim3(256), 0, h->stream, h->P, h->F);
}

// n_steps sweeps of the finite-width optimizer: [moves up to and including the next re-slicing
// sweep][re-slice] ... [the remaining moves]  (sweep k re-slices when (off + k) % every == 0)
hipError_t launch_fw_run(tnco_hip_ctx* h, const double* betas, int64_t n_steps, int prob_kind, int64_t off,
                         int64_t every, bool count_reslices = true) {
  int64_t cur = 0;
  while (cur < n_steps) {
    int64_t next = n_steps;  // first re-slicing sweep >= cur
    if (every > 0) {
      const int64_t rem = (off + cur) % every;
      next = cur + (rem == 0 ? 0 : every - rem);
    }
    const bool reslice = next < n_steps;
    const int64_t cnt = (reslice ? next + 1 : n_steps) - cur;
    const int tail_last = reslice ? 0 : 1;
    hipError_t e = h->timed(TNCO_KIND_FW_MOVE, [&]() {
#define CALL_FWM(LL, KK) launch_fw_move_lk<LL, KK>(h, betas + cur, cnt, prob_kind, tail_last)
      DISPATCH_LK(h, CALL_FWM)
#undef CALL_FWM
    });
    if (e != hipSuccess) return e;
    if (reslice) {
      // the wavefront form lists the too-wide tensors itself; the general form gets them (and the post-order) from a
      // walk kernel, unless the tree is too large for its stack fields, a leaf is too wide, or the test knob
      // TNCO_HIP_FW_STACK=0 asks for the link-walking traverse inside fw_reslice_kernel
      const bool wave = h->F.fast_ok != 0;
      const int prewalked = wave ? 3 : ((h->P.N <= 8192 && h->F.stack_cap > 0 && !h->F.leaf_wide) ? 2 : 0);
      if (prewalked == 2) {
        e = h->timed(TNCO_KIND_FW_WALK, [&]() { launch_fw_walk(h); });
        if (e != hipSuccess) return e;
      }
      if (count_reslices && wave) h->fw_wave_reslices += 1;
      h->fw_stats[wave ? 0 : 5] += h->P.R;
      e = h->timed(TNCO_KIND_FW_RESLICE, [&]() {
#define CALL_FWS(LL, KK) launch_fw_reslice_lk<LL, KK>(h, prewalked)
        DISPATCH_LK(h, CALL_FWS)
#undef CALL_FWS
      });
      if (e != hipSuccess) return e;
    }
    cur += cnt;
  }
  return hipSuccess;
}

// The replicas [r0, r0 + cnt) of a handle as a batch of their own: every per-replica array starts at r0, so the
// kernels -- which index everything by the replica's number in the launch -- run unchanged on a part of the batch.
// A finite-width handle runs its halves on two streams that way (tnco_hip_run_fw): the request-bound moves of one
// half overlap the latency-bound re-slice kernels of the other (two handles of half the replicas: +11 ... +15 %,
// tools/fw_two_handles_probe.py).
struct GroupView {
  tnco_hip_ctx* h;
  Params P0;
  FwParams F0;
  hipStream_t s0;
  GroupView(tnco_hip_ctx* h_, int64_t r0, int64_t cnt, hipStream_t s) : h(h_), P0(h_->P), F0(h_->F), s0(h_->stream) {
    Params& P = h->P;
    FwParams& F = h->F;
    const int64_t N = P.N, n = P.n, LK = (int64_t)h->L * h->K;
    P.R = cnt;
    P.blocks += r0 * P.RB; P.lpar += r0 * n * LPS; P.mt += r0 * 624; P.mtshadow += r0 * MT_SHADOW; P.rs += r0;
    P.minlinks += r0 * N; P.jlog += r0 * (int64_t)P.jcap;
    if (h->fw) {
      F.slices += r0 * 2 * LK; F.scratch_i += r0 * fw_scratch_ints((int)N, F.I64); F.scratch_d += r0 * 2 * N; F.status += r0;
      if (F.width64) F.width64 += r0 * N;
      if (F.nwide) F.nwide += r0;
      if (F.nwfront) F.nwfront += r0;
      if (F.fastflag) F.fastflag += r0;
      if (F.delta_scr) F.delta_scr += r0 * 64;
    }
    h->stream = s;
  }
  ~GroupView() {
    h->P = P0;
    h->F = F0;
    h->stream = s0;
  }
};

// checkpoint := current tree, replica state := fresh
__global__ void finish_init_kernel(Params P, const double* sum, const double* total) {
  const int64_t r = blockIdx.x;
  const int n = P.n;
  const uint8_t* blk = P.blocks + r * P.RB;
  const int32_t* lp = P.lpar + r * (int64_t)n * LPS;
  Links* ml = P.minlinks + r * (int64_t)P.N;
  for (int i = threadIdx.x; i < P.N; i += blockDim.x) {
    Links o;
    if (i < n) {
      o.left = -1; o.right = -1; o.parent = lp[(int64_t)i * LPS];
    } else {
      const NodeRec* hd = reinterpret_cast<const NodeRec*>(blk + (int64_t)(i - n) * P.BS);
      o.left = hd->left; o.right = hd->right; o.parent = hd->parent;
    }
    o.pad = 0;
    ml[i] = o;
  }
  if (threadIdx.x == 0) {
    ReplicaState* rs = P.rs + r;
    rs->min_cost = sum[r];
    rs->init_total = total[r];
    rs->n_moves = 0; rs->n_accepted = 0; rs->n_improved = 0; rs->n_randpick = 0; rs->n_fullcopy = 0;
    rs->status = 0; rs->jinvalid = 0; rs->jmin = 0; rs->jtail = 0; rs->pad0 = 0; rs->pad2 = 0;
  }
}

// the position field of a prng_state (the words are copied straight into P.mt): nothing is left to twist
__global__ void mt_pos_kernel(ReplicaState* rs, const uint32_t* pos, const int64_t* ids, int64_t k) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  ReplicaState* x = rs + (ids ? ids[i] : i);
  x->mti = (int32_t)pos[i];
  x->mtw = 624;
}

// the `min_ctree` constructor argument: checkpoint := these links, no rotation logged
__global__ void set_minlinks_kernel(Params P, const int32_t* in, int64_t stride) {
  const int64_t r = blockIdx.x;
  const int32_t* lk = in + r * stride;
  Links* ml = P.minlinks + r * (int64_t)P.N;
  for (int i = threadIdx.x; i < P.N; i += blockDim.x) {
    Links o;
    o.left = lk[i]; o.right = lk[P.N + i]; o.parent = lk[2 * (int64_t)P.N + i]; o.pad = 0;
    ml[i] = o;
  }
}
// ... min_total_cost := get_cost(min_ctree[, min_slices]); the log restarts with the next improvement,
// which takes a whole copy (the current tree is not the checkpoint's)
__global__ void restore_min_kernel(ReplicaState* rs, int64_t r0, int64_t count, const double* sum) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= count) return;
  ReplicaState* x = rs + r0 + q;
  x->min_cost = sum[q];
  x->jmin = 0; x->jtail = 0; x->jinvalid = 1;
}

// ---- host-side tree checks: Node::is_valid (include/tnco/node.hpp:72-107) +
// Tree::is_valid (include/tnco/tree.hpp:58-139) -----------------------------
const char* tree_check(int32_t N, const int32_t* left, const int32_t* right, const int32_t* parent,
                       std::vector<int32_t>& cp, std::vector<int32_t>& cc) {
  const int32_t n = (N + 1) / 2;
  int roots = 0, leaves = 0;
  for (int32_t i = 0; i < N; ++i) {
    const int32_t xs[3] = {parent[i], left[i], right[i]};
    for (int k = 0; k < 3; ++k)
      if (!(xs[k] == -1 || (xs[k] >= 0 && xs[k] < N))) return "Nodes are not valid";
    if ((left[i] < 0) != (right[i] < 0)) return "Nodes are not valid";
    if (left[i] >= 0 && left[i] == right[i]) return "Nodes are not valid";
    if (left[i] >= 0 && parent[i] >= 0 && (parent[i] == left[i] || parent[i] == right[i]))
      return "Nodes are not valid";
    roots += parent[i] < 0;
    leaves += left[i] < 0;
  }
  if (parent[N - 1] != -1) return "Last node should be root.";
  if (roots != 1) return "There should be only one root.";
  for (int32_t i = 0; i < n; ++i)
    if (left[i] >= 0) return "All leaves should be first.";
  if (leaves != n) return "Number of nodes is not constenst with the number of leaves.";
  std::fill(cp.begin(), cp.end(), 0);
  std::fill(cc.begin(), cc.end(), 0);
  for (int32_t i = 0; i < N; ++i) {
    if (left[i] >= 0) { cc[left[i]]++; cc[right[i]]++; }
    if (parent[i] >= 0) cp[parent[i]]++;
  }
  for (int32_t i = 0; i < N; ++i) {
    if (cp[i] != (left[i] < 0 ? 0 : 2)) return "Tree is not valid.";
    if (cc[i] != (parent[i] < 0 ? 0 : 1)) return "Tree is not valid.";
    if (left[i] >= 0 && (parent[left[i]] != i || parent[right[i]] != i)) return "Tree is not valid.";
  }
  return nullptr;
}

// The same checks for trees that are already in device memory (tnco_hip_greedy_trees_device leaves
// them there), one wavefront per tree.  The host version's two counters (how often a node is named as
// a parent / as a child) follow from the rest: with one root, n leaves, every child link answered by the
// child's parent pointer and left != right, the 2(n - 1) child links name 2(n - 1) different nodes
// other than the root -- each exactly once -- and the N - 1 parent pointers are those links.
// Status: 0 ok, else the number of the first check of tree_check that fails.
__global__ __launch_bounds__(64) void tree_check_kernel(int32_t N, const int32_t* links, int64_t stride, int64_t ntrees,
                                                        int32_t* out_status) {
  const int64_t r = blockIdx.x;
  if (r >= ntrees) return;
  const int32_t* left = links + r * stride;
  const int32_t* right = left + N;
  const int32_t* parent = left + 2 * (int64_t)N;
  const int32_t n = (N + 1) / 2;
  const int lane = threadIdx.x;
  int bad_node = 0, roots = 0, leaves = 0, leaf_late = 0, bad_tree = 0;
  for (int32_t i = lane; i < N; i += 64) {
    const int32_t p = parent[i], l = left[i], rr = right[i];
    const int32_t xs[3] = {p, l, rr};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (!(xs[k] == -1 || (xs[k] >= 0 && xs[k] < N))) bad_node = 1;
    if ((l < 0) != (rr < 0)) bad_node = 1;
    if (l >= 0 && l == rr) bad_node = 1;
    if (l >= 0 && p >= 0 && (p == l || p == rr)) bad_node = 1;
    roots += p < 0;
    leaves += l < 0;
    if (i < n && l >= 0) leaf_late = 1;
    if (!bad_node && l >= 0 && rr >= 0 && (parent[l] != i || parent[rr] != i)) bad_tree = 1;
  }
  for (int o = 32; o; o >>= 1) {
    roots += __shfl_xor(roots, o);
    leaves += __shfl_xor(leaves, o);
  }
  int st = 0;
  if (__any(bad_node)) st = 1;
  else if (parent[N - 1] != -1) st = 2;
  else if (roots != 1) st = 3;
  else if (__any(leaf_late)) st = 4;
  else if (leaves != n) st = 5;
  else if (__any(bad_tree)) st = 6;
  if (lane == 0) out_status[r] = st;
}
const char* tree_check_message(int st) {
  switch (st) {
    case 1: return "Nodes are not valid";
    case 2: return "Last node should be root.";
    case 3: return "There should be only one root.";
    case 4: return "All leaves should be first.";
    case 5: return "Number of nodes is not constenst with the number of leaves.";
    default: return "Tree is not valid.";
  }
}

// Post-order of include/tnco/utils.hpp:34-51.
void host_traverse(int32_t N, const int32_t* left, const int32_t* right, std::vector<int32_t>& order) {
  std::vector<int32_t> stack;
  std::vector<uint8_t> visited((size_t)N, 0);
  order.clear();
  stack.push_back(N - 1);
  while (!stack.empty()) {
    const int32_t pos = stack.back();
    if (visited[pos] || left[pos] < 0) {
      stack.pop_back();
      order.push_back(pos);
    } else {
      visited[pos] = 1;
      stack.push_back(right[pos]);
      stack.push_back(left[pos]);
    }
  }
}

// Legs of every node from the leaves (tnco/ctree.py:163-189).
void host_derive(const tnco_hip_ctx* h, const int32_t* left, const int32_t* right, uint64_t* masks) {
  const int n = h->P.n, N = h->P.N, W = h->P.W;
  std::vector<int32_t> order;
  host_traverse(N, left, right, order);
  std::vector<uint64_t> uni((size_t)N * W), outside((size_t)N * W, 0);
  std::memcpy(uni.data(), h->leafmask_w.data(), sizeof(uint64_t) * (size_t)n * W);
  std::memcpy(masks, h->leafmask_w.data(), sizeof(uint64_t) * (size_t)n * W);
  for (int32_t p : order)
    if (left[p] >= 0)
      for (int w = 0; w < W; ++w) uni[(size_t)p * W + w] = uni[(size_t)left[p] * W + w] | uni[(size_t)right[p] * W + w];
  for (int w = 0; w < W; ++w) outside[(size_t)(N - 1) * W + w] = h->outmask_w[w];
  for (auto it = order.rbegin(); it != order.rend(); ++it) {
    const int32_t p = *it;
    if (left[p] < 0) continue;
    for (int w = 0; w < W; ++w) {
      outside[(size_t)left[p] * W + w] = outside[(size_t)p * W + w] | uni[(size_t)right[p] * W + w];
      outside[(size_t)right[p] * W + w] = outside[(size_t)p * W + w] | uni[(size_t)left[p] * W + w];
    }
  }
  for (int32_t p : order)
    if (left[p] >= 0)
      for (int w = 0; w < W; ++w) {
        const uint64_t a = masks[(size_t)left[p] * W + w], b = masks[(size_t)right[p] * W + w];
        masks[(size_t)p * W + w] = (a ^ b) | (a & b & outside[(size_t)p * W + w]);
      }
}

const char* status_message(int st) {
  switch (st) {
    case 10:
    case 11: return "Contraction is not valid.";
    case 12: return "'node_masks' leaves differ from 'leaf_masks'.";
    default: return "Tree is not valid.";
  }
}

bool bad_log2(double x) {
  const double l = std::log2(x);
  return std::isinf(l) || std::isnan(l);
}

bool logclose(double x, double y, double atol) {
  if (x < 0 || y < 0) return false;
  if (x == 0 || y == 0) return x == y;
  return std::fabs(std::log(x) - std::log(y)) <= atol;
}

// finite width: problems a kernel met after `create` (a tensor with more candidate legs than the
// re-slice has scratch for); the stream must be idle
int fw_runtime_status(tnco_hip_handle h) {
  if (!h->fw) return TNCO_HIP_OK;
  std::vector<int32_t> st((size_t)h->P.R);
  HIP_TRY(hipMemcpy(st.data(), h->F.status, st.size() * 4, hipMemcpyDeviceToHost));
  for (int32_t x : st)
    if (x) return fail(TNCO_HIP_ENOTIMPL, "finite width: candidate legs beyond the re-slice scratch (internal error).");
  return TNCO_HIP_OK;
}

int fetch_rs(tnco_hip_handle h, std::vector<ReplicaState>& rs) {
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  if (int rc = fw_runtime_status(h)) return rc;
  rs.resize((size_t)h->P.R);
  HIP_TRY(hipMemcpy(rs.data(), h->P.rs, (size_t)h->P.R * sizeof(ReplicaState), hipMemcpyDeviceToHost));
  return TNCO_HIP_OK;
}

}  // namespace

extern "C" {

const char* tnco_hip_last_error(void) { return g_err.c_str(); }
const char* tnco_hip_version(void) { return "tnco_hip 0.6 (gfx950), round 5"; }

int tnco_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void tnco_hip_destroy(tnco_hip_handle h) { delete h; }

// device memory of destroyed handles kept for the next create (dev_cache.h): given back / how much is held
void tnco_hip_release_cached(void) {
  tnco::DevCache::get().release_all();
  tnco::StreamCache::get().release_all();
}
uint64_t tnco_hip_diag_cached_bytes(void) { return (uint64_t)tnco::DevCache::get().held(); }

int tnco_hip_create(const tnco_hip_desc* d, tnco_hip_handle* out) {
  if (!d || !out) return fail(TNCO_HIP_EINVAL, "null argument.");
  *out = nullptr;
  if (d->n_leaves < 2) return fail(TNCO_HIP_EINVAL, "Precision is too low.");  // total cost 0 -> log2 = -inf (optimizer.hpp:77-80)
  if (d->n_inds < 0 || d->n_replicas <= 0) return fail(TNCO_HIP_EINVAL, "'n_inds' / 'n_replicas' are not valid.");
  if (!d->leaf_masks || !d->links || (!d->seeds && !d->prng_states)) return fail(TNCO_HIP_EINVAL, "null input array.");
  if (d->prng_states)
    for (int64_t r = 0; r < d->n_replicas; ++r)
      if (d->prng_states[r * 625 + 624] > 624) return fail(TNCO_HIP_EINVAL, "prng position out of range.");
  if (d->cost_dtype != TNCO_HIP_F64 && d->cost_dtype != TNCO_HIP_F32)
    return fail(TNCO_HIP_ENOTIMPL, "cost_type must be float64 or float32.");
  const int n = d->n_leaves, N = 2 * n - 1, I = d->n_inds;
  const int W = std::max(1, (I + 63) / 64);
  if (W > 64) return fail(TNCO_HIP_ENOTIMPL, "more than 4096 indices are not supported yet.");
  if (d->sparse_mask && d->n_projs == 0) return fail(TNCO_HIP_ERUNTIME, "'n_projs' must be a positive number.");
  const int64_t R = d->n_replicas;

  // dims (include/tnco/ctree.hpp:79-89, 121-135)
  bool uniform = true;
  uint64_t dim_u = d->dim_uniform;
  if (d->dims) {
    for (int i = 0; i < I; ++i)
      if (d->dims[i] == 0) return fail(TNCO_HIP_EINVAL, "Dimensions must be positive numbers");
    for (int i = 1; i < I; ++i) uniform &= d->dims[i] == d->dims[0];
    if (I > 0 && uniform) dim_u = d->dims[0];
    if (I == 0) uniform = false;
  }
  if (uniform && dim_u == 0) return fail(TNCO_HIP_EINVAL, "Dimensions must be positive numbers");

  // finite width?  (tnco/app/app.py:866-870: finite max_width selects the finite_width optimizer)
  const bool fw = std::isfinite(d->max_width);
  if (fw) {
    // (the greedy re-slice packs a too-wide count and a shuffled rank into 16 bits each, fw_kernels.h)
    if (N > 65535) return fail(TNCO_HIP_ENOTIMPL, "finite width: more than 32768 tensors are not supported.");
    if (d->max_width < 0) return fail(TNCO_HIP_ERUNTIME, "'max_width' must be a non-negative number.");
    if (d->width_dtype != TNCO_HIP_F32 && d->width_dtype != TNCO_HIP_F64)
      return fail(TNCO_HIP_ENOTIMPL, "finite width: width_type must be float32 or float64.");
  }

  // links already in device memory (tnco_hip_greedy_trees_device): validated there, never copied
  bool links_on_device = false;
  {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, d->links) == hipSuccess) {
      if (at.type == hipMemoryTypeDevice) {
        if (at.device != d->device) return fail(TNCO_HIP_EINVAL, "'links' are in the memory of another device.");
        links_on_device = true;
      }
    } else {
      (void)hipGetLastError();  // (plain host memory: not an error)
    }
  }
  // host-side structural validation of every tree
  if (!links_on_device) {
    const int64_t ntrees = d->links_stride == 0 ? 1 : R;
    const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ntrees / 64 + 1, 16), std::thread::hardware_concurrency()));
    std::vector<const char*> errs((size_t)nth, nullptr);
    std::vector<std::thread> th;
    for (int t = 0; t < nth; ++t)
      th.emplace_back([&, t]() {
        std::vector<int32_t> cp((size_t)N), cc((size_t)N);
        for (int64_t r = t; r < ntrees && !errs[t]; r += nth) {
          const int32_t* lk = d->links + r * d->links_stride;
          errs[t] = tree_check(N, lk, lk + N, lk + 2 * (int64_t)N, cp, cc);
        }
      });
    for (auto& x : th) x.join();
    for (auto e : errs)
      if (e) return fail(TNCO_HIP_EINVAL, e);
  }

  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (d->device < 0 || d->device >= ndev) return fail(TNCO_HIP_EINVAL, "'device' is not valid.");
  HIP_TRY(hipSetDevice(d->device));

  // TNCO_HIP_DEBUG: wall time of the steps of this call on stderr (the device drained at every mark)
  const bool cdbg = std::getenv("TNCO_HIP_DEBUG") != nullptr;
  auto c_t0 = std::chrono::steady_clock::now();
  auto cmark = [&](const char* what) {
    if (!cdbg) return;
    (void)hipDeviceSynchronize();
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "create: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - c_t0).count());
    c_t0 = t;
  };
  tnco_hip_ctx* h = new tnco_hip_ctx();
  struct Guard {
    tnco_hip_ctx* h;
    ~Guard() { delete h; }
  } guard{h};
  h->device = d->device;
  HIP_TRY(tnco::StreamCache::get().take(&h->own_stream, d->device));
  h->stream = h->own_stream;

  choose_lanes(W, &h->log2l, &h->K);
  h->L = 1 << h->log2l;
  const int L = h->L * h->K;  // padded words per mask row in the shared tables

  // hyper legs present?  (an index held by more than two of {leaves, output})
  h->leafmask_w.assign(d->leaf_masks, d->leaf_masks + (size_t)n * W);
  h->outmask_w.assign((size_t)W, 0);
  if (d->output_mask) h->outmask_w.assign(d->output_mask, d->output_mask + W);
  {
    std::vector<int> cnt((size_t)W * 64, 0);
    for (int t = 0; t < n; ++t)
      for (int p = 0; p < W * 64; ++p) cnt[p] += (int)((h->leafmask_w[(size_t)t * W + (p >> 6)] >> (p & 63)) & 1);
    bool hy = d->node_masks != nullptr;
    for (int p = 0; p < W * 64; ++p) {
      const int c = cnt[p] + (int)((h->outmask_w[p >> 6] >> (p & 63)) & 1);
      if (c > 2) hy = true;
      if (p >= I && cnt[p]) return fail(TNCO_HIP_EINVAL, "index position out of range in 'leaf_masks'.");
    }
    h->hyper = hy;
  }
  const bool f32 = d->cost_dtype == TNCO_HIP_F32;
  const bool pow2u = uniform && (dim_u & (dim_u - 1)) == 0;
  h->generic = !(pow2u && !d->sparse_mask && !f32);
  // Few small trees: every replica's whole tree in LDS during a launch (sa_small.h) -- if all the replicas are resident
  // at once (decided below, where the device is known).
  h->small_tree = !fw && !h->hyper && !h->generic && W <= 2 && n >= 2 && n - 1 <= SMALL_MAX_INTERNAL && LPS == 1;
#ifdef TNCO_NO_SMALL_TREE  // (the A/B library of tools/small_tree_ab.py: `make nosmall`)
  h->small_tree = false;
#endif
  Params& P = h->P;
  P.n = n; P.N = N; P.I = I; P.W = W; P.R = R;
  P.BS = (32 + 8 * W + 31) / 32 * 32;
  // (Networks with hyper-indices, round 5: hyper[p] = legs(p) & legs(c0) & legs(c1) is derived by the kernels from legs
  //  they hold anyway, so no layout stores hyper legs: the blocks are those of a network without hyper-indices.)
  // Blocks longer than a line are PACKED (a 224-byte block at 24 mask words straddles two or three 128-byte lines
  // depending on where it starts): padding them to whole lines costs as many lines as it saves (round 3), and splitting
  // them into headers + [partial copy | legs] records loses 5-30 % to the second dirty line per move (round 5).
  P.WS = P.BS; P.WOFF = 32; P.RB = (int64_t)(n - 1) * P.BS;
  if (fw) {  // split layout (sa_kernels.h, Params)
    P.BS = 32;
    P.WS = (8 * W + 63) / 64 * 64;
    P.WOFF = ((n - 1) * 32 + 127) / 128 * 128;
    P.RB = ((int64_t)P.WOFF + (int64_t)(n - 1) * P.WS + 127) / 128 * 128;
  }
  P.f32 = f32; P.disable_shared = d->disable_shared_inds ? 1 : 0;
  P.cost_mode = uniform ? (pow2u ? 0 : 1) : 2;
  if (!uniform) {  // per-index dims, all powers of two: exponent classes only, no leg loop
    bool allp2 = true;
    for (int i = 0; i < I && allp2; ++i) {
      const uint64_t x = d->dims[i];
      allp2 = (x & (x - 1)) == 0;
    }
    if (allp2) P.cost_mode = 3;
  }
  P.log2d = 0;
  if (pow2u) while ((1ull << P.log2d) < dim_u) ++P.log2d;
  auto rc = [&](double x) { return f32 ? (double)(float)x : x; };
  P.n_projs = d->sparse_mask ? rc((double)d->n_projs) : 0.0;

  // Rotation log: one int32 per accepted move since the last re-base of the checkpoint.  When it is
  // full, the next improvement re-bases the checkpoint on the current tree (a 16 N-byte copy) and the
  // log starts again, so its size only sets how often that happens: 32 Ki entries (128 KB per replica)
  // are one re-base per ~2000 sweeps of the 512-leaf benchmark (a re-base stalls its wavefront: with
  // 16 Ki entries every replica took one in the 1200-sweep bench run, -3 %).  (Round 1 gave the log an eighth of
  // the free HBM -- 34 GB at 65536 replicas -- and hipMalloc of it took 1.4 s of every create().)
  {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    free_b += tnco::DevCache::get().held();  // (what destroyed handles left for this one counts as free)
    int64_t cap = (int64_t)(free_b / 8) / (R * 4);
    cap = std::max<int64_t>(1024, std::min<int64_t>(cap, (int64_t)1 << 15));
    if (const char* e = std::getenv("TNCO_HIP_JLOG_CAP")) cap = std::max<int64_t>(1, std::atoll(e));  // test knob
    cap = (cap + 15) & ~(int64_t)15;  // the sweep kernel appends in whole 16-entry (64-byte) pieces
    P.jcap = (int32_t)cap;
  }

  cmark("checks, stream");
  HIP_TRY(h->alloc(&P.blocks, R * h->block_bytes()));
  HIP_TRY(h->alloc(&P.lpar, R * n * LPS + 16));  // the sweep kernel reads 8 bytes at a leaf's record
  HIP_TRY(h->alloc(&P.mt, R * 624));
  HIP_TRY(h->alloc(&P.mtshadow, R * MT_SHADOW));
  HIP_TRY(h->alloc(&P.rs, R));
  HIP_TRY(h->alloc(&P.minlinks, R * N));
  HIP_TRY(h->alloc(&P.jlog, R * (int64_t)P.jcap));

  cmark("allocations");
  // shared tables
  {
    std::vector<uint64_t> lm((size_t)n * L, 0), om((size_t)L, 0), sp((size_t)L, 0);
    for (int t = 0; t < n; ++t)
      for (int w = 0; w < W; ++w) lm[(size_t)t * L + w] = h->leafmask_w[(size_t)t * W + w];
    for (int w = 0; w < W; ++w) om[w] = h->outmask_w[w];
    uint64_t *dl, *dom;
    HIP_TRY(h->alloc(&dl, (int64_t)n * L));
    HIP_TRY(h->alloc(&dom, L));
    HIP_TRY(hipMemcpy(dl, lm.data(), lm.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dom, om.data(), om.size() * 8, hipMemcpyHostToDevice));
    P.leafmask = dl;
    P.outmask = dom;
    if (d->sparse_mask) {
      for (int w = 0; w < W; ++w) sp[w] = d->sparse_mask[w];
      uint64_t* dsp;
      HIP_TRY(h->alloc(&dsp, L));
      HIP_TRY(hipMemcpy(dsp, sp.data(), sp.size() * 8, hipMemcpyHostToDevice));
      P.sparse = dsp;
    }
    if (P.cost_mode == 1) {
      // std::pow(size_t, size_t) -> double pow, converted to cost_type (simple.hpp:45)
      std::vector<double> tab((size_t)W * 64 + 1);
      for (size_t k = 0; k < tab.size(); ++k) tab[k] = rc(std::pow((double)dim_u, (double)k));
      double* dt;
      HIP_TRY(h->alloc(&dt, (int64_t)tab.size()));
      HIP_TRY(hipMemcpy(dt, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
      P.ctab = dt;
    } else if (P.cost_mode == 3 || P.cost_mode == 2) {
      // dims[p] = 2^a * odd: exponent classes (a = 1 .. max) for both modes; mode 2 also the odd parts
      // (as cost_type) and the mask of the positions that have one (simple.hpp:51-53; sa_kernels.h seq_product)
      int max_a = 0;
      std::vector<int> av((size_t)I, 0);
      std::vector<uint64_t> odd((size_t)I, 1);
      for (int i = 0; i < I; ++i) {
        uint64_t x = d->dims[i];
        while ((x & 1ull) == 0) { x >>= 1; ++av[i]; }
        odd[i] = x;
        max_a = std::max(max_a, av[i]);
      }
      if (max_a > TABS_MAXCLS)
        return fail(TNCO_HIP_ENOTIMPL, "an index dimension with a power-of-two part above 2^32 is not supported on the GPU path.");
      std::vector<uint64_t> cls((size_t)std::max(max_a, 1) * L, 0);
      for (int i = 0; i < I; ++i)
        if (av[i] > 0) cls[(size_t)(av[i] - 1) * L + (i >> 6)] |= 1ull << (i & 63);
      uint64_t* dc;
      HIP_TRY(h->alloc(&dc, (int64_t)cls.size()));
      HIP_TRY(hipMemcpy(dc, cls.data(), cls.size() * 8, hipMemcpyHostToDevice));
      P.dimclass = dc;
      P.n_dimclass = max_a;
      if (P.cost_mode == 2) {
        std::vector<double> dd((size_t)L * 64, 1.0);
        std::vector<uint64_t> om((size_t)L, 0);
        for (int i = 0; i < I; ++i) {
          dd[i] = rc((double)odd[i]);
          if (odd[i] != 1) om[i >> 6] |= 1ull << (i & 63);
        }
        double* dt;
        uint64_t* dom2;
        HIP_TRY(h->alloc(&dt, (int64_t)dd.size()));
        HIP_TRY(hipMemcpy(dt, dd.data(), dd.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(h->alloc(&dom2, (int64_t)om.size()));
        HIP_TRY(hipMemcpy(dom2, om.data(), om.size() * 8, hipMemcpyHostToDevice));
        P.dimsd = dt;
        P.oddmask = dom2;
        // a single odd part m (and exponents that fit the packed sum): chain of t factors as a table
        uint64_t m1 = 0;
        bool single = true;
        for (int i = 0; i < I; ++i)
          if (odd[i] != 1) {
            if (m1 == 0) m1 = odd[i];
            single &= odd[i] == m1;
          }
        if (single && m1 != 0 && (int64_t)max_a * W * 64 < (1 << 18) && !std::getenv("TNCO_HIP_NO_ODD_TABLE")) {
          std::vector<double> tab((size_t)W * 64 + 1);
          double c = 1.0;
          const double md = rc((double)m1);
          for (size_t t = 0; t < tab.size(); ++t) {
            tab[t] = c;
            c = rc(c * md);
          }
          double* dtab;
          HIP_TRY(h->alloc(&dtab, (int64_t)tab.size()));
          HIP_TRY(hipMemcpy(dtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
          P.ctab = dtab;
          P.odd_single = 1;
        }
      }
    }
  }

  cmark("shared tables");
  TempBufs tmp;
  // seeds -> MT state
  if (d->prng_states) {  // the string-seed form of the constructor (optimize/optimizer.hpp:68-71): whole states
    HIP_TRY(hipMemcpy2D(P.mt, (size_t)624 * 4, d->prng_states, (size_t)625 * 4, (size_t)624 * 4, (size_t)R, hipMemcpyHostToDevice));
    std::vector<uint32_t> pos((size_t)R);
    for (int64_t r = 0; r < R; ++r) pos[(size_t)r] = d->prng_states[r * 625 + 624];
    uint32_t* dpos = nullptr;
    HIP_TRY(tmp.alloc(&dpos, R));
    HIP_TRY(hipMemcpy(dpos, pos.data(), (size_t)R * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mt_pos_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, h->stream, P.rs, dpos, (const int64_t*)nullptr, R);
    HIP_TRY(h->sync_all());
  } else {
    uint32_t* dseeds = nullptr;
    HIP_TRY(tmp.alloc(&dseeds, R));
    HIP_TRY(hipMemcpy(dseeds, d->seeds, (size_t)R * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mt_seed_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, h->stream, P.mt, P.rs, dseeds, R);
    HIP_TRY(h->sync_all());
  }

  cmark("generator states");
  // links (+ optional explicit legs) -> node blocks and caches
  {
    const int64_t ntrees = d->links_stride == 0 ? 1 : R;
    int32_t* dlinks = nullptr;
    uint64_t* dmasks = nullptr;
    double *dtotal = nullptr, *dsum = nullptr;
    int32_t* dstatus = nullptr;
    std::vector<double> total((size_t)R), sum((size_t)R);
    std::vector<int32_t> status((size_t)R);
    if (links_on_device && (d->links_stride == 0 || d->links_stride == 3 * (int64_t)N)) {
      dlinks = const_cast<int32_t*>(d->links);  // (read only)
    } else {
      HIP_TRY(tmp.alloc(&dlinks, ntrees * 3 * N));
      const hipMemcpyKind kind = links_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
      if (d->links_stride == 0 || d->links_stride == 3 * (int64_t)N)
        HIP_TRY(hipMemcpy(dlinks, d->links, (size_t)ntrees * 3 * N * 4, kind));
      else
        HIP_TRY(hipMemcpy2D(dlinks, (size_t)3 * N * 4, d->links, (size_t)d->links_stride * 4, (size_t)3 * N * 4, (size_t)ntrees, kind));
    }
    if (links_on_device) {
      int32_t* dcheck = nullptr;
      HIP_TRY(tmp.alloc(&dcheck, ntrees));
      hipLaunchKernelGGL(tree_check_kernel, dim3((unsigned)ntrees), dim3(64), 0, h->stream, (int32_t)N, dlinks, 3 * (int64_t)N, ntrees, dcheck);
      HIP_TRY(hipGetLastError());
      std::vector<int32_t> chk((size_t)ntrees);
      HIP_TRY(hipMemcpyAsync(chk.data(), dcheck, (size_t)ntrees * 4, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h->sync_all());
      for (int64_t r = 0; r < ntrees; ++r)
        if (chk[r]) return fail(TNCO_HIP_EINVAL, tree_check_message(chk[r]));
    }
  cmark("trees checked");
    const int64_t nmasks = d->node_masks ? (d->node_masks_stride == 0 ? 1 : R) : 0;
    if (nmasks) {
      HIP_TRY(tmp.alloc(&dmasks, nmasks * N * W));
      if (d->node_masks_stride == 0 || d->node_masks_stride == (int64_t)N * W)
        HIP_TRY(hipMemcpy(dmasks, d->node_masks, (size_t)nmasks * N * W * 8, hipMemcpyHostToDevice));
      else
        HIP_TRY(hipMemcpy2D(dmasks, (size_t)N * W * 8, d->node_masks, (size_t)d->node_masks_stride * 8, (size_t)N * W * 8, (size_t)nmasks, hipMemcpyHostToDevice));
    }
    HIP_TRY(tmp.alloc(&dtotal, R));
    HIP_TRY(tmp.alloc(&dsum, R));
    HIP_TRY(tmp.alloc(&dstatus, R));
    BuildArgs a{};
    a.in_links = dlinks;
    a.in_links_stride = d->links_stride == 0 ? 0 : 3 * (int64_t)N;
    a.in_masks = dmasks;
    a.in_masks_stride = (d->node_masks && d->node_masks_stride != 0) ? (int64_t)N * W : 0;
    a.out_blocks = P.blocks; a.out_lpar = P.lpar;
    a.scratch = reinterpret_cast<int32_t*>(P.minlinks);  // 16 B * N per replica = 4N int32
    a.out_total = dtotal; a.out_sum = dsum; a.out_status = dstatus;
    a.r0 = 0; a.count = R;
    if (h->hyper && !d->node_masks) {  // (the second mask per node build_kernel derives the legs with)
      uint64_t* dhy = nullptr;
      HIP_TRY(tmp.alloc(&dhy, R * (int64_t)(n - 1) * W));
      a.hyper_tmp = dhy;
    }
    launch_build(h, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(h->sync_all());
  cmark("trees -> node records");
    HIP_TRY(hipMemcpy(total.data(), dtotal, (size_t)R * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sum.data(), dsum, (size_t)R * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(status.data(), dstatus, (size_t)R * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(finish_init_kernel, dim3((unsigned)R), dim3(64), 0, h->stream, P, dsum, dtotal);
    HIP_TRY(h->sync_all());
    for (int64_t r = 0; r < R; ++r)
      if (status[r]) return fail(TNCO_HIP_EINVAL, status_message(status[r]));
    if (!fw)
      for (int64_t r = 0; r < R; ++r)
        if (bad_log2(total[r]) || bad_log2(sum[r])) return fail(TNCO_HIP_EINVAL, "Precision is too low.");
  }

  cmark("costs back, best trees");
  // finite width: WidthCache, initial slices (greedy, draws from the PRNG), CostCache(slices)
  // -- finite_width/greedy/optimizer.hpp:72-115
  if (fw) {
    h->fw = true;
    FwParams& F = h->F;
    F.width_f32 = d->width_dtype == TNCO_HIP_F32 ? 1 : 0;
    F.max_width = F.width_f32 ? (double)(float)d->max_width : d->max_width;
    F.log2d = uniform ? std::log2((double)dim_u) : 0.0;
    F.log2np = d->sparse_mask ? std::log2((double)d->n_projs) : 0.0;
    F.max_new_slices = (int64_t)std::min<uint64_t>(d->max_number_new_slices, (uint64_t)1 << 40);
    F.I64 = 64 * L;
    F.leaf_wide = 0;
    if (!uniform) {
      std::vector<double> l2((size_t)L * 64, 0.0);
      for (int i = 0; i < I; ++i) l2[i] = std::log2((double)d->dims[i]);
      double* dl2;
      HIP_TRY(h->alloc(&dl2, (int64_t)l2.size()));
      HIP_TRY(hipMemcpy(dl2, l2.data(), l2.size() * 8, hipMemcpyHostToDevice));
      F.log2dims = dl2;
    }
    if (!F.width_f32) HIP_TRY(h->alloc(&F.width64, R * (int64_t)N));
    HIP_TRY(h->alloc(&F.slices, R * 2 * (int64_t)L));
    HIP_TRY(h->alloc(&F.scratch_i, R * fw_scratch_ints(N, F.I64)));
    HIP_TRY(h->alloc(&F.scratch_d, R * 2 * (int64_t)N));
    HIP_TRY(h->alloc(&F.status, R));
    HIP_TRY(h->alloc(&F.nwide, R));
    HIP_TRY(h->alloc(&F.nwfront, R));
    HIP_TRY(hipMemset(F.status, 0, (size_t)R * 4));
    // The re-slice of a replica in one wavefront, its cost cache RE-PRICED (fw_wave_kernel): costs must be powers of
    // two (uniform dims 2^k, float64, no sparse legs), a leg of a subtree decidable from the holders it contains (no
    // hyper-indices), the headers one array per replica (split layout), a lane's share of the nodes in registers.
    // TNCO_HIP_FW_WAVE=0: never (tests compare the two forms), =1: always where possible (else: by the fall-backs).
    const char* wave_env = std::getenv("TNCO_HIP_FW_WAVE");
    const int lkw = F.I64 / 64, lanes_per_mask = lkw <= 16 ? 16 : (lkw <= 32 ? 32 : 64);
    if (P.cost_mode == 0 && !P.f32 && d->sparse_mask == nullptr && P.BS == 32 && n >= 16 && n - 1 <= 64 * FWT_JMAX &&
        F.I64 <= 4096 && fww_lds_bytes(n, lanes_per_mask, h->hyper) <= 64 * 1024 && !(wave_env && std::atoi(wave_env) == 0)) {
      std::vector<int32_t> hold((size_t)F.I64 * 2, -1);
      std::vector<int32_t> cnt((size_t)I, 0);
      for (int t = 0; t < n; ++t)
        for (int i = 0; i < I; ++i)
          if ((d->leaf_masks[(size_t)t * W + (i >> 6)] >> (i & 63)) & 1ull) {
            if (cnt[i] < 2) hold[(size_t)2 * i + cnt[i]] = t;
            cnt[i] += 1;
          }
      for (int i = 0; i < I; ++i) {
        const bool is_out = d->output_mask && ((d->output_mask[i >> 6] >> (i & 63)) & 1ull);
        if (cnt[i] == 0 || cnt[i] > 2 || (cnt[i] == 2 && is_out)) hold[(size_t)2 * i] = hold[(size_t)2 * i + 1] = -1;
      }
      int32_t* dh;
      HIP_TRY(h->alloc(&dh, (int64_t)hold.size()));
      HIP_TRY(hipMemcpy(dh, hold.data(), hold.size() * 4, hipMemcpyHostToDevice));
      F.holder2 = dh;
      if (h->hyper) {  // ... every holder of an index (up to FWH_MAXH), and whether it is open
        std::vector<uint16_t> hn((size_t)F.I64 * 8, 0);
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int t = 0; t < n; ++t)
          for (int i = 0; i < I; ++i)
            if ((d->leaf_masks[(size_t)t * W + (i >> 6)] >> (i & 63)) & 1ull) {
              if (cnt[i] < FWH_MAXH) hn[(size_t)8 * i + 1 + cnt[i]] = (uint16_t)t;
              cnt[i] += 1;
            }
        for (int i = 0; i < I; ++i) {
          const bool is_out = d->output_mask && ((d->output_mask[i >> 6] >> (i & 63)) & 1ull);
          if (cnt[i] >= 1 && cnt[i] <= FWH_MAXH) hn[(size_t)8 * i] = (uint16_t)(cnt[i] | ((is_out || cnt[i] == 1) ? 0x8000 : 0));
        }
        uint16_t* dn;
        HIP_TRY(h->alloc(&dn, (int64_t)hn.size()));
        HIP_TRY(hipMemcpy(dn, hn.data(), hn.size() * 2, hipMemcpyHostToDevice));
        F.holdern = dn;
      }
      HIP_TRY(h->alloc(&F.slowstat, 4));
      HIP_TRY(hipMemset(F.slowstat, 0, 32));
      h->fw_wave_capable = true;
#ifdef TNCO_PROFILE  // (the stage counters live in the single re-slice kernel)
      h->fw_wave_capable = false;
#endif
    }
    HIP_TRY(h->alloc(&F.fastflag, R));
    HIP_TRY(hipMemset(F.fastflag, 0, (size_t)R * 4));
    HIP_TRY(h->alloc(&F.delta_scr, R * 64));
    // (word 0 of a replica = the change count of its last re-pricing, tnco_hip_diag_reslice_info: "none yet", whatever
    //  a recycled block held)
    HIP_TRY(hipMemset(F.delta_scr, 0xFF, (size_t)R * 64 * sizeof(*F.delta_scr)));
    F.stack_cap = FW_LDSPOS;
    if (const char* e = std::getenv("TNCO_HIP_FW_STACK")) F.stack_cap = std::max(0, std::min(FW_LDSPOS, std::atoi(e)));
    auto upload_mask = [&](const uint64_t* src, const uint64_t** dst) -> int {
      std::vector<uint64_t> m((size_t)L, 0);
      for (int w = 0; w < W; ++w) m[w] = src[w];
      uint64_t* dm;
      HIP_TRY(h->alloc(&dm, L));
      HIP_TRY(hipMemcpy(dm, m.data(), m.size() * 8, hipMemcpyHostToDevice));
      *dst = dm;
      return TNCO_HIP_OK;
    };
    if (d->skip_slices)
      if (int rc = upload_mask(d->skip_slices, &F.skip)) return rc;
    {  // the leaf tensors wider than max_width (usually none: the walk of the re-slice skips the leaves then)
      uint32_t* bits;
      int32_t* any;
      const int nb = (n + 31) / 32;
      HIP_TRY(h->alloc(&bits, nb));
      HIP_TRY(tmp.alloc(&any, 1));
      HIP_TRY(hipMemsetAsync(bits, 0, (size_t)nb * 4, h->stream));
      HIP_TRY(hipMemsetAsync(any, 0, 4, h->stream));
      F.leaf_bits = bits;
#define CALL_FWL(LL, KK) launch_fw_leaf_bits_lk<LL, KK>(h, bits, any)
      DISPATCH_LK(h, CALL_FWL)
#undef CALL_FWL
      HIP_TRY(hipGetLastError());
      int32_t a1 = 0;
      HIP_TRY(hipMemcpyAsync(&a1, any, 4, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h->sync_all());
      F.leaf_wide = a1 ? 1 : 0;
      if (F.leaf_wide) h->fw_wave_capable = false;  // (a too-wide leaf has no header to carry its width)
      if (h->fw_wave_capable) {
        h->fw_wave_on = true;
        h->fw_wave_lanes = lanes_per_mask;
        // the roomier configuration (up to 1 023 too-wide tensors, 512 candidate legs) where its LDS fits; taken when the
        // lean one leaves replicas behind for that reason (tnco_hip_run_fw), or from the start with TNCO_HIP_FW_BIG=1
        h->fw_wave_big_ok = fww_lds_bytes(n, lanes_per_mask, h->hyper, true) <= 64 * 1024;
        h->fw_wave_big = h->fw_wave_big_ok && std::getenv("TNCO_HIP_FW_BIG") && std::atoi(std::getenv("TNCO_HIP_FW_BIG")) != 0;
        h->set_wave_config();
      }
      F.fast_ok = h->fw_wave_on ? 1 : 0;
    }
    // rows of W words (one shared, or one per replica) -> rows of L words on the device, zero-padded
    auto upload_rows = [&](const uint64_t* src, uint64_t** dst) -> int {
      const int64_t rows = d->slices_stride == 0 ? 1 : R;
      uint64_t* dm;
      HIP_TRY(tmp.alloc(&dm, rows * L));
      HIP_TRY(hipMemset(dm, 0, (size_t)rows * L * 8));
      HIP_TRY(hipMemcpy2D(dm, (size_t)L * 8, src, (size_t)std::max<int64_t>(d->slices_stride, W) * 8, (size_t)W * 8, (size_t)rows, hipMemcpyHostToDevice));
      *dst = dm;
      return TNCO_HIP_OK;
    };
    if (d->slices_stride != 0 && d->slices_stride < W) return fail(TNCO_HIP_EINVAL, "'slices_stride' is not valid.");
    FwInitArgs a{};
    if (d->slices) {
      uint64_t* ds = nullptr;
      if (int rc = upload_rows(d->slices, &ds)) return rc;
      a.slices_in = ds;
      a.slices_in_stride = d->slices_stride == 0 ? 0 : L;
    }
    double *dtotal = nullptr, *dsum = nullptr;
    HIP_TRY(tmp.alloc(&dtotal, R));
    HIP_TRY(tmp.alloc(&dsum, R));
    a.out_total = dtotal; a.out_sum = dsum;
    launch_fw_init(h, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(h->sync_all());
    std::vector<double> total((size_t)R), sum((size_t)R);
    std::vector<int32_t> st((size_t)R);
    HIP_TRY(hipMemcpy(total.data(), dtotal, (size_t)R * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sum.data(), dsum, (size_t)R * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(st.data(), F.status, (size_t)R * 4, hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < R; ++r) {
      if (st[r]) return fail(TNCO_HIP_ENOTIMPL, "finite width: candidate legs beyond the re-slice scratch (internal error).");
      if (bad_log2(total[r]) || bad_log2(sum[r])) return fail(TNCO_HIP_EINVAL, "Precision is too low.");
    }
  }

  // The `min_ctree` / `min_slices` constructor arguments (optimize/optimizer.hpp:57-65,
  // infinite_memory/optimizer.hpp:61-88, finite_width/greedy/optimizer.hpp:72-115): the best tree so far
  // and its slices, min_total_cost = get_cost(min_ctree[, min_slices]), the checks of is_valid on them.
  if (d->min_links || (fw && d->min_slices)) {
    const int64_t LKw = L;
    if (d->min_links) {
      const int64_t ntrees = d->min_links_stride == 0 ? 1 : R;
      if (d->min_links_stride != 0 && d->min_links_stride < 3 * (int64_t)N) return fail(TNCO_HIP_EINVAL, "'min_links_stride' is not valid.");
      {
        std::vector<int32_t> cp((size_t)N), cc((size_t)N);
        for (int64_t r = 0; r < ntrees; ++r) {
          const int32_t* lk = d->min_links + r * d->min_links_stride;
          if (const char* e = tree_check(N, lk, lk + N, lk + 2 * (int64_t)N, cp, cc)) return fail(TNCO_HIP_EINVAL, e);
        }
      }
      int32_t* dmin = nullptr;
      HIP_TRY(tmp.alloc(&dmin, ntrees * 3 * N));
      if (d->min_links_stride == 0 || d->min_links_stride == 3 * (int64_t)N)
        HIP_TRY(hipMemcpy(dmin, d->min_links, (size_t)ntrees * 3 * N * 4, hipMemcpyHostToDevice));
      else
        HIP_TRY(hipMemcpy2D(dmin, (size_t)3 * N * 4, d->min_links, (size_t)d->min_links_stride * 4, (size_t)3 * N * 4, (size_t)ntrees, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(set_minlinks_kernel, dim3((unsigned)R), dim3(64), 0, h->stream, P, dmin, d->min_links_stride == 0 ? (int64_t)0 : 3 * (int64_t)N);
      HIP_TRY(hipGetLastError());
      HIP_TRY(h->sync_all());
      h->small_tree = false;  // (the LDS-resident kernel assumes the log starts at the checkpoint)
    }
    if (fw && d->min_slices) {  // row 1 of every replica's [2][L] slices record
      std::vector<uint64_t> rows((size_t)R * LKw, 0);
      for (int64_t r = 0; r < R; ++r)
        for (int w = 0; w < W; ++w) rows[(size_t)r * LKw + w] = d->min_slices[r * d->slices_stride + w];
      HIP_TRY(hipMemcpy2D(h->F.slices + LKw, (size_t)2 * LKw * 8, rows.data(), (size_t)LKw * 8, (size_t)LKw * 8, (size_t)R, hipMemcpyHostToDevice));
    }
    const int64_t per = h->block_bytes() + (int64_t)n * 4 + (int64_t)N * 16 + 64;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(R, ((int64_t)1 << 30) / per));
    uint8_t* tblk = nullptr; int32_t *tlpar = nullptr, *tscr = nullptr, *tstat = nullptr, *tbad = nullptr;
    double *ttot = nullptr, *tsum = nullptr;
    HIP_TRY(tmp.alloc(&tblk, chunk * h->block_bytes()));
    HIP_TRY(tmp.alloc(&tlpar, chunk * n * LPS));
    HIP_TRY(tmp.alloc(&tscr, chunk * 4 * N));
    HIP_TRY(tmp.alloc(&tstat, chunk));
    HIP_TRY(tmp.alloc(&tbad, chunk));
    HIP_TRY(tmp.alloc(&ttot, chunk));
    HIP_TRY(tmp.alloc(&tsum, chunk));
    uint64_t* thy = nullptr;
    if (h->hyper) HIP_TRY(tmp.alloc(&thy, chunk * (int64_t)(n - 1) * W));
    std::vector<int32_t> hs((size_t)chunk), hb((size_t)chunk);
    std::vector<double> hsum((size_t)chunk);
    for (int64_t r0 = 0; r0 < R; r0 += chunk) {
      const int64_t cnt = std::min(chunk, R - r0);
      BuildArgs a{};
      a.hyper_tmp = thy;
      a.out_blocks = tblk; a.out_lpar = tlpar; a.scratch = tscr;
      a.out_total = ttot; a.out_sum = tsum; a.out_status = tstat; a.r0 = r0; a.count = cnt;
      a.src_live = 0;
      a.src_links = P.minlinks + r0 * (int64_t)N;
      if (fw) { a.cost_slices = h->F.slices + LKw; a.cost_slices_stride = 2 * LKw; }
      launch_build(h, a);
      HIP_TRY(hipMemsetAsync(tbad, 0, (size_t)cnt * 4, h->stream));
      if (fw) launch_fw_check(h, a, 1, 1e-5, tbad);
      hipLaunchKernelGGL(restore_min_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, P.rs, r0, cnt, tsum);
      HIP_TRY(hipGetLastError());
      HIP_TRY(h->sync_all());
      HIP_TRY(hipMemcpy(hs.data(), tstat, (size_t)cnt * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(hb.data(), tbad, (size_t)cnt * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(hsum.data(), tsum, (size_t)cnt * 8, hipMemcpyDeviceToHost));
      for (int64_t q = 0; q < cnt; ++q) {
        if (hs[q]) return fail(TNCO_HIP_EINVAL, status_message(hs[q]));
        if (hb[q]) return fail(TNCO_HIP_EINVAL, "Width of the sliced minimum contraction is larger than 'max_width'.");
        if (bad_log2(hsum[q])) return fail(TNCO_HIP_EINVAL, "Precision is too low.");
      }
    }
  }

  // Infinite memory: more blocks than resident ones (65536 replicas at 512 leaves: 1024 blocks, 768 resident)?  Then
  // every step is split over two streams (host_ctx.h, tnco_hip_run), unless the last round is nearly full anyway.
  // TNCO_HIP_GROUPS=1..4 overrides.
  {
    const int64_t nblocks = (R + (SWT / h->L) - 1) / (SWT / h->L);
    int G = 1;
    if (!fw) {
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, d->device));
      // The LDS-resident kernel is bound by the instruction stream of one wavefront per SIMD (16 replicas; 1.35 us per
      // move whatever the load), the HBM kernel by memory latency per replica (3.7 us) until ~50 000 replicas saturate it
      // at 8-9e9 move-evals/s.  Up to 64 leaves a CU holds 64 LDS-resident replicas and the LDS kernel wins at every
      // replica count (1.07e10 from 16 384 replicas on); beyond, a CU holds 32 and it wins (x2.3 ... x1.15) while two
      // rounds of its blocks hold the replicas (profiles/r05_small_tree_ab.txt).
#ifndef TNCO_SMALL_TREE_ALWAYS  // (the other A/B library of tools/small_tree_ab.py: `make allsmall`)
      if (h->small_tree && n - 1 > 63 && R > 2 * (int64_t)small_replicas_per_cu(n - 1) * prop.multiProcessorCount) h->small_tree = false;
#endif
      if (h->small_tree) {
        // A batch that leaves wavefront slots empty gets fewer replicas per wavefront (the spare lane groups shadow them,
        // sa_small.h): an iteration then runs only the sections its few replicas are in -- 8.5-8.8e5 move-evals/s per
        // replica at one per wavefront (up to 1024 replicas of <= 64 leaves), 7.5e5 at four, 6.7e5 at sixteen.
        const int64_t slots = (int64_t)small_replicas_per_cu(n - 1) / (SMALL_TPB / 4) * prop.multiProcessorCount;
        while (h->small_seats > 1 && (R + h->small_seats / 2 - 1) / (h->small_seats / 2) <= slots) h->small_seats /= 2;
      }
      // Any other tree of the fast cost path whose replicas fit the CUs' LDS at once (the latency regime of the larger
      // networks: 512 leaves of 12 words are 58 KiB, two per CU, up to 512 replicas): sa_lds_kernel.
      if (!h->small_tree && !h->generic && LPS == 1 && !d->min_links && n >= 2 && h->log2l == 2 && I <= 65535 &&
          N <= 65534 && (int64_t)P.log2d * 64 * W <= 65535) {
        int deg = 0;  // index positions per leaf at most
        for (int t = 0; t < n; ++t) {
          int c = 0;
          for (int w = 0; w < W; ++w) c += __builtin_popcountll(h->leafmask_w[(size_t)t * W + w]);
          deg = std::max(deg, c);
        }
        const bool table = (int64_t)n * 32 * h->K <= 16384;  // the leaf legs themselves, [n][4 K] words
        if (table || deg <= 32) {
          const int stride = table ? 0 : std::max(1, (deg + 3) / 4);
          std::vector<uint64_t> idx(table ? (size_t)n * 4 * h->K : (size_t)n * stride, table ? 0ull : ~0ull);
          for (int t = 0; t < n; ++t) {
            if (table) {
              for (int w = 0; w < W; ++w) idx[(size_t)t * 4 * h->K + w] = h->leafmask_w[(size_t)t * W + w];
              continue;
            }
            int c = 0;
            for (int p = 0; p < I; ++p)
              if ((h->leafmask_w[(size_t)t * W + (p >> 6)] >> (p & 63)) & 1) {
                uint64_t& q = idx[(size_t)t * stride + (c >> 2)];
                q = (q & ~(0xFFFFull << (16 * (c & 3)))) | ((uint64_t)p << (16 * (c & 3)));
                ++c;
              }
          }
          const int K = h->K, ni = n - 1;
          LdsPlan pl{};
          pl.leaf_stride = stride;
          pl.leaf_words = (int)idx.size();
          pl.seat0 = (pl.leaf_words * 8 + 15) / 16 * 16;
          pl.o_part = 8 * ni;
          pl.o_legs = 16 * ni;
          pl.o_lpar = pl.o_legs + 32 * K * ni;
          pl.o_ring = pl.o_lpar + (2 * n + 7) / 8 * 8;
          pl.o_jb = pl.o_ring + 64 * 4;
          pl.seat_stride = (pl.o_jb + 16 * 4 + 15) / 16 * 16;
          // Four blocks (wavefronts of up to 16 replicas) per CU -- one per SIMD -- if they hold the batch, else two, else
          // one, and as few replicas per wavefront as spread the batch over all of them, the other lane groups shadowing
          // (sa_small.h): per-replica move-evals/s at 128 leaves, four blocks / one block per CU: 1024 replicas 7.8 / 7.1e5,
          // 2048: 7.4 / 6.6e5; 256 leaves, 1024: 7.6 / 7.0e5.  (Before the shadows it was the other way round: few active
          // lanes per CU.)  x1.6 ... x2.8 the HBM kernel per replica, so a second round of blocks would lose.
          int best = 0;
          for (int b = 4; b >= 1 && best == 0; b >>= 1) {
            const int seats = std::min(SMALL_TPB / 4, (device_lds_bytes(prop) / b - pl.seat0) / pl.seat_stride);
            if (seats > 0 && (R <= (int64_t)b * seats * prop.multiProcessorCount || b == 1)) { best = b * seats; pl.seats = seats; pl.blocks_per_cu = b; }
          }
#ifdef TNCO_NO_SMALL_TREE
          best = 0;
#endif
#ifdef TNCO_SMALL_TREE_ALWAYS
          const bool fits = best > 0;
#else
          const bool fits = best > 0 && R <= (int64_t)best * prop.multiProcessorCount;
#endif
          if (fits) {
            const int64_t waves = (int64_t)pl.blocks_per_cu * prop.multiProcessorCount;
            pl.seats = (int)std::max<int64_t>(1, std::min<int64_t>(pl.seats, (R + waves - 1) / waves));
            pl.total = pl.seat0 + pl.seats * pl.seat_stride;
            HIP_TRY(h->alloc(&h->leaf_idx, (int64_t)idx.size()));
            HIP_TRY(hipMemcpy(h->leaf_idx, idx.data(), idx.size() * 8, hipMemcpyHostToDevice));
            h->lds_plan = pl;
            h->lds_tree = lds_kernel_prepare(h, device_lds_bytes(prop)) == 0;  // (else: the HBM kernel below)
          }
        }
      }
      // A batch that leaves wavefront slots of the HBM kernel empty is spread (sa_sweep.h, SPREAD): fewer replicas per
      // wavefront, the other lane groups shadowing them.  Per-replica move-evals/s at 512 leaves, 16 (full) / 1 / 2 / 4 / 8
      // replicas per wavefront: 1024 replicas 3.2 / 5.3 / 4.4 / 3.8 / 3.5e5; 4096: 3.2 / - / 3.7 / 3.8 / 3.5; 8192: 3.2 / - /
      // 2.1 / 3.2 / 3.5; 16384: 3.2 / - / - / 1.8 / 2.9 -- one replica per wavefront while that takes at most two thirds of
      // the slots, else as few as leave ONE wavefront per SIMD (two half-filled ones lose to a full one).
      if (!h->small_tree && !h->lds_tree) {
        const int full = 64 / h->L;
        h->run_seats = 1;  // (the occupancy of the spread form)
        const int64_t wslots = (int64_t)run_blocks_per_cu(h) * prop.multiProcessorCount * (SWT / 64);
        int seats = 1;
        if (3 * R > 2 * wslots)
          while (seats < full && (R + seats - 1) / seats > 4 * (int64_t)prop.multiProcessorCount) seats *= 2;
        h->run_seats = (wslots > 0 && seats < full) ? seats : 0;
#ifdef TNCO_NO_SMALL_TREE  // (the A/B library: the batch as it was, 64 / L replicas per wavefront)
        h->run_seats = 0;
#endif
      }
      h->run_slots = run_blocks_per_cu(h) * prop.multiProcessorCount;
      if (h->run_slots > 0 && nblocks > h->run_slots) {
        const double rounds = (double)nblocks / (double)h->run_slots, part = rounds - std::floor(rounds);
        // (also with whole rounds: the blocks of a round do not end together -- the general cost path at two
        //  wavefronts per SIMD, 1024 blocks on 512 slots: +2 ... +8 % on two streams, profiles/r03_other_configs.md)
        if (part < 0.85) G = 2;
      }
    } else if (h->F.max_new_slices == 0) {
      // finite width, a batch that leaves wavefront slots of the staged moves empty: their SPREAD form, by the rule of the
      // infinite-memory kernel above -- the moves are 90 % of such a step (full wavefronts: 5.2 microseconds per move and
      // replica); otherwise two halves on two streams whatever the rounds -- what overlaps then are KERNELS of different
      // bounds (the moves wait on memory requests, get_slices and the tree kernel on LDS / instruction latency)
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, d->device));
      const int full = 64 / h->L;
      h->run_seats = 1;  // (the occupancy of the spread form)
      const int64_t wslots = (int64_t)run_blocks_per_cu(h) * prop.multiProcessorCount * (SWT / 64);
      int seats = 1;
      if (3 * R > 2 * wslots)
        while (seats < full && (R + seats - 1) / seats > 4 * (int64_t)prop.multiProcessorCount) seats *= 2;
      h->run_seats = (wslots > 0 && seats < full) ? seats : 0;
#ifdef TNCO_NO_SMALL_TREE
      h->run_seats = 0;
#endif
      if (h->run_seats == 0 && nblocks >= 64) G = 2;
    }
    if (const char* e = std::getenv("TNCO_HIP_GROUPS")) G = std::max(1, std::min((int)tnco_hip_ctx::MAX_GROUPS, std::atoi(e)));
    if (h->small_tree || h->lds_tree || h->run_seats > 0 || nblocks < 2 * G) G = 1;
    if (G > 1) {
      for (int q = 0; q < G; ++q) {
        HIP_TRY(tnco::StreamCache::get().take(&h->gstream[q], h->device));
        HIP_TRY(hipEventCreateWithFlags(&h->gjoin[q], hipEventDisableTiming));
      }
      HIP_TRY(hipEventCreateWithFlags(&h->gfork, hipEventDisableTiming));
      HIP_TRY(hipEventCreate(&h->region_a));
      HIP_TRY(hipEventCreate(&h->region_b));
      h->n_groups = G;
    }
  }

  cmark("the rest (finite width, min trees)");
  guard.h = nullptr;
  *out = h;
  return TNCO_HIP_OK;
}

int tnco_hip_set_stream(tnco_hip_handle h, void* s) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  h->close_region();
  h->stream = s ? (hipStream_t)s : h->own_stream;  // (a grouped handle forks from / joins into this stream)
  return TNCO_HIP_OK;
}

int tnco_hip_sync(tnco_hip_handle h) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  return fw_runtime_status(h);
}

int tnco_hip_run(tnco_hip_handle h, int prob_kind, const double* betas, int64_t n_steps) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  if (prob_kind < 0 || prob_kind > 2) return fail(TNCO_HIP_EINVAL, "'prob_kind' is not valid.");
  if (n_steps < 0 || (n_steps > 0 && !betas)) return fail(TNCO_HIP_EINVAL, "'betas' is not valid.");
  if (n_steps == 0) return TNCO_HIP_OK;
  if (h->fw) return fail(TNCO_HIP_EINVAL, "handle was created with 'max_width': use tnco_hip_run_fw.");
  HIP_TRY(hipSetDevice(h->device));
  // betas: pinned host ring -> device ring, no wait for the previous call (a region of the rings is
  // re-used only after a wrap, and a wrap waits for everything enqueued)
  if (n_steps > h->ring_cap / 4) {
    HIP_TRY(h->sync_all());
    if (h->beta_pin) (void)hipHostFree(h->beta_pin);
    if (h->beta_ring) (void)hipFree(h->beta_ring);
    h->beta_pin = h->beta_ring = nullptr;
    h->ring_cap = h->ring_pos = 0;
    const int64_t cap = std::max<int64_t>((int64_t)1 << 16, 4 * n_steps);
    HIP_TRY(hipHostMalloc((void**)&h->beta_pin, (size_t)cap * 8, hipHostMallocDefault));
    HIP_TRY(tnco::dev_malloc((void**)&h->beta_ring, (size_t)cap * 8));
    h->ring_cap = cap;
  }
  if (h->ring_pos + n_steps > h->ring_cap) {
    HIP_TRY(h->sync_all());
    h->ring_pos = 0;
  }
  double* dbetas = h->beta_ring + h->ring_pos;
  std::memcpy(h->beta_pin + h->ring_pos, betas, (size_t)n_steps * 8);
  HIP_TRY(hipMemcpyAsync(dbetas, h->beta_pin + h->ring_pos, (size_t)n_steps * 8, hipMemcpyHostToDevice, h->stream));
  h->ring_pos += n_steps;
  // per-launch work counters are 32-bit: at most (n_leaves - 1) moves per sweep
  const int64_t max_steps = std::max<int64_t>(1, (int64_t)0xF0000000u / std::max(1, h->P.n));
  if (h->n_groups <= 1) {
    for (int64_t s0 = 0; s0 < n_steps; s0 += max_steps) {
      const int64_t cnt = std::min(max_steps, n_steps - s0);
      HIP_TRY(h->timed(TNCO_KIND_SWEEP, [&]() { launch_run(h, dbetas + s0, cnt, prob_kind, h->stream); }));
      h->launches++;
    }
    if (h->pending.size() > 256) h->resolve_events();
    return TNCO_HIP_OK;
  }
  // Grouped: group q runs its blocks on its own stream, after this call's betas (fork event on the main
  // stream) and -- stream order -- after its own previous launch, NOT after the other groups'.  A long call
  // is cut into slices of ~100 sweeps so that the streams interleave inside one call as they do from call
  // to call (a replica's sweeps stay in order: one stream per group).
  if (h->region_open && h->region_b_set) h->close_region();  // (a join since the last call: that stretch is complete)
  if (!h->region_open) {
    HIP_TRY(hipEventRecord(h->region_a, h->stream));
    h->region_open = true;
  }
  HIP_TRY(hipEventRecord(h->gfork, h->stream));
  const int gpb = SWT / h->L;
  const int nblocks = (int)((h->P.R + gpb - 1) / gpb);
  const int64_t n_slices = std::max<int64_t>(1, std::max<int64_t>((n_steps + max_steps - 1) / max_steps, (n_steps + 50) / 100));
  for (int q = 0; q < h->n_groups; ++q) HIP_TRY(hipStreamWaitEvent(h->gstream[q], h->gfork, 0));
  for (int64_t i = 0; i < n_slices; ++i) {
    const int64_t s0 = n_steps * i / n_slices, s1 = n_steps * (i + 1) / n_slices;
    if (s1 == s0) continue;
    for (int q = 0; q < h->n_groups; ++q) {
      const int b0 = (int)((int64_t)nblocks * q / h->n_groups), b1 = (int)((int64_t)nblocks * (q + 1) / h->n_groups);
      if (b1 > b0) launch_run(h, dbetas + s0, s1 - s0, prob_kind, h->gstream[q], b0, b1 - b0);
    }
    HIP_TRY(hipGetLastError());
  }
  h->groups_dirty = true;
  h->region_calls++;
  h->launches++;
  return TNCO_HIP_OK;
}

int tnco_hip_run_fw(tnco_hip_handle h, int prob_kind, const double* betas, int64_t n_steps,
                    int64_t update_slices_every, int64_t step_offset) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  if (!h->fw) return fail(TNCO_HIP_EINVAL, "handle was created without 'max_width': use tnco_hip_run.");
  if (prob_kind < 0 || prob_kind > 2) return fail(TNCO_HIP_EINVAL, "'prob_kind' is not valid.");
  if (n_steps < 0 || (n_steps > 0 && !betas) || step_offset < 0) return fail(TNCO_HIP_EINVAL, "'betas' is not valid.");
  if (n_steps == 0) return TNCO_HIP_OK;
  if (h->fw_wave_capable && !h->fw_probed) {
    // A handle's first call: the sweeps up to its first re-slice as a call of their own, so that the rest already
    // runs in the form and configuration that re-slice's fall-backs ask for (a 1 000-tensor network sliced to 0.7
    // of its width spent its whole first call -- ten re-slices -- in the lean configuration, every replica falling
    // back: 9.4e7 move-evals/s against 1.2e9 afterwards).
    h->fw_probed = true;
    int64_t first = 0;  // sweeps up to and including the first re-slicing one (launch_fw_run: (offset + k) % every == 0)
    if (update_slices_every > 0) {
      const int64_t rem = step_offset % update_slices_every;
      first = (rem == 0 ? 0 : update_slices_every - rem) + 1;
    }
    if (first > 0 && n_steps > first) {
      if (int rc = tnco_hip_run_fw(h, prob_kind, betas, first, update_slices_every, step_offset)) return rc;
      return tnco_hip_run_fw(h, prob_kind, betas + first, n_steps - first, update_slices_every, step_offset + first);
    }
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  if (h->fw_wave_capable) {
    // The wavefront form or the general one (walk + full rebuild) for this call?  The first wins while fewer than
    // ~3 % of the replicas fall back to the full rebuild (more changed indices than it re-prices: random initial
    // trees early in a schedule) -- a fall-back costs the full rebuild on top.  The previous call's kernels are
    // complete here: its fall-backs are counted, or, after 4, 8, 16 ... calls in the other form, this call probes again.
    if (h->fw_wave_reslices > 0) {
      unsigned long long slow = 0, slow_wide = 0;
      HIP_TRY(h->collect_fw_stats(&slow, &slow_wide));
      const bool was_on = h->fw_wave_on;
      // (a replica that falls back costs its wavefront of fw_reslice_b_kernel the full rebuild -- sixteen replicas in lock
      //  step: at 3 % every third wavefront.  The roomier configuration serves networks whose full rebuild is tens of times
      //  the wavefront form: it keeps winning further out.)
      const double lim = (h->fw_wave_big ? 0.10 : 0.03) * (double)h->fw_wave_reslices * (double)h->P.R;
      if (!h->fw_wave_big && h->fw_wave_big_ok && (double)slow_wide >= lim && !std::getenv("TNCO_HIP_FW_BIG")) {
        // too many / too leggy too-wide tensors for the lean configuration: the roomier one before giving up the form
        h->fw_wave_big = true;
        h->set_wave_config();
        slow -= slow_wide;
      }
      h->fw_wave_on = (double)slow < lim;
      h->fw_probe_wait = h->fw_wave_on ? 4 : (was_on && h->fw_single_calls == 0 ? std::min(64, 2 * h->fw_probe_wait) : h->fw_probe_wait);
      h->fw_wave_reslices = 0;
      h->fw_single_calls = 0;
    } else if (!h->fw_wave_on && ++h->fw_single_calls >= h->fw_probe_wait) {
      h->fw_wave_on = true;
    }
    if (const char* e = std::getenv("TNCO_HIP_FW_WAVE")) h->fw_wave_on = std::atoi(e) != 0;  // (test knob: pin the form)
    h->F.fast_ok = h->fw_wave_on ? 1 : 0;
  }
  if (n_steps > h->betas_cap) {
    if (h->d_betas) (void)hipFree(h->d_betas);
    h->d_betas = nullptr;
    h->betas_cap = 0;
    HIP_TRY(tnco::dev_malloc((void**)&h->d_betas, (size_t)n_steps * 8));
    h->betas_cap = n_steps;
  }
  HIP_TRY(hipMemcpyAsync(h->d_betas, betas, (size_t)n_steps * 8, hipMemcpyHostToDevice, h->stream));
  const int64_t max_steps = std::max<int64_t>(1, (int64_t)0xF0000000u / std::max(1, h->P.n));
  if (h->n_groups > 1) {  // the halves of the batch on streams of their own (GroupView), after this call's betas
    if (h->region_open && h->region_b_set) h->close_region();
    if (!h->region_open) {
      HIP_TRY(hipEventRecord(h->region_a, h->stream));
      h->region_open = true;
    }
    HIP_TRY(hipEventRecord(h->gfork, h->stream));
    for (int q = 0; q < h->n_groups; ++q) HIP_TRY(hipStreamWaitEvent(h->gstream[q], h->gfork, 0));
  }
  for (int64_t s0 = 0; s0 < n_steps; s0 += max_steps) {
    const int64_t cnt = std::min(max_steps, n_steps - s0);
    if (h->n_groups > 1) {
      const int64_t R = h->P.R;
      for (int q = 0; q < h->n_groups; ++q) {
        const int64_t r0 = R * q / h->n_groups, r1 = R * (q + 1) / h->n_groups;
        GroupView gv(h, r0, r1 - r0, h->gstream[q]);
        HIP_TRY(launch_fw_run(h, h->d_betas + s0, cnt, prob_kind, step_offset + s0, update_slices_every, q == 0));
      }
      h->groups_dirty = true;
    } else {
      HIP_TRY(launch_fw_run(h, h->d_betas + s0, cnt, prob_kind, step_offset + s0, update_slices_every));
    }
    h->launches++;
  }
  if (h->n_groups > 1) h->region_calls++;
  if (h->pending.size() > 256) h->resolve_events();
  return TNCO_HIP_OK;
}

int tnco_hip_get_slices(tnco_hip_handle h, int64_t r, uint64_t* slices, uint64_t* min_slices) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  if (!h->fw) return fail(TNCO_HIP_EINVAL, "handle was created without 'max_width'.");
  if (r < 0 || r >= h->P.R) return fail(TNCO_HIP_EINVAL, "'replica' out of range.");
  const int LK = h->L * h->K, W = h->P.W;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  std::vector<uint64_t> s((size_t)2 * LK);
  HIP_TRY(hipMemcpy(s.data(), h->F.slices + r * 2 * (int64_t)LK, s.size() * 8, hipMemcpyDeviceToHost));
  if (slices) std::memcpy(slices, s.data(), (size_t)W * 8);
  if (min_slices) std::memcpy(min_slices, s.data() + LK, (size_t)W * 8);
  return TNCO_HIP_OK;
}

int tnco_hip_diag_reslice_info(tnco_hip_handle h, int32_t* how, int32_t* n_changed) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  if (!h->fw || !h->fw_wave_capable) return fail(TNCO_HIP_EINVAL, "handle has no re-pricing re-slice.");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  const int64_t R = h->P.R;
  if (how) HIP_TRY(hipMemcpy(how, h->F.fastflag, (size_t)R * 4, hipMemcpyDeviceToHost));
  if (n_changed) {  // word 0 of every replica's change list (fw_reslice_a_kernel), 512 bytes apart
    HIP_TRY(hipMemcpy2D(n_changed, 4, h->F.delta_scr, 512, 4, (size_t)R, hipMemcpyDeviceToHost));
  }
  return TNCO_HIP_OK;
}

int tnco_hip_diag_fw_stats(tnco_hip_handle h, int64_t* out8) {
  if (!h || !out8) return fail(TNCO_HIP_EINVAL, "null argument.");
  if (!h->fw) return fail(TNCO_HIP_EINVAL, "handle was created without 'max_width'.");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  if (h->F.slowstat) HIP_TRY(h->collect_fw_stats(nullptr));
  for (int i = 0; i < 8; ++i) out8[i] = h->fw_stats[i];
  return TNCO_HIP_OK;
}

int tnco_hip_diag_kernel_time(tnco_hip_handle h, double* ms, int64_t* launches, int reset) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  h->close_region();
  h->resolve_events();
  if (ms) *ms = h->kernel_ms;
  if (launches) *launches = h->launches;
  if (reset) h->reset_times();
  return TNCO_HIP_OK;
}

int tnco_hip_diag_kernel_times(tnco_hip_handle h, double* ms4, int64_t* launches4, int reset) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  h->close_region();
  h->resolve_events();
  // (a finite-width handle on two streams: the AVERAGE time a stream spent in each kernel -- the streams run
  //  concurrently, so these add up to about the device time tnco_hip_diag_kernel_time reports, not to twice it)
  const double div = (h->fw && h->n_groups > 1) ? (double)h->n_groups : 1.0;
  for (int k = 0; k < TNCO_KINDS; ++k) {
    if (ms4) ms4[k] = h->kind_ms[k] / div;
    if (launches4) launches4[k] = (int64_t)((double)h->kind_launches[k] / div);
  }
  if (reset) h->reset_times();
  return TNCO_HIP_OK;
}

int tnco_hip_diag_launch_groups(tnco_hip_handle h) { return h ? (h->small_tree || h->lds_tree ? 0 : h->n_groups) : 0; }

int64_t tnco_hip_diag_device_bytes(tnco_hip_handle h) { return h ? h->bytes : 0; }

int tnco_hip_get_costs(tnco_hip_handle h, double* total_cost, double* min_total_cost) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  if (!total_cost && !min_total_cost) return TNCO_HIP_OK;
  const int64_t R = h->P.R;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  if (int rc = fw_runtime_status(h)) return rc;
  TempBufs tmp;
  double *dt = nullptr, *dm = nullptr;
  if (total_cost) HIP_TRY(tmp.alloc(&dt, R));
  if (min_total_cost) HIP_TRY(tmp.alloc(&dm, R));
  hipLaunchKernelGGL(gather_costs_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, h->stream, h->P, dt, dm);
  HIP_TRY(hipGetLastError());
  if (total_cost) HIP_TRY(hipMemcpyAsync(total_cost, dt, (size_t)R * 8, hipMemcpyDeviceToHost, h->stream));
  if (min_total_cost) HIP_TRY(hipMemcpyAsync(min_total_cost, dm, (size_t)R * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h->sync_all());
  return TNCO_HIP_OK;
}

int tnco_hip_get_slices_many(tnco_hip_handle h, int64_t k, const int64_t* ids, uint64_t* slices, uint64_t* min_slices) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  if (!h->fw) return fail(TNCO_HIP_EINVAL, "handle was created without 'max_width'.");
  if (k < 0 || (k > 0 && !ids)) return fail(TNCO_HIP_EINVAL, "null argument.");
  for (int64_t i = 0; i < k; ++i)
    if (ids[i] < 0 || ids[i] >= h->P.R) return fail(TNCO_HIP_EINVAL, "'replica' out of range.");
  if (k == 0 || (!slices && !min_slices)) return TNCO_HIP_OK;
  const int LK = h->L * h->K, W = h->P.W;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  TempBufs tmp;
  int64_t* dids = nullptr;
  uint64_t *dc = nullptr, *dm = nullptr;
  HIP_TRY(tmp.alloc(&dids, k));
  if (slices) HIP_TRY(tmp.alloc(&dc, k * W));
  if (min_slices) HIP_TRY(tmp.alloc(&dm, k * W));
  HIP_TRY(hipMemcpyAsync(dids, ids, (size_t)k * 8, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(gather_slices_kernel, dim3((unsigned)((k * W + 255) / 256)), dim3(256), 0, h->stream, h->F.slices, LK, W,
                     dids, k, dc, dm);
  HIP_TRY(hipGetLastError());
  if (slices) HIP_TRY(hipMemcpyAsync(slices, dc, (size_t)k * W * 8, hipMemcpyDeviceToHost, h->stream));
  if (min_slices) HIP_TRY(hipMemcpyAsync(min_slices, dm, (size_t)k * W * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h->sync_all());
  return TNCO_HIP_OK;
}

int tnco_hip_get_tree(tnco_hip_handle h, int64_t r, int which, int32_t* left, int32_t* right,
                      int32_t* parent, uint64_t* masks) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  if (r < 0 || r >= h->P.R) return fail(TNCO_HIP_EINVAL, "'replica' out of range.");
  if (!left || !right || !parent) return fail(TNCO_HIP_EINVAL, "null output array.");
  const int n = h->P.n, N = h->P.N, W = h->P.W, BS = h->P.BS;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  if (which == 0) {
    std::vector<uint8_t> blk((size_t)h->block_bytes());
    HIP_TRY(hipMemcpy(blk.data(), h->P.blocks + r * h->block_bytes(), blk.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy2D(parent, 4, h->P.lpar + r * (int64_t)n * LPS, (size_t)LPS * 4, 4, (size_t)n,
                        hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) { left[i] = -1; right[i] = -1; }
    if (masks) std::memcpy(masks, h->leafmask_w.data(), (size_t)n * W * 8);
    for (int p = n; p < N; ++p) {
      const uint8_t* b = blk.data() + (size_t)(p - n) * BS;
      NodeRec hd;
      std::memcpy(&hd, b, sizeof(hd));
      left[p] = hd.left; right[p] = hd.right; parent[p] = hd.parent;
      if (masks) std::memcpy(masks + (size_t)p * W, blk.data() + (size_t)h->P.WOFF + (size_t)(p - n) * h->P.WS, (size_t)W * 8);
    }
  } else {
    // best tree = checkpoint + rotations jlog[0, jmin): Tree::swap_with_nn replayed on the host
    std::vector<Links> t((size_t)N);
    ReplicaState rs;
    HIP_TRY(hipMemcpy(&rs, h->P.rs + r, sizeof(rs), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t.data(), h->P.minlinks + r * (int64_t)N, (size_t)N * sizeof(Links), hipMemcpyDeviceToHost));
    std::vector<int32_t> lg((size_t)rs.jmin);
    if (rs.jmin)
      HIP_TRY(hipMemcpy(lg.data(), h->P.jlog + r * (int64_t)h->P.jcap, (size_t)rs.jmin * 4, hipMemcpyDeviceToHost));
    for (int32_t D : lg) {
      const int B = t[D].parent;
      const int A = t[B].parent;
      const int C = (t[A].left == B) ? t[A].right : t[A].left;
      if (t[A].left != C) t[A].right = D; else t[A].left = D;
      if (t[B].left != D) t[B].right = C; else t[B].left = C;
      t[C].parent = B;
      t[D].parent = A;
    }
    for (int i = 0; i < N; ++i) { left[i] = t[i].left; right[i] = t[i].right; parent[i] = t[i].parent; }
    if (masks) host_derive(h, left, right, masks);
  }
  return TNCO_HIP_OK;
}

int tnco_hip_get_caches(tnco_hip_handle h, int64_t r, double* ccost, double* partial, uint64_t* hyper) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  if (r < 0 || r >= h->P.R) return fail(TNCO_HIP_EINVAL, "'replica' out of range.");
  const int n = h->P.n, N = h->P.N, W = h->P.W, BS = h->P.BS;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  std::vector<uint8_t> blk((size_t)h->block_bytes());
  HIP_TRY(hipMemcpy(blk.data(), h->P.blocks + r * h->block_bytes(), blk.size(), hipMemcpyDeviceToHost));
  if (hyper) std::memset(hyper, 0, (size_t)N * W * 8);
  for (int i = 0; i < N; ++i) {
    NodeRec hd{};
    if (i >= n) std::memcpy(&hd, blk.data() + (size_t)(i - n) * BS, sizeof(hd));
    if (ccost) ccost[i] = i < n ? 0.0 : hd.ccost;
    if (partial) partial[i] = i < n ? 0.0 : hd.partial;
    if (hyper && h->hyper && i >= n) {
      {  // HyperCache (infinite_memory/utils.hpp:82-91): legs(p) & legs(c0) & legs(c1), from the stored legs
        auto legs = [&](int x) -> const uint64_t* {
          return x < n ? h->leafmask_w.data() + (size_t)x * W
                       : reinterpret_cast<const uint64_t*>(blk.data() + ((int64_t)h->P.WOFF + (int64_t)(x - n) * h->P.WS));
        };
        const uint64_t *lp = legs(i), *l0 = legs(hd.left), *l1 = legs(hd.right);
        for (int w = 0; w < W; ++w) hyper[(size_t)i * W + w] = lp[w] & l0[w] & l1[w];
      }
    }
  }
  return TNCO_HIP_OK;
}

int tnco_hip_validate(tnco_hip_handle h, double atol, int64_t* n_bad, int64_t* first_bad) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  const Params& P = h->P;
  const int64_t LK = (int64_t)h->L * h->K;
  const int n = P.n, N = P.N;
  const int64_t R = P.R;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  const int64_t per = h->block_bytes() + (int64_t)n * 4 + (int64_t)N * 32 + 64;
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(R, ((int64_t)1 << 30) / per));
  TempBufs tmp;
  uint8_t* tblk = nullptr; int32_t *tlpar = nullptr, *tscr = nullptr, *tstat = nullptr, *tbad = nullptr;
  double *ttot = nullptr, *tsum = nullptr;
  Links* tlinks = nullptr;
  HIP_TRY(tmp.alloc(&tblk, chunk * h->block_bytes()));
  HIP_TRY(tmp.alloc(&tlpar, chunk * n * LPS));
  HIP_TRY(tmp.alloc(&tscr, chunk * 4 * N));
  HIP_TRY(tmp.alloc(&tlinks, chunk * N));
  HIP_TRY(tmp.alloc(&tstat, chunk));
  HIP_TRY(tmp.alloc(&tbad, chunk));
  HIP_TRY(tmp.alloc(&ttot, chunk));
  HIP_TRY(tmp.alloc(&tsum, chunk));
  uint64_t* thy = nullptr;
  if (h->hyper) HIP_TRY(tmp.alloc(&thy, chunk * (int64_t)(n - 1) * h->P.W));
  int64_t bad = 0, first = -1;
  std::vector<ReplicaState> rs;
  if (int rc = fetch_rs(h, rs)) return rc;
  std::vector<int32_t> hb((size_t)chunk), hs((size_t)chunk);
  std::vector<double> hsum((size_t)chunk);
  for (int64_t r0 = 0; r0 < R; r0 += chunk) {
    const int64_t cnt = std::min(chunk, R - r0);
    BuildArgs a{};
    a.hyper_tmp = thy;
    a.out_blocks = tblk; a.out_lpar = tlpar; a.scratch = tscr;
    a.out_total = ttot; a.out_sum = tsum; a.out_status = tstat; a.r0 = r0; a.count = cnt;
    // (1) current tree: rebuild everything and compare
    a.src_live = 1;
    if (h->fw) { a.cost_slices = h->F.slices; a.cost_slices_stride = 2 * LK; }
    launch_build(h, a);
    launch_compare(h, a, atol, tbad);
    if (h->fw) launch_fw_check(h, a, 0, atol, tbad);
    HIP_TRY(h->sync_all());
    HIP_TRY(hipMemcpy(hb.data(), tbad, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    // (2) best tree (checkpoint + rotation log): its cost must match min_total_cost
    hipLaunchKernelGGL(materialize_min_kernel, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, h->stream, P, tlinks, r0, cnt);
    a.src_live = 0;
    a.src_links = tlinks;
    if (h->fw) { a.cost_slices = h->F.slices + LK; a.cost_slices_stride = 2 * LK; }
    launch_build(h, a);
    if (h->fw) {  // the widths of the best tree, against min_slices
      HIP_TRY(hipMemsetAsync(tbad, 0, (size_t)cnt * 4, h->stream));
      launch_fw_check(h, a, 1, atol, tbad);
      HIP_TRY(h->sync_all());
      std::vector<int32_t> hb2((size_t)cnt);
      HIP_TRY(hipMemcpy(hb2.data(), tbad, (size_t)cnt * 4, hipMemcpyDeviceToHost));
      for (int64_t q = 0; q < cnt; ++q)
        if (hb2[q]) hb[q] = hb[q] ? hb[q] : hb2[q];
    }
    HIP_TRY(h->sync_all());
    HIP_TRY(hipMemcpy(hs.data(), tstat, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hsum.data(), tsum, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    for (int64_t q = 0; q < cnt; ++q) {
      const bool b = hb[q] != 0 || hs[q] != 0 || !logclose(hsum[q], rs[r0 + q].min_cost, atol);
      if (b) { ++bad; if (first < 0) first = r0 + q; }
    }
  }
  if (n_bad) *n_bad = bad;
  if (first_bad) *first_bad = first;
  return TNCO_HIP_OK;
}

int tnco_hip_get_prng(tnco_hip_handle h, int64_t r, uint32_t* out) {
  if (!h || !out) return fail(TNCO_HIP_EINVAL, "null argument.");
  if (r < 0 || r >= h->P.R) return fail(TNCO_HIP_EINVAL, "'replica' out of range.");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  ReplicaState rs;
  HIP_TRY(hipMemcpy(&rs, h->P.rs + r, sizeof(rs), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out, h->P.mt + r * 624, 624 * 4, hipMemcpyDeviceToHost));
  // finish the lazily generated block sequence so the 624 words are what
  // libstdc++ holds after _M_gen_rand() (random.tcc:396-430)
  if (rs.mti < 624) {
    for (int k = rs.mtw; k < 624; ++k) {
      const uint32_t y = (out[k] & 0x80000000u) | (out[(k + 1) % 624] & 0x7fffffffu);
      out[k] = out[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
  }
  out[624] = (uint32_t)rs.mti;
  return TNCO_HIP_OK;
}

int tnco_hip_set_prng(tnco_hip_handle h, int64_t r, const uint32_t* in) {
  if (!h || !in) return fail(TNCO_HIP_EINVAL, "null argument.");
  if (r < 0 || r >= h->P.R) return fail(TNCO_HIP_EINVAL, "'replica' out of range.");
  if (in[624] > 624) return fail(TNCO_HIP_EINVAL, "prng position out of range.");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  ReplicaState rs;
  HIP_TRY(hipMemcpy(&rs, h->P.rs + r, sizeof(rs), hipMemcpyDeviceToHost));
  rs.mti = (int32_t)in[624];
  rs.mtw = 624;
  HIP_TRY(hipMemcpy(h->P.mt + r * 624, in, 624 * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->P.rs + r, &rs, sizeof(rs), hipMemcpyHostToDevice));
  return TNCO_HIP_OK;
}

int tnco_hip_get_prng_many(tnco_hip_handle h, int64_t k, const int64_t* ids, uint32_t* out) {
  if (!h || (k > 0 && !out)) return fail(TNCO_HIP_EINVAL, "null argument.");
  if (k < 0 || (!ids && k > h->P.R)) return fail(TNCO_HIP_EINVAL, "'k' is not valid.");
  for (int64_t i = 0; ids && i < k; ++i)
    if (ids[i] < 0 || ids[i] >= h->P.R) return fail(TNCO_HIP_EINVAL, "'replica' out of range.");
  if (k == 0) return TNCO_HIP_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  std::vector<ReplicaState> rs;
  if (int rc = fetch_rs(h, rs)) return rc;
  if (!ids) {  // replicas 0 .. k-1: one strided copy
    HIP_TRY(hipMemcpy2D(out, (size_t)625 * 4, h->P.mt, (size_t)624 * 4, (size_t)624 * 4, (size_t)k, hipMemcpyDeviceToHost));
  } else {
    for (int64_t i = 0; i < k; ++i)
      HIP_TRY(hipMemcpyAsync(out + i * 625, h->P.mt + ids[i] * 624, 624 * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h->sync_all());
  }
  // finish the lazily generated block sequence so the 624 words are what libstdc++ holds after
  // _M_gen_rand() (random.tcc:396-430)
  const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(k / 256 + 1, std::min<unsigned>(16, std::thread::hardware_concurrency())));
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&, t]() {
      for (int64_t i = t; i < k; i += nth) {
        const ReplicaState& x = rs[(size_t)(ids ? ids[i] : i)];
        uint32_t* o = out + i * 625;
        if (x.mti < 624)
          for (int j = x.mtw; j < 624; ++j) {
            const uint32_t y = (o[j] & 0x80000000u) | (o[(j + 1) % 624] & 0x7fffffffu);
            o[j] = o[(j + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
          }
        o[624] = (uint32_t)x.mti;
      }
    });
  for (auto& x : th) x.join();
  return TNCO_HIP_OK;
}

int tnco_hip_set_prng_many(tnco_hip_handle h, int64_t k, const int64_t* ids, const uint32_t* in) {
  if (!h || (k > 0 && !in)) return fail(TNCO_HIP_EINVAL, "null argument.");
  if (k < 0 || (!ids && k > h->P.R)) return fail(TNCO_HIP_EINVAL, "'k' is not valid.");
  for (int64_t i = 0; i < k; ++i) {
    if (ids && (ids[i] < 0 || ids[i] >= h->P.R)) return fail(TNCO_HIP_EINVAL, "'replica' out of range.");
    if (in[i * 625 + 624] > 624) return fail(TNCO_HIP_EINVAL, "prng position out of range.");
  }
  if (k == 0) return TNCO_HIP_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  TempBufs tmp;
  uint32_t* dpos = nullptr;
  int64_t* dids = nullptr;
  std::vector<uint32_t> pos((size_t)k);
  for (int64_t i = 0; i < k; ++i) pos[(size_t)i] = in[i * 625 + 624];
  HIP_TRY(tmp.alloc(&dpos, k));
  HIP_TRY(hipMemcpy(dpos, pos.data(), (size_t)k * 4, hipMemcpyHostToDevice));
  if (!ids) {
    HIP_TRY(hipMemcpy2D(h->P.mt, (size_t)624 * 4, in, (size_t)625 * 4, (size_t)624 * 4, (size_t)k, hipMemcpyHostToDevice));
  } else {
    HIP_TRY(tmp.alloc(&dids, k));
    HIP_TRY(hipMemcpy(dids, ids, (size_t)k * 8, hipMemcpyHostToDevice));
    for (int64_t i = 0; i < k; ++i)
      HIP_TRY(hipMemcpyAsync(h->P.mt + ids[i] * 624, in + i * 625, 624 * 4, hipMemcpyHostToDevice, h->stream));
  }
  hipLaunchKernelGGL(mt_pos_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, h->stream, h->P.rs, dpos, dids, k);
  HIP_TRY(hipGetLastError());
  HIP_TRY(h->sync_all());
  return TNCO_HIP_OK;
}

int tnco_hip_best(tnco_hip_handle h, int64_t k, double* costs, int64_t* replicas) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  if (k < 0 || k > h->P.R) return fail(TNCO_HIP_EINVAL, "'k' out of range.");
  if (k == 0) return TNCO_HIP_OK;
  HIP_TRY(hipSetDevice(h->device));
  if (k > TOPK_CHUNK / 2 || h->P.R >= ((int64_t)1 << 32)) {
    // a long head of the list: the replica records on the host (k of the order of R is a full sort anyway)
    std::vector<ReplicaState> rs;
    if (int rc = fetch_rs(h, rs)) return rc;
    std::vector<int64_t> idx((size_t)h->P.R);
    std::iota(idx.begin(), idx.end(), (int64_t)0);
    auto less = [&](int64_t a, int64_t b) {
      return rs[a].min_cost < rs[b].min_cost || (rs[a].min_cost == rs[b].min_cost && a < b);
    };
    std::partial_sort(idx.begin(), idx.begin() + k, idx.end(), less);
    for (int64_t i = 0; i < k; ++i) {
      if (costs) costs[i] = rs[idx[i]].min_cost;
      if (replicas) replicas[i] = idx[i];
    }
    return TNCO_HIP_OK;
  }
  // k-select on the device: only k (cost, replica) pairs cross PCIe
  HIP_TRY(h->sync_all());
  if (int rc = fw_runtime_status(h)) return rc;
  const int keep = (int)std::min<int64_t>(k, TOPK_CHUNK / 2);
  int64_t count = h->P.R;
  int64_t blocks = (count + TOPK_CHUNK - 1) / TOPK_CHUNK;
  TempBufs tmp;
  unsigned long long* bc[2];
  uint32_t* bi[2];
  for (int j = 0; j < 2; ++j) {
    HIP_TRY(tmp.alloc(&bc[j], blocks * keep));
    HIP_TRY(tmp.alloc(&bi[j], blocks * keep));
  }
  int cur = 0;
  hipLaunchKernelGGL(topk_pass_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, h->P.rs, nullptr, nullptr, count,
                     keep, bc[0], bi[0]);
  count = blocks * keep;
  while (blocks > 1) {
    blocks = (count + TOPK_CHUNK - 1) / TOPK_CHUNK;
    hipLaunchKernelGGL(topk_pass_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, nullptr, bc[cur], bi[cur], count,
                       keep, bc[1 - cur], bi[1 - cur]);
    cur = 1 - cur;
    count = blocks * keep;
  }
  HIP_TRY(hipGetLastError());
  std::vector<unsigned long long> hc((size_t)k);
  std::vector<uint32_t> hi((size_t)k);
  HIP_TRY(hipMemcpyAsync(hc.data(), bc[cur], (size_t)k * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(hi.data(), bi[cur], (size_t)k * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h->sync_all());
  for (int64_t i = 0; i < k; ++i) {
    double c;
    std::memcpy(&c, &hc[i], 8);
    if (costs) costs[i] = c;
    if (replicas) replicas[i] = (int64_t)hi[i];
  }
  return TNCO_HIP_OK;
}

int tnco_hip_min_cost_device(tnco_hip_handle h, void* device_dst_f64) {
  if (!h || !device_dst_f64) return fail(TNCO_HIP_EINVAL, "null argument.");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->join_groups());
  hipLaunchKernelGGL(min_cost_kernel, dim3(1), dim3(1024), 0, h->stream, h->P.rs, h->P.R, (double*)device_dst_f64);
  HIP_TRY(hipGetLastError());
  HIP_TRY(h->sync_all());  // (the caller's collective runs on another stream)
  return fw_runtime_status(h);
}

int tnco_hip_get_trees(tnco_hip_handle h, int64_t k, const int64_t* ids, int which, int32_t* links,
                       int32_t* contraction) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  if (k < 0 || (k > 0 && (!ids || !links))) return fail(TNCO_HIP_EINVAL, "null argument.");
  if (which != 0 && which != 1) return fail(TNCO_HIP_EINVAL, "'which' is not valid.");
  for (int64_t i = 0; i < k; ++i)
    if (ids[i] < 0 || ids[i] >= h->P.R) return fail(TNCO_HIP_EINVAL, "'replica' out of range.");
  if (k == 0) return TNCO_HIP_OK;
  const int n = h->P.n, N = h->P.N;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(h->sync_all());
  if (int rc = fw_runtime_status(h)) return rc;
  // in chunks, so that the device-side staging stays small whatever k
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(k, ((int64_t)256 << 20) / ((int64_t)N * 40)));
  const size_t lds_bytes = (size_t)5 * N * 4;
  const bool in_lds = lds_bytes <= 60 * 1024;
  TempBufs tmp;
  int64_t* dids = nullptr;
  int32_t *dlinks = nullptr, *dcon = nullptr, *dscr = nullptr;
  HIP_TRY(tmp.alloc(&dids, chunk));
  HIP_TRY(tmp.alloc(&dlinks, chunk * 3 * N));
  if (contraction) HIP_TRY(tmp.alloc(&dcon, chunk * 3 * (n - 1)));
  if (!in_lds) HIP_TRY(tmp.alloc(&dscr, chunk * 5 * N));
  for (int64_t k0 = 0; k0 < k; k0 += chunk) {
    const int64_t cnt = std::min(chunk, k - k0);
    HIP_TRY(hipMemcpyAsync(dids, ids + k0, (size_t)cnt * 8, hipMemcpyHostToDevice, h->stream));
    if (in_lds)
      hipLaunchKernelGGL((gather_trees_kernel<true>), dim3((unsigned)cnt), dim3(64), lds_bytes, h->stream, h->P, dids, which,
                         dlinks, dcon, nullptr);
    else
      hipLaunchKernelGGL((gather_trees_kernel<false>), dim3((unsigned)cnt), dim3(64), 0, h->stream, h->P, dids, which,
                         dlinks, dcon, dscr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(links + k0 * 3 * N, dlinks, (size_t)cnt * 3 * N * 4, hipMemcpyDeviceToHost, h->stream));
    if (contraction)
      HIP_TRY(hipMemcpyAsync(contraction + k0 * 3 * (n - 1), dcon, (size_t)cnt * 3 * (n - 1) * 4, hipMemcpyDeviceToHost,
                             h->stream));
    HIP_TRY(h->sync_all());
  }
  return TNCO_HIP_OK;
}

int tnco_hip_diag_counters(tnco_hip_handle h, uint64_t* moves, uint64_t* accepted, uint64_t* improved,
                          uint64_t* random_picks) {
  if (!h) return fail(TNCO_HIP_EINVAL, "null handle.");
  std::vector<ReplicaState> rs;
  if (int rc = fetch_rs(h, rs)) return rc;
  uint64_t m = 0, a = 0, i = 0, q = 0;
  for (auto& x : rs) { m += x.n_moves; a += x.n_accepted; i += x.n_improved; q += x.n_randpick; }
  if (moves) *moves = m;
  if (accepted) *accepted = a;
  if (improved) *improved = i;
  if (random_picks) *random_picks = q;
  return TNCO_HIP_OK;
}

int tnco_hip_diag_full_copies(tnco_hip_handle h, uint64_t* n) {
  if (!h || !n) return fail(TNCO_HIP_EINVAL, "null argument.");
  std::vector<ReplicaState> rs;
  if (int rc = fetch_rs(h, rs)) return rc;
  uint64_t c = 0;
  for (auto& x : rs) c += x.n_fullcopy;
  *n = c;
  return TNCO_HIP_OK;
}

int tnco_hip_diag_stage_cycles(tnco_hip_handle h, uint64_t* out5) {
  if (!h || !out5) return fail(TNCO_HIP_EINVAL, "null argument.");
  std::vector<ReplicaState> rs;
  if (int rc = fetch_rs(h, rs)) return rc;
  for (int k = 0; k < 5; ++k) out5[k] = 0;
  for (auto& x : rs)
    for (int k = 0; k < 5; ++k) out5[k] += x.pad1[k];
  return TNCO_HIP_OK;
}

int tnco_hip_diag_moves(tnco_hip_handle h, uint64_t* out) {
  if (!h || !out) return fail(TNCO_HIP_EINVAL, "null argument.");
  std::vector<ReplicaState> rs;
  if (int rc = fetch_rs(h, rs)) return rc;
  for (size_t r = 0; r < rs.size(); ++r) out[r] = rs[r].n_moves;
  return TNCO_HIP_OK;
}

}  // extern "C"
