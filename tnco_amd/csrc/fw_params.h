// fw_params.h -- parameters, scratch layout and width model of the finite-width optimizer
// (include/tnco/optimize/finite_width/greedy/optimizer.hpp, finite_width/cost_model/simple.hpp of the
// reference), shared by the staged sweep kernel (sa_sweep.h, FW = true) and fw_kernels.h.
#pragma once
#include "sa_kernels.h"

namespace tnco {

struct FwParams {
  double max_width;         // width_type(max_width), held as double
  double log2d;             // std::log2((double)dims), uniform dims (cost_model/simple.hpp:46)
  const double* log2dims;   // [LK*64] std::log2((double)dims[p]), per-index dims; else NULL
  double log2np;            // std::log2((double)n_projs) (simple_sparse_inds.hpp:45)
  int32_t width_f32;        // width_type float32 (else float64)
  int32_t leaf_wide;        // 1 when some leaf tensor is wider than max_width ...
  const uint32_t* leaf_bits;  // ... [ceil(n / 32)] which ones (fw_leaf_bits_kernel; leaves never change)
  int32_t I64;              // 64 * LK (padded index count)
  int32_t stack_cap;        // LDS entries of the traversal stack (<= FW_LDSPOS; 0: walk the links instead);
                            // FW_LDSPOS unless TNCO_HIP_FW_STACK says otherwise (tests of the fallbacks)
  int64_t max_new_slices;   // max_number_new_slices
  uint64_t* slices;         // [R][2][LK]  slices, min_slices
  const uint64_t* skip;     // [LK] or NULL
  int32_t* scratch_i;       // [R][fw_scratch_ints]  see FwScratch
  double* scratch_d;        // [R][2N]        see FwScratch
  double* width64;          // [R][N] widths of the internal nodes when width_type is float64
  int32_t* nwide;           // [R] too-wide tensors listed by fw_walk2_kernel for fw_reslice_kernel; from fw_wave_kernel: -1 no slices,
                            //     nothing to do; -2 left to fw_reslice_a_kernel's own traverse; -3 get_slices done
  int32_t* nwfront;         // [R] ... how many of them at the front of the list (fw_walk2_kernel: the rest at its end)
  // the re-slice of a replica in one wavefront, its cost cache re-priced from the OLD costs (fw_wave_kernel)
  int32_t fast_ok;          // this call runs it: uniform power-of-two dims, float64 cost, no sparse legs, hyper-indices on <= 7 tensors,
                            // <= 2048 tensors, no too-wide leaf, split layout; few fall-backs lately (tnco_hip_run_fw)
  const int32_t* holder2;   // [I64][2] the (one or two) tensors holding an index, -1: none / index not supported
  const uint16_t* holdern;  // [I64][8] networks with hyper-indices instead: count | open << 15 (open: an output index, or held by
                            //          one tensor), then the (up to 7) tensors holding the index; count 0: not supported
  int32_t* fastflag;        // [R] 1: fw_wave_kernel has done this replica's rebuild (+ commit)
  unsigned long long* slowstat;  // [4] replicas the re-pricing has left to the full rebuild since the host last looked; of those
                                 //     [1] too many / too deep / too leggy too-wide tensors, [2] too many changed indices or an index
                                 //     held otherwise, [3] a cost outside a double's powers of two
  uint64_t* delta_scr;      // [R][64] word 0: indices the last proposal changed (0xFFFFFFFF: not re-priced), for tnco_hip_diag_reslice_info
  int32_t* status;          // [R] runtime problems (1: candidate legs beyond the scratch; cannot happen
                            //     since the scratch holds every index)
};

// int32 scratch of one replica (FwParams::scratch_i): the too-wide counts of every index [I64],
// the candidate legs of one tensor (int16, for the shuffle when they do not fit the LDS fast path)
// [I64: every index can be one], the post-order records of the internal nodes (fw_rec, 8 bytes)
// [n - 1], the too-wide tensors in post-order [N], the deep part of the traversal stack [N].
__host__ __device__ inline int64_t fw_np(int N) { return (N + 15) & ~15; }  // (64-byte pieces: fw_walk_kernel)
__host__ __device__ inline int64_t fw_scratch_ints(int N, int I64) { return I64 + I64 / 2 + 3 * fw_np(N); }
struct FwScratch {
  int32_t* n_big;
  volatile int16_t* pos;
  uint64_t* rec;
  int32_t* wlist;
  int32_t* gstk;
  double2* cp;   // FwParams::scratch_d: rebuilt (cost, partial sum) by post-order number [n - 1]
  double* pstk;  // ... and the stack of partial sums [N]
  int32_t wcap;  // entries of wlist
  int32_t nwf;   // too-wide tensors at the front of wlist; the others are its last entries (fw_walk2_kernel)
  // the j-th of nw too-wide tensors, in post-order
  __device__ __forceinline__ int32_t wl(int j, int nw) const { return wlist[j < nwf ? j : wcap - nw + j]; }
  __device__ __forceinline__ FwScratch(const FwParams& F, int64_t r, int N) {
    int32_t* si = F.scratch_i + r * fw_scratch_ints(N, F.I64);
    n_big = si;
    pos = reinterpret_cast<volatile int16_t*>(si + F.I64);
    rec = reinterpret_cast<uint64_t*>(si + F.I64 + F.I64 / 2);
    wlist = si + F.I64 + F.I64 / 2 + fw_np(N);
    gstk = wlist + fw_np(N);
    wcap = (int32_t)fw_np(N);
    nwf = 0x7FFFFFFF;
    double* sd = F.scratch_d + r * 2 * (int64_t)N;
    cp = reinterpret_cast<double2*>(sd);
    pstk = sd + N;
  }
};

// a value of width_type, held in a double
__device__ __forceinline__ double fw_wr(const FwParams& F, double x) {
  return F.width_f32 ? (double)(float)x : x;
}

// SimpleCostModel::width, finite_width/cost_model/simple.hpp:38-57: scalar dims ->
// log2(dims) * count in double, converted to width_type; per-index dims -> running sum in
// width_type of log2(dims[p]) over ascending positions (word k*L + j is slot k of lane j).
template <int LOG2L, int K>
__device__ __forceinline__ double fw_width_simple(const FwParams& F, const Mask<K>& m, int gbase) {
  constexpr int L = 1 << LOG2L;
  if (F.log2dims == nullptr) return fw_wr(F, F.log2d * (double)gsum<LOG2L>(mpopc<K>(m)));
  double ws = 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    for (int j = 0; j < L; ++j) {
      const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)m.w[k], gbase + j);
      const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(m.w[k] >> 32), gbase + j);
      uint64_t x = ((uint64_t)hi << 32) | lo;
      const int w = k * L + j;
      while (x) {
        const int b = __ffsll((unsigned long long)x) - 1;
        ws = fw_wr(F, ws + F.log2dims[w * 64 + b]);
        x &= x - 1;
      }
    }
  }
  return ws;
}

template <int LOG2L, int K>
__device__ __forceinline__ Mask<K> fw_sparse_mask(const Params& P, int lig) {
  Mask<K> s;
#pragma unroll
  for (int k = 0; k < K; ++k) s.w[k] = P.sparse[k * (1 << LOG2L) + lig];
  return s;
}

// width of a leg set; with sparse legs: width(inds - S) + min(width(inds & S), log2(n_projs))
// (simple_sparse_inds.hpp:38-52)
template <int LOG2L, int K>
__device__ __forceinline__ double fw_width(const Params& P, const FwParams& F, const Mask<K>& m, int lig, int gbase) {
  if (P.sparse == nullptr) return fw_width_simple<LOG2L, K>(F, m, gbase);
  const Mask<K> s = fw_sparse_mask<LOG2L, K>(P, lig);
  const double w1 = fw_width_simple<LOG2L, K>(F, mandn<K>(m, s), gbase);
  const double w2 = fw_width_simple<LOG2L, K>(F, mand<K>(m, s), gbase);
  const double mn = (w2 < F.log2np) ? w2 : fw_wr(F, F.log2np);
  return fw_wr(F, w1 + mn);
}

}  // namespace tnco
