// launch_impl.h -- definitions of the per-(LOG2L, K) launchers + their explicit instantiation for
// the pair named by TNCO_INST_L / TNCO_INST_K (see inst_*.hip).
#pragma once
#include "host_ctx.h"

#include <cstdio>
#include <cstdlib>

using namespace tnco;

// The sweep kernel this handle runs (infinite memory), or nullptr for the LDS-resident small-tree kernel.
template <int LOG2L, int K>
static const void* run_kernel_ptr(const tnco_hip_ctx* h) {
  if (h->fw) {  // (finite width: the staged moves; the spread form is asked about its own occupancy)
    if (h->run_seats > 0) {
      if (h->hyper)
        return h->generic ? (const void*)fw_staged_kernel<LOG2L, K, true, true, true>() : (const void*)fw_staged_kernel<LOG2L, K, true, false, true>();
      return h->generic ? (const void*)fw_staged_kernel<LOG2L, K, false, true, true>() : (const void*)fw_staged_kernel<LOG2L, K, false, false, true>();
    }
    if (h->hyper)
      return h->generic ? (const void*)fw_staged_kernel<LOG2L, K, true, true, false>() : (const void*)fw_staged_kernel<LOG2L, K, true, false, false>();
    return h->generic ? (const void*)fw_staged_kernel<LOG2L, K, false, true, false>() : (const void*)fw_staged_kernel<LOG2L, K, false, false, false>();
  }
  if (h->run_seats > 0) {  // (a spread batch: sa_sweep.h, SPREAD)
    if (h->hyper)
      return h->generic ? (const void*)sa_run_kernel<LOG2L, K, true, true, false, true> : (const void*)sa_run_kernel<LOG2L, K, true, false, false, true>;
    return h->generic ? (const void*)sa_run_kernel<LOG2L, K, false, true, false, true> : (const void*)sa_run_kernel<LOG2L, K, false, false, false, true>;
  }
  if (h->hyper)
    return h->generic ? (const void*)sa_run_kernel<LOG2L, K, true, true, false> : (const void*)sa_run_kernel<LOG2L, K, true, false, false>;
  return h->generic ? (const void*)sa_run_kernel<LOG2L, K, false, true, false> : (const void*)sa_run_kernel<LOG2L, K, false, false, false>;
}

// Blocks of that kernel one CU holds at a time (its register budget decides: 3 at 512 leaves).
template <int LOG2L, int K>
int run_blocks_per_cu_lk(tnco_hip_ctx* h) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, run_kernel_ptr<LOG2L, K>(h), SWT, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return nb;
}

// n_steps sweeps of the replicas of blocks [block0, block0 + nblocks) on stream s (nblocks < 0: all of them).
template <int LOG2L, int K>
void launch_run_lk(tnco_hip_ctx* h, const double* betas, int64_t n_steps, int prob_kind, hipStream_t s, int block0,
                   int nblocks) {
  const Params& P = h->P;
  if constexpr (LOG2L == 2 && K == 1) {
    if (h->small_tree) {  // (one launch for all the replicas: such a handle has one group)
      const int seats = h->small_seats;
      const dim3 sgrid((unsigned)((P.R + seats - 1) / seats));
      if (P.n - 1 <= 63)
        hipLaunchKernelGGL((sa_small_kernel<63, SMALL_TPB>), sgrid, dim3(SMALL_TPB), 0, s, P, betas, n_steps, prob_kind, seats);
      else
        hipLaunchKernelGGL((sa_small_kernel<127, SMALL_TPB>), sgrid, dim3(SMALL_TPB), 0, s, P, betas, n_steps, prob_kind, seats);
      return;
    }
  }
  if constexpr (LOG2L == 2) {
    if (h->lds_tree) {
      const LdsPlan& pl = h->lds_plan;
      const void* kern = h->hyper ? (const void*)sa_lds_kernel<K, true> : (const void*)sa_lds_kernel<K, false>;
      (void)kern;  // (its dynamic-LDS limit was raised at create: lds_kernel_prepare_lk)
      const dim3 lgrid((unsigned)((P.R + pl.seats - 1) / pl.seats));
      if (h->hyper)
        hipLaunchKernelGGL((sa_lds_kernel<K, true>), lgrid, dim3(SMALL_TPB), (size_t)pl.total, s, P, betas, n_steps, prob_kind, pl, h->leaf_idx);
      else
        hipLaunchKernelGGL((sa_lds_kernel<K, false>), lgrid, dim3(SMALL_TPB), (size_t)pl.total, s, P, betas, n_steps, prob_kind, pl, h->leaf_idx);
      return;
    }
  }
  if (h->run_seats > 0) {  // a batch spread over the wavefront slots: run_seats replicas per wavefront, one launch
    const int per_block = (SWT / 64) * h->run_seats;
    const dim3 sgrid((unsigned)((P.R + per_block - 1) / per_block));
    if (h->hyper) {
      if (h->generic)
        hipLaunchKernelGGL((sa_run_kernel<LOG2L, K, true, true, false, true>), sgrid, dim3(SWT), 0, s, P, betas, n_steps, prob_kind, FwParams{}, 1, h->run_seats);
      else
        hipLaunchKernelGGL((sa_run_kernel<LOG2L, K, true, false, false, true>), sgrid, dim3(SWT), 0, s, P, betas, n_steps, prob_kind, FwParams{}, 1, h->run_seats);
    } else {
      if (h->generic)
        hipLaunchKernelGGL((sa_run_kernel<LOG2L, K, false, true, false, true>), sgrid, dim3(SWT), 0, s, P, betas, n_steps, prob_kind, FwParams{}, 1, h->run_seats);
      else
        hipLaunchKernelGGL((sa_run_kernel<LOG2L, K, false, false, false, true>), sgrid, dim3(SWT), 0, s, P, betas, n_steps, prob_kind, FwParams{}, 1, h->run_seats);
    }
    return;
  }
  const int gpb = SWT >> LOG2L;
  dim3 grid((unsigned)(nblocks >= 0 ? nblocks : (P.R + gpb - 1) / gpb));
  if (h->hyper) {
    if (h->generic)
      hipLaunchKernelGGL((sa_run_kernel<LOG2L, K, true, true, false>), grid, dim3(SWT), 0, s, P, betas, n_steps, prob_kind, FwParams{}, 1, block0);
    else
      hipLaunchKernelGGL((sa_run_kernel<LOG2L, K, true, false, false>), grid, dim3(SWT), 0, s, P, betas, n_steps, prob_kind, FwParams{}, 1, block0);
  } else {
    if (h->generic)
      hipLaunchKernelGGL((sa_run_kernel<LOG2L, K, false, true, false>), grid, dim3(SWT), 0, s, P, betas, n_steps, prob_kind, FwParams{}, 1, block0);
    else
      hipLaunchKernelGGL((sa_run_kernel<LOG2L, K, false, false, false>), grid, dim3(SWT), 0, s, P, betas, n_steps, prob_kind, FwParams{}, 1, block0);
  }
}

// More dynamic LDS than the 64 KiB a kernel gets by default.  hipFuncAttributeMaxDynamicSharedMemorySize belongs to the
// FUNCTION, process-wide, not to a handle: it is raised to what the device allows (the same value from every handle, so
// two live handles with different plans cannot lower it under one another), at create, and a handle whose plan does not
// fit -- or a runtime that refuses -- runs the HBM kernel instead.  0 = ready.
template <int LOG2L, int K>
int lds_kernel_prepare_lk(tnco_hip_ctx* h, int device_lds_bytes) {
  if constexpr (LOG2L == 2) {
    const void* kern = h->hyper ? (const void*)sa_lds_kernel<K, true> : (const void*)sa_lds_kernel<K, false>;
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, kern) != hipSuccess) { (void)hipGetLastError(); return 1; }
    const int dyn = device_lds_bytes - (int)fa.sharedSizeBytes;
    if (h->lds_plan.total > dyn) return 1;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess) { (void)hipGetLastError(); return 1; }
    return 0;
  }
  return 1;
}

template <int LOG2L, int K>
void launch_build_lk(tnco_hip_ctx* h, const BuildArgs& a) {
  const int gpb = 256 >> LOG2L;
  dim3 grid((unsigned)((a.count + gpb - 1) / gpb));
  if (h->hyper)
    hipLaunchKernelGGL((build_kernel<LOG2L, K, true>), grid, dim3(256), 0, h->stream, h->P, a);
  else
    hipLaunchKernelGGL((build_kernel<LOG2L, K, false>), grid, dim3(256), 0, h->stream, h->P, a);
}

template <int LOG2L, int K>
void launch_compare_lk(tnco_hip_ctx* h, const BuildArgs& a, double atol, int32_t* out_bad) {
  const int gpb = 256 >> LOG2L;
  dim3 grid((unsigned)((a.count + gpb - 1) / gpb));
  if (h->hyper)
    hipLaunchKernelGGL((compare_kernel<LOG2L, K, true>), grid, dim3(256), 0, h->stream, h->P, a, atol, out_bad);
  else
    hipLaunchKernelGGL((compare_kernel<LOG2L, K, false>), grid, dim3(256), 0, h->stream, h->P, a, atol, out_bad);
}


template <int LOG2L, int K>
void launch_fw_leaf_bits_lk(tnco_hip_ctx* h, uint32_t* bits, int32_t* any) {
  hipLaunchKernelGGL((fw_leaf_bits_kernel<LOG2L, K>), dim3(1), dim3(256), 0, h->stream, h->P, h->F, bits, any);
}
template <int LOG2L, int K>
void launch_fw_init_lk(tnco_hip_ctx* h, const FwInitArgs& a) {
  const int gpb = 256 >> LOG2L;
  dim3 grid((unsigned)((h->P.R + gpb - 1) / gpb));
  if (h->hyper)
    hipLaunchKernelGGL((fw_init_kernel<LOG2L, K, true>), grid, dim3(256), 0, h->stream, h->P, h->F, a);
  else
    hipLaunchKernelGGL((fw_init_kernel<LOG2L, K, false>), grid, dim3(256), 0, h->stream, h->P, h->F, a);
}
template <int LOG2L, int K>
void launch_fw_check_lk(tnco_hip_ctx* h, const BuildArgs& a, int which_min, double atol, int32_t* out_bad) {
  const int gpb = 256 >> LOG2L;
  dim3 grid((unsigned)((a.count + gpb - 1) / gpb));
  if (h->hyper)
    hipLaunchKernelGGL((fw_check_kernel<LOG2L, K, true>), grid, dim3(256), 0, h->stream, h->P, h->F, a, which_min, atol, out_bad);
  else
    hipLaunchKernelGGL((fw_check_kernel<LOG2L, K, false>), grid, dim3(256), 0, h->stream, h->P, h->F, a, which_min, atol, out_bad);
}

template <int LOG2L, int K>
void launch_fw_move_lk(tnco_hip_ctx* h, const double* betas, int64_t n_steps, int prob_kind, int tail_last) {
  const int gpb = 256 >> LOG2L, gpb_staged = SWT >> LOG2L;
  dim3 grid((unsigned)((h->P.R + gpb - 1) / gpb)), grid_staged((unsigned)((h->P.R + gpb_staged - 1) / gpb_staged));
  const bool maxnew = h->F.max_new_slices > 0;
  if (!maxnew) {  // the staged state machine (sa_sweep.h, FW = true)
#define TNCO_FW_STAGED(HY, GE)                                                                                             \
  do {                                                                                                                     \
    if (h->run_seats > 0) /* a small batch: one replica per wavefront (sa_sweep.h, SPREAD) */                              \
      hipLaunchKernelGGL((fw_staged_kernel<LOG2L, K, HY, GE, true>()),                                                     \
                         dim3((unsigned)((h->P.R + (SWT / 64) * h->run_seats - 1) / ((SWT / 64) * h->run_seats))),         \
                         dim3(SWT), 0, h->stream, h->P, betas, n_steps, prob_kind, h->F, tail_last, h->run_seats);         \
    else                                                                                                                   \
      hipLaunchKernelGGL((fw_staged_kernel<LOG2L, K, HY, GE, false>()), grid_staged, dim3(SWT), 0, h->stream, h->P, betas, n_steps, \
                         prob_kind, h->F, tail_last, 0);                                                                   \
  } while (0)
    if (h->hyper) {
      if (h->generic) TNCO_FW_STAGED(true, true); else TNCO_FW_STAGED(true, false);
    } else {
      if (h->generic) TNCO_FW_STAGED(false, true); else TNCO_FW_STAGED(false, false);
    }
#undef TNCO_FW_STAGED
    return;
  }
#define TNCO_FW_MOVE(HY, MN)                                                                                        \
  hipLaunchKernelGGL((fw_move_kernel<LOG2L, K, HY, MN>), grid, dim3(256), 0, h->stream, h->P, h->F, betas, n_steps, \
                     prob_kind, tail_last)
  if (h->hyper) {
    if (maxnew) TNCO_FW_MOVE(true, true); else TNCO_FW_MOVE(true, false);
  } else {
    if (maxnew) TNCO_FW_MOVE(false, true); else TNCO_FW_MOVE(false, false);
  }
#undef TNCO_FW_MOVE
}
template <int LOG2L, int K>
void launch_fw_reslice_lk(tnco_hip_ctx* h, int prewalked) {
  const int gpb = 256 >> LOG2L;
  dim3 grid((unsigned)((h->P.R + gpb - 1) / gpb));
#ifndef TNCO_PROFILE
  if (prewalked == 3) {
    // the re-slice of a replica in one wavefront (order | get_slices | re-pricing), then the two clean-up launches:
    // get_slices of the replicas it has left alone (usually none: 5 us of an empty launch), the full rebuild of those
    // and of the ones it could not re-price + the end of the sweep for everybody
    // lanes per leg mask: the L * K words of this translation unit's networks fit 16, 32 or 64 lanes
    constexpr int LKW = (1 << LOG2L) * K, LT = LKW <= 16 ? 4 : (LKW <= 32 ? 5 : 6);
    const size_t lb = fww_lds_bytes(h->P.n, 1 << LT, h->hyper, h->fw_wave_big);
    const int need = (h->P.n - 1 + 63) / 64;
#define TNCO_FWW(JJ, HY, BG) hipLaunchKernelGGL((fw_wave_kernel<JJ, LT, HY, BG>), dim3((unsigned)h->P.R), dim3(64), lb, h->stream, h->P, h->F, h->fw_wave_cap, h->fw_wave_maxnp)
#define TNCO_FWW_J(HY, BG)                                                                                                \
    if (need <= 2) TNCO_FWW(2, HY, BG); else if (need <= 4) TNCO_FWW(4, HY, BG); else if (need <= 6) TNCO_FWW(6, HY, BG); \
    else if (need <= 9) TNCO_FWW(9, HY, BG); else if (need <= 12) TNCO_FWW(12, HY, BG); else if (need <= 16) TNCO_FWW(16, HY, BG); \
    else if (need <= 24) TNCO_FWW(24, HY, BG); else TNCO_FWW(32, HY, BG)
    if (h->hyper) {
      if (h->fw_wave_big) { TNCO_FWW_J(true, true); } else { TNCO_FWW_J(true, false); }
      hipLaunchKernelGGL((fw_reslice_a_kernel<LOG2L, K, true>), grid, dim3(256), 0, h->stream, h->P, h->F, 2);
      hipLaunchKernelGGL((fw_reslice_b_kernel<LOG2L, K, true>), grid, dim3(256), 0, h->stream, h->P, h->F, 1);
    } else {
      if (h->fw_wave_big) { TNCO_FWW_J(false, true); } else { TNCO_FWW_J(false, false); }
      hipLaunchKernelGGL((fw_reslice_a_kernel<LOG2L, K, false>), grid, dim3(256), 0, h->stream, h->P, h->F, 2);
      hipLaunchKernelGGL((fw_reslice_b_kernel<LOG2L, K, false>), grid, dim3(256), 0, h->stream, h->P, h->F, 1);
    }
#undef TNCO_FWW_J
#undef TNCO_FWW
#ifdef TNCO_FWW_PROF
    {
      static int wcalls = 0;
      if (++wcalls % 80 == 0) {
        unsigned long long st[24];
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_fww_prof), sizeof(st)) == hipSuccess && st[8]) {
          const double d = (double)st[8];
          std::fprintf(stderr, "fw_wave after %d launches: cycles per replica: loads %.0f, ordering %.0f, legs + counts %.0f, greedy pass %.0f, "
                       "change list %.0f, marks + prices %.0f, partial sums %.0f, commit %.0f\n", wcalls, st[0] / d, st[1] / d, st[2] / d, st[3] / d,
                       st[4] / d, st[5] / d, st[6] / d, st[7] / d);
          std::fprintf(stderr, "   greedy pass: scan %.0f, draws %.0f, select by count %.0f, candidate list %.0f, permutation %.0f, tie picks %.0f; per replica "
                       "%.1f too-wide tensors, %.1f still too wide, %.1f of them with a split tie group, %.1f candidate legs each\n", st[12] / d, st[13] / d,
                       st[14] / d, st[15] / d, st[16] / d, st[17] / d, st[21] / d, st[18] / d, st[19] / d, st[18] ? (double)st[20] / st[18] : 0.0);
        }
      }
    }
#endif
    return;
  }
#endif
  if (h->hyper)
    hipLaunchKernelGGL((fw_reslice_kernel<LOG2L, K, true>), grid, dim3(256), 0, h->stream, h->P, h->F, prewalked);
  else
    hipLaunchKernelGGL((fw_reslice_kernel<LOG2L, K, false>), grid, dim3(256), 0, h->stream, h->P, h->F, prewalked);
}

template void launch_run_lk<TNCO_INST_L, TNCO_INST_K>(tnco_hip_ctx*, const double*, int64_t, int, hipStream_t, int, int);
template int run_blocks_per_cu_lk<TNCO_INST_L, TNCO_INST_K>(tnco_hip_ctx*);
template int lds_kernel_prepare_lk<TNCO_INST_L, TNCO_INST_K>(tnco_hip_ctx*, int);
template void launch_build_lk<TNCO_INST_L, TNCO_INST_K>(tnco_hip_ctx*, const BuildArgs&);
template void launch_compare_lk<TNCO_INST_L, TNCO_INST_K>(tnco_hip_ctx*, const BuildArgs&, double, int32_t*);
template void launch_fw_leaf_bits_lk<TNCO_INST_L, TNCO_INST_K>(tnco_hip_ctx*, uint32_t*, int32_t*);
template void launch_fw_init_lk<TNCO_INST_L, TNCO_INST_K>(tnco_hip_ctx*, const FwInitArgs&);
template void launch_fw_check_lk<TNCO_INST_L, TNCO_INST_K>(tnco_hip_ctx*, const BuildArgs&, int, double, int32_t*);
template void launch_fw_move_lk<TNCO_INST_L, TNCO_INST_K>(tnco_hip_ctx*, const double*, int64_t, int, int);
template void launch_fw_reslice_lk<TNCO_INST_L, TNCO_INST_K>(tnco_hip_ctx*, int);
