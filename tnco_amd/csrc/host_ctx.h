// host_ctx.h -- host-side context of a handle and the per-(LOG2L, K) launchers (internal).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <vector>

#include "sa_kernels.h"
#include "sa_sweep.h"
#include "fw_kernels.h"
#include "sa_small.h"

struct EventPair {
  hipEvent_t a, b;
  int kind;  // TNCO_KIND_*
};
enum : int { TNCO_KIND_SWEEP = 0, TNCO_KIND_FW_MOVE = 1, TNCO_KIND_FW_RESLICE = 2, TNCO_KIND_FW_WALK = 3, TNCO_KINDS = 4 };


struct tnco_hip_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  tnco::Params P{};
  int log2l = 2, K = 1, L = 4;  // lanes per replica = L, mask words per lane = K
  bool hyper = false, generic = false;
  bool fw = false;  // finite-width optimizer
  // the re-slice by re-pricing (fw_delta_kernel) pays while few replicas fall back to the full rebuild: the
  // mode of a tnco_hip_run_fw call follows the fall-backs counted during the previous one
  bool fw_delta_capable = false, fw_delta_on = false;
  int64_t fw_delta_reslices = 0;  // re-slices launched in that mode since the count was read
  int fw_single_calls = 0;        // calls in the other mode since the last probe
  int fw_probe_wait = 4;          // ... before the next probe (doubles after a probe that failed)
  bool small_tree = false;  // infinite memory, fast cost path, <= 2 mask words, <= 128 leaves: LDS-resident sweeps
  tnco::FwParams F{};
  std::vector<void*> allocs;
  int64_t bytes = 0;
  std::vector<uint64_t> leafmask_w;  // [n][W]
  std::vector<uint64_t> outmask_w;   // [W]
  double* d_betas = nullptr;
  int64_t betas_cap = 0;
  std::vector<EventPair> pending, free_events;
  double kernel_ms = 0;   // all kernels of the run calls since the last reset
  int64_t launches = 0;   // chunks of the schedule launched (one per tnco_hip_run[_fw] call unless very long)
  double kind_ms[TNCO_KINDS] = {0, 0, 0, 0};     // the same time, per kernel (HIP events around every launch)
  int64_t kind_launches[TNCO_KINDS] = {0, 0, 0, 0};

  template <typename T>
  hipError_t alloc(T** p, int64_t count) {
    void* q = nullptr;
    int64_t nb = std::max<int64_t>(count, 1) * (int64_t)sizeof(T);
    hipError_t e = hipMalloc(&q, (size_t)nb);
    if (e == hipSuccess) {
      allocs.push_back(q);
      bytes += nb;
      *p = (T*)q;
    }
    return e;
  }
  // HIP events on the handle's stream around one kernel launch
  template <typename F>
  hipError_t timed(int kind, F&& launch) {
    EventPair ev;
    if (!free_events.empty()) {
      ev = free_events.back();
      free_events.pop_back();
    } else {
      hipError_t e = hipEventCreate(&ev.a);
      if (e != hipSuccess) return e;
      e = hipEventCreate(&ev.b);
      if (e != hipSuccess) return e;
    }
    ev.kind = kind;
    hipError_t e = hipEventRecord(ev.a, stream);
    if (e != hipSuccess) return e;
    launch();
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipEventRecord(ev.b, stream);
    if (e != hipSuccess) return e;
    pending.push_back(ev);
    if (pending.size() > 1024) resolve_events();
    return hipSuccess;
  }
  void reset_times() {
    kernel_ms = 0;
    launches = 0;
    for (int k = 0; k < TNCO_KINDS; ++k) { kind_ms[k] = 0; kind_launches[k] = 0; }
  }
  void resolve_events() {
    for (auto& ev : pending) {
      float ms = 0;
      if (hipEventSynchronize(ev.b) == hipSuccess && hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
        kernel_ms += ms;
        kind_ms[ev.kind] += ms;
        kind_launches[ev.kind]++;
      }
      free_events.push_back(ev);
    }
    pending.clear();
  }
  ~tnco_hip_ctx() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    resolve_events();
    for (auto& ev : free_events) {
      (void)hipEventDestroy(ev.a);
      (void)hipEventDestroy(ev.b);
    }
    for (void* p : allocs) (void)hipFree(p);
    if (d_betas) (void)hipFree(d_betas);
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }
  int64_t block_bytes() const { return (int64_t)(P.n - 1) * P.BS; }
};


// Launchers of the kernels of one (LOG2L, K) pair; defined in launch_impl.h, instantiated once per
// pair in inst_<LOG2L>_<K>.hip so that the pairs compile in parallel.
template <int LOG2L, int K>
void launch_run_lk(tnco_hip_ctx* h, const double* betas, int64_t n_steps, int prob_kind);
template <int LOG2L, int K>
void launch_build_lk(tnco_hip_ctx* h, const tnco::BuildArgs& a);
template <int LOG2L, int K>
void launch_compare_lk(tnco_hip_ctx* h, const tnco::BuildArgs& a, double atol, int32_t* out_bad);
template <int LOG2L, int K>
void launch_fw_leaf_bits_lk(tnco_hip_ctx* h, uint32_t* bits, int32_t* any);
template <int LOG2L, int K>
void launch_fw_init_lk(tnco_hip_ctx* h, const tnco::FwInitArgs& a);
template <int LOG2L, int K>
void launch_fw_check_lk(tnco_hip_ctx* h, const tnco::BuildArgs& a, int which_min, double atol, int32_t* out_bad);
template <int LOG2L, int K>
void launch_fw_move_lk(tnco_hip_ctx* h, const double* betas, int64_t n_steps, int prob_kind, int tail_last);
template <int LOG2L, int K>
void launch_fw_reslice_lk(tnco_hip_ctx* h, int prewalked);
