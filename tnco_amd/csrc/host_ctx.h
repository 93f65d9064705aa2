// host_ctx.h -- host-side context of a handle and the per-(LOG2L, K) launchers (internal).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "dev_cache.h"
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
  // the re-slice of a replica in one wavefront (fw_wave_kernel, the cost cache re-priced) pays while few replicas fall
  // back to the full rebuild: the form of a tnco_hip_run_fw call follows the fall-backs counted during the previous one
  bool fw_wave_capable = false, fw_wave_on = false;
  int fw_wave_maxnp = 128;  // ... candidate legs of one tensor it handles (test knob TNCO_HIP_FWS_MAXNP)
  int fw_wave_cap = 0;      // ... too-wide tensors whose legs it keeps in LDS (fww_cap; test knob TNCO_HIP_FWS_CAP)
  int fw_wave_lanes = 16;   // ... lanes per leg mask
  bool fw_wave_big = false, fw_wave_big_ok = false;  // ... its roomier configuration (fw_kernels.h, BIG): in use / its LDS fits
  void set_wave_config() {
    fw_wave_cap = tnco::fww_cap(P.n, fw_wave_lanes, hyper, fw_wave_big);
    if (const char* e = std::getenv("TNCO_HIP_FWS_CAP")) fw_wave_cap = std::max(0, std::min(fw_wave_cap, std::atoi(e))) & ~7;  // (test knob: legs from memory)
    fw_wave_maxnp = fw_wave_big ? 512 : 128;
    if (const char* e = std::getenv("TNCO_HIP_FWS_MAXNP")) fw_wave_maxnp = std::max(0, std::min(fw_wave_maxnp, std::atoi(e)));
  }
  int64_t fw_wave_reslices = 0;  // re-slices launched in that form since the count was read
  int fw_single_calls = 0;        // calls in the other mode since the last probe
  int fw_probe_wait = 4;          // ... before the next probe (doubles after a probe that failed)
  // tnco_hip_diag_fw_stats: [0] replica re-slices launched in the re-pricing form, [1] of those left to the full rebuild,
  // [2..4] why (FwParams::slowstat[1..3]), [5] replica re-slices launched in the walk + full-rebuild form
  int64_t fw_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool fw_probed = false;  // the first re-slice interval of the handle has run on its own (tnco_hip_run_fw)
  unsigned long long fw_slow_pending = 0, fw_slow_wide_pending = 0;  // fall-backs collected since tnco_hip_run_fw last chose a form
  bool small_tree = false;  // small trees: LDS-resident sweeps (sa_small.h, sa_small_kernel)
  int small_seats = 16;     // ... replicas per wavefront (a batch smaller than the chip's wavefront slots is spread)
  int run_seats = 0;        // > 0: the HBM sweep kernel's SPREAD form, that many replicas per wavefront (a batch smaller than the wavefront slots)
  bool lds_tree = false;    // any tree whose replicas fit the CUs' LDS in two rounds: sa_lds_kernel with the plan below
  tnco::LdsPlan lds_plan{};
  uint64_t* leaf_idx = nullptr;  // [n][lds_plan.leaf_stride] the leaves' index positions (16 bits each), device
  tnco::FwParams F{};
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;  // (destroy hands the blocks to tnco::DevCache)
  int64_t bytes = 0;
  std::vector<uint64_t> leafmask_w;  // [n][W]
  std::vector<uint64_t> outmask_w;   // [W]
  double* d_betas = nullptr;
  int64_t betas_cap = 0;
  // Infinite memory: tnco_hip_run enqueues without waiting for the previous call.  The betas of a call go
  // through a pinned host ring into a device ring (a region is re-used only after a wrap, which waits).
  double *beta_pin = nullptr, *beta_ring = nullptr;
  int64_t ring_cap = 0, ring_pos = 0;
  // ... and a handle whose replicas do not fill whole rounds of resident blocks (65536 replicas at 512
  // leaves: 1024 blocks for 768 slots) splits every step over n_groups streams, half of the blocks each:
  // a block that ends then frees its slot for the OTHER stream's pending launch, and the chip stays full
  // from step to step instead of running the last third of every launch on a third of its CUs
  // (tools/overlap_probe.py: +10 %).  Group kernels are ordered after the main stream's work at the fork
  // (event), every other entry point joins them back first (join_groups).
  static constexpr int MAX_GROUPS = 4;
  int n_groups = 1;
  int run_slots = 0;  // resident blocks of the sweep kernel on this device
  hipStream_t gstream[MAX_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t gfork = nullptr, gjoin[MAX_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
  bool groups_dirty = false;
  // device time of grouped steps: from the fork of the first call after a reset to the join that reads it
  hipEvent_t region_a = nullptr, region_b = nullptr;
  bool region_open = false, region_b_set = false;
  int64_t region_calls = 0;
  std::vector<EventPair> pending, free_events;
  double kernel_ms = 0;   // all kernels of the run calls since the last reset
  int64_t launches = 0;   // chunks of the schedule launched (one per tnco_hip_run[_fw] call unless very long)
  double kind_ms[TNCO_KINDS] = {0, 0, 0, 0};     // the same time, per kernel (HIP events around every launch)
  int64_t kind_launches[TNCO_KINDS] = {0, 0, 0, 0};

  // the device's fall-back counters since the last look -> fw_stats; *slow = their total (may be NULL)
  hipError_t collect_fw_stats(unsigned long long* slow, unsigned long long* slow_wide = nullptr) {
    unsigned long long c[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpy(c, F.slowstat, 32, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(F.slowstat, 0, 32, stream);
    if (e != hipSuccess) return e;
    for (int i = 0; i < 4; ++i) fw_stats[1 + i] += (int64_t)c[i];
    fw_slow_pending += c[0];
    fw_slow_wide_pending += c[1];
    if (slow) {  // (the caller that decides the next call's form: everything since it last asked, whoever collected it)
      *slow = fw_slow_pending;
      fw_slow_pending = 0;
      if (slow_wide) *slow_wide = fw_slow_wide_pending;
      fw_slow_wide_pending = 0;
    }
    return hipSuccess;
  }
  template <typename T>
  hipError_t alloc(T** p, int64_t count) {
    void* q = nullptr;
    int64_t nb = std::max<int64_t>(count, 1) * (int64_t)sizeof(T);
    hipError_t e = tnco::DevCache::get().take(&q, (size_t)nb, device);
    if (e == hipSuccess) {
      allocs.push_back(q);
      alloc_bytes.push_back((size_t)nb);
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
  // Everything the group streams hold becomes work the main stream waits for.
  hipError_t join_groups() {
    if (!groups_dirty) return hipSuccess;
    for (int q = 0; q < n_groups; ++q) {
      hipError_t e = hipEventRecord(gjoin[q], gstream[q]);
      if (e != hipSuccess) return e;
      e = hipStreamWaitEvent(stream, gjoin[q], 0);
      if (e != hipSuccess) return e;
    }
    groups_dirty = false;
    if (region_open) {  // the device time of the grouped steps ends here
      hipError_t e = hipEventRecord(region_b, stream);
      if (e != hipSuccess) return e;
      region_b_set = true;
    }
    return hipSuccess;
  }
  hipError_t sync_all() {
    hipError_t e = join_groups();
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(stream);
  }
  // closes the timing region of grouped steps (called with the groups joined and the stream idle)
  void close_region() {
    if (!region_open) return;
    float ms = 0;
    if ((region_b_set || hipEventRecord(region_b, stream) == hipSuccess) && hipEventSynchronize(region_b) == hipSuccess &&
        hipEventElapsedTime(&ms, region_a, region_b) == hipSuccess) {
      kernel_ms += ms;
      if (!fw) {  // (the only kernel of an infinite-memory handle; a finite-width one times its kernels on their streams)
        kind_ms[TNCO_KIND_SWEEP] += ms;
        kind_launches[TNCO_KIND_SWEEP] += region_calls;
      }
    }
    region_open = false;
    region_b_set = false;
    region_calls = 0;
  }
  void reset_times() {
    region_open = false;
    region_b_set = false;
    region_calls = 0;
    kernel_ms = 0;
    launches = 0;
    for (int k = 0; k < TNCO_KINDS; ++k) { kind_ms[k] = 0; kind_launches[k] = 0; }
  }
  void resolve_events() {
    for (auto& ev : pending) {
      float ms = 0;
      if (hipEventSynchronize(ev.b) == hipSuccess && hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
        if (n_groups <= 1) kernel_ms += ms;  // (grouped: concurrent kernels -- the total is the region's, close_region)
        kind_ms[ev.kind] += ms;
        kind_launches[ev.kind]++;
      }
      free_events.push_back(ev);
    }
    pending.clear();
  }
  ~tnco_hip_ctx() {
    (void)hipSetDevice(device);
    for (int q = 0; q < MAX_GROUPS; ++q)
      if (gstream[q]) (void)hipStreamSynchronize(gstream[q]);
    if (stream) (void)hipStreamSynchronize(stream);
    resolve_events();
    for (int q = 0; q < MAX_GROUPS; ++q) {
      if (gstream[q]) tnco::StreamCache::get().give(gstream[q], device);
      if (gjoin[q]) (void)hipEventDestroy(gjoin[q]);
    }
    if (gfork) (void)hipEventDestroy(gfork);
    if (region_a) (void)hipEventDestroy(region_a);
    if (region_b) (void)hipEventDestroy(region_b);
    if (beta_pin) (void)hipHostFree(beta_pin);
    if (beta_ring) (void)hipFree(beta_ring);
    for (auto& ev : free_events) {
      (void)hipEventDestroy(ev.a);
      (void)hipEventDestroy(ev.b);
    }
    for (size_t i = 0; i < allocs.size(); ++i) tnco::DevCache::get().give(allocs[i], alloc_bytes[i], device);
    if (d_betas) (void)hipFree(d_betas);
    if (own_stream) tnco::StreamCache::get().give(own_stream, device);
  }
  int64_t block_bytes() const { return P.RB; }
};


// Launchers of the kernels of one (LOG2L, K) pair; defined in launch_impl.h, instantiated once per
// pair in inst_<LOG2L>_<K>.hip so that the pairs compile in parallel.
template <int LOG2L, int K>
void launch_run_lk(tnco_hip_ctx* h, const double* betas, int64_t n_steps, int prob_kind, hipStream_t s, int block0, int nblocks);
template <int LOG2L, int K>
int run_blocks_per_cu_lk(tnco_hip_ctx* h);
template <int LOG2L, int K>
int lds_kernel_prepare_lk(tnco_hip_ctx* h, int device_lds_bytes);
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
