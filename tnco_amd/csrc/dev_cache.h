// dev_cache.h -- device memory of destroyed handles kept for the next one.
//
// hipMalloc of a handle's buffers (12 GB for 65 536 replicas of the 512-leaf network) usually takes under a
// millisecond, but one create() in ten waited 1-2 s in it when the previous handle had just been freed
// (profiles/r04_e2e.txt), and hipFree of them is 6-11 ms of every destroy.  A job that builds optimizer after
// optimizer -- the components of a network, a parameter sweep -- asks for the same sizes again and again:
// blocks of >= 1 MB go to this cache on destroy and are handed out again on an exact size match.
//   * bounded: TNCO_HIP_CACHE_MB (default the smaller of an eighth of the device's memory and 32 GB -- two handles
//     of the benchmark's size; 0: no cache); when full the oldest blocks are freed;
//   * EVERY device allocation of the library goes through tnco::dev_malloc / DevCache::take: one that fails empties
//     the cache and is tried again, so the cache can never be the reason for an out-of-memory error of the library;
//   * other allocators in the process (PyTorch, RCCL) cannot do that: a process that shares the GPU with one calls
//     tnco_hip_release_cached() before handing the device over (include/tnco_hip.h says so at destroy).
// The cache object is never destroyed (at exit the HIP runtime may already be gone).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

namespace tnco {

class DevCache {
 public:
  static DevCache& get() {
    static DevCache* c = new DevCache();
    return *c;
  }
  hipError_t take(void** p, size_t bytes, int device) {
    if (bytes >= kMin) {
      std::lock_guard<std::mutex> lock(mu_);
      for (size_t i = blocks_.size(); i-- > 0;)
        if (blocks_[i].bytes == bytes && blocks_[i].device == device) {
          *p = blocks_[i].p;
          held_ -= bytes;
          blocks_.erase(blocks_.begin() + (long)i);
          return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      release_all();
      e = hipMalloc(p, bytes);
    }
    return e;
  }
  void give(void* p, size_t bytes, int device) {
    if (!p) return;
    {
      std::lock_guard<std::mutex> lock(mu_);
      const size_t cap = capacity(device);
      if (bytes >= kMin && bytes <= cap) {
        while (held_ + bytes > cap && !blocks_.empty()) {  // the oldest go first
          (void)hipFree(blocks_.front().p);
          held_ -= blocks_.front().bytes;
          blocks_.erase(blocks_.begin());
        }
        blocks_.push_back(Block{p, bytes, device});
        held_ += bytes;
        return;
      }
    }
    (void)hipFree(p);
  }
  void release_all() {
    std::lock_guard<std::mutex> lock(mu_);
    for (auto& b : blocks_) (void)hipFree(b.p);
    blocks_.clear();
    held_ = 0;
  }
  size_t held() {
    std::lock_guard<std::mutex> lock(mu_);
    return held_;
  }

 private:
  struct Block {
    void* p;
    size_t bytes;
    int device;
  };
  static constexpr size_t kMin = (size_t)1 << 20;
  size_t capacity(int device) {
    if (cap_ == (size_t)-1) {
      if (const char* e = std::getenv("TNCO_HIP_CACHE_MB")) {
        cap_ = (size_t)std::max<long long>(0, std::atoll(e)) << 20;
      } else {
        size_t free_b = 0, total_b = 0;
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        cap_ = hipMemGetInfo(&free_b, &total_b) == hipSuccess ? std::min<size_t>(total_b / 8, (size_t)32 << 30) : 0;
        if (cur != device && cur >= 0) (void)hipSetDevice(cur);
      }
    }
    return cap_;
  }
  std::mutex mu_;
  std::vector<Block> blocks_;
  size_t held_ = 0, cap_ = (size_t)-1;
};

// hipMalloc for everything that does not go through DevCache::take (blocks that are never handed back to the cache):
// what the cache holds is given up before the allocation is reported as failed.
inline hipError_t dev_malloc(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess && DevCache::get().held() > 0) {
    (void)hipGetLastError();
    DevCache::get().release_all();
    e = hipMalloc(p, bytes);
  }
  return e;
}

// Streams of destroyed handles, for the next one: hipStreamCreateWithFlags takes ~2 ms, a handle has three (6 of the
// 20 ms of a create()).  Idle (synchronised by the handle's destructor) non-blocking streams, per device; at most 16 kept.
class StreamCache {
 public:
  static StreamCache& get() {
    static StreamCache* c = new StreamCache();
    return *c;
  }
  hipError_t take(hipStream_t* s, int device) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      for (size_t i = 0; i < idle_.size(); ++i)
        if (idle_[i].second == device) {
          *s = idle_[i].first;
          idle_.erase(idle_.begin() + (long)i);
          return hipSuccess;
        }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  }
  void give(hipStream_t s, int device) {
    if (!s) return;
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (idle_.size() < 16) {
        idle_.emplace_back(s, device);
        return;
      }
    }
    (void)hipStreamDestroy(s);
  }
  void release_all() {
    std::lock_guard<std::mutex> lock(mu_);
    for (auto& e : idle_) (void)hipStreamDestroy(e.first);
    idle_.clear();
  }

 private:
  std::mutex mu_;
  std::vector<std::pair<hipStream_t, int>> idle_;
};

}  // namespace tnco
