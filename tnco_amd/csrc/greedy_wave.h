// greedy_wave.h -- constants and wave-level helpers of the initial-tree kernels (DPP reductions, scans, LDS pointer types)
// (part of greedy_device.hip, the only file that includes it: everything lives in its unnamed namespace)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tnco {
namespace {

constexpr int GREEDY_MAXH = 6;
constexpr int GREEDY_LCAP = 256;  // neighbours of one tensor the kernel handles (more: the tree goes to the host)

// -DTNCO_GREEDY_PROF: shader-clock ticks per section of greedy_kernel, summed per wavefront (diagnostic build)
#ifdef TNCO_GREEDY_PROF
#define GP_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_[i] += t_ - pt_; pt_ = t_; } while (0)
#else
#define GP_T(i)
#endif
constexpr uint32_t DEAD = 0xFFFFu;
constexpr uint64_t KMAX = ~0ull;

__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int o) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int lane) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, lane), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}
// wave reductions: four DPP steps inside each row of 16 lanes (xor 1, xor 2, half mirror, mirror), then
// the four row results through readlane -- no LDS crossbar round trips
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp64(uint64_t v) {
  return ((uint64_t)dpp<CTRL>((uint32_t)(v >> 32)) << 32) | dpp<CTRL>((uint32_t)v);
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, int lane) {
  return ((uint64_t)rdlane((uint32_t)(v >> 32), lane) << 32) | rdlane((uint32_t)v, lane);
}
// (every DPP read is evaluated ONCE, with all lanes active: a read from a lane that a branch has switched
//  off returns nothing)
template <int CTRL>
__device__ __forceinline__ uint64_t min_step(uint64_t v) {
  const uint64_t t = dpp64<CTRL>(v);
  return t < v ? t : v;
}
__device__ __forceinline__ uint32_t wsum(uint32_t v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return rdlane(v, 0) + rdlane(v, 16) + rdlane(v, 32) + rdlane(v, 48);
}
__device__ __forceinline__ uint64_t wmin64(uint64_t v) {
  v = min_step<0xB1>(v);
  v = min_step<0x4E>(v);
  v = min_step<0x141>(v);
  v = min_step<0x140>(v);
  const uint64_t a = rdlane64(v, 0), b = rdlane64(v, 16), c = rdlane64(v, 32), d = rdlane64(v, 48);
  const uint64_t ab = b < a ? b : a, cd = d < c ? d : c;
  return cd < ab ? cd : ab;
}
__device__ __forceinline__ uint64_t wxor64(uint64_t v) {
  v ^= dpp64<0xB1>(v);
  v ^= dpp64<0x4E>(v);
  v ^= dpp64<0x141>(v);
  v ^= dpp64<0x140>(v);
  return rdlane64(v, 0) ^ rdlane64(v, 16) ^ rdlane64(v, 32) ^ rdlane64(v, 48);
}
// exclusive prefix sum over the lanes
__device__ __forceinline__ uint32_t wscan_excl(uint32_t v, int lane) {
  uint32_t s = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)s, o);
    if (lane >= o) s += t;
  }
  return s - v;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS traffic of ONE wavefront needs no barrier: the LDS unit takes a wavefront's instructions in order.  (A
// __syncthreads() would also wait for the links on their way to memory -- a microsecond per contraction.)
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

typedef __attribute__((address_space(3))) volatile uint64_t* lds_u64;
typedef __attribute__((address_space(3))) volatile uint16_t* lds_u16;
typedef __attribute__((address_space(3))) volatile int16_t* lds_i16;


}  // namespace
}  // namespace tnco
