// box_probe.hip -- what THIS box's memory system gives the access pattern of the sweep kernels, measured inside the
// bench process (bench.py `calibration`; not part of libtnco_hip.so, not on the product path).
//
// The sweep kernels are bound by random fabric requests, and fresh boxes of the pool differ by ~10 % in what they
// retire.  Three rates, each a few tens of milliseconds on an 8 GiB working set (nothing of it fits the L2s):
//   [0] random 128-byte lines read, groups of 4 lanes, 4 lines in flight per group      (lines/s)
//   [1] the memory side of one infinite-memory move and nothing else: two random lines read, one header sector
//       written, on 3 of 4 moves the rest of the line and two 4-byte parent words       (moves/s)
//   [2] a streaming read of the same buffer                                              (bytes/s)
//   [3] VALU wave-instructions per second and SIMD under a full load (the clock the chip holds, / 4)
//   [4] seconds per dependent load of a cold line on an idle chip
// bench.py divides its request rate by [0] and its move rate by [1] of the box it runs on.
//
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/libbox_probe.so tools/box_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

constexpr int IN_FLIGHT = 4;

__global__ __launch_bounds__(256) void lines_read_kernel(const uint8_t* buf, uint64_t n_lines, int iters, uint64_t* sink) {
  const int lane = threadIdx.x & 3;
  const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  uint64_t acc = 0;
  for (int it = 0; it < iters; it += IN_FLIGHT) {
    uint64_t v[IN_FLIGHT][4];
#pragma unroll
    for (int u = 0; u < IN_FLIGHT; ++u) {
      const uint64_t line = mix(group * 0x9e3779b97f4a7c15ull + (uint64_t)(it + u)) % n_lines;
      const uint64_t* p = reinterpret_cast<const uint64_t*>(buf + line * 128 + lane * 32);
#pragma unroll
      for (int w = 0; w < 4; ++w) v[u][w] = p[w];
    }
#pragma unroll
    for (int u = 0; u < IN_FLIGHT; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
  }
  if (acc == 0x1234567) sink[0] = acc;
}

__global__ __launch_bounds__(256) void move_pattern_kernel(uint8_t* buf, uint64_t n_lines, int iters, uint64_t* sink) {
  const int lane = threadIdx.x & 3;
  const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  uint64_t acc = 0;
  for (int it = 0; it < iters; it += IN_FLIGHT) {
    uint64_t head[IN_FLIGHT], legs[IN_FLIGHT][4], key[IN_FLIGHT];
#pragma unroll
    for (int u = 0; u < IN_FLIGHT; ++u) {
      key[u] = mix(group * 0x9e3779b97f4a7c15ull + (uint64_t)(it + u));
      const uint64_t a = key[u] % n_lines, c = mix(key[u] + 1) % n_lines;
      head[u] = *reinterpret_cast<const uint64_t*>(buf + a * 128 + lane * 8);
#pragma unroll
      for (int w = 0; w < 4; ++w) legs[u][w] = *reinterpret_cast<const uint64_t*>(buf + c * 128 + w * 32 + lane * 8);
    }
#pragma unroll
    for (int u = 0; u < IN_FLIGHT; ++u) {
      const uint64_t b = mix(key[u] + 2) % n_lines, p0 = mix(key[u] + 3) % n_lines, p1 = mix(key[u] + 4) % n_lines;
      const uint64_t s = head[u] + legs[u][0] + legs[u][1] + legs[u][2] + legs[u][3];
      acc += s;
      *reinterpret_cast<uint64_t*>(buf + b * 128 + lane * 8) = s;
      if (((it + u) & 3) < 3) {  // an accepted move
#pragma unroll
        for (int w = 1; w < 4; ++w) *reinterpret_cast<uint64_t*>(buf + b * 128 + w * 32 + lane * 8) = s + w;
        if (lane == 0) {
          *reinterpret_cast<uint32_t*>(buf + p0 * 128 + 8) = (uint32_t)s;
          *reinterpret_cast<uint32_t*>(buf + p1 * 128 + 8) = (uint32_t)s + 1;
        }
      }
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

__global__ __launch_bounds__(256) void stream_read_kernel(const uint4* buf, uint64_t n_vec, uint64_t* sink) {
  uint64_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = buf[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 0x1234567) sink[0] = acc;
}

// [3] the clock the chip holds under a full VALU load: every wavefront runs shift-adds (8 registers per lane, one
// v_lshl_add_u32 each per round), 8 wavefronts per SIMD -- wave-instructions retired per second and SIMD = clock / 4 when the VALU is
// saturated (a wave64 instruction holds its SIMD for four clocks).
__global__ __launch_bounds__(256) void valu_kernel(uint32_t* sink, int iters, uint32_t seed) {
  uint32_t a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = seed + threadIdx.x * 8 + k;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (a[k] << 1) + a[(k + 1) & 7];  // (one v_lshl_add_u32 each)
  }
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s ^= a[k];
  if (s == 0x12345u) sink[0] = s;
}

// [4] the latency of a dependent load from cold memory on an otherwise idle chip: ONE lane chases pointers through the
// buffer (every hop a random 128-byte line of 8 GiB).
__global__ void chase_kernel(const uint64_t* buf, uint64_t n_lines, int hops, uint64_t* sink) {
  uint64_t x = 12345;
  for (int h = 0; h < hops; ++h) x = mix(x + buf[(x % n_lines) * 16]);
  sink[1] = x;
}

}  // namespace

// out[0..4] as in the header ([3] VALU wave-instructions / s / SIMD, [4] seconds per dependent load); returns 0, or a hipError_t.
extern "C" int box_probe(int device, double gib, double* out) {
#define TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { if (buf) (void)hipFree(buf); if (sink) (void)hipFree(sink); return (int)e_; } } while (0)
  uint8_t* buf = nullptr;
  uint64_t* sink = nullptr;
  TRY(hipSetDevice(device));
  const size_t bytes = (size_t)(gib * (double)(1ull << 30)) / 128 * 128;
  TRY(hipMalloc(&buf, bytes));
  TRY(hipMalloc(&sink, 64));
  TRY(hipMemset(buf, 1, bytes));
  hipEvent_t a, b;
  TRY(hipEventCreate(&a));
  TRY(hipEventCreate(&b));
  hipDeviceProp_t prop;
  TRY(hipGetDeviceProperties(&prop, device));
  const int blocks = prop.multiProcessorCount * 8;  // 8 wavefronts per SIMD
  const uint64_t n_lines = bytes / 128;
  float ms = 0;
  {  // [3], [4]
    const int simds = prop.multiProcessorCount * 4, iters = 1 << 15;
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
      TRY(hipEventRecord(a));
      valu_kernel<<<prop.multiProcessorCount * 8, 256>>>(reinterpret_cast<uint32_t*>(sink), iters, (uint32_t)rep);
      TRY(hipEventRecord(b));
      TRY(hipEventSynchronize(b));
      TRY(hipEventElapsedTime(&ms, a, b));
      // wave-instructions: blocks x 4 wavefronts x iters x 8 shift-adds
      const double rate = (double)prop.multiProcessorCount * 8 * 4 * (double)iters * 8.0 / ((double)ms * 1e-3) / simds;
      if (rep > 0 && rate > best) best = rate;
    }
    out[3] = best;
    const int hops = 4096;
    TRY(hipEventRecord(a));
    chase_kernel<<<1, 1>>>(reinterpret_cast<const uint64_t*>(buf), n_lines, hops, sink);
    TRY(hipEventRecord(b));
    TRY(hipEventSynchronize(b));
    TRY(hipEventElapsedTime(&ms, a, b));
    out[4] = (double)ms * 1e-3 / hops;
  }
  for (int which = 0; which < 3; ++which) {
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {  // (the first repetition also warms the clocks up)
      const int iters = which == 0 ? 8192 : 2048;
      TRY(hipEventRecord(a));
      if (which == 0) lines_read_kernel<<<blocks, 256>>>(buf, n_lines, iters, sink);
      else if (which == 1) move_pattern_kernel<<<blocks, 256>>>(buf, n_lines, iters, sink);
      else stream_read_kernel<<<blocks * 4, 256>>>(reinterpret_cast<const uint4*>(buf), bytes / 16, sink);
      TRY(hipEventRecord(b));
      TRY(hipEventSynchronize(b));
      TRY(hipEventElapsedTime(&ms, a, b));
      const double units = which == 2 ? (double)bytes : (double)blocks * 64.0 * iters;
      const double rate = units / ((double)ms * 1e-3);
      if (rep > 0 && rate > best) best = rate;
    }
    out[which] = best;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(buf);
  (void)hipFree(sink);
  return 0;
#undef TRY
}
