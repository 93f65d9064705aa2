// host_paths.cpp -- ContractionTree.path() (tnco/ctree.py:350-388) for a batch of trees (host).
//
// Input: get_contraction triples (child0, child1, node) in post-order, node ids local to the
// component (leaves 0..nc-1, internal nodes nc..2nc-2).  Rescaled as the reference does
// (leaf -> the component's tensor position among ALL tensors, internal -> id + n_tensors - nc), then
// linearised: every step names the POSITIONS of its two operands in the shrinking list of tensors,
// the new tensor being appended last.  `all_pos.index(x)` of the reference is a Fenwick-tree rank here.
#include "../../include/tnco_hip.h"

#include <cstdint>
#include <thread>
#include <vector>

namespace {
struct Fenwick {
  std::vector<int32_t> t;
  void reset(int n, int live) {  // entries 0..live-1 present
    t.assign((size_t)n + 1, 0);
    for (int i = 1; i <= n; ++i) {
      t[i] += (i <= live) ? 1 : 0;
      const int j = i + (i & -i);
      if (j <= n) t[j] += t[i];
    }
  }
  void add(int i, int v) {
    for (++i; i < (int)t.size(); i += i & -i) t[i] += v;
  }
  int prefix(int i) const {  // live entries among [0, i)
    int s = 0;
    for (; i > 0; i -= i & -i) s += t[i];
    return s;
  }
};
}  // namespace

extern "C" int tnco_hip_linear_paths(int32_t n_tensors, int32_t nc, const int32_t* tensors_pos, int64_t k,
                                     const int32_t* contraction, int32_t* paths, int32_t n_threads) {
  if (n_tensors < 1 || nc < 1 || nc > n_tensors || !tensors_pos || k < 0 || (k > 0 && (!contraction || !paths)))
    return TNCO_HIP_EINVAL;
  for (int32_t i = 0; i < nc; ++i)
    if (tensors_pos[i] < 0 || tensors_pos[i] >= n_tensors || (i > 0 && tensors_pos[i] <= tensors_pos[i - 1]))
      return TNCO_HIP_EINVAL;
  const int32_t steps = nc - 1;
  int nth = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  nth = (int)std::max<int64_t>(1, std::min<int64_t>(nth, k / 16 + 1));
  std::vector<int> ok((size_t)nth, 1);
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&, t]() {
      Fenwick leaves, inter;
      std::vector<int32_t> step_of((size_t)steps + 1);
      for (int64_t q = t; q < k; q += nth) {
        const int32_t* con = contraction + q * 3 * steps;
        int32_t* out = paths + q * 2 * steps;
        leaves.reset(n_tensors, n_tensors);
        inter.reset(steps + 1, 0);
        int32_t live_leaves = n_tensors;
        for (int32_t s = 0; s < steps; ++s) {
          int32_t pos[2];
          for (int j = 0; j < 2; ++j) {
            const int32_t x = con[3 * s + j];
            if (x < 0 || x >= 2 * nc - 1) { ok[t] = 0; return; }
            if (x < nc) {
              pos[j] = leaves.prefix(tensors_pos[x]);
            } else {
              pos[j] = live_leaves + inter.prefix(step_of[x - nc]);
            }
          }
          out[2 * s] = pos[0];
          out[2 * s + 1] = pos[1];
          for (int j = 0; j < 2; ++j) {
            const int32_t x = con[3 * s + j];
            if (x < nc) {
              leaves.add(tensors_pos[x], -1);
              --live_leaves;
            } else {
              inter.add(step_of[x - nc], -1);
            }
          }
          const int32_t z = con[3 * s + 2];
          if (z < nc || z >= 2 * nc - 1) { ok[t] = 0; return; }
          step_of[z - nc] = s;
          inter.add(s, 1);
        }
      }
    });
  for (auto& x : th) x.join();
  for (int v : ok)
    if (!v) return TNCO_HIP_EINVAL;
  return TNCO_HIP_OK;
}

// The same linearisation for contractions already in SSA form over ALL tensors: triples (x, y, z)
// with leaves 0 .. n_tensors-1 and the s-th step creating id n_tensors + s.  This is what
// merge_contraction_paths (tnco/utils/tn.py:334-401) computes for the concatenation of the
// components' paths: every step names the positions of its operands in one global list, sorted.
extern "C" int tnco_hip_linear_paths_ssa(int32_t n_tensors, int32_t steps, int64_t k, const int32_t* triples,
                                         int32_t* paths, int32_t n_threads) {
  if (n_tensors < 1 || steps < 0 || k < 0 || (k > 0 && steps > 0 && (!triples || !paths))) return TNCO_HIP_EINVAL;
  int nth = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  nth = (int)std::max<int64_t>(1, std::min<int64_t>(nth, k / 16 + 1));
  std::vector<int> ok((size_t)nth, 1);
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&, t]() {
      Fenwick leaves, inter;
      for (int64_t q = t; q < k; q += nth) {
        const int32_t* con = triples + q * 3 * steps;
        int32_t* out = paths + q * 2 * steps;
        leaves.reset(n_tensors, n_tensors);
        inter.reset(steps + 1, 0);
        int32_t live_leaves = n_tensors;
        for (int32_t s = 0; s < steps; ++s) {
          int32_t pos[2];
          for (int j = 0; j < 2; ++j) {
            const int32_t x = con[3 * s + j];
            if (x < 0 || x >= n_tensors + s) { ok[t] = 0; return; }
            pos[j] = x < n_tensors ? leaves.prefix(x) : live_leaves + inter.prefix(x - n_tensors);
          }
          out[2 * s] = std::min(pos[0], pos[1]);
          out[2 * s + 1] = std::max(pos[0], pos[1]);
          for (int j = 0; j < 2; ++j) {
            const int32_t x = con[3 * s + j];
            if (x < n_tensors) {
              leaves.add(x, -1);
              --live_leaves;
            } else {
              inter.add(x - n_tensors, -1);
            }
          }
          if (con[3 * s + 2] != n_tensors + s) { ok[t] = 0; return; }
          inter.add(s, 1);
        }
      }
    });
  for (auto& x : th) x.join();
  for (int v : ok)
    if (!v) return TNCO_HIP_EINVAL;
  return TNCO_HIP_OK;
}
