// host_trees.cpp -- batched seeded initial contraction trees (host, multi-threaded).
//
// Replaces, for a batch of replicas, the per-run Python of the reference
// (tnco/app/infinite_memory/sa.py:173-190: get_random_contraction_path ->
// ContractionTree).  The reference draws its initial tree from opt_einsum's
// greedy on a shuffled tensor list (tnco/utils/tn.py:195-230); opt_einsum is an
// unpinned third-party package, so the initial tree is not part of the parity
// contract and this build defines its own generator ("random Kruskal", spec in
// tnco_amd/ctree.py: random_contraction, which this file must match bit for bit).
#include "../../include/tnco_hip.h"

#include <algorithm>
#include <cstdint>
#include <random>
#include <thread>
#include <vector>

namespace {

struct Work {
  int32_t n, I;
  const int32_t* off;
  const int32_t* holders;
};

// returns false if the component is not connected
bool one_tree(const Work& w, uint32_t seed, int32_t* left, int32_t* right, int32_t* parent,
              std::vector<int32_t>& perm, std::vector<int32_t>& uf, std::vector<int32_t>& node) {
  const int32_t n = w.n, I = w.I, N = 2 * n - 1;
  std::mt19937 g;
  g.seed(seed);
  for (int32_t i = 0; i < I; ++i) perm[i] = i;
  for (int32_t i = I - 1; i >= 1; --i) {
    const uint32_t x = (uint32_t)g();
    const int32_t j = (int32_t)(x % (uint32_t)(i + 1));
    std::swap(perm[i], perm[j]);
  }
  for (int32_t t = 0; t < n; ++t) { uf[t] = t; node[t] = t; }
  for (int32_t i = 0; i < N; ++i) { left[i] = -1; right[i] = -1; parent[i] = -1; }
  auto find = [&](int32_t a) {
    while (uf[a] != a) {
      uf[a] = uf[uf[a]];
      a = uf[a];
    }
    return a;
  };
  int32_t nxt = n;
  for (int32_t q = 0; q < I; ++q) {
    const int32_t idx = perm[q];
    const int32_t b = w.off[idx], e = w.off[idx + 1];
    if (e - b < 2) continue;
    for (int32_t k = b + 1; k < e; ++k) {
      const int32_t ra = find(w.holders[b]), rb = find(w.holders[k]);
      if (ra == rb) continue;
      const int32_t x = node[ra], y = node[rb];
      left[nxt] = std::min(x, y);
      right[nxt] = std::max(x, y);
      parent[x] = nxt;
      parent[y] = nxt;
      const int32_t r = std::min(ra, rb);
      uf[ra] = r;
      uf[rb] = r;
      node[r] = nxt;
      ++nxt;
    }
  }
  return nxt == N;
}

}  // namespace

extern "C" int tnco_hip_random_trees(int32_t n_leaves, int32_t n_inds, const int32_t* holders_off,
                                     const int32_t* holders, int64_t n_replicas, const uint32_t* seeds,
                                     int32_t* links_out, int32_t n_threads) {
  if (n_leaves < 1 || n_inds < 0 || !holders_off || !holders || !seeds || !links_out || n_replicas < 0)
    return TNCO_HIP_EINVAL;
  const Work w{n_leaves, n_inds, holders_off, holders};
  const int64_t N = 2 * (int64_t)n_leaves - 1;
  int nth = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  nth = (int)std::max<int64_t>(1, std::min<int64_t>(nth, n_replicas));
  std::vector<int> ok((size_t)nth, 1);
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&, t]() {
      std::vector<int32_t> perm((size_t)std::max(n_inds, 1)), uf((size_t)n_leaves), node((size_t)n_leaves);
      for (int64_t r = t; r < n_replicas; r += nth) {
        int32_t* lk = links_out + r * 3 * N;
        if (!one_tree(w, seeds[r], lk, lk + N, lk + 2 * N, perm, uf, node)) ok[t] = 0;
      }
    });
  for (auto& x : th) x.join();
  for (int v : ok)
    if (!v) return TNCO_HIP_EINVAL;
  return TNCO_HIP_OK;
}
