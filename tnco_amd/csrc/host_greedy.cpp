// host_greedy.cpp -- batched initial contraction trees as the reference draws them (host,
// multi-threaded): CPython's Random(seed).shuffle of the component's tensors followed by
// opt_einsum's greedy path finder with every dimension 2 (tnco/utils/tn.py:189-230,
// tnco/app/infinite_memory/sa.py:173-190).  Spec: tnco_amd/ctree.py greedy_contraction /
// ssa_greedy, which this file must match tree for tree (tests/test_host.py).
//
// opt_einsum is a third-party dependency the reference does not pin (pyproject.toml:55) and that is
// absent from this image: the greedy below restates its published algorithm
// (paths.ssa_greedy_optimize, choose_fn = _simple_chooser, cost_fn = 'memory-removed') and is
// "parity unpinned" against it.  CPython's generator (Modules/_randommodule.c: init_by_array,
// genrand_uint32, getrandbits; Lib/random.py: _randbelow_with_getrandbits, shuffle) is checked
// against the interpreter of this image by the test.
#include "../../include/tnco_hip.h"
#include "greedy_key.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <queue>
#include <thread>
#include <vector>

namespace {

// MT19937 seeded the way random.seed(int) does: init_by_array(key = 32-bit digits of abs(seed)).
struct PyRandom {
  uint32_t mt[624];
  int idx = 624;
  uint64_t draws = 0;
  void init_genrand(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  explicit PyRandom(uint32_t seed) {
    const uint32_t key[1] = {seed};
    const int klen = 1;
    init_genrand(19650218u);
    int i = 1, j = 0;
    for (int k = 624 > klen ? 624 : klen; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
      if (++j >= klen) j = 0;
    }
    for (int k = 623; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int k = 0; k < 624; ++k) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    ++draws;
    return y;
  }
  // Random._randbelow_with_getrandbits(n), 0 < n <= 2^31
  uint32_t randbelow(uint32_t n) {
    int k = 0;
    while ((n >> k) != 0) ++k;  // n.bit_length()
    uint32_t r = next() >> (32 - k);
    while (r >= n) r = next() >> (32 - k);
    return r;
  }
};

struct Work {
  int32_t n, I, W;
  const int32_t* off;      // CSR index -> holders
  const int32_t* holders;
  std::vector<uint64_t> leaf;    // [n][W]
  std::vector<uint64_t> output;  // [W] (filtered output indices of the component)
};

// cost = 2^a - 2^b - 2^c, compared exactly: x < y  <=>  2^xa + 2^yb + 2^yc < 2^ya + 2^xb + 2^xc
struct Cost {
  int32_t a, b, c;
};
inline int norm3(int32_t e[3]) {  // descending distinct exponents of 2^e0 + 2^e1 + 2^e2; returns their number
  std::sort(e, e + 3, [](int32_t x, int32_t y) { return x > y; });
  int m = 3;
  for (bool again = true; again;) {
    again = false;
    for (int i = 0; i + 1 < m; ++i)
      if (e[i] == e[i + 1]) {
        e[i] += 1;
        for (int j = i + 1; j + 1 < m; ++j) e[j] = e[j + 1];
        --m;
        again = true;
        break;
      }
    std::sort(e, e + m, [](int32_t x, int32_t y) { return x > y; });
  }
  return m;
}
inline int cmp_cost(const Cost& x, const Cost& y) {
  if (x.a < 125 && x.b < 125 && x.c < 125 && y.a < 125 && y.b < 125 && y.c < 125) {
    typedef __int128 i128;
    const i128 vx = ((i128)1 << x.a) - ((i128)1 << x.b) - ((i128)1 << x.c);
    const i128 vy = ((i128)1 << y.a) - ((i128)1 << y.b) - ((i128)1 << y.c);
    return vx < vy ? -1 : (vx > vy ? 1 : 0);
  }
  int32_t l[3] = {x.a, y.b, y.c}, r[3] = {y.a, x.b, x.c};
  const int nl = norm3(l), nr = norm3(r);
  for (int i = 0; i < std::min(nl, nr); ++i)
    if (l[i] != r[i]) return l[i] < r[i] ? -1 : 1;
  return nl < nr ? -1 : (nl > nr ? 1 : 0);
}

struct Cand {
  Cost cost;
  int32_t id2, id1;  // ssa ids at push time, id1 < id2
  int32_t s1, s2;    // key slots
  int32_t k12;       // arena offset (in masks) of the resulting index set
};
struct CandGreater {  // for a min-heap on (cost, id2, id1)
  bool operator()(const Cand& x, const Cand& y) const {
    const int c = cmp_cost(x.cost, y.cost);
    if (c) return c > 0;
    if (x.id2 != y.id2) return x.id2 > y.id2;
    return x.id1 > y.id1;
  }
};

struct Scratch {
  std::vector<uint64_t> keys;     // [slot][W] index sets; a slot is an index SET, alive or not
  std::vector<int32_t> ssa, fp;   // per slot: current ssa id, |key|
  std::vector<int32_t> nz_off;    // per slot: its nonzero words are nz[nz_off[s] .. nz_off[s + 1])
  std::vector<uint8_t> nz;
  std::vector<int32_t> keepcnt;   // scratch of push_best: per word, legs of k1 that survive a contraction
  std::vector<uint8_t> alive;
  std::vector<std::vector<int32_t>> dim_keys;  // per non-output dim: the INPUT keys holding it (set-up only)
  std::vector<int32_t> cnt;       // per non-output dim: live keys holding it
  std::vector<uint64_t> nbr;      // [slot][NW] live keys that share a non-output dim with the slot's key
  std::vector<uint64_t> arena;    // k12 of the queued candidates
  std::vector<int32_t> k2s, order;
  std::vector<int32_t> table;     // open addressing: hash of the set -> slot (-1: free)
  std::vector<std::pair<int32_t, int32_t>> path;
};

inline uint64_t hash_mask(const uint64_t* m, int W) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  for (int w = 0; w < W; ++w) {
    h ^= m[w] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
  }
  return h;
}
inline int popc(const uint64_t* m, int W) {
  int c = 0;
  for (int w = 0; w < W; ++w) c += __builtin_popcountll(m[w]);
  return c;
}

// ssa path of opt_einsum's greedy over `inputs` (index sets as masks, in shuffled order)
void ssa_greedy(const Work& w, const uint64_t* inputs, Scratch& S) {
  const int n = w.n, W = w.W;
  S.path.clear();
  if (n == 1) return;
  std::vector<uint64_t> output(w.output);
  {  // dims common to all inputs join the output
    std::vector<uint64_t> all((size_t)W, ~0ull);
    for (int t = 0; t < n; ++t)
      for (int x = 0; x < W; ++x) all[x] &= inputs[(size_t)t * W + x];
    for (int x = 0; x < W; ++x) output[x] |= all[x];
  }
  S.keys.clear(); S.ssa.clear(); S.fp.clear(); S.alive.clear(); S.arena.clear();
  S.nz.clear(); S.nz_off.assign(1, 0); S.keepcnt.assign((size_t)W, 0);
  size_t tsize = 64;
  while (tsize < 8 * (size_t)n) tsize <<= 1;  // at most 2n - 1 + (outer products) index sets
  S.table.assign(tsize, -1);
  if (S.dim_keys.size() != (size_t)w.I) S.dim_keys.assign((size_t)w.I, {});
  for (auto& v : S.dim_keys) v.clear();
  int32_t next_ssa = n;
  size_t last_probe = 0;  // where find_slot stopped: the free cell a following new_slot takes
  auto find_slot = [&](const uint64_t* m) -> int32_t {
    for (size_t h = hash_mask(m, W) & (tsize - 1);; h = (h + 1) & (tsize - 1)) {
      const int32_t s = S.table[h];
      if (s < 0) {
        last_probe = h;
        return -1;
      }
      if (!std::memcmp(&S.keys[(size_t)s * W], m, (size_t)W * 8)) return s;
    }
  };
  auto new_slot = [&](const uint64_t* m) -> int32_t {  // (right after a find_slot(m) that returned -1)
    const int32_t s = (int32_t)S.ssa.size();
    S.keys.insert(S.keys.end(), m, m + W);
    S.ssa.push_back(-1);
    int fp = 0;
    for (int x = 0; x < W; ++x)
      if (m[x]) {
        fp += __builtin_popcountll(m[x]);
        S.nz.push_back((uint8_t)x);
      }
    S.fp.push_back(fp);
    S.alive.push_back(0);
    S.nz_off.push_back((int32_t)S.nz.size());
    S.table[last_probe] = s;
    return s;
  };
  S.keys.reserve((size_t)(2 * n + 8) * W);
  S.nz.reserve((size_t)(2 * n + 8) * W);
  auto for_dims = [&](const uint64_t* m, bool minus_output, auto&& f) {
    for (int x = 0; x < W; ++x) {
      uint64_t v = m[x] & (minus_output ? ~output[x] : ~0ull);
      while (v) {
        const int b = __builtin_ctzll(v);
        v &= v - 1;
        f(x * 64 + b);
      }
    }
  };
  // Bookkeeping of the published algorithm (dim -> keys holding it; the keys that share a dim with
  // the new tensor) in a form whose upkeep costs O(shared dims + neighbours) per contraction instead
  // of O(legs of both tensors): per dim only the NUMBER of live holders (that is all `_get_candidate`
  // reads: held by >= 2, by >= 3), per key the SET of its neighbours as a bitset over the slots.  When
  // k1 and k2 become k12, a dim held by one of them and kept keeps its number of holders, and every
  // neighbour of k1 or k2 is a neighbour of k12: it shares a dim held by >= 2 (or >= 3 if both hold
  // it) keys, which the result keeps.  (A queued candidate's stored result stays consistent with the
  // counts while both its keys are alive: a count only changes when two holders merge.)
  const int SMAX = 2 * n + 8;  // slots: inputs, one result per contraction (outer products keep no slot)
  const int NW = (SMAX + 63) / 64;
  S.cnt.assign((size_t)w.I, 0);
  S.nbr.assign((size_t)SMAX * NW, 0);
  auto nb = [&](int32_t s) { return &S.nbr[(size_t)s * NW]; };
  auto nb_set = [&](int32_t s, int32_t x) { nb(s)[x >> 6] |= 1ull << (x & 63); };
  auto nb_clr = [&](int32_t s, int32_t x) { nb(s)[x >> 6] &= ~(1ull << (x & 63)); };
  // eager Hadamard products of equal index sets
  for (int t = 0; t < n; ++t) {
    const uint64_t* m = inputs + (size_t)t * W;
    int32_t s = find_slot(m);
    if (s >= 0 && S.alive[s]) {
      S.path.emplace_back(S.ssa[s], t);
      S.ssa[s] = next_ssa++;
    } else {
      if (s < 0) s = new_slot(m);
      S.alive[s] = 1;
      S.ssa[s] = t;
    }
  }
  const int32_t n_init = (int32_t)S.ssa.size();
  for (int32_t s = 0; s < n_init; ++s)
    for_dims(&S.keys[(size_t)s * W], true, [&](int d) { S.dim_keys[d].push_back(s); });
  std::vector<uint64_t> ref2((size_t)W), ref3((size_t)W), k12((size_t)W);
  auto refresh_ref = [&](int d) {  // dims held by >= 2 / >= 3 live keys (never output dims)
    const int32_t c = S.cnt[d];
    const uint64_t bit = 1ull << (d & 63);
    if (c >= 2) ref2[d >> 6] |= bit; else ref2[d >> 6] &= ~bit;
    if (c >= 3) ref3[d >> 6] |= bit; else ref3[d >> 6] &= ~bit;
  };
  for (int d = 0; d < w.I; ++d) {
    const auto& v = S.dim_keys[d];
    S.cnt[d] = (int32_t)v.size();
    refresh_ref(d);
    for (size_t i = 0; i < v.size(); ++i)
      for (size_t j = i + 1; j < v.size(); ++j) {
        nb_set(v[i], v[j]);
        nb_set(v[j], v[i]);
      }
  }

  std::priority_queue<Cand, std::vector<Cand>, CandGreater> queue;
  auto result_mask = [&](int32_t s1, int32_t s2, uint64_t* out12) {
    const uint64_t *a = &S.keys[(size_t)s1 * W], *b = &S.keys[(size_t)s2 * W];
    for (int x = 0; x < W; ++x) {
      const uint64_t either = a[x] | b[x], two = a[x] & b[x], one = either & ~two;
      out12[x] = (either & output[x]) | (two & ref3[x]) | (one & ref2[x]);
    }
  };
  // The candidates of one push share k1 and the reference counts: where k2 has no legs (most words
  // of most neighbours) the result keeps k1 & (output | ref2), counted once per push; only k2's
  // nonzero words are evaluated per candidate, and the result's index set only for the winner.
  auto push_best = [&](int32_t s1, const int32_t* k2s, int cnt) {
    const uint64_t* a = &S.keys[(size_t)s1 * W];
    int32_t keep_total = 0;
    for (int x = 0; x < W; ++x) {
      S.keepcnt[x] = __builtin_popcountll(a[x] & (output[x] | ref2[x]));
      keep_total += S.keepcnt[x];
    }
    Cand best{};
    bool have = false;
    for (int i = 0; i < cnt; ++i) {
      const int32_t s2 = k2s[i];
      const uint64_t* b = &S.keys[(size_t)s2 * W];
      int32_t size12 = keep_total;
      for (int32_t j = S.nz_off[s2]; j < S.nz_off[s2 + 1]; ++j) {
        const int x = S.nz[j];
        const uint64_t either = a[x] | b[x], two = a[x] & b[x], one = either & ~two;
        size12 += __builtin_popcountll((either & output[x]) | (two & ref3[x]) | (one & ref2[x])) - S.keepcnt[x];
      }
      Cand c;
      c.cost = Cost{size12, S.fp[s1], S.fp[s2]};
      int32_t id1 = S.ssa[s1], id2 = S.ssa[s2], t1 = s1, t2 = s2;
      if (id1 > id2) { std::swap(id1, id2); std::swap(t1, t2); }
      c.id1 = id1; c.id2 = id2; c.s1 = t1; c.s2 = t2;
      if (!have || CandGreater()(best, c)) {
        best = c;
        have = true;
      }
    }
    result_mask(best.s1, best.s2, k12.data());
    best.k12 = (int32_t)(S.arena.size() / W);
    S.arena.insert(S.arena.end(), k12.begin(), k12.end());
    queue.push(best);
  };
  // initial candidates: per dim, keys sorted by ssa id, each against the later ones
  for (int d = 0; d < w.I; ++d) {
    auto& v = S.dim_keys[d];
    if (v.size() < 2) continue;
    S.order.assign(v.begin(), v.end());
    std::sort(S.order.begin(), S.order.end(), [&](int32_t x, int32_t y) { return S.ssa[x] < S.ssa[y]; });
    for (size_t i = 0; i + 1 < S.order.size(); ++i)
      push_best(S.order[i], S.order.data() + i + 1, (int)(S.order.size() - i - 1));
  }
  std::vector<uint64_t> uni((size_t)W), un((size_t)NW);
  while (!queue.empty()) {
    const Cand c = queue.top();
    queue.pop();
    if (!S.alive[c.s1] || !S.alive[c.s2]) continue;  // obsolete
    const int32_t s1 = c.s1, s2 = c.s2;
    const int32_t id1 = S.ssa[s1], id2 = S.ssa[s2];
    S.alive[s1] = 0;
    S.alive[s2] = 0;
    S.path.emplace_back(id1, id2);
    std::memcpy(k12.data(), &S.arena[(size_t)c.k12 * W], (size_t)W * 8);
    int32_t s12 = find_slot(k12.data());
    const bool merged = s12 >= 0 && S.alive[s12];  // an equal index set is live: Hadamard product with it
    if (merged) {
      S.path.emplace_back(S.ssa[s12], next_ssa++);
    } else {
      if (s12 < 0) s12 = new_slot(k12.data());
      if (s12 >= SMAX) { S.path.clear(); return; }  // (cannot happen: at most 2n - 1 distinct live / dead sets)
      S.alive[s12] = 1;
    }
    S.ssa[s12] = next_ssa++;
    const uint64_t *a = &S.keys[(size_t)s1 * W], *b = &S.keys[(size_t)s2 * W];
    // holders per dim (dim_to_keys / _update_ref_counts of the published code)
    if (merged) {  // k1 and k2 leave, nothing enters
      for (int x = 0; x < W; ++x) uni[x] = a[x] | b[x];
      for_dims(uni.data(), true, [&](int d) {
        S.cnt[d] -= (int32_t)((a[d >> 6] >> (d & 63)) & 1ull) + (int32_t)((b[d >> 6] >> (d & 63)) & 1ull);
        refresh_ref(d);
      });
    } else {  // only shared dims (two holders become one or none) and dropped dims change their number
      for (int x = 0; x < W; ++x) uni[x] = (a[x] & b[x]) | ((a[x] ^ b[x]) & ~k12[x]);
      for_dims(uni.data(), true, [&](int d) {
        const int in12 = (int)((k12[d >> 6] >> (d & 63)) & 1ull);
        S.cnt[d] -= (int32_t)((a[d >> 6] >> (d & 63)) & 1ull) + (int32_t)((b[d >> 6] >> (d & 63)) & 1ull) - in12;
        refresh_ref(d);
      });
    }
    // neighbours: those of k1 and of k2 (every one of them shares a dim the result keeps)
    uint64_t* n12 = nb(s12);
    const uint64_t *n1 = nb(s1), *n2 = nb(s2);
    for (int x = 0; x < NW; ++x) un[x] = n1[x] | n2[x];
    un[s1 >> 6] &= ~(1ull << (s1 & 63));
    un[s2 >> 6] &= ~(1ull << (s2 & 63));
    un[s12 >> 6] &= ~(1ull << (s12 & 63));
    if (merged) {  // (the live equal set keeps its neighbours, minus the two tensors that just left)
      for (int x = 0; x < NW; ++x) n12[x] |= un[x];
      n12[s1 >> 6] &= ~(1ull << (s1 & 63));
      n12[s2 >> 6] &= ~(1ull << (s2 & 63));
    }
    else { for (int x = 0; x < NW; ++x) n12[x] = un[x]; }
    S.k2s.clear();
    for (int x = 0; x < NW; ++x) {
      uint64_t v = un[x];
      while (v) {
        const int32_t y = x * 64 + __builtin_ctzll(v);
        v &= v - 1;
        nb_clr(y, s1);
        nb_clr(y, s2);
        nb_set(y, s12);
      }
      v = n12[x];
      while (v) {
        S.k2s.push_back(x * 64 + __builtin_ctzll(v));
        v &= v - 1;
      }
    }
    if (!S.k2s.empty()) push_best(s12, S.k2s.data(), (int)S.k2s.size());
  }
  // outer products of what is left, smallest output size first
  struct Rest { int32_t size, id, slot; };
  auto rest_greater = [](const Rest& x, const Rest& y) { return x.size != y.size ? x.size > y.size : x.id > y.id; };
  std::priority_queue<Rest, std::vector<Rest>, decltype(rest_greater)> rest(rest_greater);
  auto out_size = [&](const uint64_t* m) {
    int c = 0;
    for (int x = 0; x < W; ++x) c += __builtin_popcountll(m[x] & output[x]);
    return c;
  };
  for (int32_t s = 0; s < (int32_t)S.ssa.size(); ++s)
    if (S.alive[s]) rest.push(Rest{out_size(&S.keys[(size_t)s * W]), S.ssa[s], s});
  if (rest.empty()) return;
  Rest cur = rest.top();
  rest.pop();
  std::vector<uint64_t> acc(&S.keys[(size_t)cur.slot * W], &S.keys[(size_t)cur.slot * W] + W);
  std::vector<std::vector<uint64_t>> extra;  // index sets of the outer products (slot = -1 - position)
  auto mask_of = [&](const Rest& r) -> const uint64_t* {
    return r.slot >= 0 ? &S.keys[(size_t)r.slot * W] : extra[(size_t)(-1 - r.slot)].data();
  };
  while (!rest.empty()) {
    const Rest o = rest.top();
    rest.pop();
    S.path.emplace_back(std::min(cur.id, o.id), std::max(cur.id, o.id));
    std::vector<uint64_t> m((size_t)W);
    const uint64_t *a = mask_of(cur), *b = mask_of(o);
    for (int x = 0; x < W; ++x) m[x] = (a[x] | b[x]) & output[x];
    extra.push_back(m);
    const Rest nw{popc(m.data(), W), next_ssa++, -(int32_t)extra.size()};
    rest.push(nw);  // heappushpop
    cur = rest.top();
    rest.pop();
  }
}

// one replica: shuffle, greedy, links.  Returns false when the result is not a full binary tree.
bool one_tree(const Work& w, uint32_t seed, uint64_t* draws, int32_t* left, int32_t* right, int32_t* parent,
              Scratch& S, std::vector<int32_t>& perm, std::vector<uint64_t>& inputs) {
  const int32_t n = w.n, N = 2 * n - 1, W = w.W;
  PyRandom rng(seed);
  if (draws)
    for (uint64_t k = 0; k < *draws; ++k) (void)rng.next();
  for (int32_t i = 0; i < n; ++i) perm[i] = i;
  for (int32_t i = n - 1; i >= 1; --i) {  // Random.shuffle
    const uint32_t j = rng.randbelow((uint32_t)i + 1u);
    std::swap(perm[i], perm[j]);
  }
  if (draws) *draws = rng.draws;
  for (int32_t t = 0; t < n; ++t)
    std::memcpy(&inputs[(size_t)t * W], &w.leaf[(size_t)perm[t] * W], (size_t)W * 8);
  for (int32_t i = 0; i < N; ++i) { left[i] = -1; right[i] = -1; parent[i] = -1; }
  if (n == 1) return true;
  if (n == 2) {
    S.path.assign(1, {0, 1});  // contract_path does not call the optimizer for two operands
  } else {
    ssa_greedy(w, inputs.data(), S);
  }
  if ((int32_t)S.path.size() != n - 1) return false;
  for (int32_t s = 0; s < n - 1; ++s) {
    auto node = [&](int32_t x) { return x < n ? perm[x] : x; };
    const int32_t a = node(S.path[s].first), b = node(S.path[s].second), z = n + s;
    if (a < 0 || b < 0 || a >= z || b >= z || a == b || parent[a] >= 0 || parent[b] >= 0) return false;
    left[z] = std::min(a, b);
    right[z] = std::max(a, b);
    parent[a] = z;
    parent[b] = z;
  }
  return true;
}

}  // namespace

extern "C" int tnco_hip_greedy_trees(int32_t n_leaves, int32_t n_inds, const int32_t* holders_off,
                                     const int32_t* holders, const uint64_t* output_mask, int64_t n_replicas,
                                     const uint32_t* seeds, uint64_t* draws, int32_t* links_out,
                                     int32_t n_threads) {
  if (n_leaves < 1 || n_inds < 0 || !holders_off || !holders || !seeds || !links_out || n_replicas < 0)
    return TNCO_HIP_EINVAL;
  Work w;
  w.n = n_leaves; w.I = n_inds; w.W = std::max(1, (n_inds + 63) / 64);
  w.off = holders_off; w.holders = holders;
  w.leaf.assign((size_t)n_leaves * w.W, 0);
  for (int32_t i = 0; i < n_inds; ++i)
    for (int32_t k = holders_off[i]; k < holders_off[i + 1]; ++k) {
      const int32_t t = holders[k];
      if (t < 0 || t >= n_leaves) return TNCO_HIP_EINVAL;
      w.leaf[(size_t)t * w.W + (i >> 6)] |= 1ull << (i & 63);
    }
  w.output.assign((size_t)w.W, 0);
  if (output_mask)
    for (int x = 0; x < w.W; ++x) w.output[x] = output_mask[x];
  const int64_t N = 2 * (int64_t)n_leaves - 1;
  int nth = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  nth = (int)std::max<int64_t>(1, std::min<int64_t>(nth, n_replicas));
  std::vector<int> ok((size_t)nth, 1);
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&, t]() {
      Scratch S;
      std::vector<int32_t> perm((size_t)n_leaves);
      std::vector<uint64_t> inputs((size_t)n_leaves * w.W);
      for (int64_t r = t; r < n_replicas; r += nth) {
        int32_t* lk = links_out + r * 3 * N;
        if (!one_tree(w, seeds[r], draws ? draws + r : nullptr, lk, lk + N, lk + 2 * N, S, perm, inputs)) ok[t] = 0;
      }
    });
  for (auto& x : th) x.join();
  for (int v : ok)
    if (!v) return TNCO_HIP_EINVAL;
  return TNCO_HIP_OK;
}

// the order-preserving key of 2^a - 2^b - 2^c the device generator sorts its candidates by (greedy_key.h),
// for the test that checks it against exact integers
extern "C" uint64_t tnco_hip_diag_greedy_cost_key(int32_t a, int32_t b, int32_t c) { return tnco::greedy_cost_key(a, b, c); }
