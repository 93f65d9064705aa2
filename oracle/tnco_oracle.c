/*
 * oracle/tnco_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, one replica at a time) of the simulated-
 * annealing inner loop of google-research/tnco, written from the reference's
 * published algorithm.  It is the checker the HIP path is compared with: only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load
 * it.  The product (tnco_amd/) never imports, links or calls it.
 *
 * PARITY PIN STATUS (see DESIGN.md "Oracle"):
 *   - mt19937 / uniform01 / shuffle / uniform_int: pinned against the real
 *     libstdc++ of this image (tests/golden/stdlib_rng.json, generator
 *     oracle/stdlib_rng.cpp).
 *   - tree rotation + tree validity: pinned against the real reference
 *     Tree::swap_with_nn / Tree::is_valid compiled from
 *     /root/reference/include/tnco/{node,tree}.hpp (oracle/_ref/libref_tree.so).
 *   - Optimizer::update itself: PARITY UNPINNED by reference execution.  The
 *     reference hot path needs <boost/dynamic_bitset.hpp>, which this image
 *     does not have, so it is unbuildable here; update() is pinned only by the
 *     invariants the reference's own tests assert (brute-force cost recompute,
 *     lock-step determinism, greedy monotonicity, min <= cur) and the one
 *     worked example of examples/BaseOptimization.ipynb.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 *
 * Compiled twice: COST_T=double (prefix orc_f64_) and COST_T=float (orc_f32_).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef COST_T
#define COST_T double
#define PFX(name) orc_f64_##name
#endif

#undef cost_t
#define cost_t COST_T

/* ------------------------------------------------------------------------ */
/* PRNG: std::mt19937 + std::uniform_real_distribution<double> (libstdc++). */
/* include/tnco/globals.hpp:38 (prng_type = std::mt19937).                  */
/* ------------------------------------------------------------------------ */
#ifndef ORC_COMMON_DEFINED
#define ORC_COMMON_DEFINED

#define MT_N 624
#define MT_M 397

typedef struct {
  uint32_t x[MT_N];
  int32_t p; /* libstdc++ _M_p: next output position, 624 => regenerate */
} orc_mt_t;

/* /usr/include/c++/11/bits/random.tcc:326-343  mersenne_twister_engine::seed */
void orc_mt_seed(orc_mt_t* g, uint64_t seed) {
  g->x[0] = (uint32_t)(seed & 0xffffffffu);
  for (int i = 1; i < MT_N; ++i) {
    uint32_t x = g->x[i - 1];
    x ^= x >> 30;
    x *= 1812433253u;
    x += (uint32_t)i;
    g->x[i] = x;
  }
  g->p = MT_N;
}

/* random.tcc:396-430  _M_gen_rand */
static void orc_mt_gen(orc_mt_t* g) {
  const uint32_t U = 0x80000000u, L = 0x7fffffffu;
  for (int k = 0; k < MT_N - MT_M; ++k) {
    uint32_t y = (g->x[k] & U) | (g->x[k + 1] & L);
    g->x[k] = g->x[k + MT_M] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
  }
  for (int k = MT_N - MT_M; k < MT_N - 1; ++k) {
    uint32_t y = (g->x[k] & U) | (g->x[k + 1] & L);
    g->x[k] = g->x[k + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
  }
  uint32_t y = (g->x[MT_N - 1] & U) | (g->x[0] & L);
  g->x[MT_N - 1] = g->x[MT_M - 1] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
  g->p = 0;
}

/* random.tcc:453-471  operator() */
uint32_t orc_mt_next(orc_mt_t* g) {
  if (g->p >= MT_N) orc_mt_gen(g);
  uint32_t z = g->x[g->p++];
  z ^= (z >> 11) & 0xffffffffu;
  z ^= (z << 7) & 0x9d2c5680u;
  z ^= (z << 15) & 0xefc60000u;
  z ^= (z >> 18);
  return z;
}

/* random.tcc:3348-3380 generate_canonical<double,53>: two 32-bit draws, low
 * word first; used by std::uniform_real_distribution<double>{} at
 * include/tnco/optimize/infinite_memory/optimizer.hpp:100,162. */
double orc_uniform01(orc_mt_t* g) {
  double sum = 0.0, tmp = 1.0;
  sum += (double)orc_mt_next(g) * tmp;
  tmp *= 4294967296.0;
  sum += (double)orc_mt_next(g) * tmp;
  tmp *= 4294967296.0;
  double ret = sum / tmp;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret;
}

/* /usr/include/c++/11/bits/uniform_int_dist.h:246-321
 * uniform_int_distribution<unsigned long>::operator()(urng, param{0,hi}) for a
 * 32-bit generator: Lemire's nearly-divisionless method when the range fits in
 * 32 bits (the only case the hot path reaches: shuffling <= n_inds items). */
uint64_t orc_uniform_int(orc_mt_t* g, uint64_t hi /* inclusive */) {
  const uint64_t urngrange = 0xffffffffull;
  uint64_t urange = hi;
  if (urngrange > urange) {
    const uint32_t uerange = (uint32_t)(urange + 1);
    uint64_t product = (uint64_t)orc_mt_next(g) * (uint64_t)uerange;
    uint32_t low = (uint32_t)product;
    if (low < uerange) {
      uint32_t threshold = (uint32_t)(-uerange) % uerange;
      while (low < threshold) {
        product = (uint64_t)orc_mt_next(g) * (uint64_t)uerange;
        low = (uint32_t)product;
      }
    }
    return product >> 32;
  } else if (urngrange == urange) {
    return orc_mt_next(g);
  }
  /* urange > 2^32-1 is never reached on this path. */
  abort();
}

/* /usr/include/c++/11/bits/stl_algo.h:3706-3792 std::shuffle for a 32-bit
 * URBG: draws two swap positions from one variate while
 * (urngrange / urange) >= urange, else one draw per element. */
static void orc_gen_two(orc_mt_t* g, uint64_t b0, uint64_t b1, uint64_t* o0,
                        uint64_t* o1) {
  uint64_t x = orc_uniform_int(g, (b0 * b1) - 1);
  *o0 = x / b1;
  *o1 = x % b1;
}

void orc_shuffle_i32(orc_mt_t* g, int32_t* a, int64_t n) {
  if (n <= 0) return;
  const uint64_t urngrange = 0xffffffffull;
  const uint64_t urange = (uint64_t)n;
  if (urngrange / urange >= urange) {
    int64_t i = 1;
    if ((urange % 2) == 0) {
      uint64_t j = orc_uniform_int(g, 1);
      int32_t t = a[i]; a[i] = a[j]; a[j] = t;
      ++i;
    }
    while (i != n) {
      const uint64_t swap_range = (uint64_t)i + 1;
      uint64_t p0, p1;
      orc_gen_two(g, swap_range, swap_range + 1, &p0, &p1);
      int32_t t = a[i]; a[i] = a[p0]; a[p0] = t;
      ++i;
      t = a[i]; a[i] = a[p1]; a[p1] = t;
      ++i;
    }
    return;
  }
  for (int64_t i = 1; i < n; ++i) {
    uint64_t j = orc_uniform_int(g, (uint64_t)i);
    int32_t t = a[i]; a[i] = a[j]; a[j] = t;
  }
}

/* ------------------------------------------------------------------------ */
/* Bitset helpers (boost::dynamic_bitset semantics, include/tnco/bitset.hpp: */
/* bit p of a mask <-> index position p; words little-endian in p).         */
/* ------------------------------------------------------------------------ */
static inline int bs_intersects(const uint64_t* a, const uint64_t* b, int W) {
  for (int w = 0; w < W; ++w)
    if (a[w] & b[w]) return 1;
  return 0;
}
static inline int bs_count(const uint64_t* a, int W) {
  int c = 0;
  for (int w = 0; w < W; ++w) c += __builtin_popcountll(a[w]);
  return c;
}
static inline int bs_any(const uint64_t* a, int W) {
  for (int w = 0; w < W; ++w)
    if (a[w]) return 1;
  return 0;
}
static inline int bs_subset(const uint64_t* a, const uint64_t* b, int W) {
  for (int w = 0; w < W; ++w)
    if (a[w] & ~b[w]) return 0;
  return 1;
}

/* include/tnco/utils.hpp:34-51 traverse(): explicit stack, pushes child1 then
 * child0 so child0's subtree is visited first; callback at leaves and at the
 * second visit of internal nodes (post-order).  Writes the visit order. */
int orc_traverse(int32_t N, const int32_t* left, const int32_t* right,
                 int32_t* order /* [N] */) {
  int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * N + 2));
  uint8_t* visited = (uint8_t*)calloc((size_t)N, 1);
  int sp = 0, k = 0;
  stack[sp++] = N - 1;
  while (sp) {
    int32_t pos = stack[sp - 1];
    if (visited[pos] || left[pos] < 0) {
      --sp;
      order[k++] = pos;
    } else {
      visited[pos] = 1;
      stack[sp++] = right[pos];
      stack[sp++] = left[pos];
    }
  }
  free(stack);
  free(visited);
  return k;
}

/* include/tnco/utils.hpp:53-71 get_contraction(): (child0, child1, pos) for
 * every internal node in traverse order. */
int orc_get_contraction(int32_t N, const int32_t* left, const int32_t* right,
                        int32_t* out /* [(N-1)/2][3] */) {
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
  int k = orc_traverse(N, left, right, order), m = 0;
  for (int i = 0; i < k; ++i) {
    int32_t pos = order[i];
    if (left[pos] >= 0) {
      out[3 * m + 0] = left[pos];
      out[3 * m + 1] = right[pos];
      out[3 * m + 2] = pos;
      ++m;
    }
  }
  free(order);
  return m;
}

/* include/tnco/tree.hpp:141-192 Tree::swap_with_nn(pos_D): with B=parent(D),
 * A=parent(B), C=sibling(B): A.children[slot of C]=D, B.children[slot of D]=C,
 * C.parent=B, D.parent=A.  No-op if D or B is root or D out of range. */
void orc_swap_with_nn(int32_t N, int32_t* left, int32_t* right, int32_t* parent,
                      int32_t pos_D) {
  if (pos_D >= N) return;
  if (parent[pos_D] < 0) return;
  const int32_t pos_B = parent[pos_D];
  if (parent[pos_B] < 0) return;
  const int32_t pos_A = parent[pos_B];
  const int32_t pos_C = (left[pos_A] == pos_B) ? right[pos_A] : left[pos_A];
  if (left[pos_A] != pos_C) right[pos_A] = pos_D; else left[pos_A] = pos_D;
  if (left[pos_B] != pos_D) right[pos_B] = pos_C; else left[pos_B] = pos_C;
  parent[pos_C] = pos_B;
  parent[pos_D] = pos_A;
}

/* include/tnco/node.hpp:72-107 Node::is_valid + include/tnco/tree.hpp:58-139
 * Tree::is_valid.  Returns 0 if valid, else a code 1..8. */
int orc_tree_is_valid(int32_t N, const int32_t* left, const int32_t* right,
                      const int32_t* parent) {
  if (N <= 0) return 1;
  for (int32_t i = 0; i < N; ++i) {
    const int32_t xs[3] = {parent[i], left[i], right[i]};
    for (int k = 0; k < 3; ++k)
      if (!(xs[k] == -1 || (xs[k] >= 0 && xs[k] < N))) return 2;
    if ((left[i] < 0) != (right[i] < 0)) return 2;
    if (left[i] >= 0 && left[i] == right[i]) return 2;
    if (left[i] >= 0 && parent[i] >= 0 &&
        (parent[i] == left[i] || parent[i] == right[i]))
      return 2;
  }
  if (parent[N - 1] != -1) return 3;
  int roots = 0, leaves = 0;
  for (int32_t i = 0; i < N; ++i) {
    roots += parent[i] < 0;
    leaves += left[i] < 0;
  }
  if (roots != 1) return 4;
  /* n_leaves(): (size+1)/2 (tree.hpp:196-204) */
  const int32_t n_leaves = (N + 1) / 2;
  for (int32_t i = 0; i < n_leaves; ++i)
    if (left[i] >= 0) return 5;
  if (N != 2 * n_leaves - 1 || leaves != n_leaves) return 6;
  int32_t* cp = (int32_t*)calloc((size_t)N, sizeof(int32_t));
  int32_t* cc = (int32_t*)calloc((size_t)N, sizeof(int32_t));
  for (int32_t i = 0; i < N; ++i) {
    if (left[i] >= 0) { cc[left[i]]++; cc[right[i]]++; }
    if (parent[i] >= 0) cp[parent[i]]++;
  }
  /* tree.hpp:113-131 as written: the lambdas are `return c == node.is_leaf() ? 0 : 2;` and
   * `return c == node.is_root() ? 0 : 1;`, which C++ parses as `(c == is_leaf) ? 0 : 2`, so the
   * reference only requires count_parents != is_leaf and count_children != is_root (pinned by
   * oracle/_ref).  The product's own check (tnco_hip_create) is deliberately stricter. */
  int bad = 0;
  for (int32_t i = 0; i < N; ++i) {
    if (cp[i] == ((left[i] < 0) ? 1 : 0)) bad = 7;
    if (cc[i] == ((parent[i] < 0) ? 1 : 0)) bad = 8;
  }
  free(cp);
  free(cc);
  return bad;
}

/* include/tnco/ctree.hpp:101-152 ContractionTree::is_valid(check_shared):
 * for each internal node (c0^c1) subset of out subset of (c0|c1), and with
 * check_shared_inds c0 intersects c1. 0 = valid. */
int orc_ctree_is_valid(int32_t N, int32_t W, const int32_t* left,
                       const int32_t* right, const int32_t* parent,
                       const uint64_t* inds, int check_shared) {
  int rc = orc_tree_is_valid(N, left, right, parent);
  if (rc) return rc;
  for (int32_t i = 0; i < N; ++i) {
    if (left[i] < 0) continue;
    const uint64_t *a = inds + (size_t)left[i] * W, *b = inds + (size_t)right[i] * W,
                   *o = inds + (size_t)i * W;
    if (check_shared && !bs_intersects(a, b, W)) return 10;
    for (int w = 0; w < W; ++w) {
      if ((a[w] ^ b[w]) & ~o[w]) return 11;
      if (o[w] & ~(a[w] | b[w])) return 11;
    }
  }
  return 0;
}
#endif /* ORC_COMMON_DEFINED */

/* ------------------------------------------------------------------------ */
/* Problem / replica state                                                  */
/* ------------------------------------------------------------------------ */
typedef struct PFX(state) {
  int32_t n_leaves, N, n_inds, W;
  int32_t *left, *right, *parent;       /* [N] current tree              */
  uint64_t* inds;                       /* [N][W]                        */
  uint64_t* hyper;                      /* [N][W] HyperCache             */
  cost_t *ccost, *partial;              /* [N] CostCache                 */
  int32_t *min_left, *min_right, *min_parent;
  uint64_t* min_inds;
  cost_t min_total_cost;
  /* dims: uniform (dims_vec==NULL) or per index */
  uint64_t dim_uniform;
  uint64_t* dims_vec;                   /* [n_inds] or NULL              */
  /* cost model: simple, or simple_sparse_inds when sparse != NULL */
  uint64_t* sparse;                     /* [W] or NULL                   */
  uint64_t n_projs;
  int disable_shared_inds;
  orc_mt_t prng;
  uint64_t n_moves, n_accepted, n_improved;
  uint64_t* tmp;                        /* [4*W] scratch                 */
  /* ---- finite width (max_width >= 0 activates) ---- */
  int fw;
  double max_width_d;                   /* stored in width_type precision */
  int width_f32;                        /* 1: width_type float, 0: double */
  double* width;                        /* [N] WidthCache (value in width_type) */
  uint64_t *slices, *min_slices, *skip_slices; /* [W], skip may be NULL */
  uint64_t max_number_new_slices;
} PFX(state_t);

/* include/tnco/optimize/infinite_memory/cost_model/simple.hpp:37-55 get_cost:
 * scalar dims -> std::pow(size_t,size_t) = double pow, converted to cost_type
 * on return; vector dims -> running product in cost_type over ascending bits
 * (Bitset::visit order, include/tnco/bitset.hpp:95-101). */
static cost_t PFX(get_cost)(const PFX(state_t)* o, const uint64_t* m) {
  if (!o->dims_vec) {
    return (cost_t)pow((double)o->dim_uniform, (double)bs_count(m, o->W));
  }
  cost_t c = 1;
  for (int w = 0; w < o->W; ++w) {
    uint64_t x = m[w];
    while (x) {
      int b = __builtin_ctzll(x);
      c *= (cost_t)o->dims_vec[(size_t)w * 64 + b];
      x &= x - 1;
    }
  }
  return c;
}

/* simple.hpp:66-83 contraction_cost (inds_out ignored) and
 * simple_sparse_inds.hpp:37-49,66-86: cost(inds - S) * min(cost(inds & S),
 * n_projs) with the comparison done in cost_type.
 * Finite width (finite_width/cost_model/simple.hpp:127-147,
 * simple_sparse_inds.hpp:141-165): inds = in1 | in2 | slices. */
static cost_t PFX(ccost_of)(const PFX(state_t)* o, const uint64_t* a,
                            const uint64_t* b, const uint64_t* slices) {
  uint64_t* u = o->tmp + 2 * (size_t)o->W; /* tmp[2W..3W) */
  for (int w = 0; w < o->W; ++w)
    u[w] = a[w] | b[w] | (slices ? slices[w] : 0);
  if (!o->sparse) return PFX(get_cost)(o, u);
  uint64_t* v = o->tmp + 3 * (size_t)o->W; /* tmp[3W..4W) */
  for (int w = 0; w < o->W; ++w) v[w] = u[w] & ~o->sparse[w];
  cost_t c1 = PFX(get_cost)(o, v);
  for (int w = 0; w < o->W; ++w) v[w] = u[w] & o->sparse[w];
  cost_t c2 = PFX(get_cost)(o, v);
  cost_t np = (cost_t)o->n_projs;
  return c1 * (c2 < np ? c2 : np);
}

/* finite_width/cost_model/simple.hpp:38-57 get_width: scalar dims ->
 * log2(dims) * count (double arithmetic, converted to width_type on return);
 * vector dims -> running sum in width_type of log2(dims[p]) ascending.
 * simple_sparse_inds.hpp:38-52: width(inds-S) + min(width(inds&S), log2(np)).
 */
static inline double PFX(wround)(const PFX(state_t)* o, double x) {
  return o->width_f32 ? (double)(float)x : x;
}
static double PFX(get_width_simple)(const PFX(state_t)* o, const uint64_t* m) {
  if (!o->dims_vec) {
    return PFX(wround)(o, log2((double)o->dim_uniform) * (double)bs_count(m, o->W));
  }
  if (o->width_f32) {
    float wsum = 0;
    for (int w = 0; w < o->W; ++w) {
      uint64_t x = m[w];
      while (x) {
        int b = __builtin_ctzll(x);
        /* width_ += log2(dims[pos]): float += double -> computed in double,
         * rounded to float on assignment. */
        wsum = (float)((double)wsum + log2((double)o->dims_vec[(size_t)w * 64 + b]));
        x &= x - 1;
      }
    }
    return (double)wsum;
  }
  double wsum = 0;
  for (int w = 0; w < o->W; ++w) {
    uint64_t x = m[w];
    while (x) {
      int b = __builtin_ctzll(x);
      wsum += log2((double)o->dims_vec[(size_t)w * 64 + b]);
      x &= x - 1;
    }
  }
  return wsum;
}
static double PFX(get_width)(const PFX(state_t)* o, const uint64_t* m) {
  if (!o->sparse) return PFX(get_width_simple)(o, m);
  uint64_t* v = o->tmp + 3 * (size_t)o->W;
  for (int w = 0; w < o->W; ++w) v[w] = m[w] & ~o->sparse[w];
  double w1 = PFX(get_width_simple)(o, v);
  for (int w = 0; w < o->W; ++w) v[w] = m[w] & o->sparse[w];
  double w2 = PFX(get_width_simple)(o, v);
  /* min(x, y): x width_type, y = log2(size_t) double; compare in double,
   * both cast to width_type. */
  double l2 = log2((double)o->n_projs);
  double mn = (w2 < l2) ? w2 : PFX(wround)(o, l2);
  /* width_type + width_type */
  return PFX(wround)(o, w1 + mn);
}

/* include/tnco/optimize/infinite_memory/utils.hpp:31-57 CostCache ctor (with
 * slices for finite width: finite_width/utils.hpp:36-47) and :76-92 HyperCache
 * ctor. */
static void PFX(build_cost_cache)(const PFX(state_t)* o, const int32_t* left,
                                  const int32_t* right, const uint64_t* inds,
                                  const uint64_t* slices, cost_t* ccost,
                                  cost_t* partial) {
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)o->N);
  int k = orc_traverse(o->N, left, right, order);
  for (int i = 0; i < k; ++i) {
    int32_t pos = order[i];
    if (left[pos] < 0) {
      ccost[pos] = 0;
      partial[pos] = 0;
    } else {
      cost_t cA = PFX(ccost_of)(o, inds + (size_t)left[pos] * o->W,
                                inds + (size_t)right[pos] * o->W, slices);
      ccost[pos] = cA;
      partial[pos] = cA + partial[left[pos]] + partial[right[pos]];
    }
  }
  free(order);
}

/* infinite_memory/utils.hpp:102-116 get_cost: sum of contraction costs in
 * traverse order, accumulated in cost_type. */
static cost_t PFX(tree_cost)(const PFX(state_t)* o, const int32_t* left,
                             const int32_t* right, const uint64_t* inds,
                             const uint64_t* slices) {
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)o->N);
  int k = orc_traverse(o->N, left, right, order);
  cost_t total = 0;
  for (int i = 0; i < k; ++i) {
    int32_t pos = order[i];
    if (left[pos] >= 0)
      total += PFX(ccost_of)(o, inds + (size_t)left[pos] * o->W,
                             inds + (size_t)right[pos] * o->W, slices);
  }
  free(order);
  return total;
}

static void PFX(build_hyper_cache)(const PFX(state_t)* o, uint64_t* hyper) {
  for (int32_t pos = 0; pos < o->N; ++pos) {
    uint64_t* h = hyper + (size_t)pos * o->W;
    if (o->left[pos] < 0) {
      memset(h, 0, sizeof(uint64_t) * (size_t)o->W);
    } else {
      const uint64_t *x = o->inds + (size_t)pos * o->W,
                     *a = o->inds + (size_t)o->left[pos] * o->W,
                     *b = o->inds + (size_t)o->right[pos] * o->W;
      for (int w = 0; w < o->W; ++w) h[w] = x[w] & a[w] & b[w];
    }
  }
}

/* finite_width/greedy/utils.hpp:21-125 get_slices_impl. */
static void PFX(get_slices)(PFX(state_t)* o, uint64_t* out_slices);

/* ------------------------------------------------------------------------ */
/* Acceptance probabilities: include/tnco/optimize/prob/base.hpp:32-52,      */
/* greedy.hpp:33-47, mh.hpp:35-64.  kind: 0 base, 1 greedy, 2 MH.            */
/* ------------------------------------------------------------------------ */
cost_t PFX(prob)(int kind, double beta, cost_t delta, cost_t old_cost) {
  if (kind == 0) return 1;
  if (kind == 1) return delta <= 0 ? 1 : 0;
  if (delta <= 0) return (cost_t)1;
  if (old_cost == 0) return (cost_t)0;
  /* pow(cost_type, double) -> std::pow promotes to double; result converted
   * to cost_type by the return type (mh.hpp:58). */
  return (cost_t)pow((double)((cost_t)1 + (delta / old_cost)), -beta);
}

/* ------------------------------------------------------------------------ */
/* Construction: optimize/optimizer.hpp:57-82 (base: seed, min_ctree=ctree,  */
/* validity) + infinite_memory/optimizer.hpp:61-88 (caches, min_total_cost   */
/* via get_cost, "Precision is too low." check).                            */
/* Return: 0 ok; 1..11 invalid tree (orc_ctree_is_valid); 20 precision.      */
/* ------------------------------------------------------------------------ */
PFX(state_t)* PFX(create)(int32_t n_leaves, int32_t n_inds, const int32_t* left,
                          const int32_t* right, const int32_t* parent,
                          const uint64_t* inds, uint64_t dim_uniform,
                          const uint64_t* dims_vec, const uint64_t* sparse,
                          uint64_t n_projs, int disable_shared_inds,
                          uint64_t seed, const uint32_t* mt_state /* 625 or NULL */,
                          int* status) {
  PFX(state_t)* o = (PFX(state_t)*)calloc(1, sizeof(PFX(state_t)));
  const int32_t N = 2 * n_leaves - 1, W = (n_inds + 63) / 64 > 0 ? (n_inds + 63) / 64 : 1;
  o->n_leaves = n_leaves; o->N = N; o->n_inds = n_inds; o->W = W;
  size_t nb = sizeof(int32_t) * (size_t)N, mb = sizeof(uint64_t) * (size_t)N * W;
  o->left = (int32_t*)malloc(nb); o->right = (int32_t*)malloc(nb); o->parent = (int32_t*)malloc(nb);
  o->min_left = (int32_t*)malloc(nb); o->min_right = (int32_t*)malloc(nb); o->min_parent = (int32_t*)malloc(nb);
  o->inds = (uint64_t*)malloc(mb); o->min_inds = (uint64_t*)malloc(mb); o->hyper = (uint64_t*)malloc(mb);
  o->ccost = (cost_t*)malloc(sizeof(cost_t) * (size_t)N);
  o->partial = (cost_t*)malloc(sizeof(cost_t) * (size_t)N);
  o->tmp = (uint64_t*)calloc((size_t)4 * W, sizeof(uint64_t));
  memcpy(o->left, left, nb); memcpy(o->right, right, nb); memcpy(o->parent, parent, nb);
  memcpy(o->inds, inds, mb);
  o->dim_uniform = dim_uniform;
  if (dims_vec) {
    /* include/tnco/ctree.hpp:79-89: a vector of all-equal dims collapses to
     * the scalar form. */
    int all_eq = n_inds > 0;
    for (int i = 1; i < n_inds; ++i) all_eq &= dims_vec[i] == dims_vec[0];
    if (all_eq) {
      o->dim_uniform = dims_vec[0];
    } else {
      o->dims_vec = (uint64_t*)calloc((size_t)W * 64, sizeof(uint64_t));
      memcpy(o->dims_vec, dims_vec, sizeof(uint64_t) * (size_t)n_inds);
    }
  }
  if (sparse) {
    o->sparse = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)W);
    memcpy(o->sparse, sparse, sizeof(uint64_t) * (size_t)W);
    o->n_projs = n_projs;
  }
  o->disable_shared_inds = disable_shared_inds;
  if (mt_state) {
    memcpy(o->prng.x, mt_state, sizeof(uint32_t) * MT_N);
    o->prng.p = (int32_t)mt_state[MT_N];
  } else {
    orc_mt_seed(&o->prng, seed);
  }
  *status = orc_ctree_is_valid(N, W, o->left, o->right, o->parent, o->inds,
                               !disable_shared_inds);
  if (*status) return o;
  memcpy(o->min_left, left, nb); memcpy(o->min_right, right, nb); memcpy(o->min_parent, parent, nb);
  memcpy(o->min_inds, inds, mb);
  PFX(build_cost_cache)(o, o->left, o->right, o->inds, NULL, o->ccost, o->partial);
  PFX(build_hyper_cache)(o, o->hyper);
  o->min_total_cost = PFX(tree_cost)(o, o->min_left, o->min_right, o->min_inds, NULL);
  double l1 = log2((double)o->partial[N - 1]), l2 = log2((double)o->min_total_cost);
  if (isinf(l1) || isnan(l1) || isinf(l2) || isnan(l2)) *status = 20;
  return o;
}

void PFX(destroy)(PFX(state_t)* o) {
  if (!o) return;
  free(o->left); free(o->right); free(o->parent);
  free(o->min_left); free(o->min_right); free(o->min_parent);
  free(o->inds); free(o->min_inds); free(o->hyper);
  free(o->ccost); free(o->partial); free(o->tmp);
  free(o->dims_vec); free(o->sparse);
  free(o->width); free(o->slices); free(o->min_slices); free(o->skip_slices);
  free(o);
}

/* ------------------------------------------------------------------------ */
/* One sweep: include/tnco/optimize/infinite_memory/optimizer.hpp:90-221     */
/* with get_ctree_nn from include/tnco/optimize/optimizer.hpp:86-172.        */
/* ------------------------------------------------------------------------ */
void PFX(update)(PFX(state_t)* o, int prob_kind, double beta) {
  const int W = o->W;
  int32_t *left = o->left, *right = o->right, *parent = o->parent;
  /* optimizer.hpp:103 */
  int32_t pos_B = (int32_t)((uint64_t)orc_mt_next(&o->prng) % (uint64_t)o->n_leaves);
  /* :107 */
  pos_B = parent[pos_B];
  if (pos_B < 0) return;
  /* :112 */
  cost_t total_cost = o->partial[o->N - 1];
  for (;;) {
    /* optimize/optimizer.hpp:112-120 */
    if (parent[pos_B] < 0 || left[pos_B] < 0) break;
    /* :121-125 */
    const int32_t pos_A = parent[pos_B];
    int32_t pos_C = (left[pos_A] == pos_B) ? right[pos_A] : left[pos_A];
    const uint64_t* inds_C = o->inds + (size_t)pos_C * W;
    /* :128-144 */
    const int32_t pos_0 = left[pos_B], pos_1 = right[pos_B];
    const int inter_C0 = bs_intersects(o->inds + (size_t)pos_0 * W, inds_C, W);
    const int inter_C1 = bs_intersects(o->inds + (size_t)pos_1 * W, inds_C, W);
    int32_t pos_D, pos_E;
    if (o->disable_shared_inds || (inter_C0 && inter_C1)) {
      if (orc_mt_next(&o->prng) % 2) { pos_D = pos_0; pos_E = pos_1; }
      else { pos_D = pos_1; pos_E = pos_0; }
    } else if (inter_C0) { pos_D = pos_0; pos_E = pos_1; }
    else { pos_D = pos_1; pos_E = pos_0; }

    /* infinite_memory/optimizer.hpp:136-147 */
    const uint64_t* inds_A = o->inds + (size_t)pos_A * W;
    uint64_t* inds_B = o->inds + (size_t)pos_B * W;
    const uint64_t* inds_D = o->inds + (size_t)pos_D * W;
    const uint64_t* inds_E = o->inds + (size_t)pos_E * W;
    uint64_t* hyper_A = o->hyper + (size_t)pos_A * W;
    uint64_t* hyper_B = o->hyper + (size_t)pos_B * W;
    uint64_t* new_inds_B = o->tmp; /* tmp[0..W) */
    for (int w = 0; w < W; ++w)
      new_inds_B[w] = (inds_D[w] ^ inds_C[w]) | hyper_A[w] | hyper_B[w];

    /* :150-158 */
    const cost_t new_ccost_A = PFX(ccost_of)(o, new_inds_B, inds_E, NULL);
    const cost_t new_ccost_B = PFX(ccost_of)(o, inds_D, inds_C, NULL);
    const cost_t delta_cost =
        (new_ccost_B - o->ccost[pos_B]) + (new_ccost_A - o->ccost[pos_A]);
    o->n_moves++;

    /* :162 */
    const double u = orc_uniform01(&o->prng);
    if (u <= (double)PFX(prob)(prob_kind, beta, delta_cost, total_cost)) {
      /* :164 swap E<->C, :167 */
      orc_swap_with_nn(o->N, left, right, parent, pos_E);
      { int32_t t = pos_C; pos_C = pos_E; pos_E = t; }
      /* :170-172 -- inds_E / inds_C below are the PRE-swap references */
      for (int w = 0; w < W; ++w) {
        const uint64_t nb = new_inds_B[w];
        hyper_A[w] = inds_A[w] & nb & inds_E[w];
        hyper_B[w] = nb & inds_D[w] & inds_C[w];
        inds_B[w] = nb;
      }
      /* :175-177 */
      o->ccost[pos_B] = new_ccost_B;
      o->ccost[pos_A] = new_ccost_A;
      total_cost += delta_cost;
      o->n_accepted++;
    }
    /* :185-188 */
    o->partial[pos_B] = o->partial[pos_D] + o->partial[pos_E] + o->ccost[pos_B];
    o->partial[pos_A] = o->partial[pos_B] + o->partial[pos_C] + o->ccost[pos_A];
    /* :191 */
    pos_B = pos_A;
  }
  /* :198-201 */
  const cost_t tc = o->partial[o->N - 1];
  if (tc < o->min_total_cost) {
    o->min_total_cost = tc;
    size_t nb = sizeof(int32_t) * (size_t)o->N;
    memcpy(o->min_left, left, nb); memcpy(o->min_right, right, nb); memcpy(o->min_parent, parent, nb);
    memcpy(o->min_inds, o->inds, sizeof(uint64_t) * (size_t)o->N * W);
    o->n_improved++;
  }
}

/* Driver loop of tnco/app/infinite_memory/sa.py:199-209: one update per beta. */
void PFX(run)(PFX(state_t)* o, int prob_kind, const double* betas, int64_t n_steps) {
  for (int64_t i = 0; i < n_steps; ++i) PFX(update)(o, prob_kind, betas[i]);
}

/* infinite_memory/optimizer.hpp:223-251 is_valid: caches equal a from-scratch
 * rebuild (log-close within atol), hyper cache identical, min cost matches.
 * 0 = valid. */
int PFX(is_valid)(PFX(state_t)* o, double atol) {
  int rc = orc_ctree_is_valid(o->N, o->W, o->left, o->right, o->parent, o->inds,
                              !o->disable_shared_inds);
  if (rc) return rc;
  rc = orc_ctree_is_valid(o->N, o->W, o->min_left, o->min_right, o->min_parent,
                          o->min_inds, !o->disable_shared_inds);
  if (rc) return 100 + rc;
  const uint64_t* sl = o->fw ? o->slices : NULL;
  const uint64_t* msl = o->fw ? o->min_slices : NULL;
  cost_t mc = PFX(tree_cost)(o, o->min_left, o->min_right, o->min_inds, msl);
#define LOGCLOSE(x, y) (((x) < 0 || (y) < 0) ? 0 : (((x) == 0 || (y) == 0) ? ((x) == (y)) : (fabs(log((double)(x)) - log((double)(y))) <= atol)))
  if (!LOGCLOSE(mc, o->min_total_cost)) return 30;
  cost_t* cc = (cost_t*)malloc(sizeof(cost_t) * (size_t)o->N);
  cost_t* pc = (cost_t*)malloc(sizeof(cost_t) * (size_t)o->N);
  PFX(build_cost_cache)(o, o->left, o->right, o->inds, sl, cc, pc);
  int bad = 0;
  for (int32_t i = 0; i < o->N; ++i) {
    if (!LOGCLOSE(cc[i], o->ccost[i])) bad = 31;
    if (!LOGCLOSE(pc[i], o->partial[i])) bad = 32;
  }
  free(cc); free(pc);
  if (bad) return bad;
  uint64_t* hy = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)o->N * o->W);
  PFX(build_hyper_cache)(o, hy);
  if (memcmp(hy, o->hyper, sizeof(uint64_t) * (size_t)o->N * o->W)) bad = 33;
  free(hy);
  if (bad) return bad;
  if (o->fw) {
    /* finite_width/greedy/optimizer.hpp:406-423: every tensor fits after
     * slicing; :438-442 WidthCache. */
    for (int32_t i = 0; i < o->N; ++i) {
      for (int w = 0; w < o->W; ++w) o->tmp[w] = o->inds[(size_t)i * o->W + w] & ~o->slices[w];
      if (PFX(get_width)(o, o->tmp) > o->max_width_d) return 40;
      for (int w = 0; w < o->W; ++w) o->tmp[w] = o->min_inds[(size_t)i * o->W + w] & ~o->min_slices[w];
      if (PFX(get_width)(o, o->tmp) > o->max_width_d) return 41;
      if (fabs(PFX(get_width)(o, o->inds + (size_t)i * o->W) - o->width[i]) > atol) return 42;
    }
  }
  return 0;
#undef LOGCLOSE
}

/* ------------------------------------------------------------------------ */
/* Finite width twin: finite_width/greedy/optimizer.hpp.                     */
/* ------------------------------------------------------------------------ */

/* finite_width/cost_model/simple.hpp:59-76 get_delta_width and
 * simple_sparse_inds.hpp:54-82. */
static double PFX(delta_width)(PFX(state_t)* o, const uint64_t* m, int32_t pos) {
  const int test = (int)((m[pos >> 6] >> (pos & 63)) & 1);
  if (o->sparse && ((o->sparse[pos >> 6] >> (pos & 63)) & 1)) {
    uint64_t* v = o->tmp + 2 * (size_t)o->W;
    uint64_t* nv = o->tmp + 3 * (size_t)o->W;
    for (int w = 0; w < o->W; ++w) { v[w] = m[w] & o->sparse[w]; nv[w] = v[w]; }
    nv[pos >> 6] ^= (1ull << (pos & 63));
    double l2 = log2((double)o->n_projs);
    double a = PFX(get_width_simple)(o, nv), b = PFX(get_width_simple)(o, v);
    double ma = (a < l2) ? a : PFX(wround)(o, l2);
    double mb = (b < l2) ? b : PFX(wround)(o, l2);
    return PFX(wround)(o, ma - mb);
  }
  /* (1 - 2*test) * log2(dims): int * double -> double -> width_type */
  double d = o->dims_vec ? (double)o->dims_vec[pos] : (double)o->dim_uniform;
  return PFX(wround)(o, (double)(1 - (2 * test)) * log2(d));
}

typedef struct { int32_t pos; uint64_t nbig; double l2d; int vec; } PFX(sitem_t);

/* comparator `greater` of greedy/utils.hpp:50-60 */
static int PFX(greater)(const PFX(sitem_t)* x, const PFX(sitem_t)* y) {
  if (!x->vec) return x->nbig > y->nbig;
  return x->nbig == y->nbig ? (x->l2d > y->l2d) : (x->nbig > y->nbig);
}

/* std::stable_sort with comparator `greater`: any stable sort yields the same
 * permutation; insertion sort (sizes are <= n_inds). */
static void PFX(stable_sort)(PFX(sitem_t)* a, int n) {
  for (int i = 1; i < n; ++i) {
    PFX(sitem_t) key = a[i];
    int j = i - 1;
    while (j >= 0 && PFX(greater)(&key, &a[j])) { a[j + 1] = a[j]; --j; }
    a[j + 1] = key;
  }
}

static void PFX(get_slices)(PFX(state_t)* o, uint64_t* slices) {
  const int W = o->W;
  memset(slices, 0, sizeof(uint64_t) * (size_t)W);
  /* greedy/utils.hpp:41-48 */
  uint64_t* n_big = (uint64_t*)calloc((size_t)W * 64, sizeof(uint64_t));
  for (int32_t t = 0; t < o->N; ++t) {
    if (o->width[t] > o->max_width_d) {
      for (int w = 0; w < W; ++w) {
        uint64_t x = o->inds[(size_t)t * W + w];
        while (x) { n_big[(size_t)w * 64 + __builtin_ctzll(x)]++; x &= x - 1; }
      }
    }
  }
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)o->N);
  int k = orc_traverse(o->N, o->left, o->right, order);
  uint64_t* sliced_xs = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)W);
  int32_t* positions = (int32_t*)malloc(sizeof(int32_t) * (size_t)W * 64);
  PFX(sitem_t)* items = (PFX(sitem_t)*)malloc(sizeof(PFX(sitem_t)) * (size_t)W * 64);
  for (int i = 0; i < k; ++i) {
    const int32_t t = order[i];
    if (!(o->width[t] > o->max_width_d)) continue;
    /* :64-70 */
    for (int w = 0; w < W; ++w) sliced_xs[w] = o->inds[(size_t)t * W + w] & ~slices[w];
    double sliced_width = PFX(get_width)(o, sliced_xs);
    if (!(sliced_width > o->max_width_d)) continue;
    /* :72-77 positions of (sliced_xs - skip_slices), ascending */
    int np = 0;
    for (int w = 0; w < W; ++w) {
      uint64_t x = sliced_xs[w] & ~(o->skip_slices ? o->skip_slices[w] : 0);
      while (x) { positions[np++] = w * 64 + __builtin_ctzll(x); x &= x - 1; }
    }
    /* :80 */
    orc_shuffle_i32(&o->prng, positions, np);
    /* :83 */
    for (int j = 0; j < np; ++j) {
      items[j].pos = positions[j];
      items[j].nbig = n_big[positions[j]];
      items[j].vec = o->dims_vec != NULL;
      /* DimsCache log2_dims in width_type (finite_width/utils.hpp:91-108) */
      items[j].l2d = o->dims_vec ? PFX(wround)(o, log2((double)o->dims_vec[positions[j]])) : 0;
    }
    PFX(stable_sort)(items, np);
    /* :86-101 */
    for (int j = 0; j < np; ++j) {
      const int32_t xpos = items[j].pos;
      slices[xpos >> 6] |= 1ull << (xpos & 63);
      sliced_width = PFX(wround)(o, sliced_width + PFX(delta_width)(o, sliced_xs, xpos));
      sliced_xs[xpos >> 6] &= ~(1ull << (xpos & 63));
      if (sliced_width <= o->max_width_d) break;
    }
  }
  free(n_big); free(order); free(sliced_xs); free(positions); free(items);
}

/* finite_width/greedy/optimizer.hpp:72-115 ctor.  Member-init order (L61-70):
 * width_cache, dims_cache, skip_slices, slices (draws from prng unless
 * given), min_slices, cost_cache(slices), hyper_cache, min_total_cost. */
PFX(state_t)* PFX(create_fw)(int32_t n_leaves, int32_t n_inds, const int32_t* left,
                             const int32_t* right, const int32_t* parent,
                             const uint64_t* inds, uint64_t dim_uniform,
                             const uint64_t* dims_vec, const uint64_t* sparse,
                             uint64_t n_projs, int disable_shared_inds,
                             uint64_t seed, const uint32_t* mt_state,
                             double max_width, int width_f32,
                             uint64_t max_number_new_slices,
                             const uint64_t* skip_slices,
                             const uint64_t* slices_in, int* status) {
  PFX(state_t)* o = PFX(create)(n_leaves, n_inds, left, right, parent, inds,
                                dim_uniform, dims_vec, sparse, n_projs,
                                disable_shared_inds, seed, mt_state, status);
  if (*status && *status != 20) return o;
  *status = 0;
  const int W = o->W;
  o->fw = 1;
  o->width_f32 = width_f32;
  o->max_width_d = width_f32 ? (double)(float)max_width : max_width;
  o->max_number_new_slices = max_number_new_slices;
  o->width = (double*)malloc(sizeof(double) * (size_t)o->N);
  for (int32_t i = 0; i < o->N; ++i) o->width[i] = PFX(get_width)(o, o->inds + (size_t)i * W);
  if (skip_slices) {
    o->skip_slices = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)W);
    memcpy(o->skip_slices, skip_slices, sizeof(uint64_t) * (size_t)W);
  }
  o->slices = (uint64_t*)calloc((size_t)W, sizeof(uint64_t));
  o->min_slices = (uint64_t*)calloc((size_t)W, sizeof(uint64_t));
  if (slices_in) memcpy(o->slices, slices_in, sizeof(uint64_t) * (size_t)W);
  else PFX(get_slices)(o, o->slices);
  memcpy(o->min_slices, o->slices, sizeof(uint64_t) * (size_t)W);
  PFX(build_cost_cache)(o, o->left, o->right, o->inds, o->slices, o->ccost, o->partial);
  o->min_total_cost = PFX(tree_cost)(o, o->min_left, o->min_right, o->min_inds, o->min_slices);
  double l1 = log2((double)o->partial[o->N - 1]), l2 = log2((double)o->min_total_cost);
  if (isinf(l1) || isnan(l1) || isinf(l2) || isnan(l2)) *status = 20;
  return o;
}

/* The `min_ctree` (and, finite width, `min_slices`) constructor arguments, applied to a state
 * constructed without them -- nothing else of the constructors reads these members:
 * optimize/optimizer.hpp:57-65 (min_ctree{min_ctree.has_value() ? *min_ctree : ctree}),
 * infinite_memory/optimizer.hpp:61-88 and finite_width/greedy/optimizer.hpp:72-115
 * (min_slices{min_slices.has_value() ? *min_slices : slices}; min_total_cost =
 * get_cost(min_ctree[, min_slices]); "Precision is too low." on log2 of it; is_valid of the
 * tree).  These are what Optimizer.__reduce__ round-trips
 * (tnco/optimize/infinite_memory/optimizer.py:243-245, finite_width/optimizer.py:343-346).
 * min_slices == NULL: unchanged.  Return: 0 ok; 1..11 invalid tree; 20 precision. */
int PFX(set_min)(PFX(state_t)* o, const int32_t* min_left, const int32_t* min_right,
                 const int32_t* min_parent, const uint64_t* min_inds, const uint64_t* min_slices) {
  const size_t nb = sizeof(int32_t) * (size_t)o->N, mb = sizeof(uint64_t) * (size_t)o->N * o->W;
  int rc = orc_ctree_is_valid(o->N, o->W, min_left, min_right, min_parent, min_inds,
                              !o->disable_shared_inds);
  if (rc) return rc;
  memcpy(o->min_left, min_left, nb); memcpy(o->min_right, min_right, nb); memcpy(o->min_parent, min_parent, nb);
  memcpy(o->min_inds, min_inds, mb);
  if (o->fw && min_slices) memcpy(o->min_slices, min_slices, sizeof(uint64_t) * (size_t)o->W);
  o->min_total_cost = PFX(tree_cost)(o, o->min_left, o->min_right, o->min_inds, o->fw ? o->min_slices : NULL);
  const double l2 = log2((double)o->min_total_cost);
  if (isinf(l2) || isnan(l2)) return 20;
  return 0;
}

/* finite_width/greedy/optimizer.hpp:117-390 update(prob, update_slices). */
void PFX(update_fw)(PFX(state_t)* o, int prob_kind, double beta, int update_slices) {
  const int W = o->W;
  int32_t *left = o->left, *right = o->right, *parent = o->parent;
  int32_t pos_B = (int32_t)((uint64_t)orc_mt_next(&o->prng) % (uint64_t)o->n_leaves);
  pos_B = parent[pos_B];
  if (pos_B < 0) return;
  cost_t total_cost = o->partial[o->N - 1];
  uint64_t* new_inds_B = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)W);
  uint64_t* tmpm = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)W);
  for (;;) {
    if (parent[pos_B] < 0 || left[pos_B] < 0) break;
    const int32_t pos_A = parent[pos_B];
    int32_t pos_C = (left[pos_A] == pos_B) ? right[pos_A] : left[pos_A];
    const uint64_t* inds_C = o->inds + (size_t)pos_C * W;
    const int32_t pos_0 = left[pos_B], pos_1 = right[pos_B];
    const int inter_C0 = bs_intersects(o->inds + (size_t)pos_0 * W, inds_C, W);
    const int inter_C1 = bs_intersects(o->inds + (size_t)pos_1 * W, inds_C, W);
    int32_t pos_D, pos_E;
    if (o->disable_shared_inds || (inter_C0 && inter_C1)) {
      if (orc_mt_next(&o->prng) % 2) { pos_D = pos_0; pos_E = pos_1; }
      else { pos_D = pos_1; pos_E = pos_0; }
    } else if (inter_C0) { pos_D = pos_0; pos_E = pos_1; }
    else { pos_D = pos_1; pos_E = pos_0; }

    const uint64_t* inds_A = o->inds + (size_t)pos_A * W;
    uint64_t* inds_B = o->inds + (size_t)pos_B * W;
    const uint64_t* inds_D = o->inds + (size_t)pos_D * W;
    const uint64_t* inds_E = o->inds + (size_t)pos_E * W;
    uint64_t* hyper_A = o->hyper + (size_t)pos_A * W;
    uint64_t* hyper_B = o->hyper + (size_t)pos_B * W;
    /* :174-179 */
    for (int w = 0; w < W; ++w)
      new_inds_B[w] = (inds_D[w] ^ inds_C[w]) | hyper_A[w] | hyper_B[w];
    const double new_width_B = PFX(get_width)(o, new_inds_B);
    for (int w = 0; w < W; ++w) tmpm[w] = new_inds_B[w] & ~o->slices[w];
    double new_sliced_width_B = PFX(get_width)(o, tmpm);
    int skip_cost_propagation = 0;
    o->n_moves++;

    if (new_sliced_width_B <= o->max_width_d) {
      /* :190-224 */
      const cost_t new_ccost_A = PFX(ccost_of)(o, new_inds_B, inds_E, o->slices);
      const cost_t new_ccost_B = PFX(ccost_of)(o, inds_D, inds_C, o->slices);
      const cost_t delta_cost =
          (new_ccost_B - o->ccost[pos_B]) + (new_ccost_A - o->ccost[pos_A]);
      const double u = orc_uniform01(&o->prng);
      if (u <= (double)PFX(prob)(prob_kind, beta, delta_cost, total_cost)) {
        orc_swap_with_nn(o->N, left, right, parent, pos_E);
        for (int w = 0; w < W; ++w) {
          const uint64_t nb = new_inds_B[w];
          hyper_A[w] = inds_A[w] & nb & inds_E[w];
          hyper_B[w] = nb & inds_D[w] & inds_C[w];
          inds_B[w] = nb;
        }
        { int32_t t = pos_C; pos_C = pos_E; pos_E = t; }
        o->ccost[pos_B] = new_ccost_B;
        o->ccost[pos_A] = new_ccost_A;
        total_cost += delta_cost;
        o->width[pos_B] = new_width_B;
        o->n_accepted++;
      }
    } else if (o->max_number_new_slices > 0) {
      /* :226-321 random extra slices + full cost-cache rebuild */
      uint64_t* new_slices = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)W);
      memcpy(new_slices, o->slices, sizeof(uint64_t) * (size_t)W);
      int32_t* pos = (int32_t*)malloc(sizeof(int32_t) * (size_t)W * 64);
      uint64_t n_pos = 0, n_new = 0;
      for (int w = 0; w < W; ++w) {
        uint64_t x = new_inds_B[w] & ~o->slices[w] & ~(o->skip_slices ? o->skip_slices[w] : 0);
        while (x) { pos[n_pos++] = w * 64 + __builtin_ctzll(x); x &= x - 1; }
      }
      while (n_new < o->max_number_new_slices && new_sliced_width_B > o->max_width_d) {
        /* :245 prng() % n_pos (reference has UB if n_pos == 0; not reached
         * for valid inputs) */
        uint64_t j = (uint64_t)orc_mt_next(&o->prng) % n_pos;
        int32_t t = pos[j]; pos[j] = pos[n_pos - 1]; pos[n_pos - 1] = t;
        const int32_t xp = pos[n_pos - 1];
        new_slices[xp >> 6] |= 1ull << (xp & 63);
        double l2d = o->dims_vec ? PFX(wround)(o, log2((double)o->dims_vec[xp]))
                                 : PFX(wround)(o, log2((double)o->dim_uniform));
        new_sliced_width_B = PFX(wround)(o, new_sliced_width_B - l2d);
        --n_pos; ++n_new;
      }
      if (new_sliced_width_B <= o->max_width_d) {
        /* :287-290 swap inds, rotate, rebuild cost cache with new slices */
        for (int w = 0; w < W; ++w) { uint64_t t = inds_B[w]; inds_B[w] = new_inds_B[w]; new_inds_B[w] = t; }
        orc_swap_with_nn(o->N, left, right, parent, pos_E);
        cost_t* ncc = (cost_t*)malloc(sizeof(cost_t) * (size_t)o->N);
        cost_t* npc = (cost_t*)malloc(sizeof(cost_t) * (size_t)o->N);
        PFX(build_cost_cache)(o, left, right, o->inds, new_slices, ncc, npc);
        const cost_t delta_cost = npc[o->N - 1] - total_cost;
        const double u = orc_uniform01(&o->prng);
        if (u <= (double)PFX(prob)(prob_kind, beta, delta_cost, total_cost)) {
          memcpy(o->ccost, ncc, sizeof(cost_t) * (size_t)o->N);
          memcpy(o->partial, npc, sizeof(cost_t) * (size_t)o->N);
          /* :300-301 (pos_C/pos_E not swapped in this branch; inds_E, inds_C
           * are the pre-rotation references) */
          for (int w = 0; w < W; ++w) {
            hyper_A[w] = inds_A[w] & inds_B[w] & inds_E[w];
            hyper_B[w] = inds_B[w] & inds_D[w] & inds_C[w];
          }
          o->width[pos_B] = new_width_B;
          total_cost = o->partial[o->N - 1];
          memcpy(o->slices, new_slices, sizeof(uint64_t) * (size_t)W);
          skip_cost_propagation = 1;
          o->n_accepted++;
        } else {
          /* :317-318 */
          orc_swap_with_nn(o->N, left, right, parent, pos_C);
          for (int w = 0; w < W; ++w) { uint64_t t = inds_B[w]; inds_B[w] = new_inds_B[w]; new_inds_B[w] = t; }
        }
        free(ncc); free(npc);
      }
      free(new_slices); free(pos);
    }
    /* :324-331 */
    if (!skip_cost_propagation) {
      o->partial[pos_B] = o->partial[pos_D] + o->partial[pos_E] + o->ccost[pos_B];
      o->partial[pos_A] = o->partial[pos_B] + o->partial[pos_C] + o->ccost[pos_A];
    }
    pos_B = pos_A;
  }
  free(new_inds_B); free(tmpm);

  /* :360-376 */
  if (update_slices && bs_any(o->slices, W)) {
    uint64_t* new_slices = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)W);
    PFX(get_slices)(o, new_slices);
    cost_t* ncc = (cost_t*)malloc(sizeof(cost_t) * (size_t)o->N);
    cost_t* npc = (cost_t*)malloc(sizeof(cost_t) * (size_t)o->N);
    PFX(build_cost_cache)(o, left, right, o->inds, new_slices, ncc, npc);
    if (npc[o->N - 1] < o->partial[o->N - 1]) {
      memcpy(o->slices, new_slices, sizeof(uint64_t) * (size_t)W);
      memcpy(o->ccost, ncc, sizeof(cost_t) * (size_t)o->N);
      memcpy(o->partial, npc, sizeof(cost_t) * (size_t)o->N);
    }
    free(new_slices); free(ncc); free(npc);
  }
  /* :385-389 */
  const cost_t tc = o->partial[o->N - 1];
  if (tc < o->min_total_cost) {
    o->min_total_cost = tc;
    size_t nb = sizeof(int32_t) * (size_t)o->N;
    memcpy(o->min_left, left, nb); memcpy(o->min_right, right, nb); memcpy(o->min_parent, parent, nb);
    memcpy(o->min_inds, o->inds, sizeof(uint64_t) * (size_t)o->N * W);
    memcpy(o->min_slices, o->slices, sizeof(uint64_t) * (size_t)W);
    o->n_improved++;
  }
}

/* Driver loop of tnco/app/finite_width/sa.py:219-233:
 * update_slices = (n % update_slices_every == 0). */
void PFX(run_fw)(PFX(state_t)* o, int prob_kind, const double* betas, int64_t n_steps,
                 int64_t update_slices_every) {
  for (int64_t i = 0; i < n_steps; ++i)
    PFX(update_fw)(o, prob_kind, betas[i],
                   update_slices_every > 0 ? (i % update_slices_every == 0) : 0);
}

/* ------------------------------------------------------------------------ */
/* Accessors for the ctypes wrapper                                         */
/* ------------------------------------------------------------------------ */
void PFX(get_tree)(const PFX(state_t)* o, int which_min, int32_t* left, int32_t* right,
                   int32_t* parent, uint64_t* inds) {
  size_t nb = sizeof(int32_t) * (size_t)o->N;
  memcpy(left, which_min ? o->min_left : o->left, nb);
  memcpy(right, which_min ? o->min_right : o->right, nb);
  memcpy(parent, which_min ? o->min_parent : o->parent, nb);
  if (inds) memcpy(inds, which_min ? o->min_inds : o->inds, sizeof(uint64_t) * (size_t)o->N * o->W);
}
void PFX(get_caches)(const PFX(state_t)* o, double* ccost, double* partial, uint64_t* hyper) {
  for (int32_t i = 0; i < o->N; ++i) { ccost[i] = (double)o->ccost[i]; partial[i] = (double)o->partial[i]; }
  if (hyper) memcpy(hyper, o->hyper, sizeof(uint64_t) * (size_t)o->N * o->W);
}
double PFX(total_cost)(const PFX(state_t)* o) { return (double)o->partial[o->N - 1]; }
double PFX(min_total_cost)(const PFX(state_t)* o) { return (double)o->min_total_cost; }
void PFX(get_prng)(const PFX(state_t)* o, uint32_t* out625) {
  memcpy(out625, o->prng.x, sizeof(uint32_t) * MT_N);
  out625[MT_N] = (uint32_t)o->prng.p;
}
void PFX(get_counters)(const PFX(state_t)* o, uint64_t* out3) {
  out3[0] = o->n_moves; out3[1] = o->n_accepted; out3[2] = o->n_improved;
}
void PFX(get_slices_out)(const PFX(state_t)* o, uint64_t* slices, uint64_t* min_slices) {
  if (!o->fw) return;
  memcpy(slices, o->slices, sizeof(uint64_t) * (size_t)o->W);
  memcpy(min_slices, o->min_slices, sizeof(uint64_t) * (size_t)o->W);
}
void PFX(get_widths)(const PFX(state_t)* o, double* width) {
  if (!o->fw) return;
  memcpy(width, o->width, sizeof(double) * (size_t)o->N);
}

/* ------------------------------------------------------------------------ */
/* Batch driver used by bench.py's cpu_baseline leg: R independent replicas  */
/* over OpenMP threads (the reference's n_runs over joblib processes,        */
/* tnco/parallel.py:330-341).  Only the update loops are timed; tree         */
/* flattening (tnco/ctree.py:163-189 rule) and cache construction are setup. */
/* ------------------------------------------------------------------------ */
#ifndef ORC_DERIVE_DEFINED
#define ORC_DERIVE_DEFINED
#include <omp.h>
/* legs of every node from the leaves: z = (x ^ y) | (x & y & outside(z)),
 * outside(z) = output legs + legs of leaves not below z (tnco/ctree.py:163-189
 * hyper-count bookkeeping restated as sets). */
void orc_derive_inds(int32_t N, int32_t W, const int32_t* left, const int32_t* right,
                     const uint64_t* leaf_masks, const uint64_t* output_mask, uint64_t* inds) {
  const int32_t n = (N + 1) / 2;
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
  uint64_t* uni = (uint64_t*)calloc((size_t)N * W, sizeof(uint64_t));
  uint64_t* outside = (uint64_t*)calloc((size_t)N * W, sizeof(uint64_t));
  int k = orc_traverse(N, left, right, order);
  memcpy(uni, leaf_masks, sizeof(uint64_t) * (size_t)n * W);
  memcpy(inds, leaf_masks, sizeof(uint64_t) * (size_t)n * W);
  for (int i = 0; i < k; ++i) {
    int32_t p = order[i];
    if (left[p] >= 0)
      for (int w = 0; w < W; ++w) uni[(size_t)p * W + w] = uni[(size_t)left[p] * W + w] | uni[(size_t)right[p] * W + w];
  }
  if (output_mask) memcpy(outside + (size_t)(N - 1) * W, output_mask, sizeof(uint64_t) * (size_t)W);
  for (int i = k - 1; i >= 0; --i) {
    int32_t p = order[i];
    if (left[p] < 0) continue;
    for (int w = 0; w < W; ++w) {
      outside[(size_t)left[p] * W + w] = outside[(size_t)p * W + w] | uni[(size_t)right[p] * W + w];
      outside[(size_t)right[p] * W + w] = outside[(size_t)p * W + w] | uni[(size_t)left[p] * W + w];
    }
  }
  for (int i = 0; i < k; ++i) {
    int32_t p = order[i];
    if (left[p] < 0) continue;
    for (int w = 0; w < W; ++w) {
      uint64_t a = inds[(size_t)left[p] * W + w], b = inds[(size_t)right[p] * W + w];
      inds[(size_t)p * W + w] = (a ^ b) | (a & b & outside[(size_t)p * W + w]);
    }
  }
  free(order); free(uni); free(outside);
}
#endif

double PFX(run_batch)(int64_t R, int32_t n_leaves, int32_t n_inds, const int32_t* links,
                      const uint64_t* leaf_masks, const uint64_t* output_mask, uint64_t dim_uniform,
                      const uint32_t* seeds, int prob_kind, const double* betas, int64_t n_steps,
                      int n_threads, double* out_total, double* out_min, uint64_t* out_moves) {
  const int32_t N = 2 * n_leaves - 1, W = (n_inds + 63) / 64 > 0 ? (n_inds + 63) / 64 : 1;
  /* chunks of CHUNK replicas: bounded memory (a state is ~0.3 MB at 512 leaves) however large the
   * sample; only the update loops are timed */
  enum { CHUNK = 2048 };
  PFX(state_t)** st = (PFX(state_t)**)calloc((size_t)CHUNK, sizeof(void*));
  if (n_threads > 0) omp_set_num_threads(n_threads);
  double dt = 0;
  for (int64_t r0 = 0; r0 < R; r0 += CHUNK) {
    const int64_t cnt = (R - r0 < CHUNK) ? R - r0 : CHUNK;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i = 0; i < cnt; ++i) {
      const int64_t r = r0 + i;
      const int32_t* lk = links + r * 3 * (int64_t)N;
      uint64_t* inds = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N * W);
      orc_derive_inds(N, W, lk, lk + N, leaf_masks, output_mask, inds);
      int status = 0;
      st[i] = PFX(create)(n_leaves, n_inds, lk, lk + N, lk + 2 * (int64_t)N, inds, dim_uniform, NULL, NULL, 0, 0,
                          seeds[r], NULL, &status);
      free(inds);
      if (status) { PFX(destroy)(st[i]); st[i] = NULL; }
    }
    const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i = 0; i < cnt; ++i)
      if (st[i]) PFX(run)(st[i], prob_kind, betas, n_steps);
    dt += omp_get_wtime() - t0;
    for (int64_t i = 0; i < cnt; ++i) {
      const int64_t r = r0 + i;
      if (!st[i]) { if (out_total) out_total[r] = -1; continue; }
      if (out_total) out_total[r] = (double)st[i]->partial[N - 1];
      if (out_min) out_min[r] = (double)st[i]->min_total_cost;
      if (out_moves) out_moves[r] = st[i]->n_moves;
      PFX(destroy)(st[i]);
      st[i] = NULL;
    }
  }
  free(st);
  return dt;
}
