// greedy_key.h -- an order-preserving integer key for opt_einsum's greedy cost with every dimension 2,
//   v = 2^a - 2^b - 2^c   (a = |k12|, b = |k1|, c = |k2|; paths.py: _simple_chooser, 'memory-removed'),
// shared by the host generator (host_greedy.cpp) and the device one (greedy_device.hip).
//
// v is written in non-adjacent form (signed binary digits, no two neighbours nonzero): at most three
// digits, and for two numbers in that form the most significant digit in which they differ decides
// (the digits below position p are worth less than 2^(p+1) / 3 in absolute value).  So with digit k
// (k = 0 the most significant) encoded as s_k * (p_k + 1), 0 when absent, the triples compare like the
// numbers.  Positions stay below 2047 (a, b, c <= GREEDY_KEY_MAX_EXP): 12 bits per digit.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define TNCO_HD __host__ __device__
#else
#define TNCO_HD
#endif

namespace tnco {

constexpr int GREEDY_KEY_MAX_EXP = 2040;

// 36-bit key, larger key <=> larger v.  (Three digits in named variables, no indexed arrays: on the
// device an indexed array is scratch memory, and this runs once per candidate in every lane.)
TNCO_HD inline uint64_t greedy_cost_key(int a, int b, int c) {
  constexpr int GONE = -(1 << 20);  // position of an absent digit (sorts last, never equal / adjacent to a live one)
  int p0 = a, p1 = b, p2 = c;
  int s0 = 1, s1 = -1, s2 = -1;
#define TNCO_GK_CX(pa, sa, pb, sb) \
  if (pa < pb) {                   \
    const int tp_ = pa, ts_ = sa;  \
    pa = pb; sa = sb;              \
    pb = tp_; sb = ts_;            \
  }
  // (pa, sa), (pb, sb) with pa == pb or pa == pb + 1, both live -> their sum in non-adjacent form
#define TNCO_GK_JOIN(pa, sa, pb, sb)                                             \
  if (pa == pb) {                                                                \
    if (sa == sb) { pa += 1; } else { pa = GONE; sa = 0; }                       \
    pb = GONE; sb = 0;                                                           \
  } else {                                                                       \
    if (sa == sb) { pa += 1; sb = -sb; } else { pa = pb; pb = GONE; sb = 0; }    \
  }
  for (;;) {
    TNCO_GK_CX(p0, s0, p1, s1)
    TNCO_GK_CX(p1, s1, p2, s2)
    TNCO_GK_CX(p0, s0, p1, s1)
    if (s1 != 0 && p0 - p1 <= 1) {
      TNCO_GK_JOIN(p0, s0, p1, s1)
    } else if (s2 != 0 && p1 - p2 <= 1) {
      TNCO_GK_JOIN(p1, s1, p2, s2)
    } else {
      break;
    }
  }
#undef TNCO_GK_CX
#undef TNCO_GK_JOIN
  const int d0 = s0 * (p0 + 1), d1 = s1 * (p1 + 1), d2 = s2 * (p2 + 1);  // (absent: s = 0)
  return ((uint64_t)(d0 + 2048) << 24) | ((uint64_t)(d1 + 2048) << 12) | (uint64_t)(d2 + 2048);
}

// (cost, id2, id1) of a candidate as one integer: 36 + 14 + 14 bits
TNCO_HD inline uint64_t greedy_cand_key(int a, int b, int c, int id2, int id1) {
  return (greedy_cost_key(a, b, c) << 28) | ((uint64_t)id2 << 14) | (uint64_t)id1;
}

}  // namespace tnco
