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

// 36-bit key, larger key <=> larger v
TNCO_HD inline uint64_t greedy_cost_key(int a, int b, int c) {
  // digits as (position, sign), at most 3; normalise to non-adjacent form
  int p[3] = {a, b, c};
  int s[3] = {1, -1, -1};
  int m = 3;
  for (;;) {
    // order by position, descending (m <= 3)
    for (int i = 0; i + 1 < m; ++i)
      for (int j = 0; j + 1 < m - i; ++j)
        if (p[j] < p[j + 1]) {
          const int tp = p[j], ts = s[j];
          p[j] = p[j + 1]; s[j] = s[j + 1];
          p[j + 1] = tp; s[j + 1] = ts;
        }
    bool changed = false;
    for (int i = 0; i + 1 < m && !changed; ++i) {
      if (p[i] == p[i + 1]) {
        if (s[i] == s[i + 1]) {  // 2^p + 2^p = 2^(p+1)
          p[i] += 1;
          for (int j = i + 1; j + 1 < m; ++j) { p[j] = p[j + 1]; s[j] = s[j + 1]; }
          m -= 1;
        } else {                 // 2^p - 2^p = 0
          for (int j = i; j + 2 < m; ++j) { p[j] = p[j + 2]; s[j] = s[j + 2]; }
          m -= 2;
        }
        changed = true;
      } else if (p[i] == p[i + 1] + 1) {
        if (s[i] == s[i + 1]) {  // 2^(p+1) + 2^p = 2^(p+2) - 2^p
          p[i] += 1;
          s[i + 1] = -s[i + 1];
        } else {                 // 2^(p+1) - 2^p = 2^p
          p[i] = p[i + 1];
          for (int j = i + 1; j + 1 < m; ++j) { p[j] = p[j + 1]; s[j] = s[j + 1]; }
          m -= 1;
        }
        changed = true;
      }
    }
    if (!changed) break;
  }
  uint64_t key = 0;
  for (int k = 0; k < 3; ++k) {
    const int d = k < m ? s[k] * (p[k] + 1) : 0;
    key = (key << 12) | (uint64_t)(d + 2048);
  }
  return key;
}

// (cost, id2, id1) of a candidate as one integer: 36 + 14 + 14 bits
TNCO_HD inline uint64_t greedy_cand_key(int a, int b, int c, int id2, int id1) {
  return (greedy_cost_key(a, b, c) << 28) | ((uint64_t)id2 << 14) | (uint64_t)id1;
}

}  // namespace tnco
