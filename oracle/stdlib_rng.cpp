// oracle/stdlib_rng.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Emits known answers from the REAL libstdc++ of this image for the
// third-party pieces the reference hot path calls (std::mt19937,
// std::uniform_real_distribution<double>, std::shuffle,
// std::uniform_int_distribution; call sites:
// include/tnco/optimize/optimizer.hpp:73,139,
// include/tnco/optimize/infinite_memory/optimizer.hpp:100-103,162,
// include/tnco/optimize/finite_width/greedy/utils.hpp:80).
// Output (JSON on stdout) is committed as tests/golden/stdlib_rng.json and
// pins oracle/tnco_oracle.c's restatement of them.  Doubles are printed as
// C99 hex floats so the comparison is bit-exact.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <sstream>
#include <vector>

int main() {
  const unsigned long seeds[] = {0ul, 1ul, 42ul, 5489ul, 4294967295ul,
                                 4294967296ul + 7ul /* mod 2^32 -> 7 */};
  std::printf("{\n \"gcc\": \"%s\",\n \"cases\": [\n", __VERSION__);
  bool first = true;
  for (auto seed : seeds) {
    if (!first) std::printf(",\n");
    first = false;
    std::printf("  {\"seed\": %lu,\n", seed);
    // (a) raw outputs, across two regenerations
    {
      std::mt19937 g;
      g.seed(seed);
      std::printf("   \"raw\": [");
      for (int i = 0; i < 1300; ++i) std::printf("%s%lu", i ? "," : "", (unsigned long)g());
      std::printf("],\n");
    }
    // (b) uniform01
    {
      std::mt19937 g;
      g.seed(seed);
      auto u = std::uniform_real_distribution<double>{};
      std::printf("   \"uniform\": [");
      for (int i = 0; i < 400; ++i) std::printf("%s\"%a\"", i ? "," : "", u(g));
      std::printf("],\n");
    }
    // (c) interleaved draws as in update(): x % n, x % 2, uniform
    {
      std::mt19937 g;
      g.seed(seed);
      auto u = std::uniform_real_distribution<double>{};
      std::printf("   \"mixed\": [");
      for (int i = 0; i < 100; ++i) {
        unsigned long a = g() % 37ul;
        unsigned long b = g() % 2ul;
        double c = u(g);
        std::printf("%s[%lu,%lu,\"%a\"]", i ? "," : "", a, b, c);
      }
      std::printf("],\n");
    }
    // (d) std::shuffle of iota(k), consecutive shuffles from one generator
    {
      std::mt19937 g;
      g.seed(seed);
      std::printf("   \"shuffle\": [");
      const int ks[] = {0, 1, 2, 3, 4, 7, 8, 64, 65, 500, 70000};
      bool f2 = true;
      for (int k : ks) {
        std::vector<size_t> v(k);
        std::iota(v.begin(), v.end(), size_t{0});
        std::shuffle(v.begin(), v.end(), g);
        std::printf("%s{\"k\": %d, \"perm\": [", f2 ? "" : ",", k);
        f2 = false;
        // for the big case only print a digest + head
        int lim = k > 600 ? 50 : k;
        for (int i = 0; i < lim; ++i) std::printf("%s%zu", i ? "," : "", v[i]);
        uint64_t h = 1469598103934665603ull;
        for (auto x : v) { h ^= (uint64_t)x; h *= 1099511628211ull; }
        std::printf("], \"fnv\": \"%llu\", \"next\": %lu}", (unsigned long long)h, (unsigned long)g());
      }
      std::printf("],\n");
    }
    // (e) uniform_int_distribution<unsigned long> in [0, hi]
    {
      std::mt19937 g;
      g.seed(seed);
      std::printf("   \"uniform_int\": [");
      const unsigned long his[] = {1, 2, 5, 6, 99, 1000, 65535, 4294967294ul, 4294967295ul};
      bool f2 = true;
      for (auto hi : his) {
        std::uniform_int_distribution<unsigned long> d(0, hi);
        for (int i = 0; i < 20; ++i) {
          std::printf("%s[%lu,%lu]", f2 ? "" : ",", hi, d(g));
          f2 = false;
        }
      }
      std::printf("],\n");
    }
    // (f) state string after 0, 1, 624, 700 draws (format of prng_state,
    // include/tnco/optimize/optimizer.hpp:191-195)
    {
      std::printf("   \"state\": [");
      const int nd[] = {0, 1, 624, 700};
      bool f2 = true;
      for (int n : nd) {
        if (seed != 0ul && seed != 42ul) break;  // keep the fixture small
        std::mt19937 g;
        g.seed(seed);
        for (int i = 0; i < n; ++i) g();
        std::ostringstream oss;
        oss << g;
        std::printf("%s{\"draws\": %d, \"str\": \"%s\"}", f2 ? "" : ",", n, oss.str().c_str());
        f2 = false;
      }
      std::printf("]\n  }");
    }
  }
  std::printf("\n ]\n}\n");
  return 0;
}
