// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the random-number layer the reference's search path uses.
// Nothing under oracle/ is linked into, imported by, or executed from the
// product (alphazero-pybind11_amd/); only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use it, and only as the checker.
//
// What is restated here (reference file:line → libstdc++ 11 algorithm):
//   * pcg32 = setseq_xsh_rr_64_32          pcg/pcg_random.hpp:1866, 484-489, 845-872
//     (`thread_local pcg32 re`, mcts.cc:19; `re.seed(seed)`, mcts.cc:21)
//   * std::shuffle(children, re)           mcts.cc:100  → bits/stl_algo.h:3729-3792
//     with uniform_int_distribution<unsigned long> on a 32-bit URBG
//     = Lemire _S_nd<uint64_t>             bits/uniform_int_dist.h:246-270, 311-317
//   * uniform_real_distribution<float>     mcts.cc:718  → generate_canonical<float,24>,
//                                          bits/random.tcc:3348-3380 (one 32-bit draw)
//   * gamma_distribution<float>            mcts.cc:430,435 → bits/random.tcc:2335-2392
//     with its cached-pair normal_distribution (random.tcc:1800-1833)
//   * extreme_value_distribution<float>    mcts.cc:205  → bits/random.tcc:2581-2590
//
// Transcendentals: libstdc++ calls logf/powf of the host libm, whose last-bit
// behaviour is libm-version- and CPU-dispatch-dependent (glibc logf is not
// correctly rounded: 0.818 ULP).  The oracle AND the device engine therefore
// define  az_logf(x) = (float)log((double)x),  az_powf = (float)pow(double,double),
// az_expf = (float)exp((double)x).  tests/test_rng_vs_libstdcxx.py measures how
// often this differs from real libstdc++/glibc (oracle/_ref/rng_ref): integer
// paths (shuffle, uniform) are bit-exact; gamma/gumbel draws agree to <=2 ulp.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstddef>
#include <utility>

namespace orc {

inline float az_logf(float x) { return static_cast<float>(std::log(static_cast<double>(x))); }
inline float az_expf(float x) { return static_cast<float>(std::exp(static_cast<double>(x))); }
inline float az_powf(float x, float y) {
  return static_cast<float>(std::pow(static_cast<double>(x), static_cast<double>(y)));
}

struct Pcg32 {
  static constexpr uint64_t kMult = 6364136223846793005ULL;
  static constexpr uint64_t kInc = 1442695040888963407ULL;
  uint64_t state = 0;
  // engine(seed): state = bump(seed + increment)   (pcg_random.hpp:484-489)
  void seed(uint64_t s) { state = (s + kInc) * kMult + kInc; }
  // output_previous: XSH-RR of the OLD state       (pcg_random.hpp:845-872)
  uint32_t next() {
    uint64_t old = state;
    state = old * kMult + kInc;
    uint32_t xorshifted = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = static_cast<uint32_t>(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
  }
};

// uniform_int_distribution<unsigned long>{0, range-1} on a 32-bit URBG.
inline uint32_t lemire_below(Pcg32& g, uint32_t range) {
  uint64_t product = static_cast<uint64_t>(g.next()) * static_cast<uint64_t>(range);
  uint32_t low = static_cast<uint32_t>(product);
  if (low < range) {
    uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = static_cast<uint64_t>(g.next()) * static_cast<uint64_t>(range);
      low = static_cast<uint32_t>(product);
    }
  }
  return static_cast<uint32_t>(product >> 32);
}

// std::shuffle for n < 65536 (pair-swap branch).
template <class T>
inline void shuffle(T* a, size_t n, Pcg32& g) {
  if (n == 0) return;
  size_t i = 1;
  if ((n % 2) == 0) {
    uint32_t j = lemire_below(g, 2);
    std::swap(a[i], a[j]);
    ++i;
  }
  while (i != n) {
    const uint32_t swap_range = static_cast<uint32_t>(i) + 1;
    const uint32_t b1 = swap_range + 1;
    const uint32_t x = lemire_below(g, swap_range * b1);
    const uint32_t p0 = x / b1, p1 = x % b1;
    std::swap(a[i], a[p0]);
    ++i;
    std::swap(a[i], a[p1]);
    ++i;
  }
}

// generate_canonical<float, 24> with a 32-bit URBG: one draw.
inline float canonical01(Pcg32& g) {
  float sum = static_cast<float>(g.next()) * 1.0f;
  float ret = sum / 4294967296.0f;
  if (ret >= 1.0f) ret = std::nextafter(1.0f, 0.0f);
  return ret;
}

// uniform_real_distribution<float>{0,1}
inline float uniform01(Pcg32& g) { return canonical01(g) * (1.0f - 0.0f) + 0.0f; }

struct Normal01 {
  bool saved_available = false;
  float saved = 0.0f;
  float operator()(Pcg32& g) {
    float ret;
    if (saved_available) {
      saved_available = false;
      ret = saved;
    } else {
      float x, y, r2;
      do {
        x = static_cast<float>(2.0f * canonical01(g) - 1.0);
        y = static_cast<float>(2.0f * canonical01(g) - 1.0);
        r2 = x * x + y * y;
      } while (r2 > 1.0 || r2 == 0.0);
      const float mult = std::sqrt(-2 * az_logf(r2) / r2);
      saved = x * mult;
      saved_available = true;
      ret = y * mult;
    }
    ret = ret * 1.0f + 0.0f;
    return ret;
  }
};

struct Gamma {
  float alpha, beta, malpha, a2;
  Normal01 nd;
  Gamma(float a, float b) : alpha(a), beta(b) {
    malpha = alpha < 1.0 ? alpha + 1.0f : alpha;
    const float a1 = malpha - 1.0f / 3.0f;
    a2 = 1.0f / std::sqrt(9.0f * a1);
  }
  float operator()(Pcg32& g) {
    float u, v, n;
    const float a1 = malpha - 1.0f / 3.0f;
    do {
      do {
        n = nd(g);
        v = 1.0f + a2 * n;
      } while (v <= 0.0);
      v = v * v * v;
      u = canonical01(g);
    } while (u > 1.0f - 0.0331 * n * n * n * n &&
             (az_logf(u) > (0.5 * n * n + a1 * (1.0 - v + az_logf(v)))));
    if (alpha == malpha) return a1 * v * beta;
    do {
      u = canonical01(g);
    } while (u == 0.0);
    return az_powf(u, 1.0f / alpha) * a1 * v * beta;
  }
};

// extreme_value_distribution<float>{0,1}
inline float gumbel01(Pcg32& g) {
  return 0.0f - 1.0f * az_logf(-az_logf(1.0f - canonical01(g)));
}

}  // namespace orc
