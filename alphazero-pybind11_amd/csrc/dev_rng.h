// Device-side random-number layer of the self-play engine (gfx950).
//
// One pcg32 stream per game slot, kept in HBM between rounds and in registers
// inside a round.  Every lane of a slot's lane-group advances an identical copy
// of the stream (the draws are wave-uniform per group), so no cross-lane
// traffic is needed for randomness.
//
// Behaviour matched (reference file:line -> what libstdc++ 11 does there):
//   thread_local pcg32 re            mcts.cc:19      pcg_random.hpp:484-489, 845-872
//   std::shuffle(children, re)       mcts.cc:100     stl_algo.h:3729-3792 (pair-swap form)
//   uniform_int<unsigned long> on a 32-bit URBG      uniform_int_dist.h:246-270 (Lemire)
//   uniform_real_distribution<float> mcts.cc:718     random.tcc:3348-3380, one draw / 2^32
//   gamma_distribution<float>        mcts.cc:430,435 random.tcc:2335-2392 (Marsaglia-Tsang)
//   normal_distribution<float>       (inside gamma)  random.tcc:1800-1833 (cached pair)
// logf/powf/expf are taken as float(round(double fn)) — the same definition the
// parity oracle uses — because glibc's float routines are not correctly rounded
// and differ by CPU dispatch; see DESIGN.md §Numerics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace azmi {

__device__ __forceinline__ float az_logf(float x) { return static_cast<float>(log(static_cast<double>(x))); }
__device__ __forceinline__ float az_expf(float x) { return static_cast<float>(exp(static_cast<double>(x))); }
__device__ __forceinline__ float az_powf(float x, float y) {
  return static_cast<float>(pow(static_cast<double>(x), static_cast<double>(y)));
}

struct Pcg32 {
  static constexpr uint64_t kMult = 6364136223846793005ULL;
  static constexpr uint64_t kInc = 1442695040888963407ULL;
  uint64_t state;
  __host__ __device__ __forceinline__ void seed(uint64_t s) { state = (s + kInc) * kMult + kInc; }
  __host__ __device__ __forceinline__ uint32_t next() {
    const uint64_t old = state;
    state = old * kMult + kInc;
    const uint32_t xs = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = static_cast<uint32_t>(old >> 59u);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
  }
};

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
constexpr uint64_t kCoinSalt = 0x5851F42D4C957F2DULL;
constexpr uint64_t kRollSalt = 0x9E3779B97F4A7C15ULL;   // rollout stream of EvalType::PLAYOUT
__host__ __device__ __forceinline__ uint64_t slot_seed(uint64_t seed, uint32_t slot) {
  return mix64(seed + 0x9E3779B97F4A7C15ULL * (static_cast<uint64_t>(slot) + 1));
}

// uniform integer in [0, range) — Lemire's nearly-divisionless method
__device__ __forceinline__ uint32_t lemire_below(Pcg32& g, uint32_t range) {
  uint64_t product = static_cast<uint64_t>(g.next()) * static_cast<uint64_t>(range);
  uint32_t low = static_cast<uint32_t>(product);
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = static_cast<uint64_t>(g.next()) * static_cast<uint64_t>(range);
      low = static_cast<uint32_t>(product);
    }
  }
  return static_cast<uint32_t>(product >> 32);
}

__device__ __forceinline__ float canonical01(Pcg32& g) {
  const float ret = static_cast<float>(g.next()) / 4294967296.0f;
  return ret >= 1.0f ? 0.99999994f /* nextafter(1,0) */ : ret;
}

struct Normal01 {
  // The second variate of a polar pair is cached (random.tcc:1800-1833).  "No cached value" is
  // encoded as a NaN in the float itself rather than a separate bool: a bool member became an
  // i1 lane mask that hipcc (ROCm 7.2) mis-tracked across divergent groups of one wavefront.
  float saved = __builtin_nanf("");
  __device__ __forceinline__ float draw(Pcg32& g) {
    const float cached = saved;
    if (cached == cached) {
      saved = __builtin_nanf("");
      return cached * 1.0f + 0.0f;
    }
    float x, y, r2;
    do {
      x = static_cast<float>(static_cast<double>(2.0f * canonical01(g)) - 1.0);
      y = static_cast<float>(static_cast<double>(2.0f * canonical01(g)) - 1.0);
      r2 = x * x + y * y;
    } while (r2 > 1.0f || r2 == 0.0f);
    const float mult = sqrtf(-2.0f * az_logf(r2) / r2);
    saved = x * mult;
    return (y * mult) * 1.0f + 0.0f;
  }
};

struct Gamma {
  float alpha, malpha, a2;
  Normal01 nd;
  __device__ __forceinline__ explicit Gamma(float a) : alpha(a) {
    malpha = alpha < 1.0f ? alpha + 1.0f : alpha;
    const float a1 = malpha - 1.0f / 3.0f;
    a2 = 1.0f / sqrtf(9.0f * a1);
  }
  __device__ __forceinline__ float draw(Pcg32& g) {  // beta == 1
    float u, v, n;
    const float a1 = malpha - 1.0f / 3.0f;
    for (;;) {
      do {
        n = nd.draw(g);
        v = 1.0f + a2 * n;
      } while (v <= 0.0f);
      v = v * v * v;
      u = canonical01(g);
      const double dn = static_cast<double>(n);
      const bool squeeze_fail = static_cast<double>(u) > 1.0 - 0.0331 * dn * dn * dn * dn;
      if (!squeeze_fail) break;
      const double rhs = 0.5 * dn * dn +
                         static_cast<double>(a1) * (1.0 - static_cast<double>(v) + static_cast<double>(az_logf(v)));
      if (!(static_cast<double>(az_logf(u)) > rhs)) break;
    }
    if (alpha == malpha) return a1 * v * 1.0f;
    do {
      u = canonical01(g);
    } while (u == 0.0f);
    return az_powf(u, 1.0f / alpha) * a1 * v * 1.0f;
  }
};

}  // namespace azmi
