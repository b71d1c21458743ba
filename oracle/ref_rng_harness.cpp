// ORACLE — TEST INFRASTRUCTURE ONLY.
// Real-reference leg of the RNG layer: compiles the reference's OWN vendored
// generator (/root/reference/src/pcg/pcg_random.hpp, header-only, no deps) with
// this image's libstdc++ <random>/<algorithm>, i.e. exactly the code
// mcts.cc:19,100,205,430-440,718 instantiates.  No stand-in headers are used.
// Built by oracle/Makefile into oracle/_ref/librngref.so only when
// /root/reference exists; the prebuilt .so travels to the GPU box.
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <random>
#include <vector>

#include "pcg/pcg_random.hpp"

extern "C" {
void ref_pcg32_outputs(uint64_t seed, uint32_t n, uint32_t* out) {
  pcg32 re; re.seed(seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = re();
}
void ref_shuffle_iota(uint64_t seed, uint32_t n, uint32_t reps, uint32_t* out) {
  pcg32 re; re.seed(seed);
  for (uint32_t r = 0; r < reps; ++r) {
    std::vector<uint32_t> v(n);
    std::iota(v.begin(), v.end(), 0u);
    std::shuffle(v.begin(), v.end(), re);
    std::copy(v.begin(), v.end(), out + r * n);
  }
}
void ref_uniform01(uint64_t seed, uint32_t n, float* out) {
  pcg32 re; re.seed(seed);
  std::uniform_real_distribution<float> dist{0.0F, 1.0F};
  for (uint32_t i = 0; i < n; ++i) out[i] = dist(re);
}
void ref_gamma(uint64_t seed, float alpha, float beta, uint32_t n, int fresh_each, float* out) {
  pcg32 re; re.seed(seed);
  std::gamma_distribution<float> dist{alpha, beta};
  for (uint32_t i = 0; i < n; ++i) {
    if (fresh_each) { std::gamma_distribution<float> d2{alpha, beta}; out[i] = d2(re); }
    else out[i] = dist(re);
  }
}
void ref_gumbel(uint64_t seed, uint32_t n, float* out) {
  pcg32 re; re.seed(seed);
  std::extreme_value_distribution<float> dist{0.0f, 1.0f};
  for (uint32_t i = 0; i < n; ++i) out[i] = dist(re);
}
}
