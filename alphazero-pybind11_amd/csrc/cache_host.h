// Host-side allocation of a device S3-FIFO cache (shared by engine.hip and cache.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "dev_cache.h"

namespace azmi {

inline uint32_t pow2_at_least(uint32_t x) {
  uint32_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

// max_size / ghost_size are TOTALS (ShardedS3FIFOCache ctor, s3fifo_cache.h:231-241)
inline hipError_t cache_alloc(CacheView& c, std::vector<void*>& allocs, uint32_t max_size, uint32_t shards,
                              uint32_t ghost_size, uint32_t np, uint32_t nv) {
  c = CacheView{};
  c.shards = shards; c.cap = max_size / shards; c.ghost_cap = ghost_size / shards;
  c.np = np; c.nv = nv;
  c.tcap = pow2_at_least(2 * c.cap + 2);
  c.gcap = pow2_at_least(2 * c.ghost_cap + 2);
  auto get = [&](void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    allocs.push_back(*p);
    return hipMemset(*p, 0, bytes ? bytes : 16);
  };
  const size_t S = shards, C = c.cap, G = c.ghost_cap;
  hipError_t e;
#define CGET(field, count) if ((e = get(reinterpret_cast<void**>(&c.field), (count) * sizeof(*c.field))) != hipSuccess) return e
  CGET(hashes, S * C); CGET(freq, S * C); CGET(policy, S * C * np); CGET(value, S * C * nv);
  CGET(s_ring, S * C); CGET(m_ring, S * C); CGET(ghost_ring, S * G); CGET(state, S * 8);
  CGET(map_key, S * c.tcap); CGET(map_val, S * c.tcap); CGET(gset_key, S * c.gcap); CGET(stats, S * 4);
#undef CGET
  return hipStreamSynchronize(nullptr);      // the memsets above are asynchronous to the host
}

}  // namespace azmi

// the C-ABI cache object (azmi_cache_create, include/azmi.h): shared with the engine, which can run on caches it does
// not own (PlayManager(gs, params, caches), play_manager.cc:644-649)
struct azmi_cache {
  azmi::CacheView c{};
  std::vector<void*> allocs;
  int device = 0;
  uint32_t max_size = 0;
  ~azmi_cache() { for (void* p : allocs) (void)hipFree(p); }
};
