// Device-side position cache: S3-FIFO, restating S3FIFOCache / ShardedS3FIFOCache
// (/root/reference/src/s3fifo_cache.h:15-318) for HBM.
//
// position key -> (pi[num_policy], v[num_value]); per shard a Small and a Main FIFO ring of slot
// indices, a Ghost ring of evicted keys, a 2-bit frequency per slot (s3fifo_cache.h:41-59, 82-150).
// shard = key % shards, exactly as the reference.  absl's flat_hash_{map,set} become two
// open-addressing tables per shard (linear probing, backward-shift deletion, key 0 = empty).
//
// Concurrency model on the GPU:
//   * find()  — many lane-groups probe concurrently (read-only on the tables; the frequency bump
//               and the hit/miss counters are atomics).  The reference serialises finds with a
//               mutex; the end state of a set of finds is order-independent (freq saturates at 3).
//   * insert  — runs in its own kernel between rounds, ONE lane per shard walking the batch in
//               batch order (ShardedS3FIFOCache::insert_many groups by shard and keeps the order,
//               s3fifo_cache.h:259-286), so eviction order is exactly the reference's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dev_rng.h"

namespace azmi {

struct CacheView {
  uint32_t shards, cap, ghost_cap;  // per-shard capacities (max_size / shards, ghost_size / shards)
  uint32_t tcap, gcap;              // table sizes (powers of two), per shard
  uint32_t np, nv;                  // num_policy, num_value
  uint64_t* hashes;                 // [shards][cap]
  uint32_t* freq;                   // [shards][cap]
  float* policy;                    // [shards][cap][np]
  float* value;                     // [shards][cap][nv]
  uint32_t* s_ring;                 // [shards][cap]
  uint32_t* m_ring;                 // [shards][cap]
  uint64_t* ghost_ring;             // [shards][ghost_cap]
  uint32_t* state;                  // [shards][8]: s_head s_size m_head m_size next_free ghost_head ghost_count size
  uint64_t* map_key;                // [shards][tcap]
  uint32_t* map_val;                // [shards][tcap]
  uint64_t* gset_key;               // [shards][gcap]
  unsigned long long* stats;        // [shards][4]: hits misses evictions reinserts
};

enum { kSHead = 0, kSSize, kMHead, kMSize, kNextFree, kGHead, kGCount, kSize };

__device__ __forceinline__ uint64_t cache_key(uint64_t h) { return h == 0 ? 0x9E3779B97F4A7C15ULL : h; }
__device__ __forceinline__ uint32_t cache_home(uint64_t k, uint32_t mask) { return static_cast<uint32_t>(mix64(k) >> 17) & mask; }

// ---- open addressing (one table = `n` slots starting at keys/vals) ----------------------------------
__device__ inline int table_find(const uint64_t* keys, uint32_t n, uint64_t k) {
  if (n == 0) return -1;
  const uint32_t mask = n - 1;
  uint32_t i = cache_home(k, mask);
  for (uint32_t probes = 0; probes < n; ++probes, i = (i + 1) & mask) {
    const uint64_t cur = keys[i];
    if (cur == k) return static_cast<int>(i);
    if (cur == 0) return -1;
  }
  return -1;
}
__device__ inline void table_insert(uint64_t* keys, uint32_t* vals, uint32_t n, uint64_t k, uint32_t v) {
  const uint32_t mask = n - 1;
  uint32_t i = cache_home(k, mask);
  while (keys[i] != 0 && keys[i] != k) i = (i + 1) & mask;
  keys[i] = k;
  if (vals) vals[i] = v;
}
// backward-shift deletion keeps every probe chain gap-free (no tombstones)
__device__ inline bool table_erase(uint64_t* keys, uint32_t* vals, uint32_t n, uint64_t k) {
  const int at = table_find(keys, n, k);
  if (at < 0) return false;
  const uint32_t mask = n - 1;
  uint32_t hole = static_cast<uint32_t>(at);
  uint32_t j = hole;
  for (;;) {
    j = (j + 1) & mask;
    const uint64_t kj = keys[j];
    if (kj == 0) break;
    const uint32_t home = cache_home(kj, mask);
    // can kj move into the hole?  yes iff its home is cyclically outside (hole, j]
    const bool between = (hole <= j) ? (home > hole && home <= j) : (home > hole || home <= j);
    if (!between) {
      keys[hole] = kj;
      if (vals) vals[hole] = vals[j];
      hole = j;
    }
  }
  keys[hole] = 0;
  return true;
}

// ---- S3FIFOCache::find, s3fifo_cache.h:41-59 — returns the slot index (shard-relative) or -1 ----------
__device__ inline int cache_find(const CacheView& c, uint64_t hash, uint32_t* shard_out) {
  const uint64_t k = cache_key(hash);
  const uint32_t sh = static_cast<uint32_t>(hash % c.shards);
  *shard_out = sh;
  const int at = table_find(c.map_key + static_cast<size_t>(sh) * c.tcap, c.tcap, k);
  return at < 0 ? -1 : static_cast<int>(c.map_val[static_cast<size_t>(sh) * c.tcap + at]);
}
// bookkeeping of one find (call from ONE lane): counters, ghost "reinsert" stat, freq = min(freq + 1, 3)
__device__ inline void cache_find_account(const CacheView& c, uint64_t hash, uint32_t sh, int slot) {
  unsigned long long* st = c.stats + static_cast<size_t>(sh) * 4;
  if (slot < 0) {
    atomicAdd(&st[1], 1ULL);
    if (c.ghost_cap > 0 && table_find(c.gset_key + static_cast<size_t>(sh) * c.gcap, c.gcap, cache_key(hash)) >= 0)
      atomicAdd(&st[3], 1ULL);
    return;
  }
  atomicAdd(&st[0], 1ULL);
  uint32_t* f = c.freq + static_cast<size_t>(sh) * c.cap + slot;
  if (atomicAdd(f, 1u) >= 3u) atomicSub(f, 1u);
}

// ---- S3FIFOCache::insert_locked, s3fifo_cache.h:82-150 (one lane per shard) ---------------------------
struct ShardCtx {
  const CacheView& c;
  uint32_t sh;
  uint32_t* st;
  uint64_t* hashes; uint32_t* freq; uint32_t* s_ring; uint32_t* m_ring; uint64_t* ghost_ring;
  uint64_t* map_key; uint32_t* map_val; uint64_t* gset_key;
  __device__ ShardCtx(const CacheView& cv, uint32_t s)
      : c(cv), sh(s), st(cv.state + static_cast<size_t>(s) * 8), hashes(cv.hashes + static_cast<size_t>(s) * cv.cap),
        freq(cv.freq + static_cast<size_t>(s) * cv.cap), s_ring(cv.s_ring + static_cast<size_t>(s) * cv.cap),
        m_ring(cv.m_ring + static_cast<size_t>(s) * cv.cap), ghost_ring(cv.ghost_ring + static_cast<size_t>(s) * cv.ghost_cap),
        map_key(cv.map_key + static_cast<size_t>(s) * cv.tcap), map_val(cv.map_val + static_cast<size_t>(s) * cv.tcap),
        gset_key(cv.gset_key + static_cast<size_t>(s) * cv.gcap) {}
  __device__ void s_enqueue(uint32_t slot) { s_ring[(st[kSHead] + st[kSSize]) % c.cap] = slot; ++st[kSSize]; }
  __device__ uint32_t s_dequeue() { const uint32_t s = s_ring[st[kSHead]]; st[kSHead] = (st[kSHead] + 1) % c.cap; --st[kSSize]; return s; }
  __device__ void m_enqueue(uint32_t slot) { m_ring[(st[kMHead] + st[kMSize]) % c.cap] = slot; ++st[kMSize]; }
  __device__ uint32_t m_dequeue() { const uint32_t s = m_ring[st[kMHead]]; st[kMHead] = (st[kMHead] + 1) % c.cap; --st[kMSize]; return s; }
  __device__ void ghost_add(uint64_t k) {  // s3fifo_cache.h:180-194
    if (c.ghost_cap == 0) return;
    if (st[kGCount] >= c.ghost_cap) {
      table_erase(gset_key, nullptr, c.gcap, ghost_ring[st[kGHead]]);
      ghost_ring[st[kGHead]] = k;
      st[kGHead] = (st[kGHead] + 1) % c.ghost_cap;
    } else {
      ghost_ring[(st[kGHead] + st[kGCount]) % c.ghost_cap] = k;
      ++st[kGCount];
    }
    if (table_find(gset_key, c.gcap, k) < 0) table_insert(gset_key, nullptr, c.gcap, k, 0);
  }
  __device__ uint32_t evict_one() {  // s3fifo_cache.h:120-150
    unsigned long long* stats = c.stats + static_cast<size_t>(sh) * 4;
    while (st[kSSize] > 0) {
      const uint32_t slot = s_dequeue();
      if (freq[slot]) { freq[slot] = 0; m_enqueue(slot); continue; }
      if (c.ghost_cap > 0) ghost_add(hashes[slot]);
      table_erase(map_key, map_val, c.tcap, hashes[slot]);
      --st[kSize];
      stats[2] += 1;
      return slot;
    }
    for (;;) {
      const uint32_t slot = m_dequeue();
      if (freq[slot]) { --freq[slot]; m_enqueue(slot); continue; }
      table_erase(map_key, map_val, c.tcap, hashes[slot]);
      --st[kSize];
      stats[2] += 1;
      return slot;
    }
  }
  // returns the slot the caller must fill with (policy, value), or -1 if nothing is to be written
  __device__ int insert(uint64_t hash) {
    if (c.cap == 0) return -1;
    const uint64_t k = cache_key(hash);
    if (table_find(map_key, c.tcap, k) >= 0) return -1;
    bool ghost_hit = false;
    if (c.ghost_cap > 0) ghost_hit = table_erase(gset_key, nullptr, c.gcap, k);
    uint32_t slot;
    if (st[kNextFree] < c.cap) slot = st[kNextFree]++;
    else slot = evict_one();
    hashes[slot] = k;
    freq[slot] = 0;
    table_insert(map_key, map_val, c.tcap, k, slot);
    ++st[kSize];
    if (ghost_hit) m_enqueue(slot); else s_enqueue(slot);
    return static_cast<int>(slot);
  }
};

}  // namespace azmi

// =====================================================================================================
// Wave-resident shards (cap == 64): the whole bookkeeping of one shard lives in the registers of one
// wavefront — lane i holds hashes[i], freq[i], s_ring[i], m_ring[i], ghost_ring[i] and a "ghost key is
// in the set" bit — so S3FIFOCache::find / insert_locked / evict_one / ghost_add (s3fifo_cache.h:41-194)
// run as v_readlane / v_writelane / ballot sequences instead of dependent HBM accesses.  The hash map
// and the ghost set become ballots over the 64 lanes.  Semantics are exactly the generic path's
// (same rings, same counters); parity with the oracle is tested at shards = max_size / 64.
// =====================================================================================================
namespace azmi {

constexpr uint32_t kWaveCap = 64;

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// COH = true: every access is an agent-scope relaxed atomic (sc1: served by L2, written through) - the form the shard's data
// needs when wavefronts on different CUs take turns on it under the shard's insert lock (wave_shard_insert_locked)
template <bool COH>
__device__ __forceinline__ uint32_t c_ld(const uint32_t* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
template <bool COH>
__device__ __forceinline__ uint64_t c_ld(const uint64_t* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
template <bool COH>
__device__ __forceinline__ void c_st(uint32_t* p, uint32_t v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool COH>
__device__ __forceinline__ void c_st(uint64_t* p, uint64_t v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}

template <bool COH = false>
struct WaveShardT {
  const CacheView& c;
  uint32_t sh, lane;
  // v_writelane as a select (this clang has no writelane builtin): element `at` of a lane-resident array
  __device__ __forceinline__ uint32_t wl(uint32_t old, uint32_t val, uint32_t at) const { return lane == at ? val : old; }
  uint32_t hlo, hhi;     // hashes[lane] (0 = free slot)
  uint32_t fq;           // freq[lane]
  uint32_t sr, mr;       // s_ring[lane], m_ring[lane]
  uint32_t glo, ghi;     // ghost_ring[lane]
  uint32_t gin;          // ghost_ring[lane]'s key is in the ghost set
  uint32_t s_head, s_size, m_head, m_size, next_free, g_head, g_count, size;
  unsigned long long evictions;

  __device__ WaveShardT(const CacheView& cv, uint32_t s, uint32_t l) : c(cv), sh(s), lane(l), evictions(0) {
    const size_t b = static_cast<size_t>(s) * kWaveCap + l;
    const uint64_t h = c_ld<COH>(cv.hashes + b);
    hlo = static_cast<uint32_t>(h); hhi = static_cast<uint32_t>(h >> 32);
    fq = c_ld<COH>(cv.freq + b); sr = c_ld<COH>(cv.s_ring + b); mr = c_ld<COH>(cv.m_ring + b);
    uint64_t g = 0;
    gin = 0;
    if (l < cv.ghost_cap) { g = c_ld<COH>(cv.ghost_ring + static_cast<size_t>(s) * cv.ghost_cap + l); gin = c_ld<COH>(cv.map_val + b); }
    glo = static_cast<uint32_t>(g); ghi = static_cast<uint32_t>(g >> 32);
    const uint32_t* st = cv.state + static_cast<size_t>(s) * 8;
    const uint32_t stl = l < 8 ? c_ld<COH>(st + l) : 0u;       // the eight state words in one access
    s_head = rl(stl, kSHead); s_size = rl(stl, kSSize); m_head = rl(stl, kMHead); m_size = rl(stl, kMSize);
    next_free = rl(stl, kNextFree); g_head = rl(stl, kGHead); g_count = rl(stl, kGCount); size = rl(stl, kSize);
  }
  __device__ void store() const {
    const size_t b = static_cast<size_t>(sh) * kWaveCap + lane;
    c_st<COH>(c.hashes + b, static_cast<uint64_t>(hlo) | (static_cast<uint64_t>(hhi) << 32));
    c_st<COH>(c.freq + b, fq); c_st<COH>(c.s_ring + b, sr); c_st<COH>(c.m_ring + b, mr);
    if (lane < c.ghost_cap) {
      c_st<COH>(c.ghost_ring + static_cast<size_t>(sh) * c.ghost_cap + lane, static_cast<uint64_t>(glo) | (static_cast<uint64_t>(ghi) << 32));
      c_st<COH>(c.map_val + b, gin);  // map_val is unused by wave shards: it keeps the ghost-set membership bits
    }
    if (lane < 8) {
      uint32_t* st = c.state + static_cast<size_t>(sh) * 8;
      const uint32_t v = lane == kSHead ? s_head : lane == kSSize ? s_size : lane == kMHead ? m_head : lane == kMSize ? m_size
                       : lane == kNextFree ? next_free : lane == kGHead ? g_head : lane == kGCount ? g_count : size;
      c_st<COH>(st + lane, v);
    }
    if (lane == 0 && evictions) atomicAdd(&c.stats[static_cast<size_t>(sh) * 4 + 2], evictions);
  }
  __device__ __forceinline__ unsigned long long match_slots(uint32_t klo, uint32_t khi) const {
    return __ballot(hlo == klo && hhi == khi);
  }
  __device__ __forceinline__ unsigned long long match_ghost(uint32_t klo, uint32_t khi) const {
    return __ballot(lane < c.ghost_cap && gin && glo == klo && ghi == khi);
  }
  __device__ __forceinline__ void ghost_erase(uint32_t klo, uint32_t khi) {
    if (lane < c.ghost_cap && glo == klo && ghi == khi) gin = 0;
  }
  __device__ void s_enqueue(uint32_t slot) { sr = wl(sr, slot, (s_head + s_size) % kWaveCap); ++s_size; }
  __device__ uint32_t s_dequeue() { const uint32_t s = rl(sr, s_head); s_head = (s_head + 1) % kWaveCap; --s_size; return s; }
  __device__ void m_enqueue(uint32_t slot) { mr = wl(mr, slot, (m_head + m_size) % kWaveCap); ++m_size; }
  __device__ uint32_t m_dequeue() { const uint32_t s = rl(mr, m_head); m_head = (m_head + 1) % kWaveCap; --m_size; return s; }
  __device__ void ghost_add(uint32_t klo, uint32_t khi) {  // s3fifo_cache.h:180-194
    if (c.ghost_cap == 0) return;
    uint32_t pos;
    if (g_count >= c.ghost_cap) {
      const uint32_t olo = rl(glo, g_head), ohi = rl(ghi, g_head);
      ghost_erase(olo, ohi);
      pos = g_head;
      g_head = (g_head + 1) % c.ghost_cap;
    } else {
      pos = (g_head + g_count) % c.ghost_cap;
      ++g_count;
    }
    glo = wl(glo, klo, pos); ghi = wl(ghi, khi, pos);
    gin = wl(gin, 1u, pos);
  }
  __device__ void erase_slot(uint32_t slot) { hlo = wl(hlo, 0u, slot); hhi = wl(hhi, 0u, slot); --size; }
  __device__ uint32_t evict_one() {  // s3fifo_cache.h:120-150
    while (s_size > 0) {
      const uint32_t slot = s_dequeue();
      if (rl(fq, slot)) { fq = wl(fq, 0u, slot); m_enqueue(slot); continue; }
      ghost_add(rl(hlo, slot), rl(hhi, slot));
      erase_slot(slot);
      ++evictions;
      return slot;
    }
    for (;;) {
      const uint32_t slot = m_dequeue();
      const uint32_t f = rl(fq, slot);
      if (f) { fq = wl(fq, f - 1, slot); m_enqueue(slot); continue; }
      erase_slot(slot);
      ++evictions;
      return slot;
    }
  }
  // S3FIFOCache::insert_locked — returns the slot to fill or -1
  __device__ int insert(uint64_t hash) {
    const uint64_t k = cache_key(hash);
    const uint32_t klo = static_cast<uint32_t>(k), khi = static_cast<uint32_t>(k >> 32);
    if (match_slots(klo, khi)) return -1;
    const bool ghost_hit = c.ghost_cap > 0 && match_ghost(klo, khi) != 0;
    if (ghost_hit) ghost_erase(klo, khi);
    uint32_t slot;
    if (next_free < kWaveCap) slot = next_free++;
    else slot = evict_one();
    hlo = wl(hlo, klo, slot); hhi = wl(hhi, khi, slot);
    fq = wl(fq, 0u, slot);
    ++size;
    if (ghost_hit) m_enqueue(slot); else s_enqueue(slot);
    return static_cast<int>(slot);
  }
};

using WaveShard = WaveShardT<false>;

// S3FIFOCache::insert (s3fifo_cache.h:61-80: lock, insert_locked, copy the rows) by ONE wavefront while other wavefronts
// insert into the same cache: the shard's lock word (0 = free) is taken with an agent-scope compare-and-swap, the shard is
// read and written with agent-scope accesses (no fence: nothing of it sits in an L1), and the lock is released after the
// stores have drained.  The slot's key is cleared before its payload is rewritten and set after it, so a concurrent probe
// that re-reads the key after the payload (wave_shard_find's validated form) never returns a torn row.
// p_lane / v_lane: entry `lane` of the policy / value row (np, nv <= 64).  Returns false when the lock could not be had
// within `spin_cap` tries (the caller raises an error; nothing was changed).
// PROBE_SAFE = false: no probe runs while the inserts do (a kernel of inserts between two epochs): the slot's key and payload
// go out together, one wait.  Either way an entry that is already there is recognised before the lock is taken (most repeated
// positions - a whole batch of identical openings - never touch it).
template <bool PROBE_SAFE = true>
__device__ inline bool wave_shard_insert_locked(const CacheView& c, uint32_t* locks, uint64_t hash, float p_lane, float v_lane,
                                                uint32_t lane, uint32_t spin_cap = 1u << 20) {
  const uint32_t sh = static_cast<uint32_t>(hash % c.shards);
  // the key check and the first try at the lock travel together (one round trip): an entry that is already there gives the lock
  // straight back; a wavefront that waits for the lock keeps looking at the keys, so a batch of identical positions - every
  // slot's opening move - is done with the shard as soon as the first of them is in
  uint32_t got = 0;
  for (uint32_t spins = 0; spins < spin_cap; ++spins) {
    const uint64_t have = c_ld<true>(c.hashes + static_cast<size_t>(sh) * kWaveCap + lane);
    uint32_t g = 0;
    if (lane == 0) {
      uint32_t expect = 0u;
      g = __hip_atomic_compare_exchange_strong(locks + sh, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
    }
    g = uni(g);
    if (__ballot(have == cache_key(hash)) != 0ull) {
      if (g && lane == 0) __hip_atomic_store(locks + sh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    if (g) { got = 1; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  if (!got) return false;
  {
    WaveShardT<true> ws(c, sh, lane);
    const int slot = ws.insert(hash);
    if (slot >= 0) {
      const size_t e = static_cast<size_t>(sh) * kWaveCap + slot;
      if constexpr (PROBE_SAFE) {
        if (lane == 0) c_st<true>(c.hashes + e, 0ull);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (lane < c.np) c_st<true>(reinterpret_cast<uint32_t*>(c.policy) + e * c.np + lane, __float_as_uint(p_lane));
      if (lane < c.nv) c_st<true>(reinterpret_cast<uint32_t*>(c.value) + e * c.nv + lane, __float_as_uint(v_lane));
      if constexpr (PROBE_SAFE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ws.store();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(locks + sh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// find on a wave shard by a group of `G` cooperating lanes (G divides 64): each lane checks 64/G slots.
// Returns the slot or -1 (uniform over the group).  Accounting as in cache_find_account.
struct NoPreload { __device__ __forceinline__ void operator()() const {} };
// `before_loads` runs once the shard is known, right in front of the key loads: loads it issues travel with them (one round trip)
template <int G, class F = NoPreload>
__device__ __forceinline__ int wave_shard_find(const CacheView& c, uint64_t hash, uint32_t glane, uint32_t* shard_out, F&& before_loads = F()) {
  const uint64_t k = cache_key(hash);
  const uint32_t sh = static_cast<uint32_t>(hash % c.shards);
  *shard_out = sh;
  const uint64_t* hs = c.hashes + static_cast<size_t>(sh) * kWaveCap;
  // all of the lane's key loads are issued, then the caller's preloads (they return behind the keys: one round trip for both), then
  // the compares
  uint64_t hv[kWaveCap / G];
#pragma unroll
  for (int i = 0; i < static_cast<int>(kWaveCap) / G; ++i) hv[i] = hs[i * G + static_cast<int>(glane)];
  before_loads();
  int found = -1;
#pragma unroll
  for (int i = 0; i < static_cast<int>(kWaveCap) / G; ++i)
    if (hv[i] == k) found = i * G + static_cast<int>(glane);
  if constexpr (G == 8) {   // DPP moves instead of LDS-routed shuffles: xor 1, xor 2 inside quads, then the half-row mirror
    found = max(found, __builtin_amdgcn_update_dpp(0, found, 0xB1, 0xF, 0xF, true));
    found = max(found, __builtin_amdgcn_update_dpp(0, found, 0x4E, 0xF, 0xF, true));
    found = max(found, __builtin_amdgcn_update_dpp(0, found, 0x141, 0xF, 0xF, true));
  } else {
#pragma unroll
    for (int off = 1; off < G; off <<= 1) found = max(found, __shfl_xor(found, off, G));
  }
  return found;
}
__device__ inline void wave_shard_find_account(const CacheView& c, uint64_t hash, uint32_t sh, int slot) {
  unsigned long long* st = c.stats + static_cast<size_t>(sh) * 4;
  if (slot < 0) {
    atomicAdd(&st[1], 1ULL);
    if (c.ghost_cap > 0) {
      const uint64_t k = cache_key(hash);
      bool in_ghost = false;
      for (uint32_t i = 0; i < c.ghost_cap; ++i)
        in_ghost = in_ghost || (c.map_val[static_cast<size_t>(sh) * kWaveCap + i] &&
                                c.ghost_ring[static_cast<size_t>(sh) * c.ghost_cap + i] == k);
      if (in_ghost) atomicAdd(&st[3], 1ULL);
    }
    return;
  }
  atomicAdd(&st[0], 1ULL);
  uint32_t* f = c.freq + static_cast<size_t>(sh) * kWaveCap + slot;
  if (atomicAdd(f, 1u) >= 3u) atomicSub(f, 1u);
}


// Batch insert for wave shards.  One wavefront per batch element; the wave whose element is the FIRST
// of its shard in the batch becomes that shard's owner and applies all of the shard's elements in
// batch order (ShardedS3FIFOCache::insert_many, s3fifo_cache.h:259-286); every other wave exits.
// `keys[j] == 0` marks an element that is not to be inserted.  Rows j of `policy` / `value`
// (row strides np / nv floats) are the payload.  Launch: <<<ceil(n / 4), 256>>>, n <= kApplyMax.
constexpr uint32_t kApplyMax = 8192;
// dst[0..n) = src[0..n) by one wavefront with U loads in flight per lane: a plain `dst[e] = src[e]` loop waits for every
// load before its store (the two rows may alias as far as the compiler knows), i.e. one memory round trip per 64 floats -
// 27 dependent trips for a StarGambit policy row
template <uint32_t U = 8>
__device__ __forceinline__ void wave_copy_row(float* __restrict__ dst, const float* __restrict__ src, uint32_t n, uint32_t lane) {
  for (uint32_t e0 = 0; e0 < n; e0 += 64 * U) {
    float t[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) { const uint32_t e = e0 + u * 64 + lane; t[u] = e < n ? src[e] : 0.0f; }
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) { const uint32_t e = e0 + u * 64 + lane; if (e < n) dst[e] = t[u]; }
  }
}
__device__ __forceinline__ void cache_apply_batch(const CacheView& c, const uint64_t* keys, const float* policy,
                                                   const float* value, uint32_t n, uint32_t* s_sid,
                                                   const uint8_t* groups = nullptr, uint32_t group = 0) {
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t i = blockIdx.x * (blockDim.x >> 6) + wave;
  // this wave's own element - key and, when a row is at most one load per lane, the payload - is requested together with the
  // batch's keys: the insert below is a chain of dependent round trips (keys, shard state, key again, payload), and in the
  // common case of one element per shard the last two are these registers
  const bool own_rows = c.np <= 64 && c.nv <= 64;
  const uint64_t own_key = i < n ? keys[i] : 0ull;
  const float own_p = (i < n && own_rows && lane < c.np) ? policy[static_cast<size_t>(i) * c.np + lane] : 0.0f;
  const float own_v = (i < n && own_rows && lane < c.nv) ? value[static_cast<size_t>(i) * c.nv + lane] : 0.0f;
  for (uint32_t j = tid; j < n; j += blockDim.x) {
    const uint64_t k = (groups && groups[j] != group) ? 0 : keys[j];   // elements of another model group are not ours
    s_sid[j] = k ? static_cast<uint32_t>(k % c.shards) : 0xFFFFFFFFu;
  }
  __syncthreads();
  if (i >= n) return;
  const uint32_t my = s_sid[i];
  if (my == 0xFFFFFFFFu) return;
  bool earlier = false;
  for (uint32_t j = lane; j < i; j += 64) earlier = earlier || (s_sid[j] == my);
  if (__ballot(earlier)) return;  // an earlier element owns this shard
  WaveShard ws(c, my, lane);
  for (uint32_t base = (i / 64) * 64; base < n; base += 64) {
    const uint32_t j = base + lane;
    unsigned long long m = __ballot(j >= i && j < n && s_sid[j] == my);
    while (m) {
      const uint32_t b = __builtin_ctzll(m);
      m &= m - 1;
      const uint32_t jj = base + b;
      const int slot = ws.insert(jj == i ? own_key : keys[jj]);
      if (slot >= 0) {
        float* dp = c.policy + (static_cast<size_t>(my) * kWaveCap + slot) * c.np;
        float* dv = c.value + (static_cast<size_t>(my) * kWaveCap + slot) * c.nv;
        if (jj == i && own_rows) {
          if (lane < c.np) dp[lane] = own_p;
          if (lane < c.nv) dv[lane] = own_v;
        } else {
          wave_copy_row<32>(dp, policy + static_cast<size_t>(jj) * c.np, c.np, lane);   // a StarGambit row in one round trip
          for (uint32_t e = lane; e < c.nv; e += 64) dv[e] = value[static_cast<size_t>(jj) * c.nv + e];
        }
      }
    }
  }
  ws.store();
}

}  // namespace azmi
