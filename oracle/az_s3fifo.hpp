// ORACLE — TEST INFRASTRUCTURE ONLY (see az_rng.hpp header for the usage rule).
// CPU restatement of S3FIFOCache / ShardedS3FIFOCache (s3fifo_cache.h:15-318):
// position key -> (pi[num_policy], v[num_value]) with Small / Main / Ghost FIFO
// rings and a 2-bit frequency counter.  Single-threaded (the reference guards
// each shard with one mutex; per-shard operation order is what is restated).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace orc {

class S3Fifo {
 public:
  S3Fifo(uint32_t max_size, uint32_t ghost_size, uint32_t num_policy, uint32_t num_value)
      : max_size_(max_size), ghost_size_(ghost_size), np_(num_policy), nv_(num_value),
        policy_(static_cast<size_t>(max_size) * num_policy), value_(static_cast<size_t>(max_size) * num_value),
        hashes_(max_size), freq_(max_size, 0), s_ring_(max_size), m_ring_(max_size), ghost_ring_(ghost_size) {}

  // s3fifo_cache.h:41-59
  bool find(uint64_t hash, float* policy_out, float* value_out) {
    auto it = map_.find(hash);
    if (it == map_.end()) {
      ++misses_;
      if (ghost_size_ > 0 && ghost_set_.count(hash)) ++reinserts_;
      return false;
    }
    ++hits_;
    const uint32_t slot = it->second;
    if (freq_[slot] < 3) ++freq_[slot];
    std::memcpy(policy_out, &policy_[static_cast<size_t>(slot) * np_], np_ * sizeof(float));
    std::memcpy(value_out, &value_[static_cast<size_t>(slot) * nv_], nv_ * sizeof(float));
    return true;
  }

  // s3fifo_cache.h:82-111
  void insert(uint64_t hash, const float* policy, const float* value) {
    if (max_size_ == 0) return;
    if (map_.count(hash)) return;
    bool ghost_hit = false;
    if (ghost_size_ > 0) ghost_hit = ghost_set_.erase(hash) > 0;
    const uint32_t slot = alloc_slot();
    hashes_[slot] = hash;
    freq_[slot] = 0;
    std::memcpy(&policy_[static_cast<size_t>(slot) * np_], policy, np_ * sizeof(float));
    std::memcpy(&value_[static_cast<size_t>(slot) * nv_], value, nv_ * sizeof(float));
    map_[hash] = slot;
    if (ghost_hit) m_enqueue(slot);
    else s_enqueue(slot);
  }

  size_t hits() const { return hits_; }
  size_t misses() const { return misses_; }
  size_t evictions() const { return evictions_; }
  size_t reinserts() const { return reinserts_; }
  size_t size() const { return map_.size(); }
  size_t max_size() const { return max_size_; }

 private:
  uint32_t alloc_slot() {  // s3fifo_cache.h:113-118
    if (next_free_ < max_size_) return next_free_++;
    return evict_one();
  }
  uint32_t evict_one() {  // s3fifo_cache.h:120-150
    while (s_size_ > 0) {
      const uint32_t slot = s_dequeue();
      if (freq_[slot]) {
        freq_[slot] = 0;
        m_enqueue(slot);
        continue;
      }
      if (ghost_size_ > 0) ghost_add(hashes_[slot]);
      map_.erase(hashes_[slot]);
      ++evictions_;
      return slot;
    }
    while (true) {
      const uint32_t slot = m_dequeue();
      if (freq_[slot]) {
        --freq_[slot];
        m_enqueue(slot);
        continue;
      }
      map_.erase(hashes_[slot]);
      ++evictions_;
      return slot;
    }
  }
  void s_enqueue(uint32_t slot) { s_ring_[(s_head_ + s_size_) % max_size_] = slot; ++s_size_; }
  uint32_t s_dequeue() { uint32_t s = s_ring_[s_head_]; s_head_ = (s_head_ + 1) % max_size_; --s_size_; return s; }
  void m_enqueue(uint32_t slot) { m_ring_[(m_head_ + m_size_) % max_size_] = slot; ++m_size_; }
  uint32_t m_dequeue() { uint32_t s = m_ring_[m_head_]; m_head_ = (m_head_ + 1) % max_size_; --m_size_; return s; }
  void ghost_add(uint64_t hash) {  // s3fifo_cache.h:180-194
    if (ghost_size_ == 0) return;
    if (ghost_count_ >= ghost_size_) {
      ghost_set_.erase(ghost_ring_[ghost_head_]);
      ghost_ring_[ghost_head_] = hash;
      ghost_head_ = (ghost_head_ + 1) % ghost_size_;
    } else {
      ghost_ring_[(ghost_head_ + ghost_count_) % ghost_size_] = hash;
      ++ghost_count_;
    }
    ghost_set_.insert(hash);
  }

  uint32_t max_size_, ghost_size_, np_, nv_;
  std::vector<float> policy_, value_;
  std::vector<uint64_t> hashes_;
  std::vector<uint8_t> freq_;
  std::vector<uint32_t> s_ring_, m_ring_;
  uint32_t s_head_ = 0, s_size_ = 0, m_head_ = 0, m_size_ = 0, next_free_ = 0;
  std::vector<uint64_t> ghost_ring_;
  std::unordered_set<uint64_t> ghost_set_;
  uint32_t ghost_head_ = 0, ghost_count_ = 0;
  std::unordered_map<uint64_t, uint32_t> map_;
  size_t hits_ = 0, misses_ = 0, evictions_ = 0, reinserts_ = 0;
};

class ShardedS3Fifo {  // s3fifo_cache.h:229-318; shard = hash % shards
 public:
  ShardedS3Fifo(uint32_t max_size, uint32_t shards, uint32_t ghost_size, uint32_t num_policy,
                uint32_t num_value)
      : shards_(shards) {
    for (uint32_t i = 0; i < shards; ++i)
      caches_.push_back(std::make_unique<S3Fifo>(max_size / shards, ghost_size / shards, num_policy, num_value));
  }
  bool find(uint64_t h, float* p, float* v) { return caches_[h % shards_]->find(h, p, v); }
  void insert(uint64_t h, const float* p, const float* v) { caches_[h % shards_]->insert(h, p, v); }
  size_t hits() const { size_t o = 0; for (auto& c : caches_) o += c->hits(); return o; }
  size_t misses() const { size_t o = 0; for (auto& c : caches_) o += c->misses(); return o; }
  size_t evictions() const { size_t o = 0; for (auto& c : caches_) o += c->evictions(); return o; }
  size_t reinserts() const { size_t o = 0; for (auto& c : caches_) o += c->reinserts(); return o; }
  size_t size() const { size_t o = 0; for (auto& c : caches_) o += c->size(); return o; }
  size_t max_size() const { size_t o = 0; for (auto& c : caches_) o += c->max_size(); return o; }

 private:
  uint32_t shards_;
  std::vector<std::unique_ptr<S3Fifo>> caches_;
};

}  // namespace orc
