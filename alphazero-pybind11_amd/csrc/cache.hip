// Stand-alone C ABI of the device S3-FIFO cache: the counterpart of the reference's
// `S3FIFOCache` / `ShardedS3FIFOCache` Python classes (py_wrapper.cc:222-259) — used by the parity
// tests against the oracle's restatement of s3fifo_cache.h and available to callers such as
// cache_utils.cached_inference.  The PlayManager engine embeds the same structure (engine.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/azmi.h"
#include "cache_host.h"

using namespace azmi;

namespace {
thread_local std::string g_cache_err;
int cfail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_cache_err = buf;
  return code;
}

// one lane per shard, batch order preserved inside a shard (s3fifo_cache.h:259-286)
__global__ void k_cache_insert_many(CacheView c, const uint64_t* hashes, const float* policy, const float* value, uint32_t n) {
  const uint32_t sh = blockIdx.x * blockDim.x + threadIdx.x;
  if (sh >= c.shards) return;
  ShardCtx ctx(c, sh);
  for (uint32_t i = 0; i < n; ++i) {
    const uint64_t h = hashes[i];
    if (h % c.shards != sh) continue;
    const int slot = ctx.insert(h);
    if (slot < 0) continue;
    float* dp = c.policy + (static_cast<size_t>(sh) * c.cap + slot) * c.np;
    float* dv = c.value + (static_cast<size_t>(sh) * c.cap + slot) * c.nv;
    for (uint32_t j = 0; j < c.np; ++j) dp[j] = policy[static_cast<size_t>(i) * c.np + j];
    for (uint32_t j = 0; j < c.nv; ++j) dv[j] = value[static_cast<size_t>(i) * c.nv + j];
  }
}

__global__ __launch_bounds__(256) void k_cache_apply_wave(CacheView c, const uint64_t* keys, const float* policy, const float* value, uint32_t n) {
  __shared__ uint32_t s_sid[kApplyMax];
  cache_apply_batch(c, keys, policy, value, n, s_sid);
}

// 8 lanes per query (the lane-group shape the engine uses)
__global__ void k_cache_find_wave(CacheView c, const uint64_t* hashes, uint32_t n, uint8_t* hit, float* policy, float* value) {
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = gtid / 8, gl = gtid % 8;
  if (i >= n) return;
  uint32_t sh;
  const int slot = wave_shard_find<8>(c, hashes[i], gl, &sh);
  if (gl == 0) { wave_shard_find_account(c, hashes[i], sh, slot); hit[i] = slot >= 0; }
  if (slot < 0) return;
  const float* sp = c.policy + (static_cast<size_t>(sh) * kWaveCap + slot) * c.np;
  const float* sv = c.value + (static_cast<size_t>(sh) * kWaveCap + slot) * c.nv;
  for (uint32_t j = gl; j < c.np; j += 8) policy[static_cast<size_t>(i) * c.np + j] = sp[j];
  for (uint32_t j = gl; j < c.nv; j += 8) value[static_cast<size_t>(i) * c.nv + j] = sv[j];
}

__global__ void k_cache_find_many(CacheView c, const uint64_t* hashes, uint32_t n, uint8_t* hit, float* policy, float* value) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t sh;
  const int slot = cache_find(c, hashes[i], &sh);
  cache_find_account(c, hashes[i], sh, slot);
  hit[i] = slot >= 0;
  if (slot < 0) return;
  const float* sp = c.policy + (static_cast<size_t>(sh) * c.cap + slot) * c.np;
  const float* sv = c.value + (static_cast<size_t>(sh) * c.cap + slot) * c.nv;
  for (uint32_t j = 0; j < c.np; ++j) policy[static_cast<size_t>(i) * c.np + j] = sp[j];
  for (uint32_t j = 0; j < c.nv; ++j) value[static_cast<size_t>(i) * c.nv + j] = sv[j];
}
}  // namespace


extern "C" {

const char* azmi_cache_last_error(void) { return g_cache_err.c_str(); }

int azmi_cache_create(uint32_t max_size, uint32_t shards, uint32_t ghost_size, uint32_t num_policy, uint32_t num_value,
                      int device, azmi_cache** out) {
  if (!out || shards == 0) return cfail(AZMI_ERR_INVALID, "bad cache arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return cfail(AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
  if (hipSetDevice(device) != hipSuccess) return cfail(AZMI_ERR_NO_DEVICE, "hipSetDevice failed");
  auto* c = new azmi_cache();
  c->device = device;
  c->max_size = (max_size / shards) * shards;
  if (cache_alloc(c->c, c->allocs, max_size, shards, ghost_size, num_policy, num_value) != hipSuccess) {
    delete c;
    return cfail(AZMI_ERR_OOM, "cache allocation failed");
  }
  (void)hipDeviceSynchronize();
  *out = c;
  return AZMI_OK;
}
void azmi_cache_destroy(azmi_cache* c) { delete c; }

int azmi_cache_insert_many(azmi_cache* c, const uint64_t* hashes, const float* policy, const float* value, uint32_t n) {
  if (!c || (n && (!hashes || !policy || !value))) return cfail(AZMI_ERR_INVALID, "null argument");
  if (n == 0) return AZMI_OK;
  (void)hipSetDevice(c->device);
  uint64_t* dh = nullptr; float *dp = nullptr, *dv = nullptr;
  hipError_t e = hipMalloc(&dh, n * 8ULL);
  if (e == hipSuccess) e = hipMalloc(&dp, static_cast<size_t>(n) * c->c.np * 4);
  if (e == hipSuccess) e = hipMalloc(&dv, static_cast<size_t>(n) * c->c.nv * 4);
  if (e == hipSuccess) e = hipMemcpy(dh, hashes, n * 8ULL, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dp, policy, static_cast<size_t>(n) * c->c.np * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv, value, static_cast<size_t>(n) * c->c.nv * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    if (c->c.cap == kWaveCap) {
      for (uint32_t off = 0; off < n && e == hipSuccess; off += kApplyMax) {  // chunks keep batch order
        const uint32_t m = std::min<uint32_t>(kApplyMax, n - off);
        k_cache_apply_wave<<<(m + 3) / 4, 256>>>(c->c, dh + off, dp + static_cast<size_t>(off) * c->c.np, dv + static_cast<size_t>(off) * c->c.nv, m);
        e = hipDeviceSynchronize();
      }
    } else {
      k_cache_insert_many<<<(c->c.shards + 63) / 64, 64>>>(c->c, dh, dp, dv, n);
      e = hipDeviceSynchronize();
    }
  }
  (void)hipFree(dh); (void)hipFree(dp); (void)hipFree(dv);
  return e == hipSuccess ? AZMI_OK : cfail(AZMI_ERR_NO_DEVICE, "cache insert: %s", hipGetErrorString(e));
}

int azmi_cache_find_many(azmi_cache* c, const uint64_t* hashes, uint32_t n, uint8_t* hit, float* policy, float* value) {
  if (!c || (n && (!hashes || !hit || !policy || !value))) return cfail(AZMI_ERR_INVALID, "null argument");
  if (n == 0) return AZMI_OK;
  (void)hipSetDevice(c->device);
  uint64_t* dh = nullptr; uint8_t* dhit = nullptr; float *dp = nullptr, *dv = nullptr;
  hipError_t e = hipMalloc(&dh, n * 8ULL);
  if (e == hipSuccess) e = hipMalloc(&dhit, n);
  if (e == hipSuccess) e = hipMalloc(&dp, static_cast<size_t>(n) * c->c.np * 4);
  if (e == hipSuccess) e = hipMalloc(&dv, static_cast<size_t>(n) * c->c.nv * 4);
  if (e == hipSuccess) e = hipMemcpy(dh, hashes, n * 8ULL, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(dp, 0, static_cast<size_t>(n) * c->c.np * 4);
  if (e == hipSuccess) e = hipMemset(dv, 0, static_cast<size_t>(n) * c->c.nv * 4);
  if (e == hipSuccess) {
    if (c->c.cap == kWaveCap) k_cache_find_wave<<<(n * 8 + 255) / 256, 256>>>(c->c, dh, n, dhit, dp, dv);
    else k_cache_find_many<<<(n + 255) / 256, 256>>>(c->c, dh, n, dhit, dp, dv);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(hit, dhit, n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(policy, dp, static_cast<size_t>(n) * c->c.np * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(value, dv, static_cast<size_t>(n) * c->c.nv * 4, hipMemcpyDeviceToHost);
  (void)hipFree(dh); (void)hipFree(dhit); (void)hipFree(dp); (void)hipFree(dv);
  return e == hipSuccess ? AZMI_OK : cfail(AZMI_ERR_NO_DEVICE, "cache find: %s", hipGetErrorString(e));
}

int azmi_cache_stats(azmi_cache* c, uint64_t out[6]) {
  if (!c || !out) return cfail(AZMI_ERR_INVALID, "null argument");
  (void)hipSetDevice(c->device);
  std::vector<unsigned long long> st(static_cast<size_t>(c->c.shards) * 4);
  std::vector<uint32_t> state(static_cast<size_t>(c->c.shards) * 8);
  if (hipMemcpy(st.data(), c->c.stats, st.size() * 8, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(state.data(), c->c.state, state.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return cfail(AZMI_ERR_NO_DEVICE, "cache stats copy failed");
  for (int i = 0; i < 6; ++i) out[i] = 0;
  for (uint32_t s = 0; s < c->c.shards; ++s) {
    for (int j = 0; j < 4; ++j) out[j] += st[static_cast<size_t>(s) * 4 + j];
    out[4] += state[static_cast<size_t>(s) * 8 + kSize];
  }
  out[5] = static_cast<uint64_t>(c->c.cap) * c->c.shards;
  return AZMI_OK;
}

}  // extern "C"
