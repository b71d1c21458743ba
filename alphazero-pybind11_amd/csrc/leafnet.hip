// Leaf policy/value network on gfx950 matrix cores — the one dense contraction of the path.
//
// Restates the inference forward of the reference's ResNet-mode NNArch
// (/root/reference/src/neural_net.py:233-263 ResidualBlock, :448-510 NNArch.forward) and the
// probability output of NNWrapper.process (:800-823) for the trunk width the BASELINE configs
// use (64 channels, 3x3 convolutions, value head with average pool, flat policy head), with
// inference BatchNorms folded on the host (alphazero/hip_net.py):
//     stem   s  = conv3x3(x) + b0                         (bn1 folded)
//     block  t  = relu(a1 * s + b1)                       (bn1 of the block: affine on the stream)
//            u  = relu(conv3x3(t) + c1)                   (bn2 folded into conv1)
//            s  = s + conv3x3(u)
//     heads  h  = relu(conv1x1(s) + bh)                   (v_bn / pi_bn folded; 32 + 32 channels)
//            v  = softmax(W2 relu(W1 avgpool(h_v) + b1) + b2),  pi = softmax(Wp flatten(h_pi) + bp)
//
// Two kernels, one machine (a workgroup of 4 waves carries a tile of boards through the WHOLE net; weights stream from L2
// into an LDS ring by DMA; the residual stream stays in fp32 MFMA accumulators):
//   * leafnet_c4.h - the Connect4-family net (flat policy head), also launched fused with the engine's move step;
//   * leafnet_sp.h - the spatial-policy-head nets (Tafl family, StarGambit).
// This file is the host side behind azmi_net_* (include/azmi.h): validation of the descriptor, the weight image on the
// device, launches.  The fp32 correctness path lives in leafnet_f32.hip.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/azmi.h"
#include "leafnet_f32.h"
#include "leafnet_c4.h"
#include "leafnet_sp.h"

namespace {

using namespace azmi_net_dev;

thread_local std::string g_net_err;
int nfail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_net_err = buf;
  return code;
}

}  // namespace

struct azmi_net {
  NetDesc nd{};
  NetPtrs np{};
  bool x3 = false;           // precision = 2: the Connect4-family tile with split bf16 operands (leafnet_c4.h, SPLIT)
  void* f32 = nullptr;       // precision = 1: the fp32 path (leafnet_f32.hip) owns everything
  void* f32_last_stream = nullptr;
  bool f32_last_stream_set = false;
  bool spatial = false;      // spatial policy head (Tafl family, StarGambit): sp::k_leafnet_sp
  sp::SpDesc sd{};
  sp::SpPtrs sp{};
  uint32_t sp_tbw = 0;       // boards per workgroup of the instantiated tile
  size_t fc_lds = 0;         // dynamic LDS of k_heads_fc / k_heads_fc_a / _b
  bool fc_split = false;     // the heads' FC step as k_heads_fc_a + _b (big FC stacks) instead of the one k_heads_fc
  // Scratch between the kernels of one forward: PER STREAM - engines on different streams share one net object and run
  // their forwards concurrently (round 1 kept one buffer per net: the value heads of concurrent shards read each other's
  // pooled features)
  struct StreamScratch {
    float* pool = nullptr;     // [2][pool_rows][64] pooled value-head / policy-head features between k_leafnet_sp and k_heads_fc_a,
    uint32_t pool_rows = 0;    // [pool_rows][32] global-action logits and [pool_rows][hidden] last hidden layer between _a and _b
    uint32_t pool_hidden = 0;
    float *g_canon = nullptr, *g_v = nullptr, *g_pi = nullptr;   // row-list evaluation of the whole-batch kernels (azmi_net_forward_rows)
    uint32_t g_rows = 0;
  };
  std::mutex scratch_mu;
  std::map<void*, StreamScratch> scratch;
  StreamScratch& scratch_of(void* stream) { std::lock_guard<std::mutex> l(scratch_mu); return scratch[stream]; }
  void free_scratch() {
    for (auto& kv : scratch) {
      StreamScratch& sc = kv.second;
      if (sc.pool) (void)hipFree(sc.pool);
      if (sc.g_canon) (void)hipFree(sc.g_canon);
      if (sc.g_v) (void)hipFree(sc.g_v);
      if (sc.g_pi) (void)hipFree(sc.g_pi);
    }
    scratch.clear();
  }
  void* blob = nullptr;
  size_t blob_bytes = 0;
  int device = 0;
  size_t lds_bytes = 0;
};

namespace {
// compact[i] = canon[rows[i]] for i < *count, zeros behind (those rows are evaluated and dropped)
__global__ void k_gather_rows(const float* __restrict__ canon, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ count,
                              uint32_t max_rows, uint32_t chw, float* __restrict__ compact) {
  const uint32_t i = blockIdx.x;
  if (i >= max_rows) return;
  const bool on = i < *count;
  const float* src = canon + static_cast<size_t>(on ? rows[i] : 0u) * chw;
  float* dst = compact + static_cast<size_t>(i) * chw;
  for (uint32_t e = threadIdx.x; e < chw; e += blockDim.x) dst[e] = on ? src[e] : 0.0f;
}
// v[rows[i]] = v_c[i], pi[rows[i]] = pi_c[i] for i < *count: rows that are not listed are left untouched
__global__ void k_scatter_rows(const float* __restrict__ v_c, const float* __restrict__ pi_c, const uint32_t* __restrict__ rows,
                               const uint32_t* __restrict__ count, uint32_t p1, uint32_t m, float* __restrict__ v, float* __restrict__ pi) {
  const uint32_t i = blockIdx.x;
  if (i >= *count) return;
  const uint32_t r = rows[i];
  for (uint32_t e = threadIdx.x; e < p1; e += blockDim.x) v[static_cast<size_t>(r) * p1 + e] = v_c[static_cast<size_t>(i) * p1 + e];
  for (uint32_t e = threadIdx.x; e < m; e += blockDim.x) pi[static_cast<size_t>(r) * m + e] = pi_c[static_cast<size_t>(i) * m + e];
}
bool is_spatial(const azmi_net_desc* d) { return d->policy_channels > 0; }
// weight image of a spatial net (alphazero/hip_net.py fold_spatial): the chunk stream | the fp32 parameters of the tower and
// the heads | the value FC stack | pi_global
size_t spatial_blob_bytes(const azmi_net_desc* d) {
  const size_t Hd = d->v_hidden, L = d->v_fc_layers;
  size_t n = static_cast<size_t>(sp::stream_chunks(d->depth, d->in_channels, d->precision == 2)) * sp::CHUNK_BYTES;   // (bf16x3: every chunk three times)
  n += (CH + static_cast<size_t>(d->depth) * 3 * CH + 2 * sp::HCS + sp::HCS + sp::HCS + 32) * 4;
  n += (64 * Hd + Hd) * 4 + (L - 1) * (Hd * Hd + Hd) * 4 + (Hd * 16 + 16) * 4;
  if (d->num_moves > d->policy_channels * d->height * d->width) {     // pi_global: W1^T[64][Hp] b[Hp] W2^T[Hp][32] b[32] ln_g[32] ln_b[32]
    const size_t Hp = d->pi_hidden;
    n += (64 * Hp + Hp + Hp * 32 + 3 * 32) * 4;
  }
  return n;
}
}  // namespace

extern "C" {

const char* azmi_net_last_error(void) { return g_net_err.c_str(); }

size_t azmi_net_blob_bytes(const azmi_net_desc* d) {
  if (!d) return 0;
  if (d->precision == 1) return azmi_f32::blob_bytes(d);
  if (is_spatial(d)) return spatial_blob_bytes(d);
  // precision 2 (bf16x3): the stem's fragments twice (high, low parts), three chunks per tap and for the head 1x1 (leafnet_c4.h, SPLIT)
  const size_t x3 = d->precision == 2 ? 3 : 1;
  const size_t wconv = x3 * 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
  size_t n = (d->precision == 2 ? 2 : 1) * wsmall + CH * 4;
  n += static_cast<size_t>(d->depth) * (3 * CH * 4 + 2 * wconv);
  n += x3 * wsmall + CH * 4;
  n += (static_cast<size_t>(d->v_hidden) * HC + d->v_hidden) * 4;
  n += (static_cast<size_t>(d->num_players + 1) * d->v_hidden + d->num_players + 1) * 4;
  n += static_cast<size_t>(d->height) * d->width * 2 * WFRAG_BYTES + static_cast<size_t>(d->num_moves) * 4;   // policy FC: bf16 hi / lo fragments per pixel position
  return n;
}

int azmi_net_create(const azmi_net_desc* d, const void* blob, size_t blob_bytes, int device, azmi_net** out) {
  if (!d || !blob || !out) return nfail(AZMI_ERR_INVALID, "null argument");
  if (d->precision == 1) {
    const char* msg = "";
    void* impl = nullptr;
    const int rc = azmi_f32::create(d, blob, blob_bytes, device, &impl, &msg);
    if (rc != AZMI_OK) return nfail(rc, "%s", msg);
    auto net = new azmi_net();
    net->device = device; net->f32 = impl;
    *out = net;
    return AZMI_OK;
  }
  if (d->precision != 0 && d->precision != 2) return nfail(AZMI_ERR_INVALID, "precision must be 0 (bf16 MFMA), 1 (fp32) or 2 (bf16x3: split bf16 MFMA)");
  if (is_spatial(d)) {
    if (d->channels != CH || d->head_channels != sp::HCS || d->kernel_size != 3 || d->v_head_convs != 1 || d->pi_head_convs != 1)
      return nfail(AZMI_ERR_INVALID, "spatial leaf net kernel covers 64 trunk / 64 head channels, 3x3 convs, one extra conv per head");
    const bool b11 = d->height == 11 && d->width == 11, b7 = d->height == 7 && d->width == 7, b13 = d->height == 13 && d->width == 13;
    if (!b11 && !b7 && !b13) return nfail(AZMI_ERR_INVALID, "spatial leaf net kernel: board %dx%d not instantiated", d->height, d->width);
    const int tbw = b11 ? sp::Geo11::TBW : b7 ? sp::Geo7::TBW : sp::Geo13::TBW;
    const bool x3 = d->precision == 2;       // the bf16x3 tier: Geo<.., SPLIT>, 16 activation planes, one workgroup per CU
    const size_t tile_lds = x3 ? (b11 ? sp::Geo11X3::LDS_BYTES : b7 ? sp::Geo7X3::LDS_BYTES : sp::Geo13X3::LDS_BYTES)
                               : (b11 ? sp::Geo11::LDS_BYTES : b7 ? sp::Geo7::LDS_BYTES : sp::Geo13::LDS_BYTES);
    const int num_global = d->num_moves - d->policy_channels * d->height * d->width;
    if (d->policy_channels > 32 || num_global < 0 || num_global > 32)
      return nfail(AZMI_ERR_INVALID, "spatial head: policy channels <= 32, 0..32 global actions");
    if (num_global > 0 && (d->pi_hidden < 64 || d->pi_hidden > 1024 || d->pi_hidden % 64))
      return nfail(AZMI_ERR_INVALID, "spatial head with global actions: pi_hidden must be a multiple of 64 in [64, 1024]");
    if (d->in_channels < 1 || d->in_channels > 64) return nfail(AZMI_ERR_INVALID, "spatial leaf net kernel: %d input planes not supported", d->in_channels);
    if (d->depth < 1 || d->depth > sp::MAXDEPTH) return nfail(AZMI_ERR_INVALID, "spatial leaf net kernel: 1..%d residual blocks", sp::MAXDEPTH);
    if (static_cast<size_t>(tbw) * d->num_moves * 4 > sp::RING_BYTES) return nfail(AZMI_ERR_INVALID, "spatial head: logits do not fit the tile's LDS");
    if (d->v_hidden > 512 || d->v_hidden % 128 || d->v_fc_layers < 1 || d->num_players + 1 > 16) return nfail(AZMI_ERR_INVALID, "value head sizes out of range");
    if (blob_bytes != spatial_blob_bytes(d)) return nfail(AZMI_ERR_INVALID, "weight blob is %zu bytes, expected %zu", blob_bytes, spatial_blob_bytes(d));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return nfail(AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
    if (hipSetDevice(device) != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "hipSetDevice failed");
    auto net = new azmi_net();
    net->device = device; net->spatial = true; net->x3 = x3;
    net->sd = sp::SpDesc{d->in_channels, d->height, d->width, d->depth, d->num_moves, d->num_players, d->v_hidden, d->v_fc_layers, d->policy_channels,
                         num_global, num_global > 0 ? d->pi_hidden : 0};
    if (hipMalloc(&net->blob, blob_bytes) != hipSuccess) { delete net; return nfail(AZMI_ERR_OOM, "hipMalloc(weights) failed"); }
    if (hipMemcpy(net->blob, blob, blob_bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(net->blob); delete net; return nfail(AZMI_ERR_NO_DEVICE, "weight upload failed"); }
    net->blob_bytes = blob_bytes;
    const uint8_t* p = static_cast<const uint8_t*>(net->blob);
    const size_t Hd = d->v_hidden, L = d->v_fc_layers;
    sp::SpPtrs& sp = net->sp;
    auto f32p = [&](size_t count) { const float* q = reinterpret_cast<const float*>(p); p += count * 4; return q; };
    sp.stream = p; p += static_cast<size_t>(sp::stream_chunks(d->depth, d->in_channels, x3)) * sp::CHUNK_BYTES;
    sp.prm = f32p(CH + static_cast<size_t>(d->depth) * 3 * CH + 2 * sp::HCS + sp::HCS + sp::HCS + 32);
    sp.fc1_w = f32p(64 * Hd); sp.fc1_b = f32p(Hd);
    sp.fcx_w = f32p((L - 1) * Hd * Hd); sp.fcx_b = f32p((L - 1) * Hd);
    sp.fc2_w = f32p(Hd * 16); sp.fc2_b = f32p(16);
    if (num_global > 0) {
      const size_t Hp = d->pi_hidden;
      sp.pg1_w = f32p(64 * Hp); sp.pg1_b = f32p(Hp); sp.pg2_w = f32p(Hp * 32); sp.pg2_b = f32p(32); sp.pg_ln_g = f32p(32); sp.pg_ln_b = f32p(32);
    }
    net->sp_tbw = tbw;
    net->lds_bytes = tile_lds;
    const void* kernel = x3 ? (b11 ? reinterpret_cast<const void*>(&sp::k_leafnet_sp<sp::Geo11X3>)
                                   : b7 ? reinterpret_cast<const void*>(&sp::k_leafnet_sp<sp::Geo7X3>) : reinterpret_cast<const void*>(&sp::k_leafnet_sp<sp::Geo13X3>))
                            : (b11 ? reinterpret_cast<const void*>(&sp::k_leafnet_sp<sp::Geo11>)
                                   : b7 ? reinterpret_cast<const void*>(&sp::k_leafnet_sp<sp::Geo7>) : reinterpret_cast<const void*>(&sp::k_leafnet_sp<sp::Geo13>));
    net->fc_split = d->v_hidden > 256 || num_global > 0;   // more than ~0.4 MB of FC weights per group
    net->fc_lds = sp::heads_fc_lds(d->v_hidden > d->pi_hidden || num_global == 0 ? d->v_hidden : d->pi_hidden);
    // The FC kernels are shared by every spatial net of the process: the attribute is set to the most any descriptor this
    // library accepts can ask for (v_hidden / pi_hidden <= 1024), never to this net's own need - a later, smaller net must
    // not lower it under an earlier, larger one
    const int fc_max = static_cast<int>(sp::heads_fc_lds(1024));
    if (net->fc_lds > static_cast<size_t>(fc_max)) { (void)hipFree(net->blob); delete net; return nfail(AZMI_ERR_INVALID, "v_hidden / pi_hidden above 1024"); }
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(net->lds_bytes)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&sp::k_heads_fc), hipFuncAttributeMaxDynamicSharedMemorySize, fc_max) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&sp::k_heads_fc_a), hipFuncAttributeMaxDynamicSharedMemorySize, fc_max) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&sp::k_heads_fc_b), hipFuncAttributeMaxDynamicSharedMemorySize, fc_max) != hipSuccess) {
      (void)hipFree(net->blob); delete net;
      return nfail(AZMI_ERR_NO_DEVICE, "cannot reserve LDS for the spatial leaf net");
    }
    *out = net;
    return AZMI_OK;
  }
  if (d->channels != CH || d->head_channels != HC || d->kernel_size != 3)
    return nfail(AZMI_ERR_INVALID, "leaf net kernel covers 64 trunk channels, 32 head channels, 3x3 convs");
  if (!(d->height == 6 && d->width == 7)) return nfail(AZMI_ERR_INVALID, "leaf net kernel: board %dx%d not instantiated", d->height, d->width);
  if (d->in_channels != 4) return nfail(AZMI_ERR_INVALID, "leaf net kernel: %d input planes not instantiated (Connect4 has 4)", d->in_channels);
  if (d->depth < 1 || d->depth > c4::MAXDEPTH) return nfail(AZMI_ERR_INVALID, "leaf net kernel: 1..%d residual blocks", c4::MAXDEPTH);
  if (d->v_hidden > 256 || d->v_hidden % 16 || d->num_players + 1 > 4 || d->num_moves > 16) return nfail(AZMI_ERR_INVALID, "head sizes out of range (v_hidden: a multiple of 16, at most 256)");
  if (blob_bytes != azmi_net_blob_bytes(d)) return nfail(AZMI_ERR_INVALID, "weight blob is %zu bytes, expected %zu", blob_bytes, azmi_net_blob_bytes(d));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return nfail(AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
  if (hipSetDevice(device) != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "hipSetDevice failed");
  auto net = new azmi_net();
  net->device = device;
  net->nd = NetDesc{d->in_channels, d->height, d->width, d->depth, d->num_moves, d->num_players, d->v_hidden};
  if (hipMalloc(&net->blob, blob_bytes) != hipSuccess) { delete net; return nfail(AZMI_ERR_OOM, "hipMalloc(weights) failed"); }
  if (hipMemcpy(net->blob, blob, blob_bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(net->blob); delete net; return nfail(AZMI_ERR_NO_DEVICE, "weight upload failed"); }
  net->blob_bytes = blob_bytes;
  const uint8_t* p = static_cast<const uint8_t*>(net->blob);
  const bool x3 = d->precision == 2;
  net->x3 = x3;
  const size_t wconv = (x3 ? 3 : 1) * 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
  NetPtrs& np = net->np;
  np.stem_w = p; p += (x3 ? 2 : 1) * wsmall;
  np.stem_b = reinterpret_cast<const float*>(p); p += CH * 4;
  np.blocks = p; p += static_cast<size_t>(d->depth) * (3 * CH * 4 + 2 * wconv);
  np.head_w = p; p += (x3 ? 3 : 1) * wsmall;
  np.head_b = reinterpret_cast<const float*>(p); p += CH * 4;
  np.v_fc1_w = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->v_hidden) * HC * 4;
  np.v_fc1_b = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->v_hidden) * 4;
  np.v_fc2_w = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->num_players + 1) * d->v_hidden * 4;
  np.v_fc2_b = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->num_players + 1) * 4;
  np.pi_fc_w = p; p += static_cast<size_t>(d->height) * d->width * 2 * WFRAG_BYTES;
  np.pi_fc_b = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->num_moves) * 4;
  net->lds_bytes = x3 ? c4::TileBigX3::LDS_BYTES : c4::TileBig::LDS_BYTES;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&c4::k_leafnet_c4<c4::TileBig, 4, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(c4::TileBig::LDS_BYTES)) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&c4::k_leafnet_c4<c4::TileSmall, 4, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(c4::TileSmall::LDS_BYTES)) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&c4::k_leafnet_c4<c4::TileBigX3, 4, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(c4::TileBigX3::LDS_BYTES)) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&c4::k_leafnet_c4<c4::TileSmallX3, 4, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(c4::TileSmallX3::LDS_BYTES)) != hipSuccess) {
    (void)hipFree(net->blob); delete net;
    return nfail(AZMI_ERR_NO_DEVICE, "cannot reserve %zu bytes of LDS", net->lds_bytes);
  }
  *out = net;
  return AZMI_OK;
}

void azmi_net_destroy(azmi_net* net) {
  if (!net) return;
  if (net->f32) {
    (void)hipSetDevice(net->device);
    net->free_scratch();
    azmi_f32::destroy(net->f32); delete net; return;
  }
  (void)hipSetDevice(net->device);
  (void)hipFree(net->blob);
  net->free_scratch();
  delete net;
}

static int net_forward_live(azmi_net* net, const float* dev_canonical, float* dev_v, float* dev_pi, uint32_t batch, void* stream,
                            const uint32_t* live);
// Connect4-family net: 6-board tiles when the batch fills the chip's 512 workgroup slots with them, 3-board tiles below that
// (a launch of a few hundred rows is a LATENCY: leafnet_c4.h, Tile)
static void c4_launch(azmi_net* net, const float* canon, float* v, float* pi, uint32_t rows_max, const uint32_t* rows, const uint32_t* row_count,
                      hipStream_t st) {
  if (net->x3) {      // one workgroup per CU (16 activation planes): 6-board tiles once the 3-board ones would need a second pass over the chip
    if (rows_max > 256u * c4::TileSmallX3::TBW) {
      const uint32_t tiles = (rows_max + c4::TileBigX3::TBW - 1) / c4::TileBigX3::TBW;
      c4::k_leafnet_c4<c4::TileBigX3, 4, 4, 16><<<tiles, c4::NTH, c4::TileBigX3::LDS_BYTES, st>>>(net->nd, net->np, canon, v, pi, rows_max, rows, row_count);
    } else {
      const uint32_t tiles = (rows_max + c4::TileSmallX3::TBW - 1) / c4::TileSmallX3::TBW;
      c4::k_leafnet_c4<c4::TileSmallX3, 4, 4, 16><<<tiles, c4::NTH, c4::TileSmallX3::LDS_BYTES, st>>>(net->nd, net->np, canon, v, pi, rows_max, rows, row_count);
    }
  } else if (rows_max >= 512u * c4::TileBig::TBW) {
    const uint32_t tiles = (rows_max + c4::TileBig::TBW - 1) / c4::TileBig::TBW;
    c4::k_leafnet_c4<c4::TileBig, 4, 4, 16><<<tiles, c4::NTH, c4::TileBig::LDS_BYTES, st>>>(net->nd, net->np, canon, v, pi, rows_max, rows, row_count);
  } else {
    const uint32_t tiles = (rows_max + c4::TileSmall::TBW - 1) / c4::TileSmall::TBW;
    c4::k_leafnet_c4<c4::TileSmall, 4, 4, 16><<<tiles, c4::NTH, c4::TileSmall::LDS_BYTES, st>>>(net->nd, net->np, canon, v, pi, rows_max, rows, row_count);
  }
}
static int reserve_pool(azmi_net* net, void* stream, uint32_t batch) {
  azmi_net::StreamScratch& sc = net->scratch_of(stream);
  const uint32_t Hd = static_cast<uint32_t>(net->sd.v_hidden);
  if (batch <= sc.pool_rows && sc.pool_hidden == Hd) return AZMI_OK;
  // first use of this stream / a larger batch only (synchronous, outside the steady state and outside stream capture)
  (void)hipSetDevice(net->device);
  if (sc.pool) { (void)hipDeviceSynchronize(); (void)hipFree(sc.pool); sc.pool = nullptr; sc.pool_rows = 0; }
  const uint32_t want = ((batch < 4096u ? 4096u : batch) + 15u) / 16u * 16u;
  if (hipMalloc(reinterpret_cast<void**>(&sc.pool), static_cast<size_t>(want) * (2 * 64 + 32 + Hd) * sizeof(float)) != hipSuccess)
    return nfail(AZMI_ERR_OOM, "hipMalloc(head scratch) failed");
  sc.pool_rows = want; sc.pool_hidden = Hd;
  return AZMI_OK;
}
// the tiles (tower, both heads up to their pooled features / spatial logits), then the heads' FC parts batched (two small launches);
// `rows` / `row_count` (may be NULL) = the eval list (leafnet_sp.h)
static int spatial_forward(azmi_net* net, const float* dev_canonical, float* dev_v, float* dev_pi, uint32_t batch, void* stream,
                           const uint32_t* rows, const uint32_t* row_count) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rrc = reserve_pool(net, stream, batch);
  if (rrc != AZMI_OK) return rrc;
  azmi_net::StreamScratch& sc = net->scratch_of(stream);
  float* vpool = sc.pool;
  float* ppool = sc.pool + static_cast<size_t>(sc.pool_rows) * 64;
  float* glob = sc.pool + static_cast<size_t>(sc.pool_rows) * 128;
  float* hidden = sc.pool + static_cast<size_t>(sc.pool_rows) * 160;
  const uint32_t tiles = (batch + net->sp_tbw - 1) / net->sp_tbw;
  if (net->x3) {
    if (net->sd.H == 11)
      sp::k_leafnet_sp<sp::Geo11X3><<<tiles, sp::NTH, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, vpool, ppool, dev_pi, batch, rows, row_count);
    else if (net->sd.H == 13)
      sp::k_leafnet_sp<sp::Geo13X3><<<tiles, sp::NTH, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, vpool, ppool, dev_pi, batch, rows, row_count);
    else
      sp::k_leafnet_sp<sp::Geo7X3><<<tiles, sp::NTH, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, vpool, ppool, dev_pi, batch, rows, row_count);
  } else if (net->sd.H == 11)
    sp::k_leafnet_sp<sp::Geo11><<<tiles, sp::NTH, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, vpool, ppool, dev_pi, batch, rows, row_count);
  else if (net->sd.H == 13)
    sp::k_leafnet_sp<sp::Geo13><<<tiles, sp::NTH, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, vpool, ppool, dev_pi, batch, rows, row_count);
  else
    sp::k_leafnet_sp<sp::Geo7><<<tiles, sp::NTH, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, vpool, ppool, dev_pi, batch, rows, row_count);
  const uint32_t groups = (batch + 15) / 16;
  const uint32_t parts = static_cast<uint32_t>(net->sd.v_hidden / sp::HFC_SLICE) + (net->sd.num_global > 0 ? 1u : 0u);
  if (net->fc_split) {
    const int hmax = net->sd.v_hidden > net->sd.pi_hidden || net->sd.num_global == 0 ? net->sd.v_hidden : net->sd.pi_hidden;
    sp::k_heads_fc_a<<<groups * parts, sp::HFC_THREADS, sp::heads_fc_a_lds(hmax, net->sd.v_fc_layers), st>>>(net->sd, net->sp, vpool, ppool, hidden, glob, batch, row_count);
    sp::k_heads_fc_b<<<groups, sp::HFC_THREADS, sp::heads_fc_b_lds(net->sd.v_hidden), st>>>(net->sd, net->sp, hidden, glob, dev_v, dev_pi, batch, rows, row_count);
  } else {
    sp::k_heads_fc<<<groups, sp::HFC_THREADS, net->fc_lds, st>>>(net->sd, net->sp, vpool, ppool, dev_v, dev_pi, batch, rows, row_count);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "k_leafnet_sp launch: %s", hipGetErrorString(e));
  return AZMI_OK;
}
int azmi_net_reserve_stream(azmi_net* net, void* stream, uint32_t max_rows) {
  if (!net) return nfail(AZMI_ERR_INVALID, "null argument");
  return net->spatial ? reserve_pool(net, stream, max_rows) : AZMI_OK;
}
int azmi_net_forward(azmi_net* net, const float* dev_canonical, float* dev_v, float* dev_pi, uint32_t batch, void* stream) {
  return net_forward_live(net, dev_canonical, dev_v, dev_pi, batch, stream, nullptr);
}
// `live`: device-side count of real rows at the front of the batch (row-list evaluation), NULL = all of them
static int net_forward_live(azmi_net* net, const float* dev_canonical, float* dev_v, float* dev_pi, uint32_t batch, void* stream,
                            const uint32_t* live) {
  if (!net || !dev_canonical || !dev_v || !dev_pi) return nfail(AZMI_ERR_INVALID, "null argument");
  if (batch == 0) return AZMI_OK;
  if (net->f32) {
    // the fp32 path keeps ONE set of activation buffers per net: a forward on another stream first waits for the previous
    // one (a correctness path, not a throughput path)
    if (net->f32_last_stream_set && net->f32_last_stream != stream) (void)hipStreamSynchronize(static_cast<hipStream_t>(net->f32_last_stream));
    net->f32_last_stream = stream; net->f32_last_stream_set = true;
    const char* msg = "";
    const int rc = azmi_f32::forward(net->f32, dev_canonical, dev_v, dev_pi, batch, stream, &msg);
    return rc == AZMI_OK ? rc : nfail(rc, "%s", msg);
  }
  if (net->spatial) return spatial_forward(net, dev_canonical, dev_v, dev_pi, batch, stream, nullptr, nullptr);
  c4_launch(net, dev_canonical, dev_v, dev_pi, batch, nullptr, nullptr, static_cast<hipStream_t>(stream));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "k_leafnet launch: %s", hipGetErrorString(e));
  return AZMI_OK;
}

int azmi_net_forward_rows(azmi_net* net, const float* dev_canonical, float* dev_v, float* dev_pi, const uint32_t* dev_rows,
                          const uint32_t* dev_row_count, uint32_t max_rows, void* stream) {
  if (!net || !dev_canonical || !dev_v || !dev_pi || !dev_rows || !dev_row_count) return nfail(AZMI_ERR_INVALID, "null argument");
  if (max_rows == 0) return AZMI_OK;
  // Rows that are not listed are NOT touched (another model group's answers, cache hits already delivered: ADVICE r1,
  // play_manager.cc:577-597).  The MFMA kernels read and write through the list themselves; workgroups past the count exit.
  if (net->spatial) return spatial_forward(net, dev_canonical, dev_v, dev_pi, max_rows, stream, dev_rows, dev_row_count);
  if (net->f32) {
    // the fp32 kernels take a dense batch: gather the listed rows, evaluate, scatter the answers back
    uint32_t chw, p1, m;
    azmi_f32::dims(net->f32, &chw, &p1, &m);
    hipStream_t st = static_cast<hipStream_t>(stream);
    azmi_net::StreamScratch& sc = net->scratch_of(stream);
    if (max_rows > sc.g_rows) {      // first use of this stream / a larger engine only (synchronous, outside the steady state)
      (void)hipSetDevice(net->device);
      (void)hipDeviceSynchronize();
      if (sc.g_canon) (void)hipFree(sc.g_canon); if (sc.g_v) (void)hipFree(sc.g_v); if (sc.g_pi) (void)hipFree(sc.g_pi);
      sc.g_canon = sc.g_v = sc.g_pi = nullptr; sc.g_rows = 0;
      if (hipMalloc(reinterpret_cast<void**>(&sc.g_canon), static_cast<size_t>(max_rows) * chw * 4) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&sc.g_v), static_cast<size_t>(max_rows) * p1 * 4) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&sc.g_pi), static_cast<size_t>(max_rows) * m * 4) != hipSuccess)
        return nfail(AZMI_ERR_OOM, "hipMalloc(row-list staging) failed");
      sc.g_rows = max_rows;
    }
    k_gather_rows<<<max_rows, 256, 0, st>>>(dev_canonical, dev_rows, dev_row_count, max_rows, chw, sc.g_canon);
    const int rc = net_forward_live(net, sc.g_canon, sc.g_v, sc.g_pi, max_rows, stream, nullptr);
    if (rc != AZMI_OK) return rc;
    k_scatter_rows<<<max_rows, 64, 0, st>>>(sc.g_v, sc.g_pi, dev_rows, dev_row_count, p1, m, dev_v, dev_pi);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "row-list gather/scatter launch: %s", hipGetErrorString(e));
    return AZMI_OK;
  }
  c4_launch(net, dev_canonical, dev_v, dev_pi, max_rows, dev_rows, dev_row_count, static_cast<hipStream_t>(stream));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "k_leafnet launch: %s", hipGetErrorString(e));
  return AZMI_OK;
}

namespace {
struct HostEvalCtx {          // per calling thread: its own stream and staging buffers
  int device = -1;
  hipStream_t st = nullptr;
  float *c = nullptr, *v = nullptr, *pi = nullptr;
  uint32_t rows = 0, chw = 0, p1 = 0, m = 0;
  ~HostEvalCtx() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    if (c) (void)hipFree(c); if (v) (void)hipFree(v); if (pi) (void)hipFree(pi);
    if (st) (void)hipStreamDestroy(st);
  }
};
}  // namespace

void azmi_net_eval_host(const float* canonical, uint32_t n, float* v, float* pi, void* net_v) {
  azmi_net* net = static_cast<azmi_net*>(net_v);
  if (!net || !canonical || !v || !pi || n == 0) return;
  uint32_t chw, p1, m;
  if (net->f32) { azmi_f32::dims(net->f32, &chw, &p1, &m); }
  else if (net->spatial) { chw = net->sd.C_in * net->sd.H * net->sd.W; p1 = net->sd.num_players + 1; m = net->sd.num_moves; }
  else { chw = net->nd.C_in * net->nd.H * net->nd.W; p1 = net->nd.num_players + 1; m = net->nd.num_moves; }
  auto poison = [&]() {
    for (size_t i = 0; i < static_cast<size_t>(n) * p1; ++i) v[i] = __builtin_nanf("");
    for (size_t i = 0; i < static_cast<size_t>(n) * m; ++i) pi[i] = __builtin_nanf("");
  };
  thread_local HostEvalCtx ctx;
  if (hipSetDevice(net->device) != hipSuccess) { poison(); return; }
  if (ctx.device != net->device || ctx.rows < n || ctx.chw != chw || ctx.p1 != p1 || ctx.m != m) {
    if (ctx.c) (void)hipFree(ctx.c); if (ctx.v) (void)hipFree(ctx.v); if (ctx.pi) (void)hipFree(ctx.pi);
    ctx.c = ctx.v = ctx.pi = nullptr;
    if (!ctx.st && hipStreamCreateWithFlags(&ctx.st, hipStreamNonBlocking) != hipSuccess) { poison(); return; }
    const uint32_t rows = n < 256 ? 256 : n;
    if (hipMalloc(reinterpret_cast<void**>(&ctx.c), static_cast<size_t>(rows) * chw * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ctx.v), static_cast<size_t>(rows) * p1 * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ctx.pi), static_cast<size_t>(rows) * m * 4) != hipSuccess) { poison(); return; }
    ctx.device = net->device; ctx.rows = rows; ctx.chw = chw; ctx.p1 = p1; ctx.m = m;
  }
  bool ok = hipMemcpyAsync(ctx.c, canonical, static_cast<size_t>(n) * chw * 4, hipMemcpyHostToDevice, ctx.st) == hipSuccess;
  ok = ok && azmi_net_forward(net, ctx.c, ctx.v, ctx.pi, n, ctx.st) == AZMI_OK;
  ok = ok && hipMemcpyAsync(v, ctx.v, static_cast<size_t>(n) * p1 * 4, hipMemcpyDeviceToHost, ctx.st) == hipSuccess;
  ok = ok && hipMemcpyAsync(pi, ctx.pi, static_cast<size_t>(n) * m * 4, hipMemcpyDeviceToHost, ctx.st) == hipSuccess;
  ok = ok && hipStreamSynchronize(ctx.st) == hipSuccess;
  if (!ok) poison();
}

int azmi_net_c4_view_get(const azmi_net* net, azmi_net_c4_view* out) {
  if (!net || !out || net->f32 || net->spatial) return 0;
  out->nd = net->nd; out->np = net->np; out->x3 = net->x3 ? 1 : 0;
  out->lds_bytes = net->x3 ? c4::TileBigX3::LDS_BYTES : c4::TileSmall::LDS_BYTES;   // (bf16: the engine's fused launch runs the small tile)
  return 1;
}

}  // extern "C"
