// fp32 path of the leaf policy/value network: the inference forward of the reference's ResNet-mode
// NNArch (/root/reference/src/neural_net.py:233-263, 448-510) in plain fp32, layer by layer.
// This is the precision the north star's 1e-5 tolerance on (v, pi) refers to; the MFMA kernels in
// leafnet.hip are the bf16-operand fast path (what the reference runs under autocast).
// Throughput is not the point here (VALU FMAs, activations round-trip through HBM between layers).
//
// Weight blob (all f32, torch layouts, inference BatchNorms folded on the host in double):
//   stem    W[CH][Cin][3][3] b[CH]                                  (CH = NNArgs.num_channels, any width)
//   block i a1[CH] b1[CH] | W1[CH][CH][3][3] c1[CH] | W2[CH][CH][3][3]
//   value   Wv[HC][CH] bv[HC] | v_head_convs x (W[HC][HC][3][3] b[HC]) | fc1 W[Hd][HC] b[Hd] |
//           (v_fc_layers-1) x (W[Hd][Hd] b[Hd]) | fc2 W[P+1][Hd] b[P+1]
//   policy  Wp[HC][CH] bp[HC] | pi_head_convs x (W[HC][HC][3][3] b[HC]) |
//           flat: Wfc[M][HC*H*W] b[M]      spatial: Wpol[PC][HC] bpol[PC]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "leafnet_f32.h"

namespace {

thread_local std::string g_err;
int fail(const char** err, int code, const char* what, hipError_t e = hipSuccess) {
  char buf[256];
  if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof buf, "%s", what);
  g_err = buf;
  if (err) *err = g_err.c_str();
  return code;
}

// out[b][co][p] = act_out( bias[co] + sum_{ci,tap} W[co][ci][tap] * act_in(in[b][ci][p + tap]) ) (+ res[b][co][p])
// act_in: relu(a[ci] * x + b[ci]) when pre_a != nullptr (pre-activation BatchNorm + ReLU); zero padding is applied
// AFTER the input activation, as in the reference (the conv pads its already-activated input).
template <int K>
__global__ void k_conv(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                       const float* __restrict__ pre_a, const float* __restrict__ pre_b, const float* __restrict__ res,
                       float* __restrict__ out, uint32_t B, int Cin, int Cout, int H, int W, int relu_out) {
  const int HW = H * W;
  const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<size_t>(B) * Cout * HW) return;
  const int p = static_cast<int>(idx % HW), co = static_cast<int>((idx / HW) % Cout);
  const size_t b = idx / (static_cast<size_t>(HW) * Cout);
  const int h = p / W, x = p % W;
  float acc = bias ? bias[co] : 0.0f;
  const float* ib = in + b * Cin * HW;
  const float* wc = w + static_cast<size_t>(co) * Cin * K * K;
  for (int ci = 0; ci < Cin; ++ci) {
    const float a = pre_a ? pre_a[ci] : 1.0f, sh = pre_a ? pre_b[ci] : 0.0f;
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
      const int hh = h + t / K - K / 2, ww = x + t % K - K / 2;
      if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
      float v = ib[ci * HW + hh * W + ww];
      if (pre_a) v = fmaxf(a * v + sh, 0.0f);
      acc += wc[ci * K * K + t] * v;
    }
  }
  if (relu_out) acc = fmaxf(acc, 0.0f);
  if (res) acc += res[idx];
  out[idx] = acc;
}

__global__ void k_avgpool(const float* __restrict__ in, float* __restrict__ out, uint32_t B, int C, int HW) {
  const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<size_t>(B) * C) return;
  float s = 0.0f;
  for (int p = 0; p < HW; ++p) s += in[idx * HW + p];
  out[idx] = s / static_cast<float>(HW);
}

// out[b][n] = act(bias[n] + sum_k W[n][k] * in[b][k])
__global__ void k_fc(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                     float* __restrict__ out, uint32_t B, int K, int N, int relu) {
  const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<size_t>(B) * N) return;
  const int n = static_cast<int>(idx % N);
  const size_t b = idx / N;
  float acc = bias[n];
  const float* x = in + b * K;
  const float* wr = w + static_cast<size_t>(n) * K;
  for (int k = 0; k < K; ++k) acc += wr[k] * x[k];
  out[idx] = relu ? fmaxf(acc, 0.0f) : acc;
}

// one wavefront per row: out = softmax(row).  `chw` != 0: the row is read as [C][HW] and written as [HW][C]
// (spatial policy head: permute(0, 2, 3, 1) + flatten, neural_net.py:483-487)
// `glob` != nullptr: the row's last G entries are the global-action logits glob[row][G] behind the C*HW spatial ones
// (neural_net.py:486-493), the spatial block of row r then starts at in + r * C * HW
__global__ void k_softmax(const float* __restrict__ in, float* __restrict__ out, uint32_t B, int N, int C, int HW,
                          const float* __restrict__ glob = nullptr, int G = 0) {
  const uint32_t row = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x % 64;
  if (row >= B) return;
  const float* x = in + static_cast<size_t>(row) * (N - G);
  const float* gx = glob ? glob + static_cast<size_t>(row) * G : nullptr;
  auto src = [&](int e) { return e >= N - G ? gx[e - (N - G)] : (C ? x[(e % C) * HW + e / C] : x[e]); };
  float mx = -__builtin_inff();
  for (int e = lane; e < N; e += 64) mx = fmaxf(mx, src(e));
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float sum = 0.0f;
  for (int e = lane; e < N; e += 64) sum += expf(src(e) - mx);
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  for (int e = lane; e < N; e += 64) out[static_cast<size_t>(row) * N + e] = expf(src(e) - mx) / sum;
}

// nn.LayerNorm(G) over the last dimension (biased variance, eps 1e-5), one thread per row
__global__ void k_layernorm(const float* __restrict__ in, const float* __restrict__ gamma, const float* __restrict__ beta,
                            float* __restrict__ out, uint32_t B, int G) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= B) return;
  const float* x = in + static_cast<size_t>(row) * G;
  float mean = 0.0f;
  for (int i = 0; i < G; ++i) mean += x[i];
  mean /= static_cast<float>(G);
  float var = 0.0f;
  for (int i = 0; i < G; ++i) var += (x[i] - mean) * (x[i] - mean);
  var /= static_cast<float>(G);
  const float inv = 1.0f / sqrtf(var + 1e-5f);
  for (int i = 0; i < G; ++i) out[static_cast<size_t>(row) * G + i] = (x[i] - mean) * inv * gamma[i] + beta[i];
}

struct Net {
  azmi_net_desc d{};
  int device = 0;
  float* blob = nullptr;
  float* buf[3] = {nullptr, nullptr, nullptr};   // [rows][64][HW] ping-pong activations
  float* small[2] = {nullptr, nullptr};          // [rows][max(Hd, M, HC)] head vectors
  uint32_t rows = 0;
};

size_t count_floats(const azmi_net_desc* d) {
  const size_t CH = d->channels;
  const size_t HW = static_cast<size_t>(d->height) * d->width, HC = d->head_channels, Hd = d->v_hidden, P1 = d->num_players + 1;
  size_t n = static_cast<size_t>(CH) * d->in_channels * 9 + CH;
  n += static_cast<size_t>(d->depth) * (2 * CH + static_cast<size_t>(CH) * CH * 9 + CH + static_cast<size_t>(CH) * CH * 9);
  n += HC * CH + HC + static_cast<size_t>(d->v_head_convs) * (HC * HC * 9 + HC);
  n += Hd * HC + Hd + static_cast<size_t>(d->v_fc_layers - 1) * (Hd * Hd + Hd) + P1 * Hd + P1;
  n += HC * CH + HC + static_cast<size_t>(d->pi_head_convs) * (HC * HC * 9 + HC);
  if (d->policy_channels > 0) {
    n += static_cast<size_t>(d->policy_channels) * HC + d->policy_channels;
    const size_t G = static_cast<size_t>(d->num_moves) - static_cast<size_t>(d->policy_channels) * HW, Hp = d->pi_hidden;
    if (G > 0) n += Hp * HC + Hp + G * Hp + G + 2 * G;     // pi_global: Linear, Linear, LayerNorm (neural_net.py:421-426)
  } else n += static_cast<size_t>(d->num_moves) * HC * HW + d->num_moves;
  return n;
}

}  // namespace

namespace azmi_f32 {

size_t blob_bytes(const azmi_net_desc* d) { return count_floats(d) * sizeof(float); }

void dims(void* impl, uint32_t* chw, uint32_t* p1, uint32_t* m) {
  const azmi_net_desc& d = static_cast<Net*>(impl)->d;
  *chw = d.in_channels * d.height * d.width; *p1 = d.num_players + 1; *m = d.num_moves;
}

void destroy(void* impl) {
  auto* n = static_cast<Net*>(impl);
  if (!n) return;
  (void)hipSetDevice(n->device);
  if (n->blob) (void)hipFree(n->blob);
  for (float* p : n->buf) if (p) (void)hipFree(p);
  for (float* p : n->small) if (p) (void)hipFree(p);
  delete n;
}

int reserve(void* impl, uint32_t batch, const char** err) {
  auto* n = static_cast<Net*>(impl);
  if (batch <= n->rows) return AZMI_OK;
  if (hipSetDevice(n->device) != hipSuccess) return fail(err, AZMI_ERR_NO_DEVICE, "hipSetDevice failed");
  (void)hipDeviceSynchronize();
  for (float*& p : n->buf) { if (p) (void)hipFree(p); p = nullptr; }
  for (float*& p : n->small) { if (p) (void)hipFree(p); p = nullptr; }
  n->rows = 0;
  const size_t HW = static_cast<size_t>(n->d.height) * n->d.width;
  const size_t CH = std::max<size_t>(n->d.channels, n->d.head_channels);
  const size_t wide = std::max<size_t>(std::max<size_t>(std::max<size_t>(n->d.v_hidden, n->d.pi_hidden), n->d.num_moves), n->d.head_channels);
  for (float*& p : n->buf)
    if (hipMalloc(reinterpret_cast<void**>(&p), static_cast<size_t>(batch) * CH * HW * sizeof(float)) != hipSuccess)
      return fail(err, AZMI_ERR_OOM, "hipMalloc(fp32 activations) failed");
  for (float*& p : n->small)
    if (hipMalloc(reinterpret_cast<void**>(&p), static_cast<size_t>(batch) * wide * sizeof(float)) != hipSuccess)
      return fail(err, AZMI_ERR_OOM, "hipMalloc(fp32 head vectors) failed");
  n->rows = batch;
  return AZMI_OK;
}

int create(const azmi_net_desc* d, const void* blob, size_t bytes, int device, void** impl, const char** err) {
  if (d->channels < 1 || d->kernel_size != 3) return fail(err, AZMI_ERR_INVALID, "fp32 leaf net: 3x3 convolutions only");
  if (d->head_channels < 1 || d->v_fc_layers < 1 || d->v_head_convs < 0 || d->pi_head_convs < 0)
    return fail(err, AZMI_ERR_INVALID, "fp32 leaf net: head sizes out of range");
  if (d->policy_channels > 0 && d->policy_channels * d->height * d->width > d->num_moves)
    return fail(err, AZMI_ERR_INVALID, "fp32 leaf net: the spatial block exceeds num_moves");
  if (d->policy_channels > 0 && d->policy_channels * d->height * d->width < d->num_moves && d->pi_hidden < 1)
    return fail(err, AZMI_ERR_INVALID, "fp32 leaf net: global actions need pi_hidden");
  if (bytes != blob_bytes(d)) { char b[128]; snprintf(b, sizeof b, "fp32 weight blob is %zu bytes, expected %zu", bytes, blob_bytes(d)); return fail(err, AZMI_ERR_INVALID, b); }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(err, AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
  if (hipSetDevice(device) != hipSuccess) return fail(err, AZMI_ERR_NO_DEVICE, "hipSetDevice failed");
  auto* n = new Net();
  n->d = *d; n->device = device;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&n->blob), bytes);
  if (e != hipSuccess) { delete n; return fail(err, AZMI_ERR_OOM, "hipMalloc(weights)", e); }
  e = hipMemcpy(n->blob, blob, bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) { destroy(n); return fail(err, AZMI_ERR_NO_DEVICE, "weight upload", e); }
  const int rc = reserve(n, 4096, err);   // up front: forward() may run under stream capture
  if (rc != AZMI_OK) { destroy(n); return rc; }
  *impl = n;
  return AZMI_OK;
}

int forward(void* impl, const float* canon, float* v_out, float* pi_out, uint32_t B, void* stream, const char** err) {
  auto* n = static_cast<Net*>(impl);
  if (B > n->rows) { const int rc = reserve(n, B, err); if (rc != AZMI_OK) return rc; }
  const azmi_net_desc& d = n->d;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int CH = d.channels;
  const int H = d.height, W = d.width, HW = H * W, HC = d.head_channels, Hd = d.v_hidden, P1 = d.num_players + 1, M = d.num_moves;
  const float* p = n->blob;
  auto take = [&](size_t count) { const float* q = p; p += count; return q; };
  auto conv3 = [&](const float* in, const float* w, const float* b, const float* pa, const float* pb, const float* res, float* out, int cin, int cout, int relu) {
    const size_t total = static_cast<size_t>(B) * cout * HW;
    k_conv<3><<<static_cast<uint32_t>((total + 255) / 256), 256, 0, st>>>(in, w, b, pa, pb, res, out, B, cin, cout, H, W, relu);
  };
  auto conv1 = [&](const float* in, const float* w, const float* b, float* out, int cin, int cout, int relu) {
    const size_t total = static_cast<size_t>(B) * cout * HW;
    k_conv<1><<<static_cast<uint32_t>((total + 255) / 256), 256, 0, st>>>(in, w, b, nullptr, nullptr, nullptr, out, B, cin, cout, H, W, relu);
  };
  auto fc = [&](const float* in, const float* w, const float* b, float* out, int K, int N, int relu) {
    const size_t total = static_cast<size_t>(B) * N;
    k_fc<<<static_cast<uint32_t>((total + 255) / 256), 256, 0, st>>>(in, w, b, out, B, K, N, relu);
  };
  float *s = n->buf[0], *t = n->buf[1], *u = n->buf[2];
  {  // stem
    const float* w = take(static_cast<size_t>(CH) * d.in_channels * 9); const float* b = take(CH);
    conv3(canon, w, b, nullptr, nullptr, nullptr, s, d.in_channels, CH, 0);
  }
  for (int i = 0; i < d.depth; ++i) {  // out = conv2(relu(bn2(conv1(relu(bn1(x)))))) + x
    const float* a1 = take(CH); const float* b1 = take(CH);
    const float* w1 = take(static_cast<size_t>(CH) * CH * 9); const float* c1 = take(CH);
    const float* w2 = take(static_cast<size_t>(CH) * CH * 9);
    conv3(s, w1, c1, a1, b1, nullptr, t, CH, CH, 1);
    conv3(t, w2, nullptr, nullptr, nullptr, s, u, CH, CH, 0);
    float* x = s; s = u; u = x;
  }
  {  // value head
    const float* wv = take(static_cast<size_t>(HC) * CH); const float* bv = take(HC);
    conv1(s, wv, bv, t, CH, HC, 1);
    float *a = t, *b2 = u;
    for (int i = 0; i < d.v_head_convs; ++i) {
      const float* w = take(static_cast<size_t>(HC) * HC * 9); const float* b = take(HC);
      conv3(a, w, b, nullptr, nullptr, nullptr, b2, HC, HC, 1);
      float* x = a; a = b2; b2 = x;
    }
    k_avgpool<<<static_cast<uint32_t>((static_cast<size_t>(B) * HC + 255) / 256), 256, 0, st>>>(a, n->small[0], B, HC, HW);
    float *x0 = n->small[0], *x1 = n->small[1];
    const float* w = take(static_cast<size_t>(Hd) * HC); const float* b = take(Hd);
    fc(x0, w, b, x1, HC, Hd, 1);
    for (int i = 0; i + 1 < d.v_fc_layers; ++i) {
      const float* we = take(static_cast<size_t>(Hd) * Hd); const float* be = take(Hd);
      fc(x1, we, be, x0, Hd, Hd, 1);
      float* x = x0; x0 = x1; x1 = x;
    }
    const float* w2 = take(static_cast<size_t>(P1) * Hd); const float* bb = take(P1);
    fc(x1, w2, bb, x0, Hd, P1, 0);
    k_softmax<<<(B + 3) / 4, 256, 0, st>>>(x0, v_out, B, P1, 0, 0);
  }
  {  // policy head
    const float* wp = take(static_cast<size_t>(HC) * CH); const float* bp = take(HC);
    conv1(s, wp, bp, t, CH, HC, 1);
    float *a = t, *b2 = u;
    for (int i = 0; i < d.pi_head_convs; ++i) {
      const float* w = take(static_cast<size_t>(HC) * HC * 9); const float* b = take(HC);
      conv3(a, w, b, nullptr, nullptr, nullptr, b2, HC, HC, 1);
      float* x = a; a = b2; b2 = x;
    }
    if (d.policy_channels > 0) {
      const int PC = d.policy_channels;
      const float* w = take(static_cast<size_t>(PC) * HC); const float* b = take(PC);
      conv1(a, w, b, b2, HC, PC, 0);                                   // [B][PC][HW]
      const int G = M - PC * HW;
      if (G > 0) {   // pi_global over the average-pooled policy features (head_pool), neural_net.py:413-426, 486-493
        const int Hp = d.pi_hidden;
        k_avgpool<<<static_cast<uint32_t>((static_cast<size_t>(B) * HC + 255) / 256), 256, 0, st>>>(a, n->small[0], B, HC, HW);
        const float* w1 = take(static_cast<size_t>(Hp) * HC); const float* c1 = take(Hp);
        fc(n->small[0], w1, c1, n->small[1], HC, Hp, 1);
        const float* w2 = take(static_cast<size_t>(G) * Hp); const float* c2 = take(G);
        fc(n->small[1], w2, c2, n->small[0], Hp, G, 0);
        const float* lg = take(G); const float* lb = take(G);
        k_layernorm<<<(B + 63) / 64, 64, 0, st>>>(n->small[0], lg, lb, n->small[1], B, G);
        k_softmax<<<(B + 3) / 4, 256, 0, st>>>(b2, pi_out, B, M, PC, HW, n->small[1], G);
      } else
      k_softmax<<<(B + 3) / 4, 256, 0, st>>>(b2, pi_out, B, M, PC, HW);  // permuted to [HW][PC] on the fly
    } else {
      const float* w = take(static_cast<size_t>(M) * HC * HW); const float* b = take(M);
      fc(a, w, b, n->small[0], HC * HW, M, 0);                         // flatten order (c, h, w) = the buffer's
      k_softmax<<<(B + 3) / 4, 256, 0, st>>>(n->small[0], pi_out, B, M, 0, 0);
    }
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(err, AZMI_ERR_NO_DEVICE, "fp32 leaf net launch", e);
  return AZMI_OK;
}

}  // namespace azmi_f32
