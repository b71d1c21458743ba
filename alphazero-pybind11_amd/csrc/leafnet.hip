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
// One workgroup (4 waves, one per SIMD) carries a tile of TB = 8 boards through the WHOLE tower:
//   * the residual stream s stays in fp32 MFMA accumulators for the entire kernel (conv2's
//     accumulator is the stream itself: C-in = s, C-out = s + conv(u));
//   * activations that feed a convolution live in LDS as bf16 [board][padded cell][64 ch]
//     with a zero halo ring, so the 9 taps of the implicit GEMM are plain shifted reads;
//   * each convolution is D[co][pixel] = sum_k W[co][k] * X[k][pixel] on
//     v_mfma_f32_16x16x32_bf16 with A = weights (4 m-tiles = 64 output channels) and
//     B = activations (n-tiles of 16 pixels), which leaves every lane holding 4 consecutive
//     channels of one pixel — exactly the 8-byte store the LDS activation layout wants;
//   * the next convolution's 72 KB of weights are prefetched from L2 into registers while the
//     current one runs on the matrix cores, and dropped into LDS between the two barriers
//     that separate convolutions (weights are pre-swizzled on the host into MFMA fragment
//     order, so both the global load and the LDS read are flat 16 B-per-lane streams).
// LDS: 82,944 B activations + 73,728 B weights = 156,672 B of the CU's 160 KB (head scratch reuses it).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/azmi.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int CH = 64;            // trunk channels
constexpr int HC = 32;            // head channels (value / policy each)
constexpr int TB = 8;             // boards per workgroup
constexpr int CELL_BYTES = 144;   // 64 bf16 + 16 B pad (bank spread for the b128 reads)
constexpr int WFRAG_BYTES = 1024; // one MFMA A-fragment: 64 lanes x 16 B
constexpr int MT = 4;             // m-tiles (16 output channels each)

struct NetDesc {
  int C_in, H, W, depth, num_moves, num_players, v_hidden;
};

struct NetPtrs {           // device pointers into the folded weight blob
  const uint8_t* stem_w;   // [2 ks][4 mt] fragments
  const float* stem_b;     // [64]
  const uint8_t* blocks;   // per block: a1[64] b1[64] c1[64] (fp32) | conv1 frags | conv2 frags
  const uint8_t* head_w;   // [2 ks][4 mt] fragments (rows 0-31 value conv, 32-63 policy conv)
  const float* head_b;     // [64]
  const float* v_fc1_w;    // [v_hidden][32]
  const float* v_fc1_b;    // [v_hidden]
  const float* v_fc2_w;    // [P+1][v_hidden]
  const float* v_fc2_b;    // [P+1]
  const float* pi_fc_w;    // [M][32*H*W]
  const float* pi_fc_b;    // [M]
};

template <int H, int W>
struct Geo {
  static constexpr int PIX = H * W;                 // 42
  static constexpr int PH = H + 2, PW = W + 2;      // padded 8 x 9
  static constexpr int CELLS = PH * PW;             // 72
  static constexpr int NPIX = TB * PIX;             // 336 GEMM columns
  static constexpr int NT = (NPIX + 15) / 16;       // 21 n-tiles
  static constexpr int NT_W = (NT + 3) / 4;         // n-tiles per wave (6)
  static constexpr int ACT_BYTES = TB * CELLS * CELL_BYTES;  // 82,944
  static constexpr int KS3 = 9 * CH / 32;           // 18 k-steps of a 3x3 conv
  static constexpr int WCONV_BYTES = KS3 * MT * WFRAG_BYTES;  // 73,728
};

__device__ __forceinline__ bf16x8 lds_read_frag(const uint8_t* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}

template <int H, int W, int MAXP1, int MAXM>
__global__ __launch_bounds__(256, 1) void k_leafnet(NetDesc nd, NetPtrs np, const float* __restrict__ canon,
                                                     float* __restrict__ v_out, float* __restrict__ pi_out,
                                                     uint32_t batch) {
  using G = Geo<H, W>;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* act = lds;                                   // activation cells
  uint8_t* wbuf = lds + G::ACT_BYTES;                   // weights of the running convolution

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, quad = lane >> 4;
  const uint32_t board0 = blockIdx.x * TB;

  // ---- per-lane pixel geometry of the wave's n-tiles: tile t = wave + 4 j ----------------------
  int cell_off[G::NT_W];   // byte offset of the lane's pixel cell inside `act`
  bool tile_on[G::NT_W];
#pragma unroll
  for (int j = 0; j < G::NT_W; ++j) {
    const int t = wave + 4 * j;
    const int n = t * 16 + col;
    tile_on[j] = t < G::NT;
    const int nn = (t < G::NT && n < G::NPIX) ? n : 0;
    const int b = nn / G::PIX, p = nn % G::PIX;
    const int h = p / W, w = p % W;
    cell_off[j] = (b * G::CELLS + (h + 1) * G::PW + (w + 1)) * CELL_BYTES;
  }

  // ---- zero the activation buffer once (the halo ring stays zero for the whole kernel) -----------
  for (int i = tid * 16; i < G::ACT_BYTES; i += 256 * 16) *reinterpret_cast<u32x4*>(act + i) = u32x4{0, 0, 0, 0};

  // ---- stem: im2col of the C_in input planes into act as [pixel][k = tap*C_in + ci] (k < 64) ----
  // staged through `wbuf` as raw fp32 planes first
  float* raw = reinterpret_cast<float*>(wbuf + 16384);  // [TB][C_in][H][W]
  const int plane_sz = nd.C_in * G::PIX;
  for (int i = tid; i < TB * plane_sz; i += 256) {
    const uint32_t b = board0 + i / plane_sz;
    raw[i] = b < batch ? canon[static_cast<size_t>(b) * plane_sz + (i % plane_sz)] : 0.0f;
  }
  // stem weights: 2 k-steps x 4 m-tiles = 8 KB
  for (int i = tid * 16; i < 2 * MT * WFRAG_BYTES; i += 256 * 16)
    *reinterpret_cast<u32x4*>(wbuf + i) = *reinterpret_cast<const u32x4*>(np.stem_w + i);
  __syncthreads();
  {
    // the stem's B operand is laid out like a 1x1 conv input: one 128 B row per pixel, stored in the
    // pixel's own (interior) cell; the halo is not involved because the taps are already unrolled in k
    const int kdim = 9 * nd.C_in;  // <= 64
    for (int i = tid; i < G::NPIX * 64; i += 256) {
      const int n = i >> 6, k = i & 63;
      const int b = n / G::PIX, p = n % G::PIX, h = p / W, w = p % W;
      float val = 0.0f;
      if (k < kdim) {
        const int tap = k / nd.C_in, ci = k % nd.C_in;
        const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) val = raw[(b * nd.C_in + ci) * G::PIX + hh * W + ww];
      }
      const int off = (b * G::CELLS + (h + 1) * G::PW + (w + 1)) * CELL_BYTES + k * 2;
      *reinterpret_cast<__bf16*>(act + off) = static_cast<__bf16>(val);
    }
  }
  __syncthreads();

  // ---- residual stream: accumulators s[j][mt] (fp32), kept for the whole kernel -------------------
  f32x4 s[G::NT_W][MT];
  {
    f32x4 bias[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bias[mt] = *reinterpret_cast<const f32x4*>(np.stem_b + mt * 16 + quad * 4);
#pragma unroll
    for (int j = 0; j < G::NT_W; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) s[j][mt] = bias[mt];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = lds_read_frag(wbuf + (ks * MT + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) {
        if (!tile_on[j]) continue;
        const bf16x8 b = lds_read_frag(act + cell_off[j] + ks * 64 + quad * 16);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) s[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, s[j][mt], 0, 0, 0);
      }
    }
  }

  // weight prefetch registers: this thread's 288 B slice of the next convolution (18 x 16 B)
  u32x4 wnext[G::KS3];
  auto prefetch = [&](const uint8_t* src) {
#pragma unroll
    for (int i = 0; i < G::KS3; ++i) wnext[i] = *reinterpret_cast<const u32x4*>(src + (i * 256 + tid) * 16);
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < G::KS3; ++i) *reinterpret_cast<u32x4*>(wbuf + (i * 256 + tid) * 16) = wnext[i];
  };
  // epilogue helper: store one accumulator tile as bf16 into the lane's pixel cell
  auto store_tile = [&](int j, int mt, f32x4 val) {
    bf16x4 o;
    o[0] = static_cast<__bf16>(val[0]); o[1] = static_cast<__bf16>(val[1]);
    o[2] = static_cast<__bf16>(val[2]); o[3] = static_cast<__bf16>(val[3]);
    *reinterpret_cast<bf16x4*>(act + cell_off[j] + (mt * 16 + quad * 4) * 2) = o;
  };
  // one 3x3 convolution over `act` with the weights in `wbuf`, accumulating into acc[][]
  auto conv3x3 = [&](f32x4 (&acc)[G::NT_W][MT]) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int tap_off = ((tap / 3 - 1) * G::PW + (tap % 3 - 1)) * CELL_BYTES;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int ks = tap * 2 + half;
        bf16x8 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = lds_read_frag(wbuf + (ks * MT + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
        for (int j = 0; j < G::NT_W; ++j) {
          if (!tile_on[j]) continue;
          const bf16x8 b = lds_read_frag(act + cell_off[j] + tap_off + half * 64 + quad * 16);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, acc[j][mt], 0, 0, 0);
        }
      }
    }
  };

  const size_t block_stride = 3 * CH * sizeof(float) + 2 * static_cast<size_t>(G::WCONV_BYTES);
  prefetch(np.blocks + 3 * CH * sizeof(float));  // block 0 conv1 weights
  __syncthreads();                               // stem reads of act / wbuf are finished

  for (int blk = 0; blk < nd.depth; ++blk) {
    const uint8_t* bp = np.blocks + blk * block_stride;
    const float* affine = reinterpret_cast<const float*>(bp);  // a1[64] b1[64] c1[64]
    // t = relu(a1 * s + b1) -> act ; conv1 weights -> wbuf
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(affine + mt * 16 + quad * 4);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(affine + CH + mt * 16 + quad * 4);
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) {
        if (!tile_on[j]) continue;
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaxf(a1[r] * s[j][mt][r] + b1[r], 0.0f);
        store_tile(j, mt, t);
      }
    }
    commit();
    __syncthreads();
    prefetch(bp + 3 * CH * sizeof(float) + G::WCONV_BYTES);  // conv2 weights
    // u = relu(conv1(t) + c1)
    f32x4 u[G::NT_W][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(affine + 2 * CH + mt * 16 + quad * 4);
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) u[j][mt] = c1;
    }
    conv3x3(u);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) {
        if (!tile_on[j]) continue;
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaxf(u[j][mt][r], 0.0f);
        store_tile(j, mt, t);
      }
    commit();
    __syncthreads();
    // s = s + conv2(u); meanwhile fetch the next block's conv1 weights (or the head conv)
    if (blk + 1 < nd.depth) prefetch(bp + block_stride + 3 * CH * sizeof(float));
    conv3x3(s);
    __syncthreads();
  }

  // ---- heads: h = relu(conv1x1(s) + bh), 64 rows = 32 value + 32 policy channels -------------------
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < G::NT_W; ++j)
      if (tile_on[j]) store_tile(j, mt, s[j][mt]);
  for (int i = tid * 16; i < 2 * MT * WFRAG_BYTES; i += 256 * 16)
    *reinterpret_cast<u32x4*>(wbuf + i) = *reinterpret_cast<const u32x4*>(np.head_w + i);
  __syncthreads();
  f32x4 hacc[G::NT_W][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 bh = *reinterpret_cast<const f32x4*>(np.head_b + mt * 16 + quad * 4);
#pragma unroll
    for (int j = 0; j < G::NT_W; ++j) hacc[j][mt] = bh;
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = lds_read_frag(wbuf + (ks * MT + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
    for (int j = 0; j < G::NT_W; ++j) {
      if (!tile_on[j]) continue;
      const bf16x8 b = lds_read_frag(act + cell_off[j] + ks * 64 + quad * 16);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) hacc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, hacc[j][mt], 0, 0, 0);
    }
  }
  __syncthreads();
  // head activations as fp32 [pixel][64] over the (now free) act + wbuf region
  float* hbuf = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < G::NT_W; ++j) {
      const int t = wave + 4 * j, n = t * 16 + col;
      if (t < G::NT && n < G::NPIX) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaxf(hacc[j][mt][r], 0.0f);
        *reinterpret_cast<f32x4*>(hbuf + n * 64 + mt * 16 + quad * 4) = o;
      }
    }
  __syncthreads();

  // ---- value head: avgpool -> fc1 -> relu -> fc2 -> softmax (fp32 VALU) --------------------------------
  float* vpool = hbuf + G::NPIX * 64;   // [TB][32]  (LDS behind hbuf, still inside act + wbuf)
  float* vh = vpool + TB * HC;          // [TB][v_hidden]   (v_hidden <= 256)
  float* logits = vh + TB * 256;        // [TB][MAXP1 + MAXM]
  {
    const int b = tid >> 5, c = tid & 31;  // 8 boards x 32 channels = 256 threads
    float acc = 0.0f;
    for (int p = 0; p < G::PIX; ++p) acc += hbuf[(b * G::PIX + p) * 64 + c];
    vpool[b * HC + c] = acc / static_cast<float>(G::PIX);
  }
  __syncthreads();
  for (int o = tid; o < nd.v_hidden; o += 256) {
    float wrow[HC];
#pragma unroll
    for (int i = 0; i < HC; ++i) wrow[i] = np.v_fc1_w[o * HC + i];
    const float bias = np.v_fc1_b[o];
    for (int b = 0; b < TB; ++b) {
      float acc = bias;
#pragma unroll
      for (int i = 0; i < HC; ++i) acc += wrow[i] * vpool[b * HC + i];
      vh[b * 256 + o] = fmaxf(acc, 0.0f);
    }
  }
  __syncthreads();
  const int P1 = nd.num_players + 1, M = nd.num_moves;
  // value logits: (board, output) pairs, one wave-quarter (16 lanes) each
  {
    const int pair = tid >> 4, sub = tid & 15;  // 16 pairs per pass
    for (int q = pair; q < TB * P1; q += 16) {
      const int b = q / P1, o = q % P1;
      float acc = 0.0f;
      for (int i = sub; i < nd.v_hidden; i += 16) acc += np.v_fc2_w[o * nd.v_hidden + i] * vh[b * 256 + i];
      for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
      if (sub == 0) logits[b * (MAXP1 + MAXM) + o] = acc + np.v_fc2_b[o];
    }
  }
  // policy logits: flatten order (c, h, w) of the 32 policy channels -> index c * PIX + p
  {
    const int pair = tid >> 5, sub = tid & 31;  // 8 pairs per pass, 32 lanes each
    const int feat = HC * G::PIX;
    for (int q = pair; q < TB * M; q += 8) {
      const int b = q / M, m = q % M;
      float acc = 0.0f;
      for (int i = sub; i < feat; i += 32) {
        const int c = i / G::PIX, p = i % G::PIX;
        acc += np.pi_fc_w[static_cast<size_t>(m) * feat + i] * hbuf[(b * G::PIX + p) * 64 + HC + c];
      }
      for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 32);
      if (sub == 0) logits[b * (MAXP1 + MAXM) + MAXP1 + m] = acc + np.pi_fc_b[m];
    }
  }
  __syncthreads();
  if (tid < TB) {  // softmax = exp(log_softmax), neural_net.py:468,508,816
    const uint32_t b = board0 + tid;
    if (b < batch) {
      const float* lg = logits + tid * (MAXP1 + MAXM);
      float mx = lg[0];
      for (int i = 1; i < P1; ++i) mx = fmaxf(mx, lg[i]);
      float sum = 0.0f;
      for (int i = 0; i < P1; ++i) sum += expf(lg[i] - mx);
      for (int i = 0; i < P1; ++i) v_out[static_cast<size_t>(b) * P1 + i] = expf(lg[i] - mx) / sum;
      const float* lp = lg + MAXP1;
      mx = lp[0];
      for (int i = 1; i < M; ++i) mx = fmaxf(mx, lp[i]);
      sum = 0.0f;
      for (int i = 0; i < M; ++i) sum += expf(lp[i] - mx);
      for (int i = 0; i < M; ++i) pi_out[static_cast<size_t>(b) * M + i] = expf(lp[i] - mx) / sum;
    }
  }
}

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
  void* blob = nullptr;
  size_t blob_bytes = 0;
  int device = 0;
  size_t lds_bytes = 0;
};

extern "C" {

const char* azmi_net_last_error(void) { return g_net_err.c_str(); }

size_t azmi_net_blob_bytes(const azmi_net_desc* d) {
  if (!d) return 0;
  const size_t wconv = 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
  size_t n = wsmall + CH * 4;
  n += static_cast<size_t>(d->depth) * (3 * CH * 4 + 2 * wconv);
  n += wsmall + CH * 4;
  n += (static_cast<size_t>(d->v_hidden) * HC + d->v_hidden) * 4;
  n += (static_cast<size_t>(d->num_players + 1) * d->v_hidden + d->num_players + 1) * 4;
  n += (static_cast<size_t>(d->num_moves) * HC * d->height * d->width + d->num_moves) * 4;
  return n;
}

int azmi_net_create(const azmi_net_desc* d, const void* blob, size_t blob_bytes, int device, azmi_net** out) {
  if (!d || !blob || !out) return nfail(AZMI_ERR_INVALID, "null argument");
  if (d->channels != CH || d->head_channels != HC || d->kernel_size != 3)
    return nfail(AZMI_ERR_INVALID, "leaf net kernel covers 64 trunk channels, 32 head channels, 3x3 convs");
  if (!(d->height == 6 && d->width == 7)) return nfail(AZMI_ERR_INVALID, "leaf net kernel: board %dx%d not instantiated", d->height, d->width);
  if (9 * d->in_channels > 64) return nfail(AZMI_ERR_INVALID, "stem supports 9*C_in <= 64");
  if (d->v_hidden > 256 || d->num_players + 1 > 4 || d->num_moves > 16) return nfail(AZMI_ERR_INVALID, "head sizes out of range");
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
  const size_t wconv = 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
  NetPtrs& np = net->np;
  np.stem_w = p; p += wsmall;
  np.stem_b = reinterpret_cast<const float*>(p); p += CH * 4;
  np.blocks = p; p += static_cast<size_t>(d->depth) * (3 * CH * 4 + 2 * wconv);
  np.head_w = p; p += wsmall;
  np.head_b = reinterpret_cast<const float*>(p); p += CH * 4;
  np.v_fc1_w = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->v_hidden) * HC * 4;
  np.v_fc1_b = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->v_hidden) * 4;
  np.v_fc2_w = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->num_players + 1) * d->v_hidden * 4;
  np.v_fc2_b = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->num_players + 1) * 4;
  np.pi_fc_w = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->num_moves) * HC * d->height * d->width * 4;
  np.pi_fc_b = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->num_moves) * 4;
  using G = Geo<6, 7>;
  net->lds_bytes = G::ACT_BYTES + G::WCONV_BYTES;
  static_assert(G::NPIX * 64 * 4 + (TB * HC + TB * 256 + TB * (4 + 16)) * 4 <= G::ACT_BYTES + G::WCONV_BYTES, "head scratch must fit");
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_leafnet<6, 7, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(net->lds_bytes)) != hipSuccess) {
    (void)hipFree(net->blob); delete net;
    return nfail(AZMI_ERR_NO_DEVICE, "cannot reserve %zu bytes of LDS", net->lds_bytes);
  }
  *out = net;
  return AZMI_OK;
}

void azmi_net_destroy(azmi_net* net) {
  if (!net) return;
  (void)hipSetDevice(net->device);
  (void)hipFree(net->blob);
  delete net;
}

int azmi_net_forward(azmi_net* net, const float* dev_canonical, float* dev_v, float* dev_pi, uint32_t batch, void* stream) {
  if (!net || !dev_canonical || !dev_v || !dev_pi) return nfail(AZMI_ERR_INVALID, "null argument");
  if (batch == 0) return AZMI_OK;
  const uint32_t tiles = (batch + TB - 1) / TB;
  k_leafnet<6, 7, 4, 16><<<tiles, 256, net->lds_bytes, static_cast<hipStream_t>(stream)>>>(net->nd, net->np, dev_canonical, dev_v, dev_pi, batch);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "k_leafnet launch: %s", hipGetErrorString(e));
  return AZMI_OK;
}

}  // extern "C"
