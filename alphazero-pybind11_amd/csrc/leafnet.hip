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
// One workgroup (8 waves, two per SIMD) carries a tile of TB = 8 boards through the WHOLE tower:
//   * the residual stream s stays in fp32 MFMA accumulators for the entire kernel (conv2's
//     accumulator is the stream itself: C-in = s, C-out = s + conv(u));
//   * activations that feed a convolution live in LDS as bf16 in eight 8-channel planes of
//     [pixel][16 B] (bank-conflict-free fragment reads); out-of-board taps of the implicit GEMM
//     are redirected to a shared all-zero cell, in-board taps are plain shifted reads;
//   * each convolution is D[co][pixel] = sum_k W[co][k] * X[k][pixel] on
//     v_mfma_f32_16x16x32_bf16 with A = weights (4 m-tiles = 64 output channels) and
//     B = activations (n-tiles of 16 pixels), which leaves every lane holding 4 consecutive
//     channels of one pixel — exactly the 8-byte store the LDS activation layout wants;
//   * waves that own fewer n-tiles than the busiest one run the spare tile slot on dummy data
//     (branch-free MFMA stream; the spare accumulators are never stored);
//   * the next convolution's 72 KB of weights are prefetched from L2 into registers while the
//     current one runs on the matrix cores, and dropped into LDS between the two barriers
//     that separate convolutions (weights are pre-swizzled on the host into MFMA fragment
//     order, so both the global load and the LDS read are flat 16 B-per-lane streams).
// LDS: 45,056 B activations + 73,728 B weights = 118,784 B (head scratch reuses it).
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

namespace {

using namespace azmi_net_dev;

// LDS activation layout: 8 planes (one per 8-channel chunk), each [pixel slot][8 bf16 = 16 B].
// Pixel slots 0..NPIX-1 are the tile's real pixels in (board, h, w) order; slots NPIX..NPIX+15 are
// all-zero cells: an out-of-board tap is redirected to the zero cell with the SAME slot residue
// mod 16 it would have had, so the redirect never collides with another lane's bank (no halo ring).  With a plane stride that is a
// multiple of 256 B, the 16 lanes of every ds_read_b128 lane-group land on 16 distinct 16-byte
// bank slots (consecutive pixels), so fragment reads are conflict-free.
template <int H, int W, int TBG = TB>
struct Geo {
  static constexpr int PIX = H * W;                 // 42
  static constexpr int NPIX = TBG * PIX;            // 336 GEMM columns
  static constexpr int NT = (NPIX + 15) / 16;       // 21 n-tiles
  static constexpr int NT_W = (NT + NWAVES - 1) / NWAVES;  // n-tiles per wave (3)
  static constexpr int ZSLOT = NT * 16;             // first zero cell: a multiple of 16, so residues line up
  static constexpr int SLOTS = ZSLOT + 16;          // 352
  static constexpr int PLANE = SLOTS * 16;          // 5632 B, multiple of 256
  static constexpr int ZERO_OFF = ZSLOT * 16;       // 16 zero cells (slots ZSLOT..ZSLOT+15) inside every plane
  static constexpr int ACT_BYTES = 8 * PLANE;       // 45,056
  static constexpr int KS3 = 9 * CH / 32;           // 18 k-steps of a 3x3 conv
  static constexpr int WCONV_BYTES = KS3 * MT * WFRAG_BYTES;  // 73,728
  static constexpr int WREG = WCONV_BYTES / (NTHREADS * 16);  // 9 x 16 B of weights per thread
};



// =====================================================================================================
// Spatial-policy-head variant (Tafl family, configs/tawlbwrdd.yaml: 4b64c k3, head_channels 64,
// v_head_convs 1, pi_head_convs 1, v_fc_layers 2, spatial policy; neural_net.py:341-427, 448-494):
//     heads  hv = relu(conv1x1_v(s) + b)          hp = relu(conv1x1_pi(s) + b)         (v_bn / pi_bn folded)
//            v2 = relu(conv3x3(hv) + b)           p2 = relu(conv3x3(hp) + b)           (extra head convs, BN folded)
//            pooled = avgpool(v2)  -> k_value_fc  logits[h][w][c] = conv1x1(p2) + b    (pi_conv2 * pi_bn2 folded)
//            pi = softmax over all H*W*C logits of a board (index (h*W + w)*C + c = the game's move index)
// Same implicit-GEMM tower as k_leafnet with TBS boards per workgroup: 3 for 11x11 (3 * 121 = 363 pixels = 23 n-tiles,
// 3 per wave), 7 for 7x7 (343 pixels = 22 n-tiles); the value head's FC stack runs batched in k_value_fc on the exact-fp32
// matrix pipe.  Nets with fewer than 64 trunk / head channels (configs/brandubh.yaml: 32) are zero-padded to 64 by the
// host-side fold: the padded channels stay exactly 0 through every affine, ReLU and convolution.
constexpr int TBS11 = 3, TBS7 = 7, TBS13 = 2;   // 13x13 (StarGambit's canvas): 2 x 169 = 338 pixels = 22 n-tiles, 3 per wave
constexpr int HCS = 64;

struct SpatialDesc {
  int C_in, H, W, depth, num_moves, num_players, v_hidden, v_fc_layers, pol_ch;
  int num_global, pi_hidden;   // global actions behind the spatial block (StarGambit: 19) and the width of pi_global's hidden layer
};
struct SpatialPtrs {
  const uint8_t* stem_w; const float* stem_b; const uint8_t* blocks;
  const uint8_t* head_w;   // frag[2 ks][8 mt]: rows 0-63 v_conv*v_bn, rows 64-127 pi_conv*pi_bn
  const float* head_b;     // [128]
  const uint8_t* vx_w; const float* vx_b;   // value-head extra conv frag[18][4] + bias[64]
  const uint8_t* px_w; const float* px_b;   // policy-head extra conv
  const uint8_t* pol_w;    // frag[2 ks][2 mt]: rows 0..pol_ch-1 = pi_conv2*pi_bn2, zero padded to 32
  const float* pol_b;      // [32]
  const float* fc1_w;      // f32 A-fragments [v_hidden/16 tiles][64/16 groups][64 lanes][4]
  const float* fc1_b;
  const float* fcx_w;      // (v_fc_layers-1) x fragments [v_hidden/16][v_hidden/16][64][4]
  const float* fcx_b;      // (v_fc_layers-1) x [v_hidden]
  const float* fc2_w;      // fragments [1][v_hidden/16][64][4], rows >= P+1 zero
  const float* fc2_b;      // [16]
  // pi_global (neural_net.py:421-426), fp32, transposed so that consecutive lanes read consecutive outputs:
  const float* pg1_wT;     // [64][pi_hidden]
  const float* pg1_b;      // [pi_hidden]
  const float* pg2_wT;     // [pi_hidden][32] (columns >= num_global zero)
  const float* pg2_b;      // [32]
  const float* pg_ln_g;    // [32] LayerNorm weight
  const float* pg_ln_b;    // [32] LayerNorm bias
};

// `live` (may be NULL): device-side number of rows that are real; workgroups past it exit at once (a row-list evaluation
// packs the listed rows at the front of a max_rows batch)
template <int H, int W, int TBS>
__global__ __launch_bounds__(NTHREADS, 2) void k_leafnet_spatial(SpatialDesc nd, SpatialPtrs np, const float* __restrict__ canon,
                                                                  float* __restrict__ vpool_out, float* __restrict__ pi_out,
                                                                  uint32_t batch, const uint32_t* __restrict__ live) {
  using G = Geo<H, W, TBS>;
  if (live) { const uint32_t n = *live; if (blockIdx.x * TBS >= n) return; batch = n < batch ? n : batch; }
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* act = lds;
  uint8_t* wbuf = lds + G::ACT_BYTES;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, quad = lane >> 4;
  const uint32_t board0 = blockIdx.x * TBS;

  int pix_off[G::NT_W];
  uint32_t tap_ok[G::NT_W];
  bool tile_on[G::NT_W];
  uint32_t real_m = 0;       // bit j: this lane's column of tile j is a real pixel (the last tile is partial)
#pragma unroll
  for (int j = 0; j < G::NT_W; ++j) {
    const int t = wave + NWAVES * j;
    const int n = t * 16 + col;
    tile_on[j] = __builtin_amdgcn_readfirstlane(t) < G::NT;
    const bool real = t < G::NT && n < G::NPIX;
    if (real) real_m |= 1u << j;
    const int nn = real ? n : 0;
    const int p = nn % G::PIX, h = p / W, w = p % W;
    pix_off[j] = nn * 16;
    uint32_t m = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
      if (real && hh >= 0 && hh < H && ww >= 0 && ww < W) m |= 1u << tap;
    }
    tap_ok[j] = m;
  }

  for (int i = tid * 16; i < G::ACT_BYTES; i += NTHREADS * 16) *reinterpret_cast<u32x4*>(act + i) = u32x4{0, 0, 0, 0};

  // ---- stem ---------------------------------------------------------------------------------------------
  f32x4 s[G::NT_W][MT];
  const bool stem_as_conv = 9 * nd.C_in > 128;
  if (stem_as_conv) {
    // many input planes (StarGambit: 36): the stem is run as ONE MORE 64-channel convolution - the planes go into the
    // activation layout as bf16 (channels >= C_in stay zero), the weights are conv fragments with zero columns for the
    // padding (alphazero/hip_net.py) - instead of 9*C_in/64 im2col passes of scattered 2-byte LDS stores
    __syncthreads();                         // the zero fill of the planes is complete
    if (tid < G::NPIX) {
      const int n = tid, b = n / G::PIX, p = n % G::PIX;
      const bool on = board0 + b < batch;
      const float* src = canon + static_cast<size_t>(board0 + b) * nd.C_in * G::PIX + p;
      for (int ci = 0; ci < nd.C_in; ++ci)
        *reinterpret_cast<__bf16*>(act + (ci >> 3) * G::PLANE + n * 16 + (ci & 7) * 2) = static_cast<__bf16>(on ? src[ci * G::PIX] : 0.0f);
    }
    for (int i = tid * 16; i < G::WCONV_BYTES; i += NTHREADS * 16)
      *reinterpret_cast<u32x4*>(wbuf + i) = *reinterpret_cast<const u32x4*>(np.stem_w + i);
    {
      f32x4 bias[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bias[mt] = *reinterpret_cast<const f32x4*>(np.stem_b + mt * 16 + quad * 4);
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) s[j][mt] = bias[mt];
    }
    __syncthreads();
  }
  float* raw = reinterpret_cast<float*>(wbuf + 16384);
  const int plane_sz = nd.C_in * G::PIX;
  if (!stem_as_conv)
  for (int i = tid; i < TBS * plane_sz; i += NTHREADS) {
    const uint32_t b = board0 + i / plane_sz;
    raw[i] = b < batch ? canon[static_cast<size_t>(b) * plane_sz + (i % plane_sz)] : 0.0f;
  }
  // the im2col matrix has 9*C_in rows; the eight 8-row planes hold 64 of them, so a stem with more (8 input planes:
  // OpenTafl) runs in passes of 64 rows that accumulate into the same tiles
  // (StarGambit: 36 planes = 6 passes); a pass's 8 KB of weight fragments are staged at the start of the pass, so the
  // input staging behind them (wbuf + 16 KB) has the rest of the weight area whatever the number of passes
  const int npass = (9 * nd.C_in + 63) / 64;

  if (!stem_as_conv) {
    f32x4 bias[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bias[mt] = *reinterpret_cast<const f32x4*>(np.stem_b + mt * 16 + quad * 4);
#pragma unroll
    for (int j = 0; j < G::NT_W; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) s[j][mt] = bias[mt];
    for (int pass = 0; pass < npass; ++pass) {
      __syncthreads();                     // pass 0: staging done; later passes: the previous pass has read its rows
      if (pass > 0) {
        for (int i = tid * 16; i < G::ACT_BYTES; i += NTHREADS * 16) *reinterpret_cast<u32x4*>(act + i) = u32x4{0, 0, 0, 0};
        __syncthreads();
      }
      for (int i = tid * 16; i < 2 * MT * WFRAG_BYTES; i += NTHREADS * 16)
        *reinterpret_cast<u32x4*>(wbuf + i) = *reinterpret_cast<const u32x4*>(np.stem_w + pass * 2 * MT * WFRAG_BYTES + i);
      if (tid < G::NPIX) {     // the 64 im2col rows of this pass: row r = tap * C_in + ci
        const int n = tid, b = n / G::PIX, p = n % G::PIX, h = p / W, w = p % W;
        const float* rb = raw + b * plane_sz;
        const int r_end = (9 * nd.C_in < 64 * pass + 64) ? 9 * nd.C_in : 64 * pass + 64;
        int tap = (64 * pass) / nd.C_in, ci = (64 * pass) % nd.C_in;
        for (int r = 64 * pass; r < r_end; ++r) {
          const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
          const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
          const float val = ok ? rb[ci * G::PIX + hh * W + ww] : 0.0f;
          const int k = r - 64 * pass;
          *reinterpret_cast<__bf16*>(act + (k >> 3) * G::PLANE + n * 16 + (k & 7) * 2) = static_cast<__bf16>(val);
          if (++ci == nd.C_in) { ci = 0; ++tap; }
        }
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = lds_read_frag(wbuf + (ks * MT + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
        for (int j = 0; j < G::NT_W; ++j) {
          const bf16x8 b = lds_read_frag(act + (ks * 4 + quad) * G::PLANE + pix_off[j]);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) s[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, s[j][mt], 0, 0, 0);
        }
      }
    }
  }

  u32x4 wnext[G::WREG];
  auto prefetch = [&](const uint8_t* src) {
#pragma unroll
    for (int i = 0; i < G::WREG; ++i) wnext[i] = *reinterpret_cast<const u32x4*>(src + (i * NTHREADS + tid) * 16);
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < G::WREG; ++i) *reinterpret_cast<u32x4*>(wbuf + (i * NTHREADS + tid) * 16) = wnext[i];
  };
  auto store_tile = [&](int j, int mt, f32x4 val) {   // real lanes only: the partial tile's spare columns own no slot
    if (!((real_m >> j) & 1u)) return;
    bf16x4 o;
    o[0] = static_cast<__bf16>(val[0]); o[1] = static_cast<__bf16>(val[1]);
    o[2] = static_cast<__bf16>(val[2]); o[3] = static_cast<__bf16>(val[3]);
    *reinterpret_cast<bf16x4*>(act + (mt * 2 + (quad >> 1)) * G::PLANE + pix_off[j] + (quad & 1) * 8) = o;
  };
  auto store_relu = [&](f32x4 (&x)[G::NT_W][MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) {
        if (!tile_on[j]) continue;
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaxf(x[j][mt][r], 0.0f);
        store_tile(j, mt, t);
      }
  };
  auto set_bias = [&](f32x4 (&x)[G::NT_W][MT], const float* bias) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 c = *reinterpret_cast<const f32x4*>(bias + mt * 16 + quad * 4);
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) x[j][mt] = c;
    }
  };
  auto conv3x3 = [&](f32x4 (&acc)[G::NT_W][MT]) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int tap_off = ((tap / 3 - 1) * W + (tap % 3 - 1)) * 16;
      int src[G::NT_W];
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) {
        const int shifted = pix_off[j] + tap_off;
        src[j] = ((tap_ok[j] >> tap) & 1u) ? shifted : G::ZERO_OFF + (shifted & 0xF0);
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int ks = tap * 2 + half;
        bf16x8 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = lds_read_frag(wbuf + (ks * MT + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
        for (int j = 0; j < G::NT_W; ++j) {
          const bf16x8 b = lds_read_frag(act + (half * 4 + quad) * G::PLANE + src[j]);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, acc[j][mt], 0, 0, 0);
        }
      }
    }
  };

  if (stem_as_conv) conv3x3(s);
  const size_t block_stride = 3 * CH * sizeof(float) + 2 * static_cast<size_t>(G::WCONV_BYTES);
  prefetch(np.blocks + 3 * CH * sizeof(float));
  __syncthreads();

  for (int blk = 0; blk < nd.depth; ++blk) {
    const uint8_t* bp = np.blocks + blk * block_stride;
    const float* affine = reinterpret_cast<const float*>(bp);
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
    prefetch(bp + 3 * CH * sizeof(float) + G::WCONV_BYTES);
    f32x4 u[G::NT_W][MT];
    set_bias(u, affine + 2 * CH);
    conv3x3(u);
    __syncthreads();
    store_relu(u);
    commit();
    __syncthreads();
    if (blk + 1 < nd.depth) prefetch(bp + block_stride + 3 * CH * sizeof(float));
    else prefetch(np.vx_w);                       // value-head extra conv rides behind the last trunk conv
    conv3x3(s);
    __syncthreads();
  }

  // ---- head 1x1 convs: 128 rows over the raw stream ---------------------------------------------------------
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < G::NT_W; ++j)
      if (tile_on[j]) store_tile(j, mt, s[j][mt]);
  for (int i = tid * 16; i < 2 * 8 * WFRAG_BYTES; i += NTHREADS * 16)
    *reinterpret_cast<u32x4*>(wbuf + i) = *reinterpret_cast<const u32x4*>(np.head_w + i);
  __syncthreads();
  f32x4 hv[G::NT_W][MT], hp[G::NT_W][MT];
  set_bias(hv, np.head_b);
  set_bias(hp, np.head_b + HCS);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
      bf16x8 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = lds_read_frag(wbuf + (ks * 8 + hsel * 4 + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) {
        const bf16x8 b = lds_read_frag(act + (ks * 4 + quad) * G::PLANE + pix_off[j]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (hsel == 0) hv[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, hv[j][mt], 0, 0, 0);
          else hp[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, hp[j][mt], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  // ---- value head: extra conv, average pool -------------------------------------------------------------------
  store_relu(hv);
  commit();                        // vx weights
  __syncthreads();
  prefetch(np.px_w);
  set_bias(hv, np.vx_b);
  conv3x3(hv);
  __syncthreads();
  {
    float* pool_buf = reinterpret_cast<float*>(act);          // [NPIX][32] fp32, one half of the channels at a time
    float* psum = reinterpret_cast<float*>(wbuf);             // [TBS][64][4] partial sums (vx weights are dead)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) {
        if (!((real_m >> j) & 1u)) continue;
        const int n = (wave + NWAVES * j) * 16 + col;
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaxf(hv[j][half * 2 + m2][r], 0.0f);
          *reinterpret_cast<f32x4*>(pool_buf + n * 32 + m2 * 16 + quad * 4) = o;
        }
      }
      __syncthreads();
      for (int t = tid; t < TBS * 128; t += NTHREADS) {
        const int b = t / 128, r = t % 128, c = r % 32, part = r / 32;
        constexpr int per = (G::PIX + 3) / 4;
        const int p0 = part * per, p1 = (p0 + per < G::PIX) ? p0 + per : G::PIX;
        float acc = 0.0f;
        for (int p = p0; p < p1; ++p) acc += pool_buf[(b * G::PIX + p) * 32 + c];
        psum[(b * 64 + half * 32 + c) * 4 + part] = acc;
      }
      __syncthreads();
    }
    static_assert(TBS * 64 <= NTHREADS && TBS <= NWAVES, "one thread per (board, channel); one wave per board's softmax");
    if (tid < TBS * 64) {
      const int b = tid / 64, c = tid % 64;
      const float* q = psum + (b * 64 + c) * 4;
      if (board0 + b < batch) vpool_out[static_cast<size_t>(board0 + b) * 64 + c] = (((q[0] + q[1]) + q[2]) + q[3]) / static_cast<float>(G::PIX);
    }
  }
  __syncthreads();
  // ---- policy head: extra conv, 1x1 to the policy channels, softmax over the board -----------------------------
  if (tid < 8 * 16) {              // the pooling scratch ran over the zero cells of the activation planes
    const int pl = tid / 16, c = tid % 16;
    *reinterpret_cast<u32x4*>(act + pl * G::PLANE + G::ZERO_OFF + c * 16) = u32x4{0, 0, 0, 0};
  }
  store_relu(hp);
  commit();                        // px weights
  __syncthreads();
  set_bias(hp, np.px_b);
  conv3x3(hp);
  __syncthreads();
  store_relu(hp);
  for (int i = tid * 16; i < 2 * 2 * WFRAG_BYTES; i += NTHREADS * 16)
    *reinterpret_cast<u32x4*>(wbuf + i) = *reinterpret_cast<const u32x4*>(np.pol_w + i);
  __syncthreads();
  float* lg = reinterpret_cast<float*>(wbuf + 8192);           // [NPIX][pol_ch] logits, = [board][move]
  {
    f32x4 pl[G::NT_W][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const f32x4 c = *reinterpret_cast<const f32x4*>(np.pol_b + mt * 16 + quad * 4);
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) pl[j][mt] = c;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) a[mt] = lds_read_frag(wbuf + (ks * 2 + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) {
        const bf16x8 b = lds_read_frag(act + (ks * 4 + quad) * G::PLANE + pix_off[j]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) pl[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, pl[j][mt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < G::NT_W; ++j) {
      if (!((real_m >> j) & 1u)) continue;
      const int n = (wave + NWAVES * j) * 16 + col;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = mt * 16 + quad * 4 + r;
          if (c < nd.pol_ch) lg[(n / G::PIX) * nd.num_moves + (n % G::PIX) * nd.pol_ch + c] = pl[j][mt][r];
        }
    }
  }
  __syncthreads();
  if (nd.num_global > 0) {
    // ---- global actions (StarGambit: 18 deploys + end turn), neural_net.py:413-426, 486-493: the average-pooled policy
    // features -> Linear(64, pi_hidden) -> ReLU -> Linear(pi_hidden, G) -> LayerNorm(G), appended behind the spatial logits.
    // fp32 on the vector units: 64 * pi_hidden + pi_hidden * G multiply-adds per board, one wave per board.
    float* pool_buf = reinterpret_cast<float*>(act);                            // the policy 1x1 has read the planes
    float* gscr = reinterpret_cast<float*>(wbuf + 8192 + 16384);                // behind the logits: psum | pooled | hidden
    float* psum = gscr;                                                         // [TBS][64][4]
    float* pooled = gscr + TBS * 256;                                           // [TBS][64]
    float* hidden = pooled + TBS * 64;                                          // [TBS][pi_hidden]
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int j = 0; j < G::NT_W; ++j) {
        if (!((real_m >> j) & 1u)) continue;
        const int n = (wave + NWAVES * j) * 16 + col;
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaxf(hp[j][half * 2 + m2][r], 0.0f);
          *reinterpret_cast<f32x4*>(pool_buf + n * 32 + m2 * 16 + quad * 4) = o;
        }
      }
      __syncthreads();
      for (int t = tid; t < TBS * 128; t += NTHREADS) {
        const int b = t / 128, r = t % 128, c = r % 32, part = r / 32;
        constexpr int per = (G::PIX + 3) / 4;
        const int p0 = part * per, p1 = (p0 + per < G::PIX) ? p0 + per : G::PIX;
        float acc = 0.0f;
        for (int p = p0; p < p1; ++p) acc += pool_buf[(b * G::PIX + p) * 32 + c];
        psum[(b * 64 + half * 32 + c) * 4 + part] = acc;
      }
      __syncthreads();
    }
    if (tid < TBS * 64) {
      const float* q = psum + tid * 4;
      pooled[tid] = (((q[0] + q[1]) + q[2]) + q[3]) / static_cast<float>(G::PIX);
    }
    __syncthreads();
    if (wave < TBS) {
      const int Hp = nd.pi_hidden;
      const float* x = pooled + wave * 64;
      float* hrow = hidden + wave * Hp;
      for (int o = lane; o < Hp; o += 64) {
        float acc = np.pg1_b[o];
        for (int k = 0; k < 64; ++k) acc += np.pg1_wT[k * Hp + o] * x[k];
        hrow[o] = fmaxf(acc, 0.0f);
      }
    }
    __syncthreads();
    if (wave < TBS) {
      const int Hp = nd.pi_hidden, Gn = nd.num_global;
      const float* hrow = hidden + wave * Hp;
      float acc = 0.0f;
      if (lane < 32) {
        acc = np.pg2_b[lane];
        for (int k = 0; k < Hp; ++k) acc += np.pg2_wT[k * 32 + lane] * hrow[k];
      }
      const bool on = lane < Gn;
      float sum = on ? acc : 0.0f;
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
      const float mean = sum / static_cast<float>(Gn);
      float dv = on ? (acc - mean) * (acc - mean) : 0.0f;
      for (int off = 32; off > 0; off >>= 1) dv += __shfl_xor(dv, off, 64);
      const float inv = 1.0f / sqrtf(dv / static_cast<float>(Gn) + 1e-5f);
      if (on) lg[wave * nd.num_moves + G::PIX * nd.pol_ch + lane] = (acc - mean) * inv * np.pg_ln_g[lane] + np.pg_ln_b[lane];
    }
    __syncthreads();
  }
  if (wave < TBS && board0 + wave < batch) {   // one wave per board: softmax = exp(log_softmax), neural_net.py:494,816
    const int M = nd.num_moves;
    const float* row = lg + wave * M;
    float mx = -__builtin_inff();
    for (int e = lane; e < M; e += 64) mx = fmaxf(mx, row[e]);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.0f;
    for (int e = lane; e < M; e += 64) sum += expf(row[e] - mx);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    float* out = pi_out + static_cast<size_t>(board0 + wave) * M;
    for (int e = lane; e < M; e += 64) out[e] = expf(row[e] - mx) / sum;
  }
}

// Value-head FC stack over the whole batch on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32):
// x0 = pooled [b][64]; x = relu(W x + b) for fc1 and the extra layers; v = softmax(W2 x + b2).
// One workgroup = 16 boards; activations live in LDS as [k][16 boards]; weights stream from L2 transposed
// ([k][out]), 16 outputs x 4 k per MFMA.
constexpr int VFC_THREADS = 1024;
constexpr int VFC_WAVES = VFC_THREADS / 64;
__global__ __launch_bounds__(VFC_THREADS) void k_value_fc(SpatialDesc nd, SpatialPtrs np, const float* __restrict__ vpool,
                                                         float* __restrict__ v_out, uint32_t batch, const uint32_t* __restrict__ live) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  if (live) { const uint32_t n = *live; if (blockIdx.x * 16 >= n) return; batch = n < batch ? n : batch; }
  float* xa = reinterpret_cast<float*>(lds);                  // [v_hidden][16]
  float* xb = xa + nd.v_hidden * 16;
  float* part = xb + nd.v_hidden * 16;                        // [VFC_WAVES][16 outputs][16 boards]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 15, quad = lane >> 4;
  const uint32_t b0 = blockIdx.x * 16;
  for (int i = tid; i < 64 * 16; i += VFC_THREADS) {
    const int k = i / 16, b = i % 16;
    xa[i] = (b0 + b < batch) ? vpool[static_cast<size_t>(b0 + b) * 64 + k] : 0.0f;
  }
  __syncthreads();
  // hidden layers: a wave owns output tiles {wave, wave + 16, ...} two at a time, so 16 weight loads are in flight
  // per wave (the loop is bound by L2 latency, not by the matrix pipe)
  // Weights are stored on the host in MFMA A-fragment order: frag[out tile][k group of 16][lane][4], element j of lane l
  // = W[16*tile + (l & 15)][16*group + 4*j + (l >> 4)], so one 16-byte load feeds four k-steps.  A wave owns output
  // tiles {wave, wave + 16} together: the loop is bound by L2 latency, not by the matrix pipe.
  auto layer = [&](const float* wt, const float* bias, int K, int N, const float* xin, float* xout) {
    const int ntiles = N / 16, kgroups = K / 16;
    for (int t0 = wave; t0 < ntiles; t0 += 2 * VFC_WAVES) {
      const int t1 = t0 + VFC_WAVES;
      const bool two = t1 < ntiles;
      const f32x4* w0 = reinterpret_cast<const f32x4*>(wt) + static_cast<size_t>(t0) * kgroups * 64 + lane;
      const f32x4* w1 = reinterpret_cast<const f32x4*>(wt) + static_cast<size_t>(two ? t1 : t0) * kgroups * 64 + lane;
      f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
      for (int kg = 0; kg < kgroups; ++kg) {
        const f32x4 a0 = w0[kg * 64], a1 = w1[kg * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float b = xin[(kg * 16 + j * 4 + quad) * 16 + col];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b, acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o0 = t0 * 16 + quad * 4 + r;
        xout[o0 * 16 + col] = fmaxf(acc0[r] + bias[o0], 0.0f);
        if (two) { const int o1 = t1 * 16 + quad * 4 + r; xout[o1 * 16 + col] = fmaxf(acc1[r] + bias[o1], 0.0f); }
      }
    }
    __syncthreads();
  };
  const int Hd = nd.v_hidden;
  layer(np.fc1_w, np.fc1_b, 64, Hd, xa, xb);
  float *cur = xb, *nxt = xa;
  for (int l = 0; l + 1 < nd.v_fc_layers; ++l) {
    layer(np.fcx_w + static_cast<size_t>(l) * Hd * Hd, np.fcx_b + l * Hd, Hd, Hd, cur, nxt);
    float* t = cur; cur = nxt; nxt = t;
  }
  {  // output layer (one tile of 16 padded rows): k groups split over the waves, partial tiles summed in wave order
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const int gper = Hd / 16 / VFC_WAVES;
    const f32x4* w = reinterpret_cast<const f32x4*>(np.fc2_w) + lane;
    for (int kg = wave * gper; kg < (wave + 1) * gper; ++kg) {
      const f32x4 a = w[kg * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], cur[(kg * 16 + j * 4 + quad) * 16 + col], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[(wave * 16 + quad * 4 + r) * 16 + col] = acc[r];
  }
  __syncthreads();
  if (tid < 16 && b0 + tid < batch) {
    const int P1 = nd.num_players + 1;
    float lg[16];
    for (int i = 0; i < P1; ++i) {
      float a = np.fc2_b[i];
      for (int w = 0; w < VFC_WAVES; ++w) a += part[(w * 16 + i) * 16 + tid];
      lg[i] = a;
    }
    float mx = lg[0];
    for (int i = 1; i < P1; ++i) mx = fmaxf(mx, lg[i]);
    float sum = 0.0f;
    for (int i = 0; i < P1; ++i) sum += expf(lg[i] - mx);
    for (int i = 0; i < P1; ++i) v_out[static_cast<size_t>(b0 + tid) * P1 + i] = expf(lg[i] - mx) / sum;
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
  void* f32 = nullptr;       // precision = 1: the fp32 path (leafnet_f32.hip) owns everything
  void* f32_last_stream = nullptr;
  bool f32_last_stream_set = false;
  bool spatial = false;      // spatial policy head (Tafl family): k_leafnet_spatial + k_value_fc
  SpatialDesc sd{};
  SpatialPtrs sp{};
  // Scratch between the kernels of one forward: PER STREAM - engines on different streams share one net object and run
  // their forwards concurrently (round 1 kept one buffer per net: the value heads of concurrent shards read each other's
  // pooled features)
  struct StreamScratch {
    float* vpool = nullptr;    // [vpool_rows][64] pooled value-head features between k_leafnet_spatial and k_value_fc
    uint32_t vpool_rows = 0;
    float *g_canon = nullptr, *g_v = nullptr, *g_pi = nullptr;   // row-list evaluation of the whole-batch kernels (azmi_net_forward_rows)
    uint32_t g_rows = 0;
  };
  std::mutex scratch_mu;
  std::map<void*, StreamScratch> scratch;
  StreamScratch& scratch_of(void* stream) { std::lock_guard<std::mutex> l(scratch_mu); return scratch[stream]; }
  void free_scratch() {
    for (auto& kv : scratch) {
      StreamScratch& sc = kv.second;
      if (sc.vpool) (void)hipFree(sc.vpool);
      if (sc.g_canon) (void)hipFree(sc.g_canon);
      if (sc.g_v) (void)hipFree(sc.g_v);
      if (sc.g_pi) (void)hipFree(sc.g_pi);
    }
    scratch.clear();
  }
  size_t vfc_lds = 0;
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
size_t stem_passes(const azmi_net_desc* d) { return (9 * static_cast<size_t>(d->in_channels) + 63) / 64; }
// stem weights: im2col passes of 64 rows (few input planes) or one 64-channel convolution (9 * C_in > 128)
size_t stem_bytes(const azmi_net_desc* d) {
  return 9 * d->in_channels > 128 ? static_cast<size_t>(18) * MT * WFRAG_BYTES : stem_passes(d) * 2 * MT * WFRAG_BYTES;
}
size_t spatial_blob_bytes(const azmi_net_desc* d) {
  const size_t wconv = 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
  const size_t Hd = d->v_hidden, L = d->v_fc_layers;
  size_t n = stem_bytes(d) + CH * 4;                                  // stem
  n += static_cast<size_t>(d->depth) * (3 * CH * 4 + 2 * wconv);      // trunk
  n += 2 * 8 * WFRAG_BYTES + 128 * 4;                                 // head 1x1 convs
  n += 2 * (wconv + 64 * 4);                                          // extra head convs
  n += 2 * 2 * WFRAG_BYTES + 32 * 4;                                  // policy 1x1
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
  const size_t wconv = 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
  size_t n = wsmall + CH * 4;
  n += static_cast<size_t>(d->depth) * (3 * CH * 4 + 2 * wconv);
  n += wsmall + CH * 4;
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
  if (d->precision != 0) return nfail(AZMI_ERR_INVALID, "precision must be 0 (bf16 MFMA) or 1 (fp32)");
  if (is_spatial(d)) {
    if (d->channels != CH || d->head_channels != HCS || d->kernel_size != 3 || d->v_head_convs != 1 || d->pi_head_convs != 1)
      return nfail(AZMI_ERR_INVALID, "spatial leaf net kernel covers 64 trunk / 64 head channels, 3x3 convs, one extra conv per head");
    const bool b11 = d->height == 11 && d->width == 11, b7 = d->height == 7 && d->width == 7, b13 = d->height == 13 && d->width == 13;
    if (!b11 && !b7 && !b13) return nfail(AZMI_ERR_INVALID, "spatial leaf net kernel: board %dx%d not instantiated", d->height, d->width);
    const int tbs = b11 ? TBS11 : b7 ? TBS7 : TBS13;
    const int num_global = d->num_moves - d->policy_channels * d->height * d->width;
    if (d->policy_channels > 32 || num_global < 0 || num_global > 32)
      return nfail(AZMI_ERR_INVALID, "spatial head: policy channels <= 32, 0..32 global actions");
    if (num_global > 0 && (d->pi_hidden < 64 || d->pi_hidden > 1024 || d->pi_hidden % 64))
      return nfail(AZMI_ERR_INVALID, "spatial head with global actions: pi_hidden must be a multiple of 64 in [64, 1024]");
    // LDS budget of the head scratch (k_leafnet_spatial): input staging behind 16 KB of stem weights; logits behind 8 KB of
    // policy weights; the global head's pooling / hidden scratch behind the logits
    if (d->in_channels > 64 || (9 * d->in_channels <= 128 && 16384 + static_cast<size_t>(tbs) * d->in_channels * d->height * d->width * 4 > 18 * MT * WFRAG_BYTES))
      return nfail(AZMI_ERR_INVALID, "spatial leaf net kernel: %d input planes not supported", d->in_channels);
    if (num_global > 0 && (static_cast<size_t>(tbs) * d->num_moves * 4 > 16384 ||
                           8192 + 16384 + static_cast<size_t>(tbs) * (256 + 64 + d->pi_hidden) * 4 > 18 * MT * WFRAG_BYTES))
      return nfail(AZMI_ERR_INVALID, "spatial head with global actions: logits / hidden scratch do not fit");
    if (d->v_hidden > 512 || d->v_hidden % 256 || d->v_fc_layers < 1 || d->num_players + 1 > 16) return nfail(AZMI_ERR_INVALID, "value head sizes out of range");
    if (blob_bytes != spatial_blob_bytes(d)) return nfail(AZMI_ERR_INVALID, "weight blob is %zu bytes, expected %zu", blob_bytes, spatial_blob_bytes(d));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return nfail(AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
    if (hipSetDevice(device) != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "hipSetDevice failed");
    auto net = new azmi_net();
    net->device = device; net->spatial = true;
    net->sd = SpatialDesc{d->in_channels, d->height, d->width, d->depth, d->num_moves, d->num_players, d->v_hidden, d->v_fc_layers, d->policy_channels,
                          num_global, num_global > 0 ? d->pi_hidden : 0};
    if (hipMalloc(&net->blob, blob_bytes) != hipSuccess) { delete net; return nfail(AZMI_ERR_OOM, "hipMalloc(weights) failed"); }
    if (hipMemcpy(net->blob, blob, blob_bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(net->blob); delete net; return nfail(AZMI_ERR_NO_DEVICE, "weight upload failed"); }
    net->blob_bytes = blob_bytes;
    const uint8_t* p = static_cast<const uint8_t*>(net->blob);
    const size_t wconv = 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
    const size_t Hd = d->v_hidden, L = d->v_fc_layers;
    SpatialPtrs& sp = net->sp;
    auto f32p = [&](size_t count) { const float* q = reinterpret_cast<const float*>(p); p += count * 4; return q; };
    sp.stem_w = p; p += stem_bytes(d); sp.stem_b = f32p(CH);
    sp.blocks = p; p += static_cast<size_t>(d->depth) * (3 * CH * 4 + 2 * wconv);
    sp.head_w = p; p += 2 * 8 * WFRAG_BYTES; sp.head_b = f32p(128);
    sp.vx_w = p; p += wconv; sp.vx_b = f32p(64);
    sp.px_w = p; p += wconv; sp.px_b = f32p(64);
    sp.pol_w = p; p += 2 * 2 * WFRAG_BYTES; sp.pol_b = f32p(32);
    sp.fc1_w = f32p(64 * Hd); sp.fc1_b = f32p(Hd);
    sp.fcx_w = f32p((L - 1) * Hd * Hd); sp.fcx_b = f32p((L - 1) * Hd);
    sp.fc2_w = f32p(Hd * 16); sp.fc2_b = f32p(16);
    if (num_global > 0) {
      const size_t Hp = d->pi_hidden;
      sp.pg1_wT = f32p(64 * Hp); sp.pg1_b = f32p(Hp); sp.pg2_wT = f32p(Hp * 32); sp.pg2_b = f32p(32); sp.pg_ln_g = f32p(32); sp.pg_ln_b = f32p(32);
    }
    auto reserve = [&](auto geo, const void* kernel) {
      using GS = decltype(geo);
      static_assert(GS::NPIX <= NTHREADS, "stem im2col maps one thread to one pixel");
      static_assert(GS::NPIX * 32 * 4 <= GS::ACT_BYTES, "pooling scratch must fit the activation planes");
      static_assert(8192 + GS::NPIX * 32 * 4 <= GS::WCONV_BYTES, "policy logits must fit the weight area");
      net->lds_bytes = GS::ACT_BYTES + GS::WCONV_BYTES;
      return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(net->lds_bytes)) == hipSuccess;
    };
    static_assert(2 * MT * WFRAG_BYTES <= 16384, "one stem pass of weights sits in front of the input staging");
    net->vfc_lds = (2 * Hd * 16 + VFC_WAVES * 256) * sizeof(float);
    const bool reserved = b11 ? reserve(Geo<11, 11, TBS11>{}, reinterpret_cast<const void*>(&k_leafnet_spatial<11, 11, TBS11>))
                        : b7 ? reserve(Geo<7, 7, TBS7>{}, reinterpret_cast<const void*>(&k_leafnet_spatial<7, 7, TBS7>))
                             : reserve(Geo<13, 13, TBS13>{}, reinterpret_cast<const void*>(&k_leafnet_spatial<13, 13, TBS13>));
    if (!reserved || hipFuncSetAttribute(reinterpret_cast<const void*>(&k_value_fc), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(net->vfc_lds)) != hipSuccess) {
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
  np.pi_fc_w = p; p += static_cast<size_t>(d->height) * d->width * 2 * WFRAG_BYTES;
  np.pi_fc_b = reinterpret_cast<const float*>(p); p += static_cast<size_t>(d->num_moves) * 4;
  net->lds_bytes = c4::LDS_BYTES;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&c4::k_leafnet_c4<4, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(net->lds_bytes)) != hipSuccess) {
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
  if (net->spatial) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    azmi_net::StreamScratch& sc = net->scratch_of(stream);
    if (batch > sc.vpool_rows) {   // first use of this stream / a larger batch only (synchronous, outside the steady state and outside stream capture)
      (void)hipSetDevice(net->device);
      if (sc.vpool) { (void)hipDeviceSynchronize(); (void)hipFree(sc.vpool); sc.vpool = nullptr; sc.vpool_rows = 0; }
      const uint32_t want = batch < 4096u ? 4096u : batch;
      if (hipMalloc(reinterpret_cast<void**>(&sc.vpool), static_cast<size_t>(want) * 64 * sizeof(float)) != hipSuccess)
        return nfail(AZMI_ERR_OOM, "hipMalloc(value-head scratch) failed");
      sc.vpool_rows = want;
    }
    if (net->sd.H == 11)
      k_leafnet_spatial<11, 11, TBS11><<<(batch + TBS11 - 1) / TBS11, NTHREADS, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, sc.vpool, dev_pi, batch, live);
    else if (net->sd.H == 13)
      k_leafnet_spatial<13, 13, TBS13><<<(batch + TBS13 - 1) / TBS13, NTHREADS, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, sc.vpool, dev_pi, batch, live);
    else
      k_leafnet_spatial<7, 7, TBS7><<<(batch + TBS7 - 1) / TBS7, NTHREADS, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, sc.vpool, dev_pi, batch, live);
    k_value_fc<<<(batch + 15) / 16, VFC_THREADS, net->vfc_lds, st>>>(net->sd, net->sp, sc.vpool, dev_v, batch, live);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "k_leafnet_spatial launch: %s", hipGetErrorString(e));
    return AZMI_OK;
  }
  const uint32_t tiles = (batch + c4::TBW - 1) / c4::TBW;
  c4::k_leafnet_c4<4, 4, 16><<<tiles, c4::NTH, net->lds_bytes, static_cast<hipStream_t>(stream)>>>(net->nd, net->np, dev_canonical, dev_v, dev_pi, batch, nullptr, nullptr);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "k_leafnet launch: %s", hipGetErrorString(e));
  return AZMI_OK;
}

int azmi_net_forward_rows(azmi_net* net, const float* dev_canonical, float* dev_v, float* dev_pi, const uint32_t* dev_rows,
                          const uint32_t* dev_row_count, uint32_t max_rows, void* stream) {
  if (!net || !dev_canonical || !dev_v || !dev_pi || !dev_rows || !dev_row_count) return nfail(AZMI_ERR_INVALID, "null argument");
  if (max_rows == 0) return AZMI_OK;
  if (net->f32 || net->spatial) {
    // these kernels take a dense batch: gather the listed rows, evaluate, scatter the answers back.  Rows that are not
    // listed are NOT touched (another model group's answers, cache hits already delivered: ADVICE r1, play_manager.cc:577-597)
    uint32_t chw, p1, m;
    if (net->f32) azmi_f32::dims(net->f32, &chw, &p1, &m);
    else { chw = net->sd.C_in * net->sd.H * net->sd.W; p1 = net->sd.num_players + 1; m = net->sd.num_moves; }
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
    const int rc = net_forward_live(net, sc.g_canon, sc.g_v, sc.g_pi, max_rows, stream, net->f32 ? nullptr : dev_row_count);
    if (rc != AZMI_OK) return rc;
    k_scatter_rows<<<max_rows, 64, 0, st>>>(sc.g_v, sc.g_pi, dev_rows, dev_row_count, p1, m, dev_v, dev_pi);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "row-list gather/scatter launch: %s", hipGetErrorString(e));
    return AZMI_OK;
  }
  const uint32_t tiles = (max_rows + c4::TBW - 1) / c4::TBW;
  c4::k_leafnet_c4<4, 4, 16><<<tiles, c4::NTH, net->lds_bytes, static_cast<hipStream_t>(stream)>>>(net->nd, net->np, dev_canonical, dev_v, dev_pi, max_rows, dev_rows, dev_row_count);
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
  out->nd = net->nd; out->np = net->np; out->lds_bytes = net->lds_bytes;
  return 1;
}

}  // extern "C"
