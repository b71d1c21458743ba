// Device code of the spatial-policy-head leaf nets (Tafl family 7x7 / 11x11, StarGambit 13x13): sp::tile / k_leafnet_sp.
// Architecture restated (reference neural_net.py:341-427, 448-494; configs/tawlbwrdd.yaml, open_tafl.yaml, brandubh.yaml,
// star_gambit_unified.yaml: 4b64c k3, head_channels 64, one extra conv per head, v_fc_layers >= 1, spatial policy):
//     s  = conv3x3(x) * bn1 + b                                     (stem; BatchNorms folded on the host)
//     per block:  t = relu(a1 s + b1);  u = relu(conv1(t) + c1);  s = s + conv2(u)
//     heads  hv = relu(conv1x1_v(s) + b)          hp = relu(conv1x1_pi(s) + b)
//            v2 = relu(conv3x3(hv) + b)           p2 = relu(conv3x3(hp) + b)
//            pooled = avgpool(v2) -> FC stack -> softmax = v      logits[h][w][c] = conv1x1(p2) + b
//            [StarGambit: avgpool(p2) -> Linear -> ReLU -> Linear -> LayerNorm = the global actions behind the spatial logits]
//            pi = softmax over all logits of a board (index (h*W + w)*C + c = the game's move index)
//
// Same machine as the Connect4 tile (leafnet_c4.h): ONE workgroup of 4 waves (one per SIMD) carries a tile of boards through
// the whole net; 80 KB of LDS and <= 256 VGPRs, so two workgroups share a CU and one's matrix phases cover the other's
// barriers, epilogues and heads.  Every weight fragment of the net - stem, tower, head 1x1s, the heads' extra convs, the
// policy 1x1 - is ONE stream of 8 KB chunks (one 3x3 tap, or one 1x1 convolution) that the waves DMA from L2 straight into a
// 5-slot LDS ring three chunks ahead of the matrix cores (global_load_lds_dwordx4, counted vmcnt, one raw s_barrier per chunk);
// the k-loop is software-pipelined by hand.  The predecessor (8 waves, a whole convolution's 72 KB of weights staged through
// registers into LDS, 118 KB of LDS = one workgroup per CU, the value FC stack as a second launch) reached 24 % of the MFMA
// peak alone and owned its CU: DESIGN.md §4.6.
// Tiles: 11x11 -> 2 boards (242 pixels = 16 n-tiles, 4 per wave), 7x7 -> 5 boards (245 pixels), 13x13 -> 1 board (169 pixels =
// 11 of 12 n-tiles, 3 per wave).  Every reduction runs in an order that does not depend on where a board sits in its tile or
// batch, so a position's (v, pi) is bit-identical wherever the engine's unordered eval list places it.
#pragma once
#include <type_traits>

#include "leafnet_c4.h"

namespace azmi_net_dev {
namespace sp {

using c4::barrier_lds;
using c4::wait_vm;
// c4::dma16x2 with the destination as an LDS byte address: the pointer form casts flat -> LDS per call, and where the optimizer
// moves that cast behind the readfirstlane of a wave-dependent address the backend emits an illegal compare against
// src_shared_base ("Operand has incorrect register class"; seen when the tile grew a tail) - here the ring's LDS address is
// taken once (tile) and the slots are integer arithmetic on it
__device__ __forceinline__ void dma16x2_at(const uint8_t* src_lane, uint32_t dst_lds) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(dst_lds);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "global_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src_lane), "s"(dst) : "memory");
}

constexpr int NTH = 256, NWV = 4;
constexpr int CHUNK_KS = 2;
constexpr int CHUNK_BYTES = CHUNK_KS * MT * WFRAG_BYTES;   // 8,192
constexpr int PIECES = CHUNK_BYTES / (NWV * WFRAG_BYTES);  // 2
constexpr int NRING = 5;
constexpr int RING_BYTES = NRING * CHUNK_BYTES;            // 40,960
constexpr int MAXDEPTH = 6;
constexpr int HCS = 64;                                    // head channels (value / policy each)
// fp32 parameters, in LDS for the whole kernel (a global load inside the weight stream would join its vmcnt queue)
constexpr int PRM_STEM = 0, PRM_BLOCKS = CH, PRM_HEAD = PRM_BLOCKS + MAXDEPTH * 3 * CH, PRM_VX = PRM_HEAD + 2 * HCS,
              PRM_PX = PRM_VX + HCS, PRM_POL = PRM_PX + HCS, PRM_FLOATS = PRM_POL + 32;   // 1,504

// SPLIT (the "bf16x3" precision tier, as c4::Tile's): weights and activations as bf16 HIGH + LOW parts (x = hi + lo to ~16 bits of
// mantissa), every product as three MFMAs hi*hi + hi*lo + lo*hi - each chunk of the weight stream three times ([W_hi][W_hi][W_lo])
// against the activation planes [X_hi][X_lo][X_hi] (planes 8-15 hold the low parts): 16 planes, so one workgroup per CU.
template <int H_, int W_, int TBW_, int NTW_, int SPLIT_ = 0>
struct Geo {
  static constexpr int H = H_, W = W_, TBW = TBW_, NTW = NTW_, SPLIT = SPLIT_;
  static constexpr int NPLANES = SPLIT_ ? 16 : 8;
  static constexpr int PIX = H * W, NPIX = TBW * PIX;
  static constexpr int NT = NWV * NTW;                // n-tiles of the workgroup
  static constexpr int ZSLOT = NT * 16;               // first all-zero cell
  static constexpr int SLOTS = ZSLOT + 16;
  static constexpr int PLANE = SLOTS * 16;            // a multiple of 256 B
  static constexpr int ZERO_OFF = ZSLOT * 16;
  static constexpr int ACT_BYTES = NPLANES * PLANE;
  static constexpr int LDS_BYTES = ACT_BYTES + RING_BYTES + PRM_FLOATS * 4;
  static constexpr int PITER = (TBW * 128 + NTH - 1) / NTH;   // pooling passes: one thread per (board, channel of a half, quarter)
  static_assert(NPIX <= NT * 16 && NPIX <= NTH, "one thread per pixel; the tile's pixels fit its n-tiles");
  static_assert(PLANE % 256 == 0, "conflict-free fragment reads need a plane stride that is a multiple of 256 B");
  static_assert((SPLIT_ ? 1 : 2) * LDS_BYTES <= 160 * 1024, "two workgroups per CU (SPLIT: one)");
  static_assert(NT * 16 * 32 * 4 <= ACT_BYTES, "pooling scratch fits the activation planes");
  static_assert(NPIX * 32 * 4 <= RING_BYTES, "policy logits fit the ring");
};
#ifndef AZMI_SP11_TBW
#define AZMI_SP11_TBW 2        // boards per 11 x 11 tile (experiment builds: 1 - half the tile latency, two n-tiles per wave)
#endif
using Geo11 = Geo<11, 11, AZMI_SP11_TBW, 2 * AZMI_SP11_TBW>;
using Geo7 = Geo<7, 7, 5, 4>;
using Geo13 = Geo<13, 13, 1, 3>;
using Geo11X3 = Geo<11, 11, AZMI_SP11_TBW, 2 * AZMI_SP11_TBW, 1>;
using Geo7X3 = Geo<7, 7, 5, 4, 1>;
using Geo13X3 = Geo<13, 13, 1, 3, 1>;

struct SpDesc {
  int C_in, H, W, depth, num_moves, num_players, v_hidden, v_fc_layers, pol_ch;
  int num_global, pi_hidden;   // global actions behind the spatial block (StarGambit: 19) and the width of pi_global's hidden layer
};
struct SpPtrs {
  // the weight stream, 8 KB chunks = frag[2 k-steps][4 m-tiles]: stem (2 or 9 chunks, stem_chunks) |
  // per block conv1 (9) conv2 (9) | value-head 1x1 (1) | policy-head 1x1 (1) | value extra conv (9) | policy extra conv (9) |
  // policy 1x1 (1; rows >= pol_ch zero)
  const uint8_t* stream;
  const float* prm;        // stem_b[64] | per block a1 b1 c1 [3][64] | head_b[128] (value, policy) | vx_b[64] | px_b[64] | pol_b[32]
  const float* fc1_w;      // f32 A-fragments [v_hidden/16 tiles][64/16 groups][64 lanes][4]
  const float* fc1_b;
  const float* fcx_w;      // (v_fc_layers-1) x fragments [v_hidden/16][v_hidden/16][64][4]
  const float* fcx_b;      // (v_fc_layers-1) x [v_hidden]
  const float* fc2_w;      // fragments [1][v_hidden/16][64][4], rows >= P+1 zero
  const float* fc2_b;      // [16]
  // pi_global (neural_net.py:421-426), fp32 A-fragments like the value FC stack:
  const float* pg1_w;      // [pi_hidden/16][64/16][64][4]
  const float* pg1_b;      // [pi_hidden]
  const float* pg2_w;      // [2][pi_hidden/16][64][4] (rows >= num_global zero)
  const float* pg2_b;      // [32]
  const float* pg_ln_g;    // [32] LayerNorm weight
  const float* pg_ln_b;    // [32] LayerNorm bias
};
// stem: up to 8 input planes sit in ONE activation plane, so the 3x3 stem is an implicit GEMM with k = tap * 8 + ci (K = 72,
// padded to 128 = 2 chunks) whose B fragments are that plane read at the tap of each lane group; more planes (StarGambit: 36)
// run as one more 64-channel convolution (9 chunks)
__host__ __device__ inline int stem_chunks(int c_in) { return c_in <= 8 ? 2 : 9; }
__host__ __device__ inline int stream_chunks(int depth, int c_in, int split = 0) { return (split ? 3 : 1) * (stem_chunks(c_in) + 2 * depth * 9 + 2 + 9 + 9 + 1); }

// rows != nullptr: evaluate only the rows listed in rows[0 .. *row_count) (the engine's eval list); the canonical planes are
// read from, and (v, pi) written to, the LISTED row; workgroups past the end of the list leave at once.
template <class G>
__device__ __forceinline__ void tile(const SpDesc& nd, const SpPtrs& np, const float* __restrict__ canon,
                                     float* __restrict__ vpool, float* __restrict__ ppool, float* __restrict__ pi_out,
                                     uint32_t batch, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ row_count,
                                     const uint32_t tile_index, uint8_t* const lds) {
  constexpr int NTW = G::NTW, TBW = G::TBW, PIX = G::PIX, NPIX = G::NPIX, PLANE = G::PLANE, BW = G::W, BH = G::H;
  if (rows) { const uint32_t n = *row_count; batch = n < batch ? n : batch; }
  if (tile_index * TBW >= batch) return;
  uint8_t* const act = lds;
  uint8_t* const ring = lds + G::ACT_BYTES;
  float* const prm = reinterpret_cast<float*>(lds + G::ACT_BYTES + RING_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, quad = lane >> 4;
  const uint32_t board0 = tile_index * TBW;
  const int depth = nd.depth;
  constexpr bool SPLIT = G::SPLIT != 0;
  const int nchunks = stream_chunks(depth, nd.C_in, G::SPLIT);

  // ---- small fp32 parameters -> LDS (plain loads, before any DMA is in flight) ------------------------------------
  {
    const int nfront = CH + depth * 3 * CH;
    for (int i = tid; i < nfront; i += NTH) prm[i] = np.prm[i];
    for (int i = tid; i < PRM_FLOATS - PRM_HEAD; i += NTH) prm[PRM_HEAD + i] = np.prm[nfront + i];
  }
  // this thread's pixel of the tile's input planes (boards past the end of the batch are all zero and never stored)
  const int in_b = tid < NPIX ? tid / PIX : 0, in_p = tid < NPIX ? tid % PIX : 0;
  const bool in_on = tid < NPIX && board0 + in_b < batch;
  const uint32_t in_row = in_on ? (rows ? rows[board0 + in_b] : board0 + in_b) : 0u;
  // ---- zero the activation planes once (covers the zero cells and the channels the input planes do not fill) -------
  for (int i = tid * 16; i < G::ACT_BYTES; i += NTH * 16) *reinterpret_cast<u32x4*>(act + i) = u32x4{0, 0, 0, 0};

  // ---- weight stream: chunk g lives in ring slot g % NRING; every wave moves 2 of a chunk's 8 pieces ---------------
  const uint32_t ring_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((c4::lptr_t)ring));
  const uint8_t* wnext = np.stream + (wave * PIECES) * WFRAG_BYTES + lane * 16;
  int issued = 0;
  auto issue_next = [&](int slot) {
    dma16x2_at(wnext, ring_lds + slot * CHUNK_BYTES + (wave * PIECES) * WFRAG_BYTES);
    // after the run's last chunk the stream keeps re-sending that chunk into slots nobody reads any more: the tail of the
    // run then needs no special cases - one wait count, no branch around the issue (see conv)
    ++issued;
    if (issued < nchunks) wnext += CHUNK_BYTES;
  };
  issue_next(0);
  issue_next(1);
  issue_next(2);

  // ---- per-lane pixel geometry of the wave's n-tiles: tile t = wave * NTW + j --------------------------------------
  const int pix0 = (wave * NTW * 16 + col) * 16;
  uint32_t tap_ok[NTW];  // bit tap: the 3x3 neighbour (tap/3-1, tap%3-1) is on the board; 0 for the unused columns
  uint32_t real_m = 0;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int n = (wave * NTW + j) * 16 + col;
    const bool real = n < NPIX;
    if (real) real_m |= 1u << j;
    const int p = n % PIX, h = p / BW, w = p % BW;
    uint32_t m = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
      if (real && hh >= 0 && hh < BH && ww >= 0 && ww < BW) m |= 1u << tap;
    }
    tap_ok[j] = m;
  }

  // ---- input planes -> activation layout as bf16 (channel ci = plane ci / 8, element ci % 8), one thread per pixel ---
  barrier_lds();                     // the zero fill is complete
  {
    const float* src = canon + static_cast<size_t>(in_row) * nd.C_in * PIX + in_p;
    const int nc8 = (nd.C_in + 7) >> 3;
    for (int c8 = 0; c8 < nc8; ++c8) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (in_on && c8 * 8 + e < nd.C_in) ? src[(c8 * 8 + e) * PIX] : 0.0f;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = static_cast<__bf16>(x[e]);
      if (tid < NPIX) *reinterpret_cast<bf16x8*>(act + c8 * PLANE + tid * 16) = o;
      if constexpr (SPLIT) {          // (an input plane need not be 0 / 1: OpenTafl's turn / max_turns plane) the low parts
        bf16x8 l;
#pragma unroll
        for (int e = 0; e < 8; ++e) l[e] = static_cast<__bf16>(x[e] - static_cast<float>(o[e]));
        if (tid < NPIX) *reinterpret_cast<bf16x8*>(act + (8 + c8) * PLANE + tid * 16) = l;
      }
    }
  }

  // epilogue helper: 4 consecutive channels (mt*16 + quad*4 ..) of the lane's pixel as bf16, into plane `plane0 + quad/2`
  // (round 6: the convolution epilogues store EVERY lane's pixel, the few padding pixels of a tile (n >= NPIX) included - until then
  // each of an epilogue's 16 stores sat in an exec-mask region of its own (s_and_saveexec / branch / s_or per store: the lone tile
  // 62.5 -> 59.3 us, profiles/r6_tile_exp_noguard.txt).  A padding pixel reads all-zero cells at every tap (tap_ok = 0), so its
  // values are biases: finite, written to slots no real pixel's tap ever reads (taps stay on the reader's own board), never pooled,
  // never stored.  The heads' scratch (value rows, policy planes, logits) keeps its test: there a padding pixel's index is out of range.)
  auto store4 = [&](int j, int plane0, f32x4 val) {
    bf16x4 o;
    o[0] = static_cast<__bf16>(val[0]); o[1] = static_cast<__bf16>(val[1]);
    o[2] = static_cast<__bf16>(val[2]); o[3] = static_cast<__bf16>(val[3]);
    *reinterpret_cast<bf16x4*>(act + (plane0 + (quad >> 1)) * PLANE + pix0 + j * 256 + (quad & 1) * 8) = o;
    if constexpr (SPLIT) {        // the low parts: what the bf16 rounding left, in the same cell of plane + 8
      bf16x4 l;
      l[0] = static_cast<__bf16>(val[0] - static_cast<float>(o[0])); l[1] = static_cast<__bf16>(val[1] - static_cast<float>(o[1]));
      l[2] = static_cast<__bf16>(val[2] - static_cast<float>(o[2])); l[3] = static_cast<__bf16>(val[3] - static_cast<float>(o[3]));
      *reinterpret_cast<bf16x4*>(act + (8 + plane0 + (quad >> 1)) * PLANE + pix0 + j * 256 + (quad & 1) * 8) = l;
    }
  };
  auto store_relu = [&](f32x4 (&x)[NTW][MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        // (no `real_m` test here since round 6: see the note at store4)
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaxf(x[j][mt][r], 0.0f);
        store4(j, mt * 2, t);
      }
  };
  auto set_bias = [&](f32x4 (&x)[NTW][MT], const float* bias) {     // bias: 64 floats in LDS
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 c = *reinterpret_cast<const f32x4*>(bias + mt * 16 + quad * 4);
#pragma unroll
      for (int j = 0; j < NTW; ++j) x[j][mt] = c;
    }
  };

  // B-fragment addressing (c4::tile): a tap reads the lane's pixel + the tap offset when that neighbour is on the board, else an
  // all-zero cell with the 16-byte-slot residue of the cell the tap would have read (no bank conflict with the lanes beside it).
  // The choice is made once per tap and tile and serves both k-steps of the tap: 6 VALU per k-step instead of 16.
  const uint8_t* const po = act + quad * PLANE + pix0;     // tile j: + j * 256, an immediate of the read
  const uint8_t* const wlane = ring + lane * 16;

  // One convolution over `act`, accumulating into acc[][]: NCH weight chunks starting in ring slot `slot0` - 9 (a 3x3: chunk =
  // tap), 1 (a 1x1: the centre tap) or 2 (the stem over <= 8 input planes: k = tap * 8 + ci, lane group `quad` of k-step ks reads
  // plane 0 at tap 4 * ks + quad; taps >= 9 read zeros against zero weights).  Software-pipelined over its 2 * NCH k-steps exactly like c4::tile's conv3x3: the A
  // (ring) and B (activation) fragments of k-step i + 1 are read while the MFMAs of k-step i issue; the barrier that opens
  // chunk c + 1 sits BEFORE the second k-step of chunk c: passing it means chunk c + 1 has landed for every wave and every wave
  // is done with the slot of chunk c - 1, which the DMA of chunk c + 4 refills.  The last such barrier of a convolution comes
  // after every LDS read of the convolution has RETURNED (lgkmcnt(0)), so the epilogue may overwrite the planes without
  // another barrier.  Precondition: the activations are visible and the first chunk has landed for all waves.
  // Returns the ring slot of the next convolution's first chunk.
  auto conv = [&](auto nch_tag, f32x4 (&acc)[NTW][MT], int slot0) -> int {
    // SPLIT: every chunk of the convolution three times - stream chunk c = the convolution's chunk c / 3, pass c % 3 =
    // (W_hi, X_hi), (W_hi, X_lo), (W_lo, X_hi)
    constexpr int NCH0 = decltype(nch_tag)::value, NCH = (SPLIT ? 3 : 1) * NCH0, NKS = NCH * CHUNK_KS;
    bf16x8 a[2][MT], b[2][NTW];
    int slot = slot0;
    auto load_a = [&](int ksl, const uint8_t* wsl, bf16x8 (&fa)[MT]) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = lds_read_frag(wsl + (ksl * MT + mt) * WFRAG_BYTES);
    };
    int sel[NTW];                  // per tile j: the plane offset the current tap's reads go through
    auto load_b = [&](int ks_stream, bf16x8 (&fb)[NTW]) {
      const int chunk_s = ks_stream >> 1, pass = SPLIT ? chunk_s % 3 : 0;
      const int ks = SPLIT ? ((chunk_s / 3) << 1) | (ks_stream & 1) : ks_stream;      // the k-step of the convolution itself
      const int lo_planes = (SPLIT && pass == 1) ? 8 * PLANE : 0;                      // pass 1 reads the activations' low parts
      if constexpr (NCH0 == 2) {
        const int tap = 4 * ks + quad, th = tap / 3, tw = tap - 3 * th;
        const int tap_off = ((th - 1) * BW + (tw - 1)) * 16;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const int d = (tap < 9 && ((tap_ok[j] >> tap) & 1u)) ? tap_off : (G::ZERO_OFF - pix0 - j * 256) + ((pix0 + tap_off) & 0xF0);
          fb[j] = lds_read_frag(act + pix0 + d + j * 256 + lo_planes);
        }
      } else {
        const int tap = NCH0 == 9 ? (ks >> 1) : 4, half = ks & 1;
        const int tap_off = ((tap / 3 - 1) * BW + (tap % 3 - 1)) * 16;
        if (half == 0 && pass == 0) {       // a new tap
          int zs = (pix0 + tap_off) & 0xF0;   // slot residue of the cell this tap reads on the board (the same for every tile j)
          asm volatile("" : "+v"(zs));        // keeps the selects here: hoisted out of the block loop, a convolution's bases cost VGPRs (spills)
          const int zc = quad * PLANE + G::ZERO_OFF + zs;          // byte offsets into the planes (32-bit: a pointer here costs two registers)
          const int pot = quad * PLANE + pix0 + tap_off;
#pragma unroll
          for (int j = 0; j < NTW; ++j) sel[j] = ((tap_ok[j] >> tap) & 1u) ? pot : zc - j * 256;
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) fb[j] = lds_read_frag(act + sel[j] + (j * 256 + half * 4 * PLANE + lo_planes));
      }
    };
    load_a(0, wlane + slot * CHUNK_BYTES, a[0]);
    load_b(0, b[0]);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int cur = ks & 1, c = ks / CHUNK_KS, ksl = ks % CHUNK_KS;
      if (ksl == CHUNK_KS - 1) {            // open chunk c + 1 (the next convolution's first chunk when c == NCH - 1)
        wait_vm<4>();                       // chunk g + 1 has landed; g + 2 and g + 3 (2 pieces each) may still be in flight
        if (c == NCH - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue_next(slot == 0 ? NRING - 1 : slot - 1);     // chunk g + 4 (past the end: a harmless repeat)
        slot = slot == NRING - 1 ? 0 : slot + 1;          // ring slot of chunk c + 1
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < NKS) {
        load_a((ks + 1) % CHUNK_KS, wlane + slot * CHUNK_BYTES, a[cur ^ 1]);
        load_b(ks + 1, b[cur ^ 1]);
      }
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[cur][mt], b[cur][j], acc[j][mt], 0, 0, 0);
      if (ks + 1 < NKS) {
        // issue order inside the k-step: MFMAs and one A fragment of the next k-step (x4: its first MFMAs need all four),
        // then MFMAs, the address arithmetic and the read of one B fragment (x NTW)
        constexpr int MF_A = NTW * MT >= 12 ? 2 : 1, MF_B = (NTW * MT - 4 * MF_A) / NTW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, MF_A, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, MF_B, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return slot;
  };
  const std::integral_constant<int, 9> k3x3{};
  const std::integral_constant<int, 1> k1x1{};
  const std::integral_constant<int, 2> kstem8{};

  // ---- stem (see stem_chunks) --------------------------------------------------------------------------------------------
  f32x4 s[NTW][MT];
  set_bias(s, prm + PRM_STEM);
  wait_vm<4>();                      // chunk 0 (chunks 1 and 2 may still be in flight)
  barrier_lds();                     // input planes visible, chunk 0 landed for every wave, parameters in place
  issue_next(3);
  int slot = nd.C_in <= 8 ? conv(kstem8, s, 0) : conv(k3x3, s, 0);

  for (int blk = 0; blk < depth; ++blk) {
    const float* affine = prm + PRM_BLOCKS + blk * 3 * CH;        // a1[64] b1[64] c1[64]
    // t = relu(a1 * s + b1) -> act
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(affine + mt * 16 + quad * 4);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(affine + CH + mt * 16 + quad * 4);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        // (no `real_m` test here since round 6: see the note at store4)
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaxf(a1[r] * s[j][mt][r] + b1[r], 0.0f);
        store4(j, mt * 2, t);
      }
    }
    // u = relu(conv1(t) + c1)
    f32x4 u[NTW][MT];
    set_bias(u, affine + 2 * CH);
    barrier_lds();
    slot = conv(k3x3, u, slot);
    store_relu(u);
    barrier_lds();
    // s = s + conv2(u)
    slot = conv(k3x3, s, slot);
  }

  // ---- head 1x1 convs over the raw stream: hv (value rows), hp (policy rows) -------------------------------------------
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
      store4(j, mt * 2, s[j][mt]);
  f32x4 hv[NTW][MT], hp[NTW][MT];
  set_bias(hv, prm + PRM_HEAD);
  set_bias(hp, prm + PRM_HEAD + HCS);
  barrier_lds();
  slot = conv(k1x1, hv, slot);
  slot = conv(k1x1, hp, slot);

  // average pool of relu(x) over every board, channel halves of 32 through the (dead) activation planes as fp32
  // [pixel][32]: thread (board, channel, quarter) sums its quarter of the board's pixels in pixel order, the quarters are
  // added as (q0 + q1) + (q2 + q3): out[half][it] of thread t = tid + it * NTH is feature half*32 + (t >> 2 & 31) of board t >> 7
  auto pool = [&](f32x4 (&x)[NTW][MT], float (&out)[2][G::PITER]) {
    float* pool_buf = reinterpret_cast<float*>(act);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        if (!((real_m >> j) & 1u)) continue;
        const int n = (wave * NTW + j) * 16 + col;
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaxf(x[j][half * 2 + m2][r], 0.0f);
          *reinterpret_cast<f32x4*>(pool_buf + n * 32 + m2 * 16 + quad * 4) = o;
        }
      }
      barrier_lds();
#pragma unroll
      for (int it = 0; it < G::PITER; ++it) {
        const int t = tid + it * NTH;
        const int b = t >> 7, c = (t >> 2) & 31, part = t & 3;
        constexpr int per = (PIX + 3) / 4;
        const int p0 = part * per, p1 = (p0 + per < PIX) ? p0 + per : PIX;
        // (round 6: the quarter's reads are all issued before the first add - as a loop with a run-time trip count it was 31
        // dependent LDS round trips per half, 4 us of a 61 us Tawlbwrdd tile.  The sum is the same sum in the same order: the entries
        // past the quarter's end are + 0.0f onto a sum of relu outputs)
        float acc = 0.0f;
        if (b < TBW) {
          float x[per];
#pragma unroll
          for (int i = 0; i < per; ++i) x[i] = (p0 + i < p1) ? pool_buf[(b * PIX + p0 + i) * 32 + c] : 0.0f;
#pragma unroll
          for (int i = 0; i < per; ++i) acc += x[i];
        }
        const float s01 = acc + __shfl_xor(acc, 1, 64);
        out[half][it] = (s01 + __shfl_xor(s01, 2, 64)) / static_cast<float>(PIX);
      }
      barrier_lds();
    }
  };
  auto rezero_cells = [&]() {          // the pooling scratch ran over the zero cells of the activation planes
    if (tid < G::NPLANES * 16) *reinterpret_cast<u32x4*>(act + (tid >> 4) * PLANE + G::ZERO_OFF + (tid & 15) * 16) = u32x4{0, 0, 0, 0};
  };

  // ---- value head: extra conv, average pool (the FC stack runs last) -----------------------------------------------------
  store_relu(hv);
  set_bias(hv, prm + PRM_VX);
  barrier_lds();
  slot = conv(k3x3, hv, slot);
  float vp[2][G::PITER];
  pool(hv, vp);
  rezero_cells();
  // ---- policy head: extra conv, 1x1 to the policy channels -----------------------------------------------------------------
  store_relu(hp);
  set_bias(hp, prm + PRM_PX);
  barrier_lds();
  slot = conv(k3x3, hp, slot);
  store_relu(hp);
  f32x4 pl[NTW][MT];                   // rows >= pol_ch have zero weights
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
    if (mt < 2) c = *reinterpret_cast<const f32x4*>(prm + PRM_POL + mt * 16 + quad * 4);
#pragma unroll
    for (int j = 0; j < NTW; ++j) pl[j][mt] = c;
  }
  barrier_lds();
  slot = conv(k1x1, pl, slot);
  wait_vm<0>();                        // the stream's trailing repeats have landed: nothing is in flight any more
  barrier_lds();                       // ... for every wave: the ring is free
  float* const lg = reinterpret_cast<float*>(ring);            // [TBW][num_moves] logits
  const int M = nd.num_moves;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    if (!((real_m >> j) & 1u)) continue;
    const int n = (wave * NTW + j) * 16 + col;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = mt * 16 + quad * 4 + r;
        if (c < nd.pol_ch) lg[(n / PIX) * M + (n % PIX) * nd.pol_ch + c] = pl[j][mt][r];
      }
  }
  // pooled value-head features of the tile's boards -> vpool[list position][64] (the FC stack runs batched in k_heads_fc_a / _b)
  auto put_pooled = [&](float* __restrict__ dst, const float (&x)[2][G::PITER]) {
#pragma unroll
    for (int it = 0; it < G::PITER; ++it) {
      const int t = tid + it * NTH;
      const int b = t >> 7, c = (t >> 2) & 31;
      if ((t & 3) == 0 && b < TBW && board0 + b < batch) {
        dst[static_cast<size_t>(board0 + b) * 64 + c] = x[0][it];
        dst[static_cast<size_t>(board0 + b) * 64 + 32 + c] = x[1][it];
      }
    }
  };
  put_pooled(vpool, vp);
  if (nd.num_global > 0) {
    // global actions (StarGambit: 18 deploys + end turn), neural_net.py:413-426, 486-493: the average-pooled policy features
    // go to ppool, the raw spatial logits to the pi row; k_heads_fc_a / _b run pi_global and the softmax over the whole row
    float pp[2][G::PITER];
    barrier_lds();                     // the logits are written (the policy 1x1 is done with the planes: its last barrier)
    pool(hp, pp);
    put_pooled(ppool, pp);
    const int S = PIX * nd.pol_ch;
    for (int b = wave; b < TBW; b += NWV) {
      if (board0 + b >= batch) continue;
      const uint32_t out_row = rows ? rows[board0 + b] : board0 + b;
      const float* row = lg + b * M;
      float* out = pi_out + static_cast<size_t>(out_row) * M;
      for (int e = lane; e < S; e += 64) out[e] = row[e];
    }
    return;
  }
  barrier_lds();
  // softmax = exp(log_softmax), neural_net.py:494,816.  Round 6: every exponential is taken ONCE (it goes back into the logits' place in
  // LDS and is scaled in the last pass; rounds 2-5 took it twice) and, where the tile has fewer boards than waves (11 x 11: two boards),
  // a board's row is cut into PB contiguous parts, one wave each - the two 2662-entry rows of a Tawlbwrdd tile kept two of the four
  // waves busy for 8.8 us of a 61 us tile (profiles/r6_sp_tile_timing.txt).  The maximum and the sum of a row are combined from the
  // parts in part order, each part's sum in the order of rounds 2-5 (lane-strided, then the butterfly): a function of the row alone,
  // wherever the board sits in its tile or batch.
  constexpr int PB = (TBW < NWV && NWV % TBW == 0) ? NWV / TBW : 1;     // waves per board
  if constexpr (PB > 1) {
    float* const red = prm;                                   // [TBW][PB] maxima, then sums (the parameters are dead by now)
    const int b = wave / PB, part = wave % PB;
    const bool on = board0 + b < batch;
    const int per = (M + PB - 1) / PB, e0 = part * per, e1 = (e0 + per < M) ? e0 + per : M;
    float* const row = lg + b * M;
    float mx = -__builtin_inff();
    for (int e = e0 + lane; e < e1; e += 64) mx = fmaxf(mx, row[e]);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) red[wave] = mx;
    barrier_lds();
    mx = red[b * PB];
#pragma unroll
    for (int q = 1; q < PB; ++q) mx = fmaxf(mx, red[b * PB + q]);
    float sum = 0.0f;
    for (int e = e0 + lane; e < e1; e += 64) { const float x = expf(row[e] - mx); row[e] = x; sum += x; }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    barrier_lds();                                            // every wave has read the maxima
    if (lane == 0) red[wave] = sum;
    barrier_lds();
    sum = red[b * PB];
#pragma unroll
    for (int q = 1; q < PB; ++q) sum += red[b * PB + q];
    if (on) {
      const uint32_t out_row = rows ? rows[board0 + b] : board0 + b;
      float* out = pi_out + static_cast<size_t>(out_row) * M;
      for (int e = e0 + lane; e < e1; e += 64) out[e] = row[e] / sum;
    }
  } else {
    for (int b = wave; b < TBW; b += NWV) {        // one wave per board
      if (board0 + b >= batch) continue;
      const uint32_t out_row = rows ? rows[board0 + b] : board0 + b;
      float* const row = lg + b * M;
      float mx = -__builtin_inff();
      for (int e = lane; e < M; e += 64) mx = fmaxf(mx, row[e]);
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      float sum = 0.0f;
      for (int e = lane; e < M; e += 64) { const float x = expf(row[e] - mx); row[e] = x; sum += x; }
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
      float* out = pi_out + static_cast<size_t>(out_row) * M;
      for (int e = lane; e < M; e += 64) out[e] = row[e] / sum;
    }
  }
}

template <class G>
__global__ __launch_bounds__(NTH, G::SPLIT ? 1 : 2) void k_leafnet_sp(SpDesc nd, SpPtrs np, const float* __restrict__ canon,
                                                        float* __restrict__ vpool, float* __restrict__ ppool, float* __restrict__ pi_out,
                                                        uint32_t batch, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ row_count) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_sp[];
  tile<G>(nd, np, canon, vpool, ppool, pi_out, batch, rows, row_count, blockIdx.x, lds_sp);
}

// The heads' fully connected parts, batched over groups of 16 list entries (the boards are the 16 columns of the exact-fp32
// v_mfma_f32_16x16x4_f32), because their weights (0.3 - 1.5 MB fp32) would otherwise be re-read from L2 by every tile of 1 - 5
// boards.  A workgroup streams its weights through the fabric at ~60 GB/s (`profiles/r2_pmc_traffic_*`: no L2 sharing between
// the workgroups of a launch), so the LATENCY of this step is bytes per workgroup - and the step is split over workgroups:
//   k_heads_fc_a, workgroup (group, part):
//     part < NS = v_hidden / 128: value head (neural_net.py:448-470) x0 = vpool [64]; x = relu(W x + b) for fc1 and the extra
//       layers - all of them but the LAST hidden layer in full (small), of the last one only output units [128 part, 128 part
//       + 128) -> hidden[group][unit][16] in HBM;
//     part == NS (StarGambit): pi_global (neural_net.py:413-426, 486-494) ppool [64] -> Linear(64, pi_hidden) -> ReLU ->
//       Linear(pi_hidden, G) -> LayerNorm(G) = the logits of the global actions -> glob[group][board][32];
//   k_heads_fc_b, workgroup = group: v = softmax(W2 hidden + b2); with global actions the softmax over the whole pi row
//     ([0, S) = the tile's raw spatial logits, then the global logits).
// Activations live in LDS as [k][16]; weights stream in A-fragment order: frag[out tile][k group of 16][lane][4], element j of
// lane l = W[16*tile + (l & 15)][16*group + 4*j + (l >> 4)], so one 16-byte load feeds four k-steps; up to 16 loads in flight
// per lane and <= 128 VGPRs (two of these waves fit a SIMD beside one tile workgroup).
constexpr int HFC_THREADS = 512, HFC_WAVES = HFC_THREADS / 64;
constexpr int HFC_SLICE = 128;     // units of the last hidden layer per workgroup of k_heads_fc_a (8 tiles: one per wave)
__host__ __device__ inline size_t heads_fc_lds(int hidden) { return (2 * static_cast<size_t>(hidden) * 16 + HFC_WAVES * 256 + 16 * 32) * sizeof(float); }
// what the two halves of the split form really use (LDS is what the kernels of a wide-game round compete for: two tile workgroups
// fill a CU's 160 KB, and so do six tree wavefronts): k_heads_fc_a with two value layers keeps the 64 pooled inputs in its
// first buffer and one hidden layer in its second; k_heads_fc_b holds one hidden layer.
__host__ __device__ inline int heads_fc_a_first(int hidden, int layers) { return layers > 2 ? hidden : 64; }
__host__ __device__ inline size_t heads_fc_a_lds(int hidden, int layers) {
  return ((static_cast<size_t>(heads_fc_a_first(hidden, layers)) + hidden) * 16 + HFC_WAVES * 256 + 16 * 32) * sizeof(float);
}
__host__ __device__ inline size_t heads_fc_b_lds(int hidden) { return (static_cast<size_t>(hidden) * 16 + HFC_WAVES * 256 + 16 * 32) * sizeof(float); }

// x_out[unit][16] = act(W x_in + b) for output tiles [t_begin, t_begin + t_count) of a layer with K inputs; x_out is indexed
// by the layer's unit number (LDS, or the group's hidden block in HBM).  A wave owns tiles {t, t + HFC_WAVES} together.
__device__ __forceinline__ void fc_tiles(const float* __restrict__ wt, const float* __restrict__ bias, int K, int t_begin, int t_count,
                                         const float* xin, float* xout, bool relu) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 15, quad = lane >> 4;
  const int kgroups = K / 16;
  for (int i = wave; i < t_count; i += 2 * HFC_WAVES) {
    const int t0 = t_begin + i, t1 = t0 + HFC_WAVES;
    const bool two = i + HFC_WAVES < t_count;
    const f32x4* w0 = reinterpret_cast<const f32x4*>(wt) + static_cast<size_t>(t0) * kgroups * 64 + lane;
    const f32x4* w1 = reinterpret_cast<const f32x4*>(wt) + static_cast<size_t>(two ? t1 : t0) * kgroups * 64 + lane;
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int kg0 = 0; kg0 < kgroups; kg0 += 8) {
      f32x4 a0[8], a1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (kg0 + u < kgroups) { a0[u] = w0[(kg0 + u) * 64]; a1[u] = w1[(kg0 + u) * 64]; }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (kg0 + u < kgroups) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float bq = xin[((kg0 + u) * 16 + j * 4 + quad) * 16 + col];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u][j], bq, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u][j], bq, acc1, 0, 0, 0);
          }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o0 = t0 * 16 + quad * 4 + r;
      const float y0 = acc0[r] + bias[o0];
      xout[o0 * 16 + col] = relu ? fmaxf(y0, 0.0f) : y0;
      if (two) { const int o1 = t1 * 16 + quad * 4 + r; const float y1 = acc1[r] + bias[o1]; xout[o1 * 16 + col] = relu ? fmaxf(y1, 0.0f) : y1; }
    }
  }
}
// 16 outputs x 16 boards of a layer with ONE output tile per `tile` index, K split over the waves that share the tile:
// wave w takes tile (w % ntile), K part (w / ntile) of HFC_WAVES / ntile; partial tiles go to part[w][16][16]
__device__ __forceinline__ void fc_ksplit(const float* __restrict__ wt, int K, int ntile, const float* xin, float* part) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 15, quad = lane >> 4;
  const int kgroups = K / 16, kparts = HFC_WAVES / ntile, t = wave % ntile, kp = wave / ntile;
  const int gper = kgroups / kparts;
  const f32x4* w = reinterpret_cast<const f32x4*>(wt) + static_cast<size_t>(t) * kgroups * 64 + lane;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int kg0 = kp * gper; kg0 < (kp + 1) * gper; kg0 += 8) {
    f32x4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (kg0 + u < (kp + 1) * gper) a[u] = w[(kg0 + u) * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (kg0 + u < (kp + 1) * gper) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], xin[((kg0 + u) * 16 + j * 4 + quad) * 16 + col], acc, 0, 0, 0);
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[(wave * 16 + quad * 4 + r) * 16 + col] = acc[r];
}

__global__ __launch_bounds__(HFC_THREADS) void k_heads_fc_a(SpDesc nd, SpPtrs np, const float* __restrict__ vpool, const float* __restrict__ ppool,
                                                            float* __restrict__ hidden, float* __restrict__ glob, uint32_t batch,
                                                            const uint32_t* __restrict__ row_count) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_fc[];
  if (row_count) { const uint32_t n = *row_count; batch = n < batch ? n : batch; }
  const int Hd = nd.v_hidden, NS = Hd / HFC_SLICE, parts = NS + (nd.num_global > 0 ? 1 : 0);
  const uint32_t group = blockIdx.x / parts;
  const int part_id = blockIdx.x % parts;
  const uint32_t b0 = group * 16;
  if (b0 >= batch) return;
  const int tid = threadIdx.x;
  const int Hmax = nd.v_hidden > nd.pi_hidden ? nd.v_hidden : nd.pi_hidden;
  float* xa = reinterpret_cast<float*>(lds_fc);               // [64 inputs, or a hidden layer when there are more than two][16]
  float* xb = xa + heads_fc_a_first(Hmax, nd.v_fc_layers) * 16;   // [hidden][16]
  float* part = xb + Hmax * 16;                               // [HFC_WAVES][16 outputs][16 boards]
  const float* pooled = part_id < NS ? vpool : ppool;
  for (int i = tid; i < 64 * 16; i += HFC_THREADS) {
    const int k = i / 16, b = i % 16;
    xa[i] = (b0 + b < batch) ? pooled[static_cast<size_t>(b0 + b) * 64 + k] : 0.0f;
  }
  __syncthreads();
  if (part_id < NS) {
    float* hid = hidden + static_cast<size_t>(group) * Hd * 16;   // the group's last hidden layer, [unit][16]
    const int L = nd.v_fc_layers;
    if (L == 1) { fc_tiles(np.fc1_w, np.fc1_b, 64, part_id * (HFC_SLICE / 16), HFC_SLICE / 16, xa, hid, true); return; }
    fc_tiles(np.fc1_w, np.fc1_b, 64, 0, Hd / 16, xa, xb, true);
    __syncthreads();
    float *cur = xb, *nxt = xa;
    for (int l = 0; l + 2 < L; ++l) {
      fc_tiles(np.fcx_w + static_cast<size_t>(l) * Hd * Hd, np.fcx_b + l * Hd, Hd, 0, Hd / 16, cur, nxt, true);
      __syncthreads();
      float* t = cur; cur = nxt; nxt = t;
    }
    fc_tiles(np.fcx_w + static_cast<size_t>(L - 2) * Hd * Hd, np.fcx_b + (L - 2) * Hd, Hd, part_id * (HFC_SLICE / 16), HFC_SLICE / 16, cur, hid, true);
    return;
  }
  // ---- pi_global -------------------------------------------------------------------------------------------------------
  const int Hp = nd.pi_hidden, Gn = nd.num_global;
  fc_tiles(np.pg1_w, np.pg1_b, 64, 0, Hp / 16, xa, xb, true);
  __syncthreads();
  fc_ksplit(np.pg2_w, Hp, 2, xb, part);                         // 32 outputs = 2 tiles, K over 4 waves each
  __syncthreads();
  if (tid < 256) {                   // LayerNorm over the G outputs of a board: 16 lanes per board, two outputs per lane
    const int b = tid >> 4, i = tid & 15;
    float y0 = np.pg2_b[i], y1 = np.pg2_b[i + 16];
    for (int kp = 0; kp < HFC_WAVES / 2; ++kp) { y0 += part[((kp * 2 + 0) * 16 + i) * 16 + b]; y1 += part[((kp * 2 + 1) * 16 + i) * 16 + b]; }
    if (i >= Gn) y0 = 0.0f;
    if (i + 16 >= Gn) y1 = 0.0f;
    float sum = y0 + y1;
    for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / static_cast<float>(Gn);
    float dv = (i < Gn ? (y0 - mean) * (y0 - mean) : 0.0f) + (i + 16 < Gn ? (y1 - mean) * (y1 - mean) : 0.0f);
    for (int off = 8; off > 0; off >>= 1) dv += __shfl_xor(dv, off, 64);
    const float inv = 1.0f / sqrtf(dv / static_cast<float>(Gn) + 1e-5f);
    float* g = glob + (static_cast<size_t>(group) * 16 + b) * 32;
    if (i < Gn) g[i] = (y0 - mean) * inv * np.pg_ln_g[i] + np.pg_ln_b[i];
    if (i + 16 < Gn) g[i + 16] = (y1 - mean) * inv * np.pg_ln_g[i + 16] + np.pg_ln_b[i + 16];
  }
}

__global__ __launch_bounds__(HFC_THREADS) void k_heads_fc_b(SpDesc nd, SpPtrs np, const float* __restrict__ hidden, const float* __restrict__ glob,
                                                            float* __restrict__ v_out, float* __restrict__ pi_out, uint32_t batch,
                                                            const uint32_t* __restrict__ rows, const uint32_t* __restrict__ row_count) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_fc[];
  if (rows) { const uint32_t n = *row_count; batch = n < batch ? n : batch; }
  const uint32_t group = blockIdx.x, b0 = group * 16;
  if (b0 >= batch) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int Hd = nd.v_hidden;
  float* xa = reinterpret_cast<float*>(lds_fc);               // [v_hidden][16]
  float* part = xa + Hd * 16;                                 // [HFC_WAVES][16 outputs][16 boards]
  const float* hid = hidden + static_cast<size_t>(group) * Hd * 16;
  for (int i = tid; i < Hd * 16; i += HFC_THREADS) xa[i] = hid[i];
  __syncthreads();
  fc_ksplit(np.fc2_w, Hd, 1, xa, part);                       // one tile of 16 padded outputs, K over the 8 waves
  __syncthreads();
  if (tid < 256) {                   // lane group of 16 = one board; lane i of the group = output i
    const int P1 = nd.num_players + 1, b = tid >> 4, i = tid & 15;
    const bool on = i < P1;
    float a = on ? np.fc2_b[i] : 0.0f;
    for (int w = 0; w < HFC_WAVES; ++w) a += part[(w * 16 + i) * 16 + b];
    float mx = on ? a : -__builtin_inff();
    for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float e = on ? expf(a - mx) : 0.0f;
    float sum = e;
    for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (on && b0 + b < batch) {
      const uint32_t out_row = rows ? rows[b0 + b] : b0 + b;
      v_out[static_cast<size_t>(out_row) * P1 + i] = e / sum;
    }
  }
  if (nd.num_global == 0) return;
  const int Gn = nd.num_global, M = nd.num_moves, S = M - Gn;
  for (int b = wave; b < 16; b += HFC_WAVES) {     // one wave per board: softmax = exp(log_softmax), neural_net.py:494,816
    if (b0 + b >= batch) continue;
    const uint32_t out_row = rows ? rows[b0 + b] : b0 + b;
    float* row = pi_out + static_cast<size_t>(out_row) * M;   // [0, S): the tile's raw spatial logits
    const float g = lane < Gn ? glob[(static_cast<size_t>(group) * 16 + b) * 32 + lane] : -__builtin_inff();
    constexpr int RU = 28;           // the row in registers when it fits (StarGambit: 1690 spatial logits = 27 per lane): one
    if (S <= 64 * RU) {              // round trip for the whole row instead of three passes over it
      float r[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) { const int e = lane + 64 * u; r[u] = e < S ? row[e] : -__builtin_inff(); }
      float mx = g;
#pragma unroll
      for (int u = 0; u < RU; ++u) mx = fmaxf(mx, r[u]);
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float eg = lane < Gn ? expf(g - mx) : 0.0f;
      float sum = eg;                // (same order of additions as the three-pass form: the global term, then the row)
#pragma unroll
      for (int u = 0; u < RU; ++u) { const int e = lane + 64 * u; r[u] = e < S ? expf(r[u] - mx) : 0.0f; if (e < S) sum += r[u]; }
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
#pragma unroll
      for (int u = 0; u < RU; ++u) { const int e = lane + 64 * u; if (e < S) row[e] = r[u] / sum; }
      if (lane < Gn) row[S + lane] = eg / sum;
      continue;
    }
    float mx = g;
    for (int e = lane; e < S; e += 64) mx = fmaxf(mx, row[e]);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = lane < Gn ? expf(g - mx) : 0.0f;
    for (int e = lane; e < S; e += 64) sum += expf(row[e] - mx);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    for (int e = lane; e < S; e += 64) row[e] = expf(row[e] - mx) / sum;
    if (lane < Gn) row[S + lane] = expf(g - mx) / sum;
  }
}

// The same step as ONE launch, a workgroup per group running every layer (round 2's first form): for a small FC stack
// (v_hidden <= 256, no global actions - Brandubh: 0.3 MB of weights), where the split's second launch costs more than its
// shorter weight streams gain.  The 512-wide stacks (Tawlbwrdd, OpenTafl: 1.2 MB; StarGambit: 1.5 MB with pi_global) take the
// split form above (StarGambit 18.4 -> 19.7 games/s; Tawlbwrdd's FC step 58 -> 38 + 13 us in the mix).
__global__ __launch_bounds__(HFC_THREADS) void k_heads_fc(SpDesc nd, SpPtrs np, const float* __restrict__ vpool, const float* __restrict__ ppool,
                                                          float* __restrict__ v_out, float* __restrict__ pi_out, uint32_t batch,
                                                          const uint32_t* __restrict__ rows, const uint32_t* __restrict__ row_count) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_fc[];
  if (rows) { const uint32_t n = *row_count; batch = n < batch ? n : batch; }
  const uint32_t b0 = blockIdx.x * 16;
  if (b0 >= batch) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 15, quad = lane >> 4;
  const int Hmax = nd.v_hidden > nd.pi_hidden ? nd.v_hidden : nd.pi_hidden;
  float* xa = reinterpret_cast<float*>(lds_fc);               // [hidden][16]
  float* xb = xa + Hmax * 16;
  float* part = xb + Hmax * 16;                               // [HFC_WAVES][16 outputs][16 boards]
  (void)ppool; (void)pi_out;
  auto load_x0 = [&](const float* pooled) {
    for (int i = tid; i < 64 * 16; i += HFC_THREADS) {
      const int k = i / 16, b = i % 16;
      xa[i] = (b0 + b < batch) ? pooled[static_cast<size_t>(b0 + b) * 64 + k] : 0.0f;
    }
    __syncthreads();
  };
  auto layer = [&](const float* wt, const float* bias, int K, int N, const float* xin, float* xout, bool relu) {
    const int ntiles = N / 16, kgroups = K / 16;
    for (int t0 = wave; t0 < ntiles; t0 += 2 * HFC_WAVES) {
      const int t1 = t0 + HFC_WAVES;
      const bool two = t1 < ntiles;
      const f32x4* w0 = reinterpret_cast<const f32x4*>(wt) + static_cast<size_t>(t0) * kgroups * 64 + lane;
      const f32x4* w1 = reinterpret_cast<const f32x4*>(wt) + static_cast<size_t>(two ? t1 : t0) * kgroups * 64 + lane;
      f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int kg0 = 0; kg0 < kgroups; kg0 += 8) {            // 16 weight loads in flight per lane (and <= 128 VGPRs: two of these
        f32x4 a0[8], a1[8];                                   // waves fit a SIMD beside one tile workgroup's)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (kg0 + u < kgroups) { a0[u] = w0[(kg0 + u) * 64]; a1[u] = w1[(kg0 + u) * 64]; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (kg0 + u < kgroups) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float bq = xin[((kg0 + u) * 16 + j * 4 + quad) * 16 + col];
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u][j], bq, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u][j], bq, acc1, 0, 0, 0);
            }
          }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o0 = t0 * 16 + quad * 4 + r;
        const float y0 = acc0[r] + bias[o0];
        xout[o0 * 16 + col] = relu ? fmaxf(y0, 0.0f) : y0;
        if (two) { const int o1 = t1 * 16 + quad * 4 + r; const float y1 = acc1[r] + bias[o1]; xout[o1 * 16 + col] = relu ? fmaxf(y1, 0.0f) : y1; }
      }
    }
    __syncthreads();
  };
  // ---- value head ----------------------------------------------------------------------------------------------------
  {
    const int Hd = nd.v_hidden;
    load_x0(vpool);
    layer(np.fc1_w, np.fc1_b, 64, Hd, xa, xb, true);
    float *cur = xb, *nxt = xa;
    for (int l = 0; l + 1 < nd.v_fc_layers; ++l) {
      layer(np.fcx_w + static_cast<size_t>(l) * Hd * Hd, np.fcx_b + l * Hd, Hd, Hd, cur, nxt, true);
      float* t = cur; cur = nxt; nxt = t;
    }
    {  // output layer (one tile of 16 padded rows): k groups split over the waves, partial tiles summed in wave order
      f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
      const int gper = Hd / 16 / HFC_WAVES;
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
    if (tid < 256) {                   // lane group of 16 = one board; lane i of the group = output i
      const int P1 = nd.num_players + 1, b = tid >> 4, i = tid & 15;
      const bool on = i < P1;
      float a = on ? np.fc2_b[i] : 0.0f;
      for (int w = 0; w < HFC_WAVES; ++w) a += part[(w * 16 + i) * 16 + b];
      float mx = on ? a : -__builtin_inff();
      for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float e = on ? expf(a - mx) : 0.0f;
      float sum = e;
      for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
      if (on && b0 + b < batch) {
        const uint32_t out_row = rows ? rows[b0 + b] : b0 + b;
        v_out[static_cast<size_t>(out_row) * P1 + i] = e / sum;
      }
    }
  }
}   // (nets with global actions take the split form: leafnet.hip, fc_split)

}  // namespace sp
}  // namespace azmi_net_dev
