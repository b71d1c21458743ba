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
#include <string>
#include <vector>

#include "../../include/azmi.h"
#include "leafnet_f32.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int CH = 64;            // trunk channels
constexpr int HC = 32;            // head channels (value / policy each)
constexpr int TB = 8;             // boards per workgroup
constexpr int NTHREADS = 512;     // 8 waves: two per SIMD, so one wave's LDS latency hides under the other's MFMAs
constexpr int NWAVES = NTHREADS / 64;
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
  const float* v_fc1_w;    // [32][v_hidden]  (transposed on the host)
  const float* v_fc1_b;    // [v_hidden]
  const float* v_fc2_w;    // [P+1][v_hidden]
  const float* v_fc2_b;    // [P+1]
  const uint8_t* pi_fc_w;  // per pixel position p: bf16 A-fragment of W_p[m][c] = W[m][c*H*W + p] (rows >= M zero), high parts, then low parts
  const float* pi_fc_b;    // [M]
};

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

__device__ __forceinline__ bf16x8 lds_read_frag(const uint8_t* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}

// =====================================================================================================
// Connect4-family net (6x7 board, 64 trunk / 32 head channels, flat policy head): k_leafnet_c4
//
// One workgroup = 4 waves (one per SIMD) carries a tile of 6 boards (252 pixels = 16 n-tiles, 4 per wave, exactly) through
// the WHOLE tower; it needs 79 KB of LDS and at most 256 VGPRs, so TWO workgroups share a CU and one's matrix phases
// cover the other's barriers, epilogues, stem and heads (the 8-wave / 8-board predecessor owned a CU alone: its matrix pipe
// was busy 49 % of the time and a launch had an 88 us floor).
//   * weights never pass through registers: every wave streams its share of the 8 KB weight chunks (2 k-steps x 4 m-tiles
//     of MFMA A-fragments) from L2 straight into a 5-slot LDS ring with global_load_lds_dwordx4, three chunks ahead of the
//     matrix cores; one s_barrier per chunk (counted vmcnt, raw barrier: the DMA stays in flight across it);
//   * the k-loop is software-pipelined by hand: the fragments of k-step i + 1 are read, one ds_read_b128 per two MFMAs,
//     while the 16 MFMAs of k-step i issue (a lone wave cannot hide an LDS round trip any other way);
//   * activations: bf16, eight 8-channel planes of [pixel slot][16 B] (conflict-free ds_read_b128 B-fragments, out-of-board
//     taps redirected to all-zero cells), the residual stream in fp32 accumulators for the whole kernel;
//   * policy head: logits[m][board] = sum_p W_p[m][c] h[c][board, p] as 42 accumulating 16x16x32 MFMAs whose B operand IS the
//     activation-plane format (no transposition); W and h are split into bf16 high + low parts (3 MFMAs per pixel
//     position: hi*hi + hi*lo + lo*hi), which keeps the flat head at fp32-like accuracy on the bf16 pipe;
//   * every reduction runs in an order that does not depend on where a board sits in its tile or batch, so a position's
//     (v, pi) is bit-identical wherever the engine's unordered eval list places it.
namespace c4 {
constexpr int TBW = 6;                 // boards per workgroup
constexpr int NTH = 256, NWV = 4;      // threads, waves (one per SIMD)
constexpr int NTW = 4;                 // n-tiles per wave
constexpr int BH = 6, BW = 7, PIX = BH * BW, NPIX = TBW * PIX;   // 252 GEMM columns
constexpr int NT = 16;                 // n-tiles (256 columns, the last 4 unused)
constexpr int ZSLOT = NT * 16;         // first all-zero cell
constexpr int SLOTS = ZSLOT + 16;      // 272
constexpr int PLANE = SLOTS * 16;      // 4352 B, a multiple of 256
constexpr int ZERO_OFF = ZSLOT * 16;
constexpr int ACT_BYTES = 8 * PLANE;   // 34,816
constexpr int CHUNK_KS = 2;            // k-steps per weight chunk (= one 3x3 tap)
constexpr int CHUNK_BYTES = CHUNK_KS * MT * WFRAG_BYTES;   // 8,192
constexpr int PIECES = CHUNK_BYTES / (NWV * WFRAG_BYTES);  // 1 KB DMA pieces per wave and chunk: 2
constexpr int NRING = 5;               // one chunk being read, one landed, two in flight, one being refilled
constexpr int RING_BYTES = NRING * CHUNK_BYTES;            // 40,960
constexpr int CHUNKS_PER_CONV = 18 / CHUNK_KS;             // 9
constexpr int MAXDEPTH = 6;
constexpr int PRM_FLOATS = CH + MAXDEPTH * 3 * CH + CH;    // stem bias | per block a1 b1 c1 | head bias
constexpr int LDS_BYTES = ACT_BYTES + RING_BYTES + PRM_FLOATS * 4;   // 80,896: two workgroups per CU
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
static_assert(NPIX * HC * 4 + NWV * 256 * 4 <= RING_BYTES, "value-head scratch + policy partials reuse the ring");

typedef __attribute__((address_space(3))) void* lptr_t;
// 64 lanes x 16 B from per-lane global addresses into LDS at the wave-uniform `dst` + lane * 16, no VGPR destination.
// Issued from inline asm on purpose: hipcc models the builtin form (__builtin_amdgcn_global_load_lds) as a FLAT access that
// may touch LDS, and while one is pending every LDS wait it inserts becomes lgkmcnt(0) - the software pipeline of the
// convolutions (fragments of the next k-step in flight behind the MFMAs of this one) then stalls on every other k-step.
// The DMA is ordered by hand instead: counted s_waitcnt vmcnt + s_barrier (wait_vm / conv3x3).
__device__ __forceinline__ void dma16(const uint8_t* src_lane, uint8_t* dst_wave) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lptr_t)dst_wave)));
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src_lane), "s"(dst) : "memory");
}
// two consecutive 1 KB pieces (source and destination both advance by 1024: the instruction offset applies to both sides)
__device__ __forceinline__ void dma16x2(const uint8_t* src_lane, uint8_t* dst_wave) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lptr_t)dst_wave)));
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "global_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src_lane), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else static_assert(N == 0, "add the count");
}
// workgroup barrier that leaves LDS-DMA in flight: this wave's LDS accesses retire, then a raw s_barrier (__syncthreads()
// would also drain vmcnt, i.e. the prefetched weight chunks)
__device__ __forceinline__ void barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// DBG != 0: timing-only variants (wrong results): 1 = no DMA waits, 2 = no DMA waits and no chunk barriers, 3 = no weight DMA at all,
// 4 = no fragment reads in the convolutions (MFMA stream alone), 5 = no MFMAs in the convolutions (fragment reads alone),
// 6 = stem only, 7 = stem + trunk
template <int CIN, int MAXP1, int MAXM, int DBG = 0>
__global__ __launch_bounds__(NTH, 2) void k_leafnet_c4(NetDesc nd, NetPtrs np, const float* __restrict__ canon,
                                                        float* __restrict__ v_out, float* __restrict__ pi_out, uint32_t batch,
                                                        const uint32_t* __restrict__ rows, const uint32_t* __restrict__ row_count) {
  static_assert(9 * CIN <= 64, "stem im2col fits one 64-row k-chunk");
  // rows != nullptr: evaluate only the rows listed in rows[0 .. *row_count) (the engine's eval list: slots whose
  // pending leaf really needs the net); workgroups past the end of the list leave at once
  // (the list length and this thread's list entry are loaded together: one memory round trip, not two)
  const uint32_t max_rows = batch;
  uint32_t in_row = blockIdx.x * TBW + threadIdx.x / (CIN * PIX / 4);
  if (in_row >= max_rows) in_row = max_rows - 1;
  if (rows) { batch = *row_count; in_row = rows[in_row]; }
  if (blockIdx.x * TBW >= batch) return;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* const act = lds;
  uint8_t* const ring = lds + ACT_BYTES;
  float* const prm = reinterpret_cast<float*>(lds + ACT_BYTES + RING_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, quad = lane >> 4;
  const uint32_t board0 = blockIdx.x * TBW;
  const int depth = nd.depth;
  const int nchunks = 2 * depth * CHUNKS_PER_CONV + 1;        // + the head 1x1 weights, streamed as one more chunk
  const size_t block_stride = 3 * CH * sizeof(float) + 2 * static_cast<size_t>(18 * MT * WFRAG_BYTES);

  // ---- small fp32 parameters -> LDS (plain loads, before any DMA is in flight) ------------------------------------
  if (tid < CH) { prm[tid] = np.stem_b[tid]; prm[CH + MAXDEPTH * 3 * CH + tid] = np.head_b[tid]; }
  if (tid < 3 * CH)
    for (int blk = 0; blk < depth; ++blk) prm[CH + blk * 3 * CH + tid] = reinterpret_cast<const float*>(np.blocks + blk * block_stride)[tid];
  // source row of this thread's 16-byte piece of the tile's input planes
  // (boards past the end of the batch read whatever row their stale list entry names - any slot's planes - and are never stored)
  const int in_piece = tid % (CIN * PIX / 4);
  static_assert((CIN * PIX) % 4 == 0 && TBW * (CIN * PIX / 4) <= NTH, "one 16-byte input piece per thread");
  // ---- zero the activation planes once (covers the zero cells and the k rows the stem does not use) ----------------
  for (int i = tid * 16; i < ACT_BYTES; i += NTH * 16) *reinterpret_cast<u32x4*>(act + i) = u32x4{0, 0, 0, 0};

  // ---- weight stream: chunk g of the run = 8 KB of fragments (one tap of one convolution); every wave moves 2 of its
  // 8 pieces.  Chunk g lives in ring slot g % NRING.
  // The stream's chunks are consecutive in the weight blob except for the 768 bytes of fp32 block parameters in front of
  // every block's fragments; the head's fragments come last.  `wnext` walks the blob: source of the next chunk to issue.
  const uint8_t* wnext = np.blocks + 3 * CH * sizeof(float) + (wave * PIECES) * WFRAG_BYTES;
  int issued = 0, in_block = 0;            // chunks issued so far; how many of them belong to the current block
  auto issue_next = [&](int slot) {
    const uint8_t* src = wnext + lane * 16;
    uint8_t* dst = ring + slot * CHUNK_BYTES + (wave * PIECES) * WFRAG_BYTES;
    static_assert(PIECES == 2, "dma16x2 moves the wave's two pieces");
    dma16x2(src, dst);
    // after the run's last chunk (the head's) the stream keeps re-sending that chunk into slots nobody reads any more: the
    // tail of the run then needs no special cases - one wait count, no branch around the issue (see conv3x3)
    ++issued; ++in_block;
    if (issued < nchunks - 1) {
      wnext += CHUNK_BYTES;
      if (in_block == 2 * CHUNKS_PER_CONV) { in_block = 0; wnext += 3 * CH * sizeof(float); }
    } else {
      wnext = np.head_w + (wave * PIECES) * WFRAG_BYTES;
    }
  };
  // ring slots 3 and 4 first hold the stem's operands: 8 KB of stem fragments, and the tile's raw input planes
  uint8_t* const stem_w_lds = ring + 3 * CHUNK_BYTES;
  float* const raw = reinterpret_cast<float*>(ring + 4 * CHUNK_BYTES);
  static_assert(2 * MT * WFRAG_BYTES <= CHUNK_BYTES && TBW * CIN * PIX * 4 <= CHUNK_BYTES, "stem operands fit two ring slots");
  if (tid < TBW * (CIN * PIX / 4))
    dma16(reinterpret_cast<const uint8_t*>(canon + static_cast<size_t>(in_row) * (CIN * PIX)) + in_piece * 16,
          reinterpret_cast<uint8_t*>(raw) + wave * 1024);
#pragma unroll
  for (int i = 0; i < 2; ++i) dma16(np.stem_w + (wave * 2 + i) * WFRAG_BYTES + lane * 16, stem_w_lds + (wave * 2 + i) * WFRAG_BYTES);
  issue_next(0);
  issue_next(1);
  issue_next(2);

  // ---- per-lane pixel geometry of the wave's n-tiles: tile t = wave * 4 + j ---------------------------------------
  // the lane's pixel of tile j sits at byte pix0 + j * 256 of a plane (tiles are 16 consecutive pixel slots)
  const int pix0 = (wave * NTW * 16 + col) * 16;
  uint32_t tap_ok[NTW];  // bit tap: the 3x3 neighbour (tap/3-1, tap%3-1) is on the board; 0 for the 4 unused columns
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

  // ---- stem: im2col of the CIN input planes, B[k = tap*CIN + ci][pixel] (k < 64), one thread per pixel --------------
  wait_vm<6>();                      // input planes + stem fragments have landed (chunks 0-2 stay in flight)
  barrier_lds();
  if (tid < NPIX) {
    const int b = tid / PIX, p = tid % PIX, h = p / BW, w = p % BW;
    const float* rb = raw + b * (CIN * PIX);
#pragma unroll
    for (int pl = 0; pl < (9 * CIN + 7) / 8; ++pl) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = pl * 8 + e, tap = k / CIN, ci = k % CIN;
        float val = 0.0f;
        if (k < 9 * CIN) {
          const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
          if (hh >= 0 && hh < BH && ww >= 0 && ww < BW) val = rb[ci * PIX + hh * BW + ww];
        }
        o[e] = static_cast<__bf16>(val);
      }
      *reinterpret_cast<bf16x8*>(act + pl * PLANE + tid * 16) = o;
    }
  }
  barrier_lds();

  // ---- residual stream: accumulators s[j][mt] (fp32), kept for the whole kernel -----------------------------------
  f32x4 s[NTW][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 bias = *reinterpret_cast<const f32x4*>(prm + mt * 16 + quad * 4);
#pragma unroll
    for (int j = 0; j < NTW; ++j) s[j][mt] = bias;
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 a[MT], b[NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = lds_read_frag(stem_w_lds + (ks * MT + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
    for (int j = 0; j < NTW; ++j) b[j] = lds_read_frag(act + (ks * 4 + quad) * PLANE + pix0 + j * 256);
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) s[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[j], s[j][mt], 0, 0, 0);
  }

  // epilogue helper: 4 consecutive channels (mt*16 + quad*4 ..) of the lane's pixel as bf16, into plane `plane0 + quad/2`
  auto store4 = [&](int j, int plane0, f32x4 val) {
    bf16x4 o;
    o[0] = static_cast<__bf16>(val[0]); o[1] = static_cast<__bf16>(val[1]);
    o[2] = static_cast<__bf16>(val[2]); o[3] = static_cast<__bf16>(val[3]);
    *reinterpret_cast<bf16x4*>(act + (plane0 + (quad >> 1)) * PLANE + pix0 + j * 256 + (quad & 1) * 8) = o;
  };

  // B-fragment addressing: the lane's pixel in plane `quad` (po), and the distance from there to the all-zero cell with
  // the pixel's own 16-byte-slot residue (zd): a tap reads po + (on the board ? tap offset : zd), two VALU per fragment
  const uint8_t* const po = act + quad * PLANE + pix0;     // tile j: + j * 256, an immediate of the read
  int zd[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) zd[j] = ZERO_OFF + (pix0 & 0xF0) - pix0 - j * 256;
  const uint8_t* const wlane = ring + lane * 16;

  // One 3x3 convolution over `act`, accumulating into acc[][]; g0 = index of its first weight chunk in the run, slot0 = that
  // chunk's ring slot.  Software-pipelined over its 18 k-steps (k-step = half a tap, chunk = one tap): the A (ring) and B
  // (activation) fragments of k-step i + 1 are read, interleaved one ds_read per two MFMAs, while the 16 MFMAs of k-step i
  // issue - across chunk boundaries too.  The barrier that opens chunk c + 1 therefore sits BEFORE the second k-step of
  // chunk c (whose fragments are already on their way): passing it means chunk c + 1 has landed for every wave (each
  // waited for its own pieces) and every wave is done with the slot of chunk c - 1, which the DMA of chunk c + 4 refills.
  // The last such barrier of a convolution comes after every activation read of the convolution has RETURNED
  // (lgkmcnt(0)), so the epilogue may overwrite the planes without another barrier.
  // Precondition: the activations are visible and chunk g0 has landed for all waves.
  auto conv3x3 = [&](f32x4 (&acc)[NTW][MT], int slot0) {
    bf16x8 a[2][MT], b[2][NTW];
    int slot = slot0;
    auto load_a = [&](int ksl, const uint8_t* wsl, bf16x8 (&fa)[MT]) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = lds_read_frag(wsl + (ksl * MT + mt) * WFRAG_BYTES);
    };
    auto load_b = [&](int ks, bf16x8 (&fb)[NTW]) {
      const int tap = ks >> 1, half = ks & 1;
      const int tap_off = ((tap / 3 - 1) * BW + (tap % 3 - 1)) * 16;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        int z = zd[j];
        asm volatile("" : "+v"(z));     // keep the select here: hoisted out of the block loop, the 36 offsets of a convolution cost 36 VGPRs (spills)
        const int d = ((tap_ok[j] >> tap) & 1u) ? tap_off : z;
        fb[j] = lds_read_frag(po + d + (j * 256 + half * 4 * PLANE));
      }
    };
    load_a(0, wlane + slot * CHUNK_BYTES, a[0]);
    load_b(0, b[0]);
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
      const int cur = ks & 1, c = ks / CHUNK_KS, ksl = ks % CHUNK_KS;
      if (ksl == CHUNK_KS - 1) {            // open chunk c + 1 (the next convolution's first chunk when c == 8)
        if constexpr (DBG == 0) wait_vm<4>();      // chunk g + 1 has landed; g + 2 and g + 3 (2 pieces each) may still be in flight
        if (c == CHUNKS_PER_CONV - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (DBG != 2) { __builtin_amdgcn_s_barrier(); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG != 3) issue_next(slot == 0 ? NRING - 1 : slot - 1);     // chunk g + 4 (past the end: a harmless repeat)
        slot = slot == NRING - 1 ? 0 : slot + 1;                 // ring slot of chunk c + 1
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DBG != 4) {
        if (ks + 1 < 18) {
          load_a((ks + 1) % CHUNK_KS, wlane + slot * CHUNK_BYTES, a[cur ^ 1]);
          load_b(ks + 1, b[cur ^ 1]);
        }
      }
      if constexpr (DBG == 5) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) asm volatile("" :: "v"(b[cur][j]));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" :: "v"(a[cur][mt]));
      } else {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[cur][mt], b[cur][j], acc[j][mt], 0, 0, 0);
        if (ks + 1 < 18) {
          // issue order inside the k-step: 2 MFMAs and one A fragment of the next k-step (x4: its first MFMAs need all
          // four), then 2 MFMAs, the address arithmetic and the read of one B fragment (x4)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return slot;
  };

  if constexpr (DBG == 6) { if (s[0][0][0] == 12345.678f) v_out[0] = s[1][1][1] + s[2][2][2] + s[3][3][3]; return; }   // timing: stem only
  int slot = 0;
  for (int blk = 0; blk < depth; ++blk) {
    const float* affine = prm + CH + blk * 3 * CH;        // a1[64] b1[64] c1[64]
    if (blk == 0) barrier_lds();                          // the stem's reads of the planes (later blocks: the convolution's last barrier)
    // t = relu(a1 * s + b1) -> act
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(affine + mt * 16 + quad * 4);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(affine + CH + mt * 16 + quad * 4);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        if (!((real_m >> j) & 1u)) continue;
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaxf(a1[r] * s[j][mt][r] + b1[r], 0.0f);
        store4(j, mt * 2, t);
      }
    }
    // u = relu(conv1(t) + c1)
    f32x4 u[NTW][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(affine + 2 * CH + mt * 16 + quad * 4);
#pragma unroll
      for (int j = 0; j < NTW; ++j) u[j][mt] = c1;
    }
    if (blk == 0) { if constexpr (DBG == 0) wait_vm<4>(); }      // chunk 0 (chunks 1 and 2 stay in flight)
    barrier_lds();                                        // barrier C: planes visible (and, block 0, chunk 0 landed for every wave)
    if (blk == 0) { if constexpr (DBG != 3) issue_next(3); }  // the stem is done with slot 3 (slot 4: chunk 4, at the first chunk barrier)
    slot = conv3x3(u, slot);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        if (!((real_m >> j) & 1u)) continue;
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaxf(u[j][mt][r], 0.0f);
        store4(j, mt * 2, t);
      }
    barrier_lds();
    // s = s + conv2(u)
    slot = conv3x3(s, slot);
  }

  if constexpr (DBG == 7) { if (s[0][0][0] == 12345.678f) v_out[0] = s[1][1][1] + s[2][2][2] + s[3][3][3]; return; }   // timing: stem + trunk
  // ---- heads: h = relu(conv1x1(s) + bh), 64 rows = 32 value + 32 policy channels -------------------------------------
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
      if ((real_m >> j) & 1u) store4(j, mt * 2, s[j][mt]);
  f32x4 hacc[NTW][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 bh = *reinterpret_cast<const f32x4*>(prm + CH + MAXDEPTH * 3 * CH + mt * 16 + quad * 4);
#pragma unroll
    for (int j = 0; j < NTW; ++j) hacc[j][mt] = bh;
  }
  wait_vm<0>();                       // the head fragments (the run's last chunk, ring slot `slot`): nothing is in flight any more
  // Every global operand of the heads is requested HERE, together, and lands while the head convolution runs: the
  // heads are a chain of small dependent steps, and one memory round trip per step used to cost more than the arithmetic.
  const int P1 = nd.num_players + 1, M = nd.num_moves, Hd = nd.v_hidden;
  const int p0 = wave < 2 ? wave * 11 : 22 + (wave - 2) * 10, npos = wave < 2 ? 11 : 10;   // this wave's pixel positions of the policy FC
  bf16x8 pw[11][2];
  {
    const uint8_t* wp = np.pi_fc_w + static_cast<size_t>(p0) * 2 * WFRAG_BYTES + lane * 16;
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      const int ii = i < npos ? i : 0;
      pw[i][0] = *reinterpret_cast<const bf16x8*>(wp + (ii * 2) * WFRAG_BYTES);
      pw[i][1] = *reinterpret_cast<const bf16x8*>(wp + (ii * 2 + 1) * WFRAG_BYTES);
    }
  }
  const float pib = tid < TBW * M ? np.pi_fc_b[tid % M] : 0.0f;
  uint32_t out_row = board0 + tid / (MAXP1 + MAXM);          // output row of the softmax threads (TBW x (MAXP1 + MAXM) of them)
  const bool out_on = tid < TBW * (MAXP1 + MAXM) && out_row < batch;
  if (out_on && rows) out_row = rows[out_row];
  barrier_lds();
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 a[MT], b[NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = lds_read_frag(ring + slot * CHUNK_BYTES + (ks * MT + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
    for (int j = 0; j < NTW; ++j) b[j] = lds_read_frag(act + (ks * 4 + quad) * PLANE + pix0 + j * 256);
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) hacc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[j], hacc[j][mt], 0, 0, 0);
  }
  barrier_lds();
  // (the value head's FC operands are requested now - the head accumulators are about to die - and land during the policy FC)
  float w1[HC], w1b = 0.0f;           // value fc1 column of this thread (weights transposed on the host: [32][v_hidden])
  const int o1 = tid < Hd ? tid : 0;
#pragma unroll
  for (int i = 0; i < HC; ++i) w1[i] = np.v_fc1_w[i * Hd + o1];
  w1b = np.v_fc1_b[o1];
  float w2[MAXP1][4], w2b[MAXP1];     // value fc2: this lane's hidden units lane + 64 k
#pragma unroll
  for (int o = 0; o < MAXP1; ++o) {
    w2b[o] = o < P1 ? np.v_fc2_b[o] : 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) w2[o][k] = (o < P1 && lane + 64 * k < Hd) ? np.v_fc2_w[o * Hd + lane + 64 * k] : 0.0f;
  }
  // value channels (rows 0-31) as fp32 [pixel][32] over the (now free) ring; policy channels (rows 32-63) back into the
  // activation planes as bf16 high parts (planes 0-3) and low parts (planes 4-7): h = hi + lo to ~16 bits
  float* const vbuf = reinterpret_cast<float*>(ring);
  float* const part = vbuf + NPIX * HC;                   // [NWV][16 moves][16 boards] partial policy logits
  float* const vpool = prm;                               // [TBW][32]
  float* const logits = prm + TBW * HC;                   // [TBW][MAXP1 + MAXM]
  float* const vh = reinterpret_cast<float*>(act);        // [TBW][256] hidden layer of the value head (after the policy MFMAs)
  static_assert(TBW * HC + TBW * (4 + 16) <= PRM_FLOATS && TBW * 256 * 4 <= ACT_BYTES, "head scratch must fit");
  static_assert(TBW * (4 + 16) <= NTH, "one softmax thread per output");
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    if (!((real_m >> j) & 1u)) continue;
    const int n = (wave * NTW + j) * 16 + col;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaxf(hacc[j][mt][r], 0.0f);
      *reinterpret_cast<f32x4*>(vbuf + n * HC + mt * 16 + quad * 4) = o;
    }
#pragma unroll
    for (int mt = 2; mt < 4; ++mt) {
      f32x4 hi, lo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = fmaxf(hacc[j][mt][r], 0.0f);
        hi[r] = static_cast<float>(static_cast<__bf16>(x));
        lo[r] = x - hi[r];
      }
      store4(j, (mt - 2) * 2, hi);
      store4(j, 4 + (mt - 2) * 2, lo);
    }
  }
  barrier_lds();
  {
    // policy logits: wave w sums pixel positions [p0, p0 + npos) (11, 11, 10, 10), partial tiles are added in wave order
    const int bsrc0 = col < TBW ? col * PIX * 16 : ZERO_OFF + col * 16;      // board `col`, or an all-zero cell
    const int bstep = col < TBW ? 16 : 0;
    f32x4 pacc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      if (i < npos) {
        const int bsrc = bsrc0 + (p0 + i) * bstep;
        const bf16x8 bh = lds_read_frag(act + quad * PLANE + bsrc);
        const bf16x8 bl = lds_read_frag(act + (4 + quad) * PLANE + bsrc);
        pacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw[i][1], bh, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw[i][0], bl, pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw[i][0], bh, pacc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[(wave * 16 + quad * 4 + r) * 16 + col] = pacc[r];
  }
  if (tid < TBW * HC) {      // average pool of the value channels: pixel order, four interleaved partial sums
    const int b = tid >> 5, c = tid & 31;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    const float* vb = vbuf + (b * PIX) * HC + c;
#pragma unroll
    for (int p = 0; p < 40; p += 4) { a0 += vb[p * HC]; a1 += vb[(p + 1) * HC]; a2 += vb[(p + 2) * HC]; a3 += vb[(p + 3) * HC]; }
    a0 += vb[40 * HC]; a1 += vb[41 * HC];
    vpool[b * HC + c] = ((a0 + a1) + (a2 + a3)) / static_cast<float>(PIX);
  }
  barrier_lds();
  if (tid < TBW * M) {
    const int b = tid / M, m = tid % M;
    float a = pib;
#pragma unroll
    for (int w = 0; w < NWV; ++w) a += part[(w * 16 + m) * 16 + b];
    logits[b * (MAXP1 + MAXM) + MAXP1 + m] = a;
  }
  if (tid < Hd) {            // value fc1
    float acc[TBW];
#pragma unroll
    for (int b = 0; b < TBW; ++b) acc[b] = w1b;
#pragma unroll
    for (int i = 0; i < HC; ++i) {
#pragma unroll
      for (int b = 0; b < TBW; ++b) acc[b] += w1[i] * vpool[b * HC + i];
    }
#pragma unroll
    for (int b = 0; b < TBW; ++b) vh[b * 256 + tid] = fmaxf(acc[b], 0.0f);
  }
  barrier_lds();
  for (int b = wave; b < TBW; b += NWV) {      // value fc2: a wave per board, lane l sums hidden units l, l + 64, ...
    float acc[MAXP1];
#pragma unroll
    for (int o = 0; o < MAXP1; ++o) acc[o] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float x = lane + 64 * k < Hd ? vh[b * 256 + lane + 64 * k] : 0.0f;
#pragma unroll
      for (int o = 0; o < MAXP1; ++o) acc[o] += w2[o][k] * x;
    }
#pragma unroll
    for (int o = 0; o < MAXP1; ++o) {
      if (o < P1) {
        float a = acc[o];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if (lane == 0) logits[b * (MAXP1 + MAXM) + o] = a + w2b[o];
      }
    }
  }
  barrier_lds();
  // softmax = exp(log_softmax), neural_net.py:468,508,816: one thread per output entry; every thread of a group walks the
  // group's logits in the same order, so the shared maximum and sum are bit-identical across the group
  if (out_on) {
    const int b = tid / (MAXP1 + MAXM), k = tid % (MAXP1 + MAXM);
    const bool is_v = k < MAXP1;
    const int idx = is_v ? k : k - MAXP1, cnt = is_v ? P1 : M;
    if (idx < cnt) {
      const float* lg = logits + b * (MAXP1 + MAXM) + (is_v ? 0 : MAXP1);
      float mx = lg[0];
      for (int i = 1; i < cnt; ++i) mx = fmaxf(mx, lg[i]);
      float sum = 0.0f;
      for (int i = 0; i < cnt; ++i) sum += expf(lg[i] - mx);
      const float pr = expf(lg[idx] - mx) / sum;
      if (is_v) v_out[static_cast<size_t>(out_row) * P1 + idx] = pr;
      else pi_out[static_cast<size_t>(out_row) * M + idx] = pr;
    }
  }
}
}  // namespace c4

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
constexpr int TBS11 = 3, TBS7 = 7;
constexpr int HCS = 64;

struct SpatialDesc {
  int C_in, H, W, depth, num_moves, num_players, v_hidden, v_fc_layers, pol_ch;
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
};

template <int H, int W, int TBS>
__global__ __launch_bounds__(NTHREADS, 2) void k_leafnet_spatial(SpatialDesc nd, SpatialPtrs np, const float* __restrict__ canon,
                                                                  float* __restrict__ vpool_out, float* __restrict__ pi_out,
                                                                  uint32_t batch) {
  using G = Geo<H, W, TBS>;
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
  float* raw = reinterpret_cast<float*>(wbuf + 16384);
  const int plane_sz = nd.C_in * G::PIX;
  for (int i = tid; i < TBS * plane_sz; i += NTHREADS) {
    const uint32_t b = board0 + i / plane_sz;
    raw[i] = b < batch ? canon[static_cast<size_t>(b) * plane_sz + (i % plane_sz)] : 0.0f;
  }
  // the im2col matrix has 9*C_in rows; the eight 8-row planes hold 64 of them, so a stem with more (8 input planes:
  // OpenTafl) runs in passes of 64 rows that accumulate into the same tiles
  const int npass = (9 * nd.C_in + 63) / 64;
  for (int i = tid * 16; i < npass * 2 * MT * WFRAG_BYTES; i += NTHREADS * 16)
    *reinterpret_cast<u32x4*>(wbuf + i) = *reinterpret_cast<const u32x4*>(np.stem_w + i);

  f32x4 s[G::NT_W][MT];
  {
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
      if (tid < G::NPIX) {
        const int n = tid, b = n / G::PIX, p = n % G::PIX, h = p / W, w = p % W;
        const float* rb = raw + b * plane_sz;
        for (int tap = 0; tap < 9; ++tap) {
          const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
          const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
          for (int ci = 0; ci < nd.C_in; ++ci) {
            const int k = tap * nd.C_in + ci - 64 * pass;
            if (k < 0 || k >= 64) continue;
            const float val = ok ? rb[ci * G::PIX + hh * W + ww] : 0.0f;
            *reinterpret_cast<__bf16*>(act + (k >> 3) * G::PLANE + n * 16 + (k & 7) * 2) = static_cast<__bf16>(val);
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = lds_read_frag(wbuf + ((pass * 2 + ks) * MT + mt) * WFRAG_BYTES + lane * 16);
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
          if (c < nd.pol_ch) lg[n * nd.pol_ch + c] = pl[j][mt][r];
        }
    }
  }
  __syncthreads();
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
                                                         float* __restrict__ v_out, uint32_t batch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
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
  bool spatial = false;      // spatial policy head (Tafl family): k_leafnet_spatial + k_value_fc
  SpatialDesc sd{};
  SpatialPtrs sp{};
  float* vpool = nullptr;    // [vpool_rows][64] pooled value-head features between the two kernels
  uint32_t vpool_rows = 0;
  size_t vfc_lds = 0;
  void* blob = nullptr;
  size_t blob_bytes = 0;
  int device = 0;
  size_t lds_bytes = 0;
};

namespace {
bool is_spatial(const azmi_net_desc* d) { return d->policy_channels > 0; }
size_t stem_passes(const azmi_net_desc* d) { return (9 * static_cast<size_t>(d->in_channels) + 63) / 64; }
size_t spatial_blob_bytes(const azmi_net_desc* d) {
  const size_t wconv = 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
  const size_t Hd = d->v_hidden, L = d->v_fc_layers;
  size_t n = stem_passes(d) * wsmall + CH * 4;                        // stem: one 64-row k-chunk pair per pass
  n += static_cast<size_t>(d->depth) * (3 * CH * 4 + 2 * wconv);      // trunk
  n += 2 * 8 * WFRAG_BYTES + 128 * 4;                                 // head 1x1 convs
  n += 2 * (wconv + 64 * 4);                                          // extra head convs
  n += 2 * 2 * WFRAG_BYTES + 32 * 4;                                  // policy 1x1
  n += (64 * Hd + Hd) * 4 + (L - 1) * (Hd * Hd + Hd) * 4 + (Hd * 16 + 16) * 4;
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
    const bool b11 = d->height == 11 && d->width == 11, b7 = d->height == 7 && d->width == 7;
    if (!b11 && !b7) return nfail(AZMI_ERR_INVALID, "spatial leaf net kernel: board %dx%d not instantiated", d->height, d->width);
    if (9 * d->in_channels > 128 || d->policy_channels > 32 || d->policy_channels * d->height * d->width != d->num_moves)
      return nfail(AZMI_ERR_INVALID, "spatial head: 9*C_in <= 128, policy channels <= 32, no global actions");
    if (d->v_hidden > 512 || d->v_hidden % 256 || d->v_fc_layers < 1 || d->num_players + 1 > 16) return nfail(AZMI_ERR_INVALID, "value head sizes out of range");
    if (blob_bytes != spatial_blob_bytes(d)) return nfail(AZMI_ERR_INVALID, "weight blob is %zu bytes, expected %zu", blob_bytes, spatial_blob_bytes(d));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return nfail(AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
    if (hipSetDevice(device) != hipSuccess) return nfail(AZMI_ERR_NO_DEVICE, "hipSetDevice failed");
    auto net = new azmi_net();
    net->device = device; net->spatial = true;
    net->sd = SpatialDesc{d->in_channels, d->height, d->width, d->depth, d->num_moves, d->num_players, d->v_hidden, d->v_fc_layers, d->policy_channels};
    if (hipMalloc(&net->blob, blob_bytes) != hipSuccess) { delete net; return nfail(AZMI_ERR_OOM, "hipMalloc(weights) failed"); }
    if (hipMemcpy(net->blob, blob, blob_bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(net->blob); delete net; return nfail(AZMI_ERR_NO_DEVICE, "weight upload failed"); }
    net->blob_bytes = blob_bytes;
    const uint8_t* p = static_cast<const uint8_t*>(net->blob);
    const size_t wconv = 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
    const size_t Hd = d->v_hidden, L = d->v_fc_layers;
    SpatialPtrs& sp = net->sp;
    auto f32p = [&](size_t count) { const float* q = reinterpret_cast<const float*>(p); p += count * 4; return q; };
    sp.stem_w = p; p += stem_passes(d) * wsmall; sp.stem_b = f32p(CH);
    sp.blocks = p; p += static_cast<size_t>(d->depth) * (3 * CH * 4 + 2 * wconv);
    sp.head_w = p; p += 2 * 8 * WFRAG_BYTES; sp.head_b = f32p(128);
    sp.vx_w = p; p += wconv; sp.vx_b = f32p(64);
    sp.px_w = p; p += wconv; sp.px_b = f32p(64);
    sp.pol_w = p; p += 2 * 2 * WFRAG_BYTES; sp.pol_b = f32p(32);
    sp.fc1_w = f32p(64 * Hd); sp.fc1_b = f32p(Hd);
    sp.fcx_w = f32p((L - 1) * Hd * Hd); sp.fcx_b = f32p((L - 1) * Hd);
    sp.fc2_w = f32p(Hd * 16); sp.fc2_b = f32p(16);
    auto reserve = [&](auto geo, const void* kernel) {
      using GS = decltype(geo);
      static_assert(GS::NPIX <= NTHREADS, "stem im2col maps one thread to one pixel");
      static_assert(GS::NPIX * 32 * 4 <= GS::ACT_BYTES, "pooling scratch must fit the activation planes");
      static_assert(8192 + GS::NPIX * 32 * 4 <= GS::WCONV_BYTES, "policy logits must fit the weight area");
      static_assert(16384 + GS::NPIX * 128 * 4 / 9 <= GS::WCONV_BYTES, "input staging (<= 14 planes) must fit behind two stem passes");
      net->lds_bytes = GS::ACT_BYTES + GS::WCONV_BYTES;
      return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(net->lds_bytes)) == hipSuccess;
    };
    static_assert(2 * 2 * MT * WFRAG_BYTES <= 16384, "two stem passes of weights sit in front of the input staging");
    net->vfc_lds = (2 * Hd * 16 + VFC_WAVES * 256) * sizeof(float);
    // scratch for 16384 positions up front: forward() may run under stream capture, where hipMalloc is not allowed
    net->vpool_rows = 16384;
    if (hipMalloc(reinterpret_cast<void**>(&net->vpool), static_cast<size_t>(net->vpool_rows) * 64 * sizeof(float)) != hipSuccess) {
      (void)hipFree(net->blob); delete net;
      return nfail(AZMI_ERR_OOM, "hipMalloc(value-head scratch) failed");
    }
    const bool reserved = b11 ? reserve(Geo<11, 11, TBS11>{}, reinterpret_cast<const void*>(&k_leafnet_spatial<11, 11, TBS11>))
                              : reserve(Geo<7, 7, TBS7>{}, reinterpret_cast<const void*>(&k_leafnet_spatial<7, 7, TBS7>));
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
  if (net->f32) { azmi_f32::destroy(net->f32); delete net; return; }
  (void)hipSetDevice(net->device);
  (void)hipFree(net->blob);
  if (net->vpool) (void)hipFree(net->vpool);
  delete net;
}

int azmi_net_forward(azmi_net* net, const float* dev_canonical, float* dev_v, float* dev_pi, uint32_t batch, void* stream) {
  if (!net || !dev_canonical || !dev_v || !dev_pi) return nfail(AZMI_ERR_INVALID, "null argument");
  if (batch == 0) return AZMI_OK;
  if (net->f32) {
    const char* msg = "";
    const int rc = azmi_f32::forward(net->f32, dev_canonical, dev_v, dev_pi, batch, stream, &msg);
    return rc == AZMI_OK ? rc : nfail(rc, "%s", msg);
  }
  if (net->spatial) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (batch > net->vpool_rows) {   // grows on first use / larger batches only (synchronous, outside steady state)
      (void)hipSetDevice(net->device);
      if (net->vpool) { (void)hipDeviceSynchronize(); (void)hipFree(net->vpool); net->vpool = nullptr; net->vpool_rows = 0; }
      if (hipMalloc(reinterpret_cast<void**>(&net->vpool), static_cast<size_t>(batch) * 64 * sizeof(float)) != hipSuccess)
        return nfail(AZMI_ERR_OOM, "hipMalloc(value-head scratch) failed");
      net->vpool_rows = batch;
    }
    if (net->sd.H == 11)
      k_leafnet_spatial<11, 11, TBS11><<<(batch + TBS11 - 1) / TBS11, NTHREADS, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, net->vpool, dev_pi, batch);
    else
      k_leafnet_spatial<7, 7, TBS7><<<(batch + TBS7 - 1) / TBS7, NTHREADS, net->lds_bytes, st>>>(net->sd, net->sp, dev_canonical, net->vpool, dev_pi, batch);
    k_value_fc<<<(batch + 15) / 16, VFC_THREADS, net->vfc_lds, st>>>(net->sd, net->sp, net->vpool, dev_v, batch);
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
  if (net->f32 || net->spatial)   // those paths evaluate the whole slot-indexed batch (rows not listed keep valid rows too)
    return azmi_net_forward(net, dev_canonical, dev_v, dev_pi, max_rows, stream);
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

}  // extern "C"
