// The CONVEYOR: the Connect4-family leaf net (leafnet_c4.h: stem, pre-activation residual tower, both heads - neural_net.py:233-263,
// 448-510, 800-823) as a chain of weight-stationary wavefronts, for the asynchronous pipeline's net side (pipeline.hip, round 5).
//
// Why: the tile of leafnet_c4.h carries 3 or 6 boards through the whole net inside one workgroup - every wave re-reads all 4 KB of
// A fragments per k-step from an LDS ring that is refilled from L2 for every tile, one barrier per 8 KB chunk, stem and heads on
// a lone wave per SIMD: 0.45 of the MFMA peak alone on the chip, 0.31 beside the tree kernel.  Here a wavefront OWNS
// one 3x3 convolution for the whole epoch: its 72 A fragments sit in its registers (256 accumulator + 32 vector registers; gfx950:
// one 512-entry file per SIMD, MFMA A operands may be AGPRs), it streams n-tiles of 16 pixels through them - per k-step ONE
// ds_read_b128 (the B fragment) and four v_mfma_f32_16x16x32_bf16, no weight DMA, no chunk barrier, no workgroup barrier at all -
// and hands every finished n-tile to the next layer's wave through a ring in LDS (next CU: through HBM).  Measured alone
// (scripts/micro/conv_stationary.hip, profiles/r5_conv_stationary_gate.txt): 83 % matrix issue, 1.80 PFLOP/s chip-wide.
//
//   service workgroup (k_cv_service)           LINE = depth / 2 conv workgroups (k_cv_line), one per CU, 4 waves = 4 layers
//   wave "stem" : request ring -> 3-board GROUPS  ---X0--->  [conv1 conv2 conv1 conv2] ---X1---> [...] ---X2---> [...] ---X3--+
//   wave "head" : 1x1 conv, value / policy FCs, softmax, result granules, answer table, READY tokens  <------------------------+
//
// * The pixel stream: a GROUP = 3 boards = 126 pixels = 8 n-tiles (2 padding pixels); groups follow each other without gaps.  A
//   wave may start n-tile k once its producer has finished k + 1 (the 3x3 halo; the last n-tile of a group needs nothing beyond the
//   group): a layer lags its producer by about two n-tiles, not by a tile of boards.
// * Rings: activations as eight 8-channel bf16 planes of [pixel slot][16 B] (the tile's format), 8 n-tiles per ring, guard cells at
//   both ends (copies of the other end's pixels: a tap read never wraps) and 16 all-zero cells for taps that leave the board; the
//   residual stream crosses from a block's conv2 wave to the next block's conv2 wave as fp32 accumulator images (in LDS inside a
//   workgroup, in HBM between workgroups).  Counters "n-tiles present" / "n-tiles consumed" per ring, in LDS (same-wave LDS
//   operations execute in order: data, then counter - no wait) or in HBM (sc1 stores, drained before the counter moves).
// * Arithmetic: the SAME MFMAs on the same operands in the same order per accumulator as leafnet_c4.h's tile, the same epilogue
//   expressions: bit-identical answers (tests/test_gpu_conveyor.py), so every fixture and parity tier carries over.
#pragma once
#include "../../alphazero-pybind11_amd/csrc/leafnet_c4.h"
#include "../../alphazero-pybind11_amd/csrc/pipe_types.h"

namespace azmi_net_dev {
namespace cv {

using c4::PIX; using c4::BW; using c4::BH;
constexpr int GT = 8;                          // n-tiles per group
constexpr int GB = 3;                          // boards per group
constexpr int GPIX = GB * PIX;                 // 126 real pixels of a group's 128
constexpr int RNT = 8;                         // n-tiles of an LDS ring
constexpr int RPIX = RNT * 16;
constexpr int GUARD = 8;
constexpr int PLANE = (GUARD + RPIX + GUARD + 16) * 16;       // 2,560 B = 10 x 256: conflict-free B-fragment reads
constexpr int ZERO_OFF = (GUARD + RPIX + GUARD) * 16;
constexpr int RING_BYTES = 8 * PLANE;                          // 20,480
constexpr int RES_BYTES = RNT * 4096;                          // residual ring: fp32 accumulator images [n-tile][m tile][lane][16 B]
constexpr int LDS_RINGS = 0, LDS_RES = 4 * RING_BYTES, LDS_TAPS = LDS_RES + RES_BYTES, LDS_CTR = LDS_TAPS + 512;
constexpr int LDS_AFF = LDS_CTR + 128;                         // a1 | b1 of the block a conv2 wave feeds: [2 waves][2][64] floats
constexpr int LDS_RESIN = LDS_AFF + 1024;                      // the residual images that arrive from the workgroup before (LDS-DMA by wave 0)
constexpr int LINE_LDS_BYTES = LDS_RESIN + RES_BYTES;          // 149,120: one conv workgroup per CU
static_assert(PLANE % 256 == 0, "plane stride");
// cross-workgroup ring (HBM): a header (all headers in one block: one memset per epoch), bf16 plane images [XNT][2048], fp32
// accumulator images [XNT][4096]
constexpr int XNT = 32;
constexpr int X_HDR = 512, X_T = XNT * 2048, X_S = XNT * 4096;
// header words (u32 index; each hot word on its own 128-byte line): n-tiles out | the stream has ended (1; 2 = with an error) |
// n-tiles the two consumers are done with (planes / stream; the last ring: the two head waves)
constexpr int XH_PROD = 0, XH_END = 1, XH_CONS_T = 32, XH_CONS_S = 64;
// group metadata (stem -> head), per line a ring of MGRP groups of 128 bytes: per board b the words 8 b + {0 slot, 1 sequence number,
// 2 / 3 position key, 4 answer-table entry}
constexpr int MGRP = 32, META_BYTES = 128;
constexpr uint32_t kSpinSleep = 8;

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

struct CvArgs {
  NetDesc nd;
  NetPtrs np;
  uint32_t* xh;               // [lines][nwg + 1][128] ring headers (zeroed before every epoch)
  uint8_t* xt;                // [lines][nwg + 1][X_T] plane images
  uint8_t* xs;                // [lines][nwg + 1][X_S] stream images
  uint32_t* meta;             // [lines][MGRP][32]
  uint32_t lines, nwg;
  uint32_t dbg_flags;         // timing experiments (wrong answers): bit 0 = wave 0 stages nothing from HBM
  uint32_t* err;              // PipeCtl::err (sticky)
  uint32_t* stop;             // PipeEpoch::stop
  unsigned long long cap_ticks;
  unsigned long long* stat;   // [0] conv-wave n-tiles, [1] of those drained (epilogue on its own), [2] stem groups, [3] boards, [4] head groups; 100 MHz ticks: [5..9) conv waves by role waiting for input, [9..13) for output room, [13] stems claiming, [14] stems waiting for room, [15] heads waiting, [16..20) conv-wave lifetimes by role, [20] stem, [21] head lifetimes
};

__device__ __forceinline__ size_t x_index(const CvArgs& a, uint32_t line, uint32_t b) { return static_cast<size_t>(line) * (a.nwg + 1u) + b; }
__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long wall100() { return __builtin_amdgcn_s_memrealtime(); }

// MFMAs whose A operand is pinned: "a" = an accumulator register (the compiler cannot move it into the vector file), "v" = vector
__device__ __forceinline__ void mfma_a(f32x4& acc, const u32x4& wa, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x4& acc, const u32x4& wv, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wv), "v"(b));
}
__device__ __forceinline__ void mfma_a_init(f32x4& acc, const u32x4& wa, const bf16x8& b, const f32x4& c) {
  // (s_nop: the initial value may have just been moved by a VALU instruction; inside asm the compiler pads no hazard)
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(wa), "v"(b), "v"(c));
}

// 64 lanes x 16 B from per-lane global addresses straight into LDS at the wave-uniform `dst_lds` (+ lane * 16), L1 bypassed (sc1: the
// bytes were written by another CU); no register destination - counted by the issuing wave's vmcnt like any load
__device__ __forceinline__ void dma16_sc1(const uint8_t* src_lane, uint32_t dst_lds_) {
  const uint32_t dst_lds = __builtin_amdgcn_readfirstlane(dst_lds_);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src_lane), "s"(dst_lds) : "memory");
}

// ---- one conv wavefront ------------------------------------------------------------------------------------------------------------
// ROLE = the wave's index in its workgroup = its layer there: 0 conv1 of block 2c (input staged from HBM), 1 conv2 of block 2c
// (residual from HBM), 2 conv1 of block 2c + 1, 3 conv2 of block 2c + 1 (residual from the LDS residual ring, output to HBM).
// LAST: the line's last workgroup (role 3 then writes only the stream: the heads take it from there).
template <int ROLE, bool LAST>
__device__ __forceinline__ void conv_wave(const CvArgs& a, uint8_t* const lds, const uint32_t line, const uint32_t cwg) {
  constexpr bool IN_X = ROLE == 0;                 // input planes staged from the cross-workgroup ring
  constexpr bool EPI_B = (ROLE & 1) != 0;          // conv2: the stream out (fp32) + relu(a1 s + b1) of the NEXT block as planes
  constexpr bool RES_L = ROLE == 1 || ROLE == 3;     // the stream comes in through an LDS ring: ROLE 1 the one wave 0 fills from HBM, ROLE 3 the one wave 1 writes
  constexpr bool OUT_X = ROLE == 3;
  constexpr int PD = 2;                            // B fragments are read this many k-steps ahead (three buffers: 18 k-steps leave them in place)
  const int lane = threadIdx.x & 63, col = lane & 15, quad = lane >> 4;
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((c4::lptr_t)lds));
  (void)lds0;
  uint8_t* const in = lds + LDS_RINGS + ROLE * RING_BYTES;
  uint8_t* const out = lds + LDS_RINGS + (ROLE < 3 ? ROLE + 1 : 0) * RING_BYTES;
  uint8_t* const res = lds + LDS_RES;                                   // written by ROLE 1, read by ROLE 3
  uint8_t* const resin = lds + LDS_RESIN;                                // filled by ROLE 0 (DMA), read by ROLE 1
  const uint8_t* const res_rd = ROLE == 1 ? resin : res;
  constexpr int RES_P = ROLE == 1 ? 0 : 2, RES_C = ROLE == 1 ? 9 : 8;    // counters: images present (= ring 0 staged / ring 2 written), images consumed
  const uint32_t resin_lds = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>((c4::lptr_t)resin)));
  (void)resin_lds; (void)res_rd;
  // (an LDS pointer by TYPE: through a generic pointer a volatile access compiles to flat_load / flat_store, which count on vmcnt AND
  // lgkmcnt - every look at a counter then drains all the fragment reads in flight)
  typedef volatile __attribute__((address_space(3))) uint32_t* lctr_t;
  const lctr_t ctr = (lctr_t)(lds + LDS_CTR);
  // counters: P[i] = ctr[i] n-tiles present in ring i (i = 0: staged by wave 0), C[i] = ctr[4 + i] n-tiles of ring i consumed by wave i,
  // ctr[8] = residual ring consumed (wave 3), ctr[12 + i] = END: ring i's producer has delivered its last n-tile
  const size_t xi = x_index(a, line, cwg);
  const uint8_t* const xin_s = a.xs + xi * X_S + lane * 16;
  (void)xin_s;
  uint32_t* const xin_h = a.xh + xi * 128;
  uint32_t* const xout_h = a.xh + (xi + 1) * 128;
  uint8_t* const xout_t = a.xt + (xi + 1) * X_T;
  const auto r_xin_t = __builtin_amdgcn_make_buffer_rsrc(a.xt + xi * X_T, 0, X_T, 0x00020000);
  const auto r_xout_s = __builtin_amdgcn_make_buffer_rsrc(a.xs + (xi + 1) * X_S, 0, X_S, 0x00020000);
  (void)r_xin_t; (void)r_xout_s; (void)xin_h; (void)xout_h; (void)xout_t; (void)res;

  // ---- the layer's weights and parameters ----------------------------------------------------------------------------------------
  const uint32_t block = 2u * cwg + (ROLE >> 1);
  const size_t block_stride = 3 * CH * sizeof(float) + 2 * static_cast<size_t>(18) * MT * WFRAG_BYTES;
  const uint8_t* const wsrc = a.np.blocks + block * block_stride + 3 * CH * sizeof(float) + (ROLE & 1) * (18 * MT * WFRAG_BYTES) + lane * 16;
  u32x4 WA[16][4], WV[2][4];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      u32x4 t = *reinterpret_cast<const u32x4*>(wsrc + (ks * 4 + mt) * WFRAG_BYTES);
      asm volatile("" : "=a"(WA[ks][mt]) : "0"(t));
    }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      u32x4 t = *reinterpret_cast<const u32x4*>(wsrc + ((16 + ks) * 4 + mt) * WFRAG_BYTES);
      asm volatile("" : "=v"(WV[ks][mt]) : "0"(t));
    }
  // conv1: the accumulators start from c1 (bn2 folded); conv2: a1 / b1 of the NEXT block (the planes it feeds) - kept in LDS and
  // read four channels at a time while the epilogue runs (32 registers less on the waves that also carry the residual images)
  f32x4 cinit[4];
  float* const aff = reinterpret_cast<float*>(lds + LDS_AFF + (ROLE >> 1) * 512);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    cinit[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (!EPI_B) {
      cinit[mt] = *reinterpret_cast<const f32x4*>(a.np.blocks + block * block_stride + (2 * CH + mt * 16 + quad * 4) * sizeof(float));
    } else if (!(LAST && ROLE == 3)) {
      const uint8_t* nb = a.np.blocks + (block + 1u) * block_stride;
      if (col == 0) {
        *reinterpret_cast<f32x4*>(aff + mt * 16 + quad * 4) = *reinterpret_cast<const f32x4*>(nb + (mt * 16 + quad * 4) * sizeof(float));
        *reinterpret_cast<f32x4*>(aff + CH + mt * 16 + quad * 4) = *reinterpret_cast<const f32x4*>(nb + (CH + mt * 16 + quad * 4) * sizeof(float));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  f32x4 eaA = {0.f, 0.f, 0.f, 0.f}, ebA = eaA, eaB = eaA, ebB = eaA;      // a1 / b1 of the m tile the epilogue is at (A: even m, B: odd m)
  auto aff_load = [&](int m) __attribute__((always_inline)) {
    if constexpr (EPI_B && !(LAST && ROLE == 3)) {
      if (m & 1) { eaB = *reinterpret_cast<const f32x4*>(aff + m * 16 + quad * 4); ebB = *reinterpret_cast<const f32x4*>(aff + CH + m * 16 + quad * 4); }
      else { eaA = *reinterpret_cast<const f32x4*>(aff + m * 16 + quad * 4); ebA = *reinterpret_cast<const f32x4*>(aff + CH + m * 16 + quad * 4); }
    }
  };

  // ---- per-lane geometry -------------------------------------------------------------------------------------------------------------
  const uint32_t lanebase = quad * PLANE + GUARD * 16 + col * 16 - 128;      // the lane's pixel of ring n-tile 0 in plane `quad`, minus the read immediates' bias
  uint32_t zb[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int off = (tap / 3 - 1) * BW + (tap % 3 - 1);
    zb[tap] = quad * PLANE + ZERO_OFF + ((col + off) & 15) * 16 - (off * 16 + 128);
  }
  const uint32_t* const mtab = reinterpret_cast<const uint32_t*>(lds + LDS_TAPS) + col;
  const uint32_t outbase = (quad >> 1) * PLANE + GUARD * 16 + col * 16 + (quad & 1) * 8;     // the lane's 8 bytes of ring n-tile 0, plane quad / 2 (+ 2 mt planes)
  const int dup_lo = col < 8 ? RPIX * 16 : 0, dup_hi = col >= 8 ? -RPIX * 16 : 0;                 // guard copies of ring n-tile 0's first / n-tile 7's last 8 pixels

  auto read_b = [&](const uint32_t (&addr)[9], int ks) __attribute__((always_inline)) -> bf16x8 {
    const int tap = ks >> 1, off = (tap / 3 - 1) * BW + (tap % 3 - 1);
    return *reinterpret_cast<const bf16x8*>(in + addr[tap] + (off * 16 + 128 + (ks & 1) * 4 * PLANE));
  };
  auto taps_all = [&](uint32_t k, uint32_t (&addr)[9]) __attribute__((always_inline)) {
    const uint32_t tm = mtab[(k & 7u) * 16u];
    const uint32_t pb = lanebase + (k & 7u) * 256u;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const uint32_t sel = (tm >> tap) & 1u ? 0xFFFFFFFFu : 0u;
      addr[tap] = tap == 4 ? pb : ((pb & sel) | (zb[tap] & ~sel));
    }
  };

  // ---- input side: how many n-tiles of my ring are there ---------------------------------------------------------------------------
  uint32_t staged = 0;                // ROLE 0: n-tiles staged into ring 0 (= ctr[0])
  uint32_t xprod = 0;                 // ROLE 0: the cross ring's producer count as last read
  bool stg_pending = false;           // ROLE 0: the planes of n-tile `staged` are on their way into `stg`
  u32x4 stg[2];
  auto present = [&]() __attribute__((always_inline)) -> uint32_t { if constexpr (IN_X) return staged; else return uni(ctr[ROLE]); };
  auto need_of = [&](uint32_t k) __attribute__((always_inline)) -> uint32_t { return (k & 7u) == 7u ? k + 1u : k + 2u; };
  // ROLE 0: one n-tile of planes from the cross ring into ring 0 (two 1 KB pieces: planes 0-3 and 4-7, this lane's pixel lane & 15)
  auto stage_load = [&](uint32_t k) __attribute__((always_inline)) {
    if constexpr (IN_X) {
      if (a.dbg_flags & 1u) return;
      // the stream's fp32 image straight into the residual-in ring (four 1 KB pieces, no registers), then the planes: loads return in
      // order, so when the planes are in `stg` the image has landed
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) dma16_sc1(xin_s + (k % XNT) * 4096u + mt * 1024u, resin_lds + (k & 7u) * 4096u + mt * 1024u);
#pragma unroll
      for (int i = 0; i < 2; ++i) stg[i] = __builtin_amdgcn_raw_buffer_load_b128(r_xin_t, (k % XNT) * 2048u + i * 1024u + lane * 16u, 0, 16);
    }
  };
  auto stage_write = [&](uint32_t k) __attribute__((always_inline)) {
    if constexpr (IN_X) {
      const uint32_t j = k & 7u;
      uint8_t* p = lds + LDS_RINGS + quad * PLANE + GUARD * 16 + (j * 16 + col) * 16;
      *reinterpret_cast<u32x4*>(p) = stg[0];
      *reinterpret_cast<u32x4*>(p + 4 * PLANE) = stg[1];
      if (j == 0u || j == 7u) {         // guard copies: ring n-tile 0's first / n-tile 7's last eight pixels
        const int dup = j == 0u ? dup_lo : dup_hi;
        if (dup != 0) { *reinterpret_cast<u32x4*>(p + dup) = stg[0]; *reinterpret_cast<u32x4*>(p + 4 * PLANE + dup) = stg[1]; }
      }
    }
  };

  // ---- output side -----------------------------------------------------------------------------------------------------------------------
  uint32_t xcons = 0;                 // ROLE 3: the cross ring's consumer count as last read
  auto xcons_load = [&]() __attribute__((always_inline)) -> uint32_t {
    return uni(min(ld_sc1(xout_h + XH_CONS_T), ld_sc1(xout_h + XH_CONS_S)));
  };
  // space for the output of n-tile k: ring slots (LDS: the consumer is past k - RNT + 2, guard copies included; HBM: past k - XNT)
  auto space_for = [&](uint32_t k, bool refresh) __attribute__((always_inline)) -> bool {
    if constexpr (OUT_X) {
      if (refresh) xcons = xcons_load();
      return static_cast<int32_t>(k - xcons) < XNT;
    } else {
      uint32_t c = ctr[4 + ROLE + 1];
      if constexpr (ROLE == 1) c = min(c, static_cast<uint32_t>(ctr[8]));
      return static_cast<int32_t>(k - uni(c)) <= RNT - 2;
    }
  };
  // the four values (channels mt * 16 + quad * 4 ..) of the lane's pixel of n-tile k, as two packed bf16 pairs, into the output planes
  auto store_act = [&](uint32_t k, int mt, uint32_t p0, uint32_t p1) __attribute__((always_inline)) {
    if constexpr (OUT_X) {
      if constexpr (!LAST) {
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        gu64* dst = (gu64*)(xout_t + (k % XNT) * 2048u + (mt * 2 + (quad >> 1)) * 256u + col * 16u + (quad & 1) * 8u);
        __hip_atomic_store(dst, static_cast<unsigned long long>(p0) | (static_cast<unsigned long long>(p1) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      const uint32_t j = k & 7u;
      uint8_t* p = out + outbase + j * 256u + mt * 2 * PLANE;
      *reinterpret_cast<u32x2*>(p) = u32x2{p0, p1};
      if (j == 0u || j == 7u) {         // guard copies: ring n-tile 0's first / n-tile 7's last eight pixels
        const int dup = j == 0u ? dup_lo : dup_hi;
        if (dup != 0) *reinterpret_cast<u32x2*>(p + dup) = u32x2{p0, p1};
      }
    }
  };
  auto store_res = [&](uint32_t k, int mt, const f32x4& v) __attribute__((always_inline)) {
    if constexpr (OUT_X) {
      u32x4 u; u[0] = __float_as_uint(v[0]); u[1] = __float_as_uint(v[1]); u[2] = __float_as_uint(v[2]); u[3] = __float_as_uint(v[3]);
      __builtin_amdgcn_raw_buffer_store_b128(u, r_xout_s, (k % XNT) * 4096u + mt * 1024u + lane * 16u, 0, 16);
    } else if constexpr (ROLE == 1) {
      *reinterpret_cast<f32x4*>(res + (k & 7u) * 4096u + mt * 1024u + lane * 16u) = v;
    }
  };
  constexpr int X_STORES = OUT_X ? (LAST ? 4 : 8) : 0;       // HBM stores of one n-tile's epilogue
  // n-tiles 0 .. k are out (LDS: the stores above are ahead of this one in the wave's LDS queue; HBM: the caller has drained them)
  auto publish = [&](uint32_t k) __attribute__((always_inline)) {
    if constexpr (OUT_X) st_sc1(xout_h + XH_PROD, k + 1u);
    else ctr[ROLE + 1] = k + 1u;
  };

  // ---- the residual stream as the accumulators' initial value -----------------------------------------------------------------------
  uint32_t rl = 0;                    // residual images requested so far (n-tiles 0 .. rl - 1)
  auto res_load = [&](uint32_t k, f32x4 (&r)[4]) __attribute__((always_inline)) {
    if constexpr (RES_L) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) r[mt] = *reinterpret_cast<const f32x4*>(res_rd + (k & 7u) * 4096u + mt * 1024u + lane * 16u);
    }
  };
  // is n-tile n's residual image there?  (ROLE 1: wave 0 has staged n - planes and image land together; ROLE 3: wave 1's count)
  auto res_there = [&](uint32_t n) __attribute__((always_inline)) -> bool { return static_cast<int32_t>(uni(ctr[RES_P]) - n) > 0; };

  // ---- epilogue of one n-tile (the tile's expressions: leafnet_c4.h, residual block loop) ----------------------------------------
  float ev[16]; uint32_t pk[8];
  auto epi_value = [&](const f32x4 (&acc)[4], int v) __attribute__((always_inline)) {          // value v = (mt, r) = (v / 4, v % 4)
    const int mt = v >> 2, r = v & 3;
    float x = acc[mt][r];
    if constexpr (EPI_B && !(LAST && ROLE == 3)) x = (mt & 1) ? eaB[r] * x + ebB[r] : eaA[r] * x + ebA[r];
    asm("v_max_f32 %0, 0, %1" : "=v"(ev[v]) : "v"(x));
  };
  auto epi_pack = [&](int pair) __attribute__((always_inline)) { asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[pair]) : "v"(ev[2 * pair]), "v"(ev[2 * pair + 1])); };
  auto epilogue_all = [&](uint32_t k, const f32x4 (&acc)[4]) __attribute__((always_inline)) {
    asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");           // the last MFMAs' results (asm: the compiler pads nothing)
#pragma unroll
    for (int v = 0; v < 16; ++v) { if ((v & 3) == 0) aff_load(v >> 2); epi_value(acc, v); }
#pragma unroll
    for (int p = 0; p < 8; ++p) epi_pack(p);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if constexpr (EPI_B) store_res(k, mt, acc[mt]);
      if constexpr (!(LAST && ROLE == 3)) store_act(k, mt, pk[2 * mt], pk[2 * mt + 1]);
    }
    if constexpr (OUT_X) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish(k);
  };

  // ---- the stream ----------------------------------------------------------------------------------------------------------------------
  const unsigned long long t_start = wall100();
  uint32_t k = 0;                       // next n-tile
  uint32_t addrA[9], addrB[9];          // tap bases of the even / odd n-tile in flight
  bf16x8 b[PD + 1];
  static_assert(18 % (PD + 1) == 0, "the fragment buffers stay in place from n-tile to n-tile");
  f32x4 accA[4], accB[4], rinA[4], rinB[4];
  unsigned long long n_body = 0, n_drain = 0, t_wait_in = 0, t_wait_out = 0, t_go = 0, n_go = 0;
  bool failed = false;

  // one n-tile: 72 MFMAs into `acc`; every gap carries at most a few other instructions (a lone wave issues in order): the B read of
  // k-step ks + PD, the epilogue of n-tile k - 1 (`accp`), the tap bases and first reads of n-tile k + 1, staging / residual traffic.
  // WITH_PREV: n-tile k - 1's epilogue is pending.  Returns: go straight on with k + 1 (its input is there, k's output has room).
  // rin = the residual image of n-tile k (and, once consumed, the buffer of k + 2's), rin_next = that of k + 1.
  auto body = [&](auto with_prev_tag, f32x4 (&acc)[4], f32x4 (&accp)[4], f32x4 (&rin)[4], f32x4 (&rin_next)[4], uint32_t (&addr)[9], uint32_t (&addr_n)[9]) __attribute__((always_inline)) -> bool {
    constexpr bool WITH_PREV = decltype(with_prev_tag)::value;
    uint32_t tm_next = 0, pb_n = 0, p_in = 0, xc_new = 0;
    uint32_t v_pin = 0, v_room = 0, v_room2 = 0, v_resp = 0, v_c9 = 0;      // counters, read a few k-steps before they are looked at
    bool room = false, xc_asked = false;
    (void)rin_next; (void)xc_new; (void)xc_asked; (void)v_pin; (void)v_room; (void)v_room2; (void)v_resp; (void)v_c9;
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
      const bf16x8 bc = b[ks % (PD + 1)];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (ks == 0) {
          if constexpr (EPI_B) mfma_a_init(acc[mt], WA[0][mt], bc, rin[mt]); else mfma_a_init(acc[mt], WA[0][mt], bc, cinit[mt]);
        } else if (ks < 16) mfma_a(acc[mt], WA[ks][mt], bc);
        else mfma_v(acc[mt], WV[ks - 16][mt], bc);
        // ---- the gap behind MFMA (ks, mt) ----
        if (mt == 0) {
          if (ks + PD < 18) b[(ks + PD) % (PD + 1)] = read_b(addr, ks + PD);
          else b[(ks + PD) % (PD + 1)] = read_b(addr_n, ks + PD - 18);
          if (ks == 5) {
            if constexpr (!IN_X) v_pin = ctr[ROLE];
            if constexpr (!OUT_X) v_room = ctr[4 + ROLE + 1];
            if constexpr (ROLE == 1) v_room2 = ctr[8];
            if constexpr (RES_L) v_resp = ctr[RES_P];
            if constexpr (IN_X) v_c9 = ctr[9];
          }
          if (ks == 9) { if constexpr (IN_X) p_in = staged; else p_in = uni(v_pin); }
          if constexpr (WITH_PREV) { if (ks == 2 || ks == 6 || ks == 10) aff_load((ks + 2) >> 2); }
          if (ks == 16) aff_load(0);
        } else if (mt == 1) {
          if constexpr (WITH_PREV) { if (ks < 16) epi_value(accp, ks); }
          if constexpr (IN_X) { if (ks == 16) { st_sc1(xin_h + XH_CONS_T, staged); st_sc1(xin_h + XH_CONS_S, staged); } }
          if constexpr (RES_L) { if (ks == 16) ctr[RES_C] = k + 1u; }
          if (ks == 17) ctr[4 + ROLE] = k;           // n-tiles before k are no longer read (k - 1's last reads are long out)
        } else if (mt == 2) {
          if constexpr (WITH_PREV) { if (ks < 16 && (ks & 1)) epi_pack(ks >> 1); }
          if constexpr (IN_X) {
            // one more n-tile into ring 0 per body, in three steps a few hundred cycles apart: the planes requested in the last body
            // land in the ring, the producer's count is read, the next n-tile's planes are requested
            if (ks == 6) { if (stg_pending) { stage_write(staged); staged += 1u; ctr[0] = staged; stg_pending = false; } }
            if (ks == 0) { if (static_cast<int32_t>(xprod - staged) <= 1) xprod = uni(ld_sc1(xin_h + XH_PROD)); }
            if (ks == 7) { if (static_cast<int32_t>(xprod - staged) > 0 && static_cast<int32_t>(staged - k) <= RNT - 2 && static_cast<int32_t>(staged - uni(v_c9)) < RNT) { stage_load(staged); stg_pending = true; } }
          }
          if constexpr (OUT_X) {
            if (ks == 0) { if (static_cast<int32_t>(k - xcons) >= XNT / 2) { xc_new = xcons_load(); xc_asked = true; } }
          }
        } else {
          constexpr int what_of[18] = {-1, -2, 0, -3, 1, 2, 3, -3, 5, 6, 7, -3, 8, -5, -4, -3, -4, -4};
          const int what = what_of[ks];
          if (what == -1) tm_next = mtab[((k + 1u) & 7u) * 16u];
          else if (what == -2) { pb_n = lanebase + ((k + 1u) & 7u) * 256u; addr_n[4] = pb_n; }
          else if (what == -3) {
            if constexpr (WITH_PREV) {
              const int m = ks >> 2;
              if constexpr (EPI_B) store_res(k - 1u, m, accp[m]);
              if constexpr (!(LAST && ROLE == 3)) store_act(k - 1u, m, pk[2 * m], pk[2 * m + 1]);
              if (ks == 15) {
                if constexpr (OUT_X) {
                  // write-through stores take longer than an n-tile: the count moves three n-tiles behind the stores - everything but
                  // the stores (and counts) of this body and the two before it is done, so n-tile k - 4 is whole
                  if constexpr (X_STORES == 8) asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                  if (k >= 4u) publish(k - 4u);
                } else publish(k - 1u);
              }
            }
          } else if (what == -5) {
            if constexpr (OUT_X) { if (xc_asked) xcons = xc_new; room = space_for(k, false); }
            else {
              uint32_t c = v_room;
              if constexpr (ROLE == 1) c = min(c, v_room2);
              room = static_cast<int32_t>(k - uni(c)) <= RNT - 2;
            }
          } else if (what >= 0) {
            uint32_t sel;
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(tm_next), "i"(what));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(addr_n[what]) : "v"(sel), "v"(pb_n), "v"(zb[what]));
          }
          // the residual image of the next n-tile
          if constexpr (RES_L) { if (ks == 14) { if (rl == k + 1u && static_cast<int32_t>(uni(v_resp) - rl) > 0) { res_load(rl, rin_next); rl += 1u; } } }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return room && static_cast<int32_t>(p_in - need_of(k + 1u)) >= 0;
  };

  for (;;) {
    // ---- wait for n-tile k's input (and stage, ROLE 0) ---------------------------------------------------------------------------
    bool end = false;
    const unsigned long long tw0 = wall100();
    for (uint32_t spins = 0;; ++spins) {
      if constexpr (IN_X) {
        if (stg_pending) { stage_write(staged); staged += 1u; stg_pending = false; }
        xprod = uni(ld_sc1(xin_h + XH_PROD));
        while (static_cast<int32_t>(xprod - staged) > 0 && static_cast<int32_t>(staged - k) <= RNT - 2 && static_cast<int32_t>(staged - uni(ctr[9])) < RNT) {
          stage_load(staged); stage_write(staged); staged += 1u;
        }
        ctr[0] = staged;
        st_sc1(xin_h + XH_CONS_T, staged); st_sc1(xin_h + XH_CONS_S, staged);
      }
      const uint32_t p = present();
      if (static_cast<int32_t>(p - need_of(k)) >= 0) break;
      // the producer has delivered its last n-tile: what is there is all there will be (a group is always whole: p == k then)
      bool e;
      if constexpr (IN_X) e = uni(ld_sc1(xin_h + XH_END)) != 0u && uni(ld_sc1(xin_h + XH_PROD)) == staged; else e = uni(ctr[12 + ROLE]) != 0u && uni(ctr[ROLE]) == p;
      if (e) { end = static_cast<int32_t>(p - k) <= 0; break; }
      if ((spins & 63u) == 63u) {
        if (uni(ld_sc1(a.err)) != 0u) { failed = true; break; }
        if (wall100() - t_start > 8u * a.cap_ticks) { atomicOr(a.err, 64u); st_sc1(a.stop, 1u); failed = true; break; }
      }
      __builtin_amdgcn_s_sleep(kSpinSleep);
    }
    t_wait_in += wall100() - tw0;
    if (end || failed) break;
    // ---- prologue: tap bases and the first B fragments of n-tile k, the residual image ---------------------------------------------
    if (k & 1u) { taps_all(k, addrB);
#pragma unroll
      for (int i = 0; i < PD; ++i) b[i] = read_b(addrB, i);
    } else { taps_all(k, addrA);
#pragma unroll
      for (int i = 0; i < PD; ++i) b[i] = read_b(addrA, i);
    }
    if constexpr (RES_L) { if (rl == k) { if (k & 1u) res_load(k, rinB); else res_load(k, rinA); rl = k + 1u; } }
    // ---- run: n-tile after n-tile while the input is there and the output has room -----------------------------------------------
    bool go = (k & 1u) ? body(std::false_type{}, accB, accA, rinB, rinA, addrB, addrA) : body(std::false_type{}, accA, accB, rinA, rinB, addrA, addrB);
    n_body += 1;
    while (go) {
      const unsigned long long tg0 = (a.dbg_flags & 2u) ? wall100() : 0ull;
      k += 1u;
      if constexpr (RES_L) { if (rl == k) { if (k & 1u) res_load(k, rinB); else res_load(k, rinA); rl = k + 1u; } }      // (not read ahead: an LDS round trip in the open, rare)
      go = (k & 1u) ? body(std::true_type{}, accB, accA, rinB, rinA, addrB, addrA) : body(std::true_type{}, accA, accB, rinA, rinB, addrA, addrB);
      n_body += 1;
      if (a.dbg_flags & 2u) { t_go += wall100() - tg0; n_go += 1; }
    }
    // ---- drain: n-tile k's epilogue on its own ------------------------------------------------------------------------------------
    const unsigned long long tw1 = wall100();
    for (uint32_t spins = 0; !space_for(k, true); ++spins) {
      if ((spins & 63u) == 63u) {
        if (uni(ld_sc1(a.err)) != 0u) { failed = true; break; }
        if (wall100() - t_start > 8u * a.cap_ticks) { atomicOr(a.err, 64u); st_sc1(a.stop, 1u); failed = true; break; }
      }
      __builtin_amdgcn_s_sleep(kSpinSleep);
    }
    t_wait_out += wall100() - tw1;
    if (failed) break;
    if (k & 1u) epilogue_all(k, accB); else epilogue_all(k, accA);
    n_drain += 1;
    ctr[4 + ROLE] = k;               // (n-tile k itself is still the halo of k + 1)
    k += 1u;
  }
  // ---- the end of the stream goes down the line ---------------------------------------------------------------------------------------
  if constexpr (OUT_X) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_sc1(xout_h + XH_PROD, k); st_sc1(xout_h + XH_END, failed ? 2u : 1u); }
  else { ctr[ROLE + 1] = k; ctr[12 + ROLE + 1] = 1u; }
  ctr[4 + ROLE] = k;
  if constexpr (RES_L) ctr[RES_C] = k;
  if (lane == 0 && a.stat) {
    atomicAdd(a.stat + 0, n_body); atomicAdd(a.stat + 1, n_drain);
    atomicAdd(a.stat + 5 + ROLE, t_wait_in); atomicAdd(a.stat + 9 + ROLE, t_wait_out); atomicAdd(a.stat + 16 + ROLE, wall100() - t_start);
    atomicAdd(a.stat + 22 + ROLE, t_go); atomicAdd(a.stat + 26 + ROLE, n_go);
  }
}

template <bool LAST>
__device__ __forceinline__ void conv_wg(const CvArgs& a, uint8_t* lds, uint32_t line, uint32_t cwg) {
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  if (wave == 0) conv_wave<0, LAST>(a, lds, line, cwg);
  else if (wave == 1) conv_wave<1, LAST>(a, lds, line, cwg);
  else if (wave == 2) conv_wave<2, LAST>(a, lds, line, cwg);
  else conv_wave<3, LAST>(a, lds, line, cwg);
}

// grid = lines x nwg workgroups of 256 threads; one workgroup per CU (512 registers per lane, 115 KB of LDS)
__device__ __forceinline__ void line_wg(const CvArgs& a, uint8_t* const lds_cv) {
  const uint32_t line = blockIdx.x / a.nwg, cwg = blockIdx.x % a.nwg;
  // zero the rings (zero cells, guard cells), the counters; the tap masks [8 positions][16 columns]
  for (int i = threadIdx.x * 16; i < LINE_LDS_BYTES; i += 256 * 16) *reinterpret_cast<u32x4*>(lds_cv + i) = u32x4{0, 0, 0, 0};
  __syncthreads();
  if (threadIdx.x < 128) {
    const int pos = threadIdx.x >> 4, c = threadIdx.x & 15, gp = pos * 16 + c, p = gp % PIX, h = p / BW, x = p % BW;
    uint32_t m = 0;
    for (int tap = 0; tap < 9; ++tap) {
      const int hh = h + tap / 3 - 1, xx = x + tap % 3 - 1;
      if (gp < GPIX && hh >= 0 && hh < BH && xx >= 0 && xx < BW) m |= 1u << tap;
    }
    reinterpret_cast<uint32_t*>(lds_cv + LDS_TAPS)[threadIdx.x] = m;
  }
  __syncthreads();
  if (cwg + 1u == a.nwg) conv_wg<true>(a, lds_cv, line, cwg); else conv_wg<false>(a, lds_cv, line, cwg);
}


// =====================================================================================================================================
// The service side: per LINE one workgroup of three waves - "stem" (request ring -> groups -> the stem convolution -> X0), and two
// "head" waves that take the line's finished groups in turn (even / odd): 1x1 head convolution, value pool + FCs, policy FC,
// softmaxes (the tile's head section statement by statement, for one 3-board group on ONE wave), then the answers: result granules,
// answer-table granules, READY tokens - what k_pipe_net's tile did.
constexpr int SVC_THREADS = 192;
// LDS of a head wave: policy planes [8][144 slots][16 B] (slots 128.. = zero cells) | scratch 4 KB (the 1x1 convolution's operand planes,
// then the value channels of an n-tile for the pool; later the policy partials) | value hidden layer [3][256] | pooled [3][32] |
// logits [3][20]
constexpr int H_PLANE = 144 * 16, H_PLANES = 8 * H_PLANE, H_SCR = 4096, H_VH = GB * 256 * 4, H_POOL = GB * 32 * 4, H_LOG = GB * 20 * 4;
constexpr int H_OFF_SCR = H_PLANES, H_OFF_VH = H_OFF_SCR + H_SCR, H_OFF_POOL = H_OFF_VH + H_VH, H_OFF_LOG = H_OFF_POOL + H_POOL;
constexpr int HEAD_LDS = H_OFF_LOG + 256;                       // 26,368
constexpr int STEM_LDS = 256;
constexpr int SVC_LDS_BYTES = STEM_LDS + 2 * HEAD_LDS;          // 52,992: three service workgroups per CU

struct SvcPipe {               // what the service side needs of the pipeline (pipe_types.h), as plain pointers
  const unsigned long long* ring;      // request ring [kPipeRing][kReqGranules]
  uint32_t* head; uint32_t* tail;      // its free-running positions
  uint32_t* stop; uint32_t* tree_done; uint32_t* tree_arrived;
  unsigned long long* res;             // [S][kResStride] result granules
  unsigned long long* l0; uint32_t l0_mask;
  unsigned long long* rring; uint32_t rshift; uint32_t n_tree_wgs; uint32_t* wg_rtail;     // READY rings: rtail of workgroup w at wg_rtail[w * 32 + 1]
  unsigned long long* tiles; unsigned long long* tile_boards; uint32_t* lost; uint32_t* lost_total;
  uint32_t* dbg;
  uint32_t n_heads;                    // head waves per line: 1 or 2
  uint32_t census_hold; unsigned long long* t0; uint32_t* arrived; uint32_t* late_n; uint32_t* late;
};

// ---- stem ------------------------------------------------------------------------------------------------------------------------------
template <class KeyFn>
__device__ __forceinline__ void stem_wave(const CvArgs& a, const SvcPipe& sp, uint8_t* const sl, const uint32_t line, KeyFn key_of) {
  using namespace azmi;
  const int lane = threadIdx.x & 63, col = lane & 15, quad = lane >> 4;
  const size_t xi = x_index(a, line, 0);
  uint32_t* const xh = a.xh + xi * 128;
  uint8_t* const xt = a.xt + xi * X_T;
  const auto r_xs = __builtin_amdgcn_make_buffer_rsrc(a.xs + xi * X_S, 0, X_S, 0x00020000);
  unsigned long long* const bbs = reinterpret_cast<unsigned long long*>(sl);        // [0..4) stones of player 0, [4..8) of player 1 (board 3 = empty: padding pixels)
  uint32_t* const pls = reinterpret_cast<uint32_t*>(sl + 64);                         // [0..4) player to move
  // stem weights (2 k-steps x 4 m tiles), bias, block 0's a1 / b1
  bf16x8 wf[2][4];
  f32x4 bias[4], ea[4], eb[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) wf[ks][mt] = *reinterpret_cast<const bf16x8*>(a.np.stem_w + (ks * MT + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    bias[mt] = *reinterpret_cast<const f32x4*>(a.np.stem_b + mt * 16 + quad * 4);
    ea[mt] = *reinterpret_cast<const f32x4*>(a.np.blocks + (mt * 16 + quad * 4) * sizeof(float));
    eb[mt] = *reinterpret_cast<const f32x4*>(a.np.blocks + (CH + mt * 16 + quad * 4) * sizeof(float));
  }
  const unsigned long long t_start = wall100();
  constexpr uint64_t kPatienceTicks = 150;
  uint32_t w0 = 0, wn = 0, wdone = 0;       // the window: GB ring positions drawn with one fetch-add, served as their requests arrive
  uint32_t k = 0, grp = 0;                  // n-tiles, groups sent
  unsigned long long n_boards = 0, t_claim = 0, t_room = 0;
  bool failed = false;
  for (;;) {
    // ---- claim up to three requests (k_pipe_net's window rule) --------------------------------------------------------------------
    uint32_t n = 0, sl_ = 0xFFFFFFFFu, sq = 0, pl = 0;
    unsigned long long b0 = 0, b1 = 0;
    const unsigned long long tc0 = wall100();
    {
      uint64_t t_first = 0, t_empty = 0;
      uint32_t final_looks = 0;
      for (;;) {
        if (wdone == wn) {
          uint32_t h = 0;
          if (lane == 0) h = atomicAdd(sp.head, static_cast<uint32_t>(GB));
          w0 = uni(h); wn = GB; wdone = 0;
        }
        const uint32_t left = wn - wdone;
        bool here = false;
        if (static_cast<uint32_t>(lane) < left) {
          const uint32_t pos = w0 + wdone + lane;
          const unsigned long long want = pipe_lap_tag(pos);
          const unsigned long long* e = sp.ring + static_cast<size_t>(pos & (kPipeRing - 1u)) * kReqGranules;
          const unsigned long long a0 = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a1 = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                   a2 = __hip_atomic_load(e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a3 = __hip_atomic_load(e + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          here = (a0 >> 48) == want && (a1 >> 48) == want && (a2 >> 48) == want && (a3 >> 48) == want;
          if (here) {
            b0 = a0 & ((1ull << 48) - 1ull); b1 = a1 & ((1ull << 48) - 1ull);
            sl_ = static_cast<uint32_t>(a2 & 0xFFFFull); pl = static_cast<uint32_t>((a2 >> 16) & 1ull);
            sq = static_cast<uint32_t>(a3);
          }
        }
        const uint32_t hm = static_cast<uint32_t>(__ballot(here)) & 0xFFu;
        const uint32_t kk = static_cast<uint32_t>(__builtin_ctz(~hm));
        const uint64_t now = wall100();
        if (kk == left) { n = kk; break; }
        if (kk != 0u) {
          if (t_first == 0) t_first = now;
          if (now - t_first > kPatienceTicks) { n = kk; break; }
          continue;
        }
        uint32_t over = 0, stale = 0;
        if (lane == 0) {
          if (t_empty == 0) t_empty = now;
          if (now - t_empty > 50000ull && static_cast<int32_t>(ld_sc1(sp.tail) - (w0 + wdone)) > static_cast<int32_t>(kPipeRing / 2u)) stale = 1;
          if (ld_sc1(a.err)) over = 1;
          else if (ld_sc1(sp.stop) != 0u && ld_sc1(sp.tree_done) >= ld_sc1(sp.tree_arrived)) {
            const uint32_t t2 = ld_sc1(sp.tail);
            if (static_cast<int32_t>(t2 - (w0 + wdone)) <= 0) over = 1;
            else if (++final_looks >= 2u) stale = 1;
          }
          if (!over && now - t_start > 8u * a.cap_ticks) {
            if (atomicAdd(&sp.dbg[7], 1u) == 0u) {
              sp.dbg[8] = ld_sc1(sp.stop); sp.dbg[9] = ld_sc1(sp.tree_done); sp.dbg[10] = ld_sc1(sp.tree_arrived);
              sp.dbg[11] = ld_sc1(sp.tail); sp.dbg[12] = w0 + wdone; sp.dbg[13] = static_cast<uint32_t>((now - t_start) / 1000u);
            }
            atomicOr(a.err, static_cast<uint32_t>(kPipeErrNetTimeout)); st_sc1(sp.stop, 1u); over = 1;
          }
        }
        if (uni(over)) { n = 0; break; }
        if (uni(stale)) {
          if (lane == 0) { atomicAdd(sp.lost, left); atomicAdd(sp.lost_total, left); }
          wdone = wn;
          continue;
        }
        __builtin_amdgcn_s_sleep(16);
      }
    }
    t_claim += wall100() - tc0;
    if (n == 0u) break;
    wdone += n;
    // ---- the group's boards: packed positions into LDS, metadata for the head wave ------------------------------------------------
    if (lane < 4) {
      const bool mine = static_cast<uint32_t>(lane) < n;
      bbs[lane] = mine ? b0 : 0ull; bbs[4 + lane] = mine ? b1 : 0ull; pls[lane] = mine ? pl : 0u;
      if (lane < GB) {
        uint32_t* m = a.meta + (static_cast<size_t>(line) * MGRP + (grp % MGRP)) * 32 + lane * 8;
        unsigned long long k64 = 0; uint32_t l0e = 0;
        if (mine && sp.l0) { k64 = key_of(b0, b1, pl); l0e = pipe_l0_entry(k64, sp.l0_mask); }
        st_sc1(m + 0, mine ? sl_ : 0xFFFFFFFFu); st_sc1(m + 1, sq);
        st_sc1(m + 2, static_cast<uint32_t>(k64)); st_sc1(m + 3, static_cast<uint32_t>(k64 >> 32)); st_sc1(m + 4, l0e);
      }
    }
    if (lane == 0) { atomicAdd(sp.tiles, 1ull); atomicAdd(sp.tile_boards, static_cast<unsigned long long>(n)); }
    n_boards += n;
    // ---- room in X0 for eight more n-tiles ----------------------------------------------------------------------------------------------
    const unsigned long long tr0 = wall100();
    for (uint32_t spins = 0;; ++spins) {
      const uint32_t c = uni(min(ld_sc1(xh + XH_CONS_T), ld_sc1(xh + XH_CONS_S)));
      if (static_cast<int32_t>(k + GT - c) <= XNT) break;
      if ((spins & 63u) == 63u) {
        if (uni(ld_sc1(a.err)) != 0u) { failed = true; break; }
        if (wall100() - t_start > 8u * a.cap_ticks) { atomicOr(a.err, 64u); st_sc1(sp.stop, 1u); failed = true; break; }
      }
      __builtin_amdgcn_s_sleep(kSpinSleep);
    }
    t_room += wall100() - tr0;
    if (failed) break;
    // ---- the stem convolution, n-tile by n-tile (leafnet_c4.h, PIPE stem: the im2col operand straight from the stone bits) ---------
    for (uint32_t j = 0; j < GT; ++j, ++k) {
      const int gp = j * 16 + col;
      const int bd = (gp >= PIX) + (gp >= 2 * PIX) + (gp >= 3 * PIX);
      const int p = gp - bd * PIX, h = (p * 37) >> 8, w = p - h * BW;
      const unsigned long long s0 = bbs[bd], s1 = bbs[4 + bd];
      const uint32_t plr = pls[bd];
      const uint32_t one2 = plr == 0u ? 0x3F80u : 0u, one3 = plr == 1u ? 0x3F80u : 0u;
      auto tap_words = [&](int tap, uint32_t& e01, uint32_t& on) __attribute__((always_inline)) {       // (ci 0, ci 1) and (ci 2, ci 3) of one tap, as bf16 pairs
        const int dh = ((tap * 11) >> 5) - 1, dw = tap - 3 * ((tap * 11) >> 5) - 1;
        const int hh = h + dh, ww = w + dw;
        const bool ok = tap < 9 && bd < GB && hh >= 0 && hh < BH && ww >= 0 && ww < BW;
        const int q = ok ? hh * BW + ww : 0;
        const uint32_t v0 = ok ? static_cast<uint32_t>(s0 >> q) & 1u : 0u, v1 = ok ? static_cast<uint32_t>(s1 >> q) & 1u : 0u;
        e01 = v0 * 0x3F80u | v1 * 0x3F800000u;
        on = ok ? (one2 | one3 << 16) : 0u;
      };
      // k-step ks, lane quad: the 8-element plane entry pl = ks * 4 + quad = taps 2 pl, 2 pl + 1 (k = tap * 4 + ci; taps >= 9: zero)
      u32x4 bw[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int pl_ = ks * 4 + quad;
        uint32_t e0, n0, e1, n1;
        tap_words(2 * pl_, e0, n0);
        tap_words(2 * pl_ + 1, e1, n1);
        bw[ks] = u32x4{e0, n0, e1, n1};
      }
      f32x4 acc[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = bias[mt];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 bf; __builtin_memcpy(&bf, &bw[ks], 16);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][mt], bf, acc[mt], 0, 0, 0);
      }
      // the stream (fp32 image) and block 0's planes t = relu(a1 s + b1)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        u32x4 u; u[0] = __float_as_uint(acc[mt][0]); u[1] = __float_as_uint(acc[mt][1]); u[2] = __float_as_uint(acc[mt][2]); u[3] = __float_as_uint(acc[mt][3]);
        __builtin_amdgcn_raw_buffer_store_b128(u, r_xs, (k % XNT) * 4096u + mt * 1024u + lane * 16u, 0, 16);
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = static_cast<__bf16>(fmaxf(ea[mt][r] * acc[mt][r] + eb[mt][r], 0.0f));
        unsigned long long ob; __builtin_memcpy(&ob, &o, 8);
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        __hip_atomic_store((gu64*)(xt + (k % XNT) * 2048u + (mt * 2 + (quad >> 1)) * 256u + col * 16u + (quad & 1) * 8u), ob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // the count moves two n-tiles behind the stores (write-through stores take longer than a turn): everything but this turn's and
      // the last turn's stores (and the last count) is done
      asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
      if (k >= 1u) st_sc1(xh + XH_PROD, k - 1u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_sc1(xh + XH_PROD, k);
    grp += 1u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st_sc1(xh + XH_PROD, k);
  st_sc1(xh + XH_END, failed ? 2u : 1u);
  if (lane == 0 && a.stat) {
    atomicAdd(a.stat + 2, static_cast<unsigned long long>(grp)); atomicAdd(a.stat + 3, n_boards);
    atomicAdd(a.stat + 13, t_claim); atomicAdd(a.stat + 14, t_room); atomicAdd(a.stat + 20, wall100() - t_start);
  }
}

// ---- head --------------------------------------------------------------------------------------------------------------------------------
// which = this wave's turn among the line's n_heads head waves: it takes groups which, which + n_heads, ...
template <int MAXP1, int MAXM>
__device__ __forceinline__ void head_wave(const CvArgs& a, const SvcPipe& sp, uint8_t* const hl, const uint32_t line, const uint32_t which) {
  using namespace azmi;
  const int lane = threadIdx.x & 63, col = lane & 15, quad = lane >> 4;
  const size_t xi = x_index(a, line, a.nwg);
  uint32_t* const xh = a.xh + xi * 128;
  uint32_t* const my_cons = xh + (which == 0u ? XH_CONS_S : XH_CONS_T);
  const auto r_xs = __builtin_amdgcn_make_buffer_rsrc(a.xs + xi * X_S, 0, X_S, 0x00020000);
  uint8_t* const planes = hl;                                               // policy channels: planes 0-3 high parts, 4-7 low parts
  uint8_t* const scr = hl + H_OFF_SCR;
  float* const vh = reinterpret_cast<float*>(hl + H_OFF_VH);
  float* const vpool = reinterpret_cast<float*>(hl + H_OFF_POOL);
  float* const logits = reinterpret_cast<float*>(hl + H_OFF_LOG);
  const int P1 = a.nd.num_players + 1, M = a.nd.num_moves, Hd = a.nd.v_hidden;
  const int ntile = Hd >> 4;
  // zero cells of the policy planes (slots 128 .. 143) - and the two padding pixels' cells, which nothing writes
  for (int i = lane; i < 8 * 18; i += 64) *reinterpret_cast<u32x4*>(planes + (i / 18) * H_PLANE + (GPIX + i % 18) * 16) = u32x4{0, 0, 0, 0};
  // head 1x1 convolution: fragments, bias
  bf16x8 hw[2][4];
  f32x4 bh[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) hw[ks][mt] = *reinterpret_cast<const bf16x8*>(a.np.head_w + (ks * MT + mt) * WFRAG_BYTES + lane * 16);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) bh[mt] = *reinterpret_cast<const f32x4*>(a.np.head_b + mt * 16 + quad * 4);
  float w2[MAXP1][4], w2b[MAXP1];
#pragma unroll
  for (int o = 0; o < MAXP1; ++o) {
    w2b[o] = o < P1 ? a.np.v_fc2_b[o] : 0.0f;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) w2[o][kq] = (o < P1 && lane + 64 * kq < Hd) ? a.np.v_fc2_w[o * Hd + lane + 64 * kq] : 0.0f;
  }
  f32x4 pib4;         // policy bias of moves quad * 4 + r
#pragma unroll
  for (int r = 0; r < 4; ++r) pib4[r] = quad * 4 + r < M ? a.np.pi_fc_b[quad * 4 + r] : 0.0f;
  const uint32_t nh = sp.n_heads;
  if (nh == 1u && lane == 0) st_sc1(xh + XH_CONS_T, 0x7FFFFFF0u);          // (no second head wave: its counter never holds the producer back)
  if (lane == 0) st_sc1(my_cons, which * GT);                                 // I need nothing below my first group
  const unsigned long long t_start = wall100();
  unsigned long long n_groups = 0, t_hwait = 0;
  bool failed = false;
  for (uint32_t g = which;; g += nh) {
    const uint32_t k0 = g * GT;
    float pa_[4] = {0.f, 0.f, 0.f, 0.f};       // the pool's four running sums of (board in flight, channel lane & 31)
    bool over = false, have_next = false;
    f32x4 svn[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      const uint32_t k = k0 + j;
      // ---- wait for n-tile k of the stream; its image was requested while the n-tile before was worked on when it was there already,
      // and the next one's is requested now (an HBM round trip per n-tile in the open made a group take 19 us) ---------------------------
      const unsigned long long th0 = wall100();
      uint32_t pr = 0;
      for (uint32_t spins = 0;; ++spins) {
        pr = uni(ld_sc1(xh + XH_PROD));
        if (static_cast<int32_t>(pr - k) > 0) break;
        if (uni(ld_sc1(xh + XH_END)) != 0u && uni(ld_sc1(xh + XH_PROD)) == pr) { over = true; break; }
        if ((spins & 63u) == 63u) {
          if (uni(ld_sc1(a.err)) != 0u) { failed = true; break; }
          if (wall100() - t_start > 8u * a.cap_ticks) { atomicOr(a.err, 64u); st_sc1(sp.stop, 1u); failed = true; break; }
        }
        __builtin_amdgcn_s_sleep(kSpinSleep);
      }
      t_hwait += wall100() - th0;
      if (over || failed) break;
      // ---- h = relu(conv1x1(s) + bh): the stream's n-tile as bf16 operand planes (scratch), two k-steps x four m tiles ------------
      f32x4 sv[4];
      if (!have_next) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r_xs, (k % XNT) * 4096u + mt * 1024u + lane * 16u, 0, 16);
          sv[mt] = f32x4{__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
        }
      } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) sv[mt] = svn[mt];
      }
      have_next = j + 1 < GT && static_cast<int32_t>(pr - (k + 1u)) > 0;
      if (have_next) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r_xs, ((k + 1u) % XNT) * 4096u + mt * 1024u + lane * 16u, 0, 16);
          svn[mt] = f32x4{__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
        }
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = static_cast<__bf16>(sv[mt][r]);
        *reinterpret_cast<bf16x4*>(scr + (mt * 2 + (quad >> 1)) * 256 + col * 16 + (quad & 1) * 8) = o;
      }
      f32x4 hacc[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) hacc[mt] = bh[mt];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(scr + (ks * 4 + quad) * 256 + col * 16);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) hacc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hw[ks][mt], bfr, hacc[mt], 0, 0, 0);
      }
      // value channels (rows 0-31) of the 16 pixels as fp32 [pixel][32] into the scratch (behind the operand planes: 2 KB each);
      // policy channels (rows 32-63) into the policy planes as bf16 high parts (planes 0-3) and low parts (4-7)
      float* const vb = reinterpret_cast<float*>(scr + 2048);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaxf(hacc[mt][r], 0.0f);
        *reinterpret_cast<f32x4*>(vb + col * HC + mt * 16 + quad * 4) = o;
      }
#pragma unroll
      for (int mt = 2; mt < 4; ++mt) {
        bf16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = fmaxf(hacc[mt][r], 0.0f);
          hi[r] = static_cast<__bf16>(x);
          lo[r] = static_cast<__bf16>(x - static_cast<float>(hi[r]));
        }
        uint8_t* pp = planes + ((mt - 2) * 2 + (quad >> 1)) * H_PLANE + (j * 16 + col) * 16 + (quad & 1) * 8;
        *reinterpret_cast<bf16x4*>(pp) = hi;
        *reinterpret_cast<bf16x4*>(pp + 4 * H_PLANE) = lo;
      }
      // ---- the pool, in the tile's order: per (board, channel) four interleaved sums over p = 0 .. 41 (p % 4), then
      // ((a0 + a1) + (a2 + a3)) / 42.  Lanes 0-31 = the channel; the pixels of this n-tile one after the other ------------------------
      {
        const int c = lane & 31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int gp = j * 16 + i;
          if (gp >= GPIX) continue;
          const int bd = gp / PIX, p = gp % PIX;
          pa_[p & 3] += vb[i * HC + c];
          if (p == PIX - 1) {
            if (lane < 32) vpool[bd * HC + c] = ((pa_[0] + pa_[1]) + (pa_[2] + pa_[3])) / static_cast<float>(PIX);
            pa_[0] = pa_[1] = pa_[2] = pa_[3] = 0.0f;
          }
        }
      }
    }
    if (over || failed) break;
    if (lane == 0) st_sc1(my_cons, (g + nh) * GT);        // (every load of this group has returned: its values were used above)
    // ---- policy logits: 42 pixel positions x (lo * hi, hi * lo, hi * hi), summed in four runs (11, 11, 10, 10 positions) that are
    // added in order - the tile's four waves' partial tiles.  The 84 KB of weight fragments come from L2 six positions at a time, the
    // next six on their way while these are multiplied (the pointer is made opaque per group: hoisted out of the group loop the
    // fragments would sit in 336 registers) ---------------------------------------------------------------------------------------------
    f32x4 lg4;
    {
      const int bsrc0 = col < GB ? col * PIX * 16 : GPIX * 16 + col * 16;      // board `col`, or an all-zero cell
      const int bstep = col < GB ? 16 : 0;
      const uint8_t* wp = a.np.pi_fc_w + lane * 16;
      asm volatile("" : "+v"(wp));
      constexpr int PC = 6, NCH = PIX / PC;
      static_assert(PIX % PC == 0, "whole chunks");
      bf16x8 wA[PC][2], wB[PC][2];
      f32x4 run[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      auto wload = [&](int c, bf16x8 (&w)[PC][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PC; ++i) {
          w[i][0] = *reinterpret_cast<const bf16x8*>(wp + ((c * PC + i) * 2) * WFRAG_BYTES);
          w[i][1] = *reinterpret_cast<const bf16x8*>(wp + ((c * PC + i) * 2 + 1) * WFRAG_BYTES);
        }
      };
      auto wmul = [&](int c, const bf16x8 (&w)[PC][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PC; ++i) {
          const int pz = c * PC + i;
          const int rw = pz < 11 ? 0 : pz < 22 ? 1 : pz < 32 ? 2 : 3;
          const int bsrc = bsrc0 + pz * bstep;
          const bf16x8 xh_ = *reinterpret_cast<const bf16x8*>(planes + quad * H_PLANE + bsrc);
          const bf16x8 xl_ = *reinterpret_cast<const bf16x8*>(planes + (4 + quad) * H_PLANE + bsrc);
          run[rw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][1], xh_, run[rw], 0, 0, 0);
          run[rw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][0], xl_, run[rw], 0, 0, 0);
          run[rw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][0], xh_, run[rw], 0, 0, 0);
        }
      };
      wload(0, wA);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) { if (c & 1) wload(c + 1, wA); else wload(c + 1, wB); }
        __builtin_amdgcn_sched_barrier(0);
        if (c & 1) wmul(c, wB); else wmul(c, wA);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { float x = pib4[r]; x += run[0][r]; x += run[1][r]; x += run[2][r]; x += run[3][r]; lg4[r] = x; }
    }
    if (col < GB) {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (quad * 4 + r < M) logits[col * (MAXP1 + MAXM) + MAXP1 + quad * 4 + r] = lg4[r];
    }
    // ---- value fc1 on the exact-fp32 matrix pipe (boards = columns), four output tiles of 16 units at a time with the next four's
    // weights on their way; fc2 as the tile's lane sums + butterfly -------------------------------------------------------------------
    {
      float bq[2][4];
#pragma unroll
      for (int gq = 0; gq < 2; ++gq)
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) bq[gq][jq] = col < GB ? vpool[col * HC + gq * 16 + jq * 4 + quad] : 0.0f;
      const f32x4* w1 = reinterpret_cast<const f32x4*>(a.np.v_fc1_w) + lane;
      const float* b1 = a.np.v_fc1_b + quad * 4;
      asm volatile("" : "+v"(w1), "+v"(b1));
      constexpr int TC = 4;
      f32x4 fA[TC][3], fB[TC][3];
      auto floadt = [&](int t0, f32x4 (&f)[TC][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TC; ++i) {
          const int t = t0 + i < ntile ? t0 + i : 0;
          f[i][0] = w1[(t * 2 + 0) * 64]; f[i][1] = w1[(t * 2 + 1) * 64]; f[i][2] = *reinterpret_cast<const f32x4*>(b1 + t * 16);
        }
      };
      auto fmul = [&](int t0, const f32x4 (&f)[TC][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TC; ++i) {
          const int t = t0 + i;
          if (t < ntile) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f[i][0][jq], bq[0][jq], acc, 0, 0, 0);
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f[i][1][jq], bq[1][jq], acc, 0, 0, 0);
            if (col < GB) {
#pragma unroll
              for (int r = 0; r < 4; ++r) vh[col * 256 + t * 16 + quad * 4 + r] = fmaxf(acc[r] + f[i][2][r], 0.0f);
            }
          }
        }
      };
      floadt(0, fA);
      for (int t0 = 0; t0 < ntile; t0 += 2 * TC) {
        floadt(t0 + TC, fB);
        __builtin_amdgcn_sched_barrier(0);
        fmul(t0, fA);
        __builtin_amdgcn_sched_barrier(0);
        floadt(t0 + 2 * TC, fA);
        __builtin_amdgcn_sched_barrier(0);
        fmul(t0 + TC, fB);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int bq_ = 0; bq_ < GB; ++bq_) {
      float acc[MAXP1];
#pragma unroll
      for (int o = 0; o < MAXP1; ++o) acc[o] = 0.0f;
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        const float x = lane + 64 * kq < Hd ? vh[bq_ * 256 + lane + 64 * kq] : 0.0f;
#pragma unroll
        for (int o = 0; o < MAXP1; ++o) acc[o] += w2[o][kq] * x;
      }
#pragma unroll
      for (int o = 0; o < MAXP1; ++o) {
        if (o < P1) {
          float s_ = acc[o];
          for (int off = 32; off > 0; off >>= 1) s_ += __shfl_xor(s_, off, 64);
          if (lane == 0) logits[bq_ * (MAXP1 + MAXM) + o] = s_ + w2b[o];
        }
      }
    }
    // ---- softmaxes and the answers: one lane per output entry ------------------------------------------------------------------------
    const uint32_t* const meta = a.meta + (static_cast<size_t>(line) * MGRP + (g % MGRP)) * 32;
    if (lane < GB * (MAXP1 + MAXM)) {
      const int bq_ = lane / (MAXP1 + MAXM), kq = lane % (MAXP1 + MAXM);
      const bool is_v = kq < MAXP1;
      const int idx = is_v ? kq : kq - MAXP1, cnt = is_v ? P1 : M;
      const uint32_t slot = ld_sc1(meta + bq_ * 8 + 0);
      if (idx < cnt && slot != 0xFFFFFFFFu) {
        const float* lg = logits + bq_ * (MAXP1 + MAXM) + (is_v ? 0 : MAXP1);
        float mx = lg[0];
        for (int i = 1; i < cnt; ++i) mx = fmaxf(mx, lg[i]);
        float sum = 0.0f;
        for (int i = 0; i < cnt; ++i) sum += expf(lg[i] - mx);
        const float pr = expf(lg[idx] - mx) / sum;
        const uint32_t seq = ld_sc1(meta + bq_ * 8 + 1);
        const uint32_t gk = is_v ? kResV + idx : idx;
        __hip_atomic_store(sp.res + static_cast<size_t>(slot) * kResStride + gk, (static_cast<unsigned long long>(seq) << 32) | __float_as_uint(pr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sp.l0) {
          const unsigned long long key = static_cast<unsigned long long>(ld_sc1(meta + bq_ * 8 + 2)) | (static_cast<unsigned long long>(ld_sc1(meta + bq_ * 8 + 3)) << 32);
          __hip_atomic_store(sp.l0 + static_cast<size_t>(ld_sc1(meta + bq_ * 8 + 4)) * kResStride + gk,
                             (static_cast<unsigned long long>(pipe_l0_tag(key, gk)) << 32) | __float_as_uint(pr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // READY tokens of the answered slots, each into the ring of its slot's home workgroup (the tree wavefront that draws one checks
    // the granules' tags itself: no ordering behind the stores above)
    if (lane < GB) {
      const uint32_t slot = ld_sc1(meta + lane * 8 + 0);
      if (slot != 0xFFFFFFFFu) {
        const uint32_t seq = ld_sc1(meta + lane * 8 + 1);
        const uint32_t home = slot % sp.n_tree_wgs;
        const uint32_t pos = atomicAdd(sp.wg_rtail + home * 32u + 1u, 1u);
        __hip_atomic_store(sp.rring + ((static_cast<size_t>(home) << sp.rshift) + (pos & ((1u << sp.rshift) - 1u))),
                           (pipe_lap_tag_r(pos, sp.rshift) << 48) | static_cast<unsigned long long>(slot) | (static_cast<unsigned long long>(seq) << 16),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    n_groups += 1;
  }
  if (lane == 0) st_sc1(my_cons, 0x7FFFFFF0u);
  if (lane == 0 && a.stat) { atomicAdd(a.stat + 4, n_groups); atomicAdd(a.stat + 15, t_hwait); atomicAdd(a.stat + 21, wall100() - t_start); }
}

// grid = lines workgroups of 192 threads: wave 0 = the line's stem, waves 1 / 2 = its head waves
template <class KeyFn, int MAXP1, int MAXM>
__device__ __forceinline__ void service_wg(const CvArgs& a, const SvcPipe& sp, uint8_t* lds, KeyFn key_of) {
  const uint32_t line = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  if (wave == 0) stem_wave(a, sp, lds, line, key_of);
  else if (static_cast<uint32_t>(wave) <= sp.n_heads) head_wave<MAXP1, MAXM>(a, sp, lds + STEM_LDS + (wave - 1) * HEAD_LDS, line, static_cast<uint32_t>(wave - 1));
}

}  // namespace cv
}  // namespace azmi_net_dev
