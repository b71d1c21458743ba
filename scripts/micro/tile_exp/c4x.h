// Device code of the Connect4-family leaf net (k_leafnet_c4), shared by leafnet.hip (the plain launch behind azmi_net_forward*)
// and engine.hip (the fused net + move-step launch of a split round).  See leafnet.hip for the architecture it restates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#ifndef X_NRING
#define X_NRING 5
#endif
#ifndef X_NWV
#define X_NWV 4            // waves per workgroup: 8 = two per SIMD inside ONE tile (n-tiles per wave halved, 128 registers)
#endif
#ifndef X_STAMP
#define X_STAMP 0          // 1: wave 0 stamps s_memtime at the phase boundaries; the per-phase cycle sums of every tile go to stamp_out[tile][16]
#endif
#ifndef X_NOGUARD
#define X_NOGUARD 0        // 1: the conv epilogues store every lane's pixel (the 2-4 padding pixels of a tile too: bias-only values nobody reads) - no exec-mask region per store
#endif
#ifndef X_FMA
#define X_FMA 0            // 1: the pre-activation affine of an epilogue as ONE fma per value (the library is built with -ffp-contract=off: a multiply and an add)
#endif
#ifndef X_STAGGER
#define X_STAGGER 0        // shader cycles the second half of the grid (blockIdx >= gridDim / 2: with one round of 2 x CUs tiles, the CU partners of the first half) waits before it starts: deterministic anti-phase of the two workgroups of a CU
#endif
#ifndef X_SWP
#define X_SWP 1            // 1: fragments of k-step i + 1 read while the MFMAs of k-step i issue (double buffer); 0: read, wait, multiply
#endif
#ifndef X_SCHED
#define X_SCHED 0
#endif
#ifndef X_M0
#define X_M0 0
#endif
#ifndef X_PRMFAKE
#define X_PRMFAKE 0
#endif
namespace azmi_net_dev {

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
  const float* v_fc1_w;    // f32 A-fragments [v_hidden / 16][2][64 lanes][4] (hip_net._f32_frags)
  const float* v_fc1_b;    // [v_hidden]
  const float* v_fc2_w;    // [P+1][v_hidden]
  const float* v_fc2_b;    // [P+1]
  const uint8_t* pi_fc_w;  // per pixel position p: bf16 A-fragment of W_p[m][c] = W[m][c*H*W + p] (rows >= M zero), high parts, then low parts
  const float* pi_fc_b;    // [M]
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
#if X_STAMP
__device__ unsigned long long* g_stamp_out;
#define X_ST(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define X_ST(i) do {} while (0)
#endif
constexpr int NWV = X_NWV, NTH = 64 * NWV;      // waves, threads
constexpr int BH = 6, BW = 7, PIX = BH * BW;
constexpr int CHUNK_KS = 2;            // k-steps per weight chunk (= one 3x3 tap)
constexpr int CHUNK_BYTES = CHUNK_KS * MT * WFRAG_BYTES;   // 8,192
constexpr int PIECES = CHUNK_BYTES / (NWV * WFRAG_BYTES);  // 1 KB DMA pieces per wave and chunk: 2
constexpr int NRING = X_NRING;               // one chunk being read, one landed, two in flight, one being refilled
constexpr int RING_BYTES = NRING * CHUNK_BYTES;            // 40,960
constexpr int CHUNKS_PER_CONV = 18 / CHUNK_KS;             // 9
constexpr int MAXDEPTH = 6;
constexpr int PRM_FLOATS = X_PRMFAKE ? (CH + 2 * 3 * CH + CH) : (CH + MAXDEPTH * 3 * CH + CH);
constexpr int PRM_BLK(int blk) { return X_PRMFAKE ? (blk & 1) : blk; }
constexpr int PRM_HEAD = X_PRMFAKE ? (CH + 2 * 3 * CH) : (CH + MAXDEPTH * 3 * CH);    // stem bias | per block a1 b1 c1 | head bias
// Tile geometry.  Big: 6 boards (252 pixels = 16 n-tiles, 4 per wave, exactly) - the most boards per weight byte, for batches
// that fill the chip.  Small: 3 boards (126 pixels = 8 n-tiles, 2 per wave) - half the matrix work per k-step behind the same
// A-fragment reads, i.e. a shorter dependent chain per tile (the LDS, not the matrix pipe, then sets the k-step) for twice
// the workgroups: what the engine's rounds want, where a launch is a few hundred rows and its LATENCY is what the round
// waits for (at one workgroup per CU a 6-board tile takes 68 us, 55 of them matrix-pipe bound).
// SPLIT (the "bf16x3" precision tier): weights and activations are carried as bf16 HIGH + LOW parts (x = hi + lo to ~16 bits of
// mantissa) and every product runs as three MFMAs, hi*hi + hi*lo + lo*hi - the K dimension of each convolution tripled: per tap
// three 8 KB chunks [W_hi][W_hi][W_lo] against the activation planes [X_hi][X_lo][X_hi].  Twice the activation planes (16), so one
// workgroup per CU; everything else (ring, barriers, software pipeline) is the bf16 tile's.
template <int TBW_, int NTW_, int SPLIT_ = 0>
struct Tile {
  static constexpr int TBW = TBW_;                 // boards per workgroup
  static constexpr int NTW = NTW_;                 // n-tiles per wave
  static constexpr int SPLIT = SPLIT_;
  static constexpr int NPLANES = SPLIT_ ? 16 : 8;  // 8-channel activation planes (SPLIT: planes 8-15 hold the low parts)
  static constexpr int NPIX = TBW * PIX;           // GEMM columns in use
  static constexpr int NT = NWV * NTW;             // n-tiles
  static constexpr int ZSLOT = NT * 16;            // first all-zero cell
  static constexpr int SLOTS = ZSLOT + 16;
  static constexpr int PLANE = SLOTS * 16;         // a multiple of 256 B
  static constexpr int ZERO_OFF = ZSLOT * 16;
  static constexpr int ACT_BYTES = NPLANES * PLANE;
  static constexpr int ACT_OFF = 256;              // the planes start 256 B into the workgroup's LDS: a B-fragment read's base is the lane's pixel - 128 (conv_run)
  static constexpr int RING_OFF = ACT_OFF + ACT_BYTES;
  static constexpr int LDS_BYTES = RING_OFF + RING_BYTES + PRM_FLOATS * 4;   // big: 81,152 - two workgroups per CU
  static_assert(NPIX <= NT * 16 && NPIX <= NTH, "the tile's pixels fit its n-tiles; one thread per pixel");
  static_assert(PLANE % 256 == 0, "conflict-free fragment reads need a plane stride that is a multiple of 256 B");
  static_assert((SPLIT_ ? 1 : 2) * LDS_BYTES <= 160 * 1024, "two workgroups per CU (SPLIT: one)");
  static_assert(X_NRING < 5 || NPIX * HC * 4 + NWV * 256 * 4 <= RING_BYTES, "value-head scratch + policy partials reuse the ring");
};
using TileBig = Tile<6, 4>;
using TileSmall = Tile<3, 2>;
using TileBigX3 = Tile<6, 4, 1>;
using TileSmallX3 = Tile<3, 2, 1>;

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
// (dst = the LDS byte address as a number: a generic-to-LDS pointer cast per call carries a null check, four scalar instructions
// in the middle of the matrix stream)
__device__ __forceinline__ void dma16x2(const uint8_t* src_lane, uint32_t dst) {
#if X_M0
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
               "global_load_lds_dwordx4 %0, off offset:1024"
               :: "v"(src_lane), "s"(dst) : "memory", "m0");
#else
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "global_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src_lane), "s"(dst) : "memory");
#endif
}
__device__ __forceinline__ void dma16x1(const uint8_t* src_lane, uint32_t dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src_lane), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
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
// 6 = stem only, 7 = stem + trunk, 8 = ... + head 1x1 convolution, 9 = ... + policy FC partials and value pool, 10 = ... + value FCs
// `tile_index` = which tile of TBW rows this workgroup takes (the block index of the plain launch; the fused net + move-step
// launch of the engine passes its own); `lds` = the workgroup's LDS_BYTES of dynamic LDS.
// PIPE (the asynchronous pipeline's persistent net workgroups, pipeline.hip): the caller has already written the tile's raw
// input planes into their LDS staging area (ring slot 4: [board][CIN][PIX] floats) - there is no DMA from `canon` -, and the
// outputs leave as tagged result granules instead of rows: entry k of board b (pi entries first, then the value entries) goes to
// pio->res[pio->slot[b] * pio->stride + k] as ONE 8-byte agent-scope store {pio->seq[b] << 32 | float bits}; boards whose
// slot is 0xFFFFFFFF are padding.  pio->slot / pio->seq live in LDS behind the tile's own LDS_BYTES.
struct PipeIO {
  unsigned long long* res;
  const uint32_t* slot;
  const uint32_t* seq;
  uint32_t stride, v_first;
  // in-epoch answer table (pipe_types.h): every output entry also goes to l0[l0_entry[b] * stride + k] as {l0_tag(key[b], k) | float bits};
  // l0 == nullptr: off.  l0_entry / key live in LDS beside slot / seq.
  unsigned long long* l0;
  const uint32_t* l0_entry;
  const unsigned long long* key;
  // the tile's input, as the request carried it: stones of player 0 at bb[b], of player 1 at bb[8 + b] (bit = h * 7 + w), the player to
  // move at player[b] (connect4_gs.cc:131-149: planes 0 / 1 the stones, plane 2 + player all ones) - the stem builds its im2col
  // operand straight from these bits (round 4: no float planes staged in LDS, no reads of them: ~3 us of a 60 us tile)
  const unsigned long long* bb;
  const uint32_t* player;
};
__device__ __forceinline__ uint32_t pipe_io_tag(unsigned long long key, uint32_t k) {     // = pipe_l0_tag (pipe_types.h)
  unsigned long long x = key + 0x9E3779B97F4A7C15ULL * (k + 1u);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31;
  return static_cast<uint32_t>(x >> 32) | 1u;
}
template <class TG, int CIN, int MAXP1, int MAXM, int DBG = 0, bool PIPE = false>
__device__ __forceinline__ void tile(const NetDesc& nd, const NetPtrs& np, const float* __restrict__ canon,
                                     float* __restrict__ v_out, float* __restrict__ pi_out, uint32_t batch,
                                     const uint32_t* __restrict__ rows, const uint32_t* __restrict__ row_count,
                                     const uint32_t tile_index, uint8_t* const lds, const PipeIO* pio = nullptr) {
  static_assert(9 * CIN <= 64, "stem im2col fits one 64-row k-chunk");
  constexpr int TBW = TG::TBW, NTW = TG::NTW, NPIX = TG::NPIX, PLANE = TG::PLANE, ZERO_OFF = TG::ZERO_OFF, ACT_BYTES = TG::ACT_BYTES;
  // rows != nullptr: evaluate only the rows listed in rows[0 .. *row_count) (the engine's eval list: slots whose
  // pending leaf really needs the net); workgroups past the end of the list leave at once
  // (the list length and this thread's list entry are loaded together: one memory round trip, not two)
  const uint32_t max_rows = batch;
  uint32_t in_row = tile_index * TBW + threadIdx.x / (CIN * PIX / 4);
  if (in_row >= max_rows) in_row = max_rows - 1;
  if (rows) { batch = *row_count; in_row = rows[in_row]; }
  if (tile_index * TBW >= batch) return;
#if X_STAGGER
  #ifndef X_STAGGER_WHO
#define X_STAGGER_WHO 0      // which workgroups wait: 0 = the second half of the grid, 1 = odd block indices, 2 = odd (block index / 8) (XCD round-robin), 3 = the SIMD's second wave slot (HW_REG_HW_ID wave id bit 0)
#endif
  if (X_STAGGER_WHO == 0 ? blockIdx.x >= gridDim.x / 2 : X_STAGGER_WHO == 1 ? (blockIdx.x & 1u) != 0u : X_STAGGER_WHO == 2 ? ((blockIdx.x >> 3) & 1u) != 0u
                         : (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 1u) != 0u) { const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); while (__builtin_amdgcn_s_memtime() - t0_ < X_STAGGER) __builtin_amdgcn_s_sleep(4); }
#endif
  uint8_t* const act = lds + TG::ACT_OFF;
  uint8_t* const ring = lds + TG::RING_OFF;
  float* const prm = reinterpret_cast<float*>(lds + TG::RING_OFF + RING_BYTES);

  int tid_ = threadIdx.x;
  // a persistent caller runs this body in a loop: everything below that depends on the thread index alone would be hoisted out
  // of that loop and spilled (64 scratch stores in its preheader); an opaque copy keeps it where it is
  if constexpr (PIPE) asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63;
#if X_STAMP
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, quad = lane >> 4;
  const uint32_t board0 = tile_index * TBW;
  const int depth = nd.depth;
  constexpr bool SPLIT = TG::SPLIT != 0;
  constexpr int CPC = SPLIT ? 3 * CHUNKS_PER_CONV : CHUNKS_PER_CONV;     // weight chunks per 3x3 convolution
  constexpr int HEADCH = SPLIT ? 3 : 1;                                  // ... of the head 1x1 convolution
    const size_t block_stride = 3 * CH * sizeof(float) + 2 * static_cast<size_t>(CPC) * CHUNK_BYTES;

  // ---- small fp32 parameters -> LDS (plain loads, before any DMA is in flight) ------------------------------------
  if (tid < CH) { prm[tid] = np.stem_b[tid]; prm[PRM_HEAD + tid] = np.head_b[tid]; }
  if (tid < 3 * CH)
    for (int blk = 0; blk < depth; ++blk) prm[CH + PRM_BLK(blk) * 3 * CH + tid] = reinterpret_cast<const float*>(np.blocks + blk * block_stride)[tid];
  // source row of this thread's 16-byte piece of the tile's input planes
  // (boards past the end of the batch read whatever row their stale list entry names - any slot's planes - and are never stored)
  const int in_piece = tid % (CIN * PIX / 4);
  static_assert((CIN * PIX) % 4 == 0 && TBW * (CIN * PIX / 4) <= NTH, "one 16-byte input piece per thread");
  // ---- zero the activation planes once (covers the zero cells and the k rows the stem does not use) ----------------
  for (int i = tid * 16; i < ACT_BYTES; i += NTH * 16) *reinterpret_cast<u32x4*>(act + i) = u32x4{0, 0, 0, 0};

  // ---- weight stream: a chunk = 8 KB of fragments (one tap of one convolution; SPLIT: one of a tap's three passes); every wave
  // moves 2 of its 8 pieces.  The chunks of a convolution are consecutive in the blob, so are the two convolutions of a block;
  // 768 bytes of fp32 block parameters sit in front of every block's fragments and the head's fragments come last.  A convolution
  // therefore issues from two bases with compile-time offsets (conv_run: its own chunks, then the next convolution's) - the
  // bookkeeping of a running pointer cost ~25 scalar instructions per chunk in the middle of the matrix stream.
  const size_t wave_off = static_cast<size_t>(wave * PIECES) * WFRAG_BYTES;
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lptr_t)ring))) + static_cast<uint32_t>(wave_off);
  auto issue_chunk = [&](const uint8_t* chunk_src, int slot) {
    static_assert(PIECES == 2 || PIECES == 1, "one or two 1 KB pieces per wave");
    if constexpr (PIECES == 2) dma16x2(chunk_src + wave_off + lane * 16, ring_lds + static_cast<uint32_t>(slot) * CHUNK_BYTES);
    else dma16x1(chunk_src + wave_off + lane * 16, ring_lds + static_cast<uint32_t>(slot) * CHUNK_BYTES);
  };
  const uint8_t* const conv0_w = np.blocks + 3 * CH * sizeof(float);      // block 0, conv1
  // ring slots 3 and 4 first hold the stem's operands: 8 KB of stem fragments, and the tile's raw input planes
  uint8_t* const stem_w_lds = ring + (NRING - 2) * CHUNK_BYTES;
  float* const raw = reinterpret_cast<float*>(ring + (NRING - 1) * CHUNK_BYTES);
  static_assert(2 * MT * WFRAG_BYTES <= CHUNK_BYTES && TBW * CIN * PIX * 4 <= CHUNK_BYTES, "stem operands fit two ring slots");
  if constexpr (!PIPE) {
    if (tid < TBW * (CIN * PIX / 4))
      dma16(reinterpret_cast<const uint8_t*>(canon + static_cast<size_t>(in_row) * (CIN * PIX)) + in_piece * 16,
            reinterpret_cast<uint8_t*>(raw) + wave * 1024);
  }
#pragma unroll
  for (int i = 0; i < PIECES; ++i) dma16(np.stem_w + (wave * PIECES + i) * WFRAG_BYTES + lane * 16, stem_w_lds + (wave * PIECES + i) * WFRAG_BYTES);
#pragma unroll
  for (int i = 0; i < NRING - 2; ++i) issue_chunk(conv0_w + i * CHUNK_BYTES, i);

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
  wait_vm<PIECES * (NRING - 2)>();                      // input planes + stem fragments have landed (chunks 0-2 stay in flight)
  barrier_lds();
  if constexpr (PIPE) {
    static_assert(CIN == 4, "the packed-position stem is Connect4's: two stone planes + two player planes");
    if (tid < NPIX) {
      const int b = tid / PIX, p = tid % PIX, h = p / BW, w = p % BW;
      const unsigned long long s0 = pio->bb[b], s1 = pio->bb[8 + b];
      const uint32_t one2 = pio->player[b] == 0u ? 0x3F80u : 0u, one3 = pio->player[b] == 1u ? 0x3F80u : 0u;     // bf16 1.0 = 0x3F80
      // per tap: is the neighbour on the board, and the two stone bits there (k = tap * 4 + ci: two taps per 8-element plane entry)
      uint32_t e01[9], on[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
        const bool ok = hh >= 0 && hh < BH && ww >= 0 && ww < BW;
        const int q = ok ? hh * BW + ww : 0;
        const uint32_t v0 = ok ? static_cast<uint32_t>(s0 >> q) & 1u : 0u, v1 = ok ? static_cast<uint32_t>(s1 >> q) & 1u : 0u;
        e01[tap] = v0 * 0x3F80u | v1 * 0x3F800000u;      // (ci 0, ci 1) as two bf16
        on[tap] = ok ? (one2 | one3 << 16) : 0u;          // (ci 2, ci 3)
      }
#pragma unroll
      for (int pl = 0; pl < (9 * CIN + 7) / 8; ++pl) {
        u32x4 o;
        o[0] = e01[2 * pl]; o[1] = on[2 * pl];
        o[2] = 2 * pl + 1 < 9 ? e01[2 * pl + 1 < 9 ? 2 * pl + 1 : 0] : 0u; o[3] = 2 * pl + 1 < 9 ? on[2 * pl + 1 < 9 ? 2 * pl + 1 : 0] : 0u;
        *reinterpret_cast<u32x4*>(act + pl * PLANE + tid * 16) = o;
      }
    }
  } else if (tid < NPIX) {
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
  if constexpr (SPLIT) {
    // the input planes are 0 / 1 (exact in bf16): only the stem's weights have a low part.  Its fragments follow the high ones in
    // the blob and take their place in the ring slot once every wave has read those; one exposed DMA round trip per tile
    barrier_lds();
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(np.stem_w + 2 * MT * WFRAG_BYTES + (wave * 2 + i) * WFRAG_BYTES + lane * 16, stem_w_lds + (wave * 2 + i) * WFRAG_BYTES);
    wait_vm<0>();
    barrier_lds();
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
  }

  // epilogue helper: 4 consecutive channels (mt*16 + quad*4 ..) of the lane's pixel as bf16, into plane `plane0 + quad/2`
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

  // B-fragment addressing: a tap reads the lane's pixel + the tap offset when that neighbour is on the board, else an all-zero
  // cell with the 16-byte-slot residue of the cell the tap WOULD have read.  (Round 2 used the residue of the lane's OWN pixel:
  // beside lanes that read their shifted on-board neighbour that was a 2-way bank conflict wherever a lane group mixed the two -
  // every border pixel -, 28 % of the convolutions' LDS cycles in SQ_LDS_BANK_CONFLICT: profiles/r3_c4_tile_pmc.csv.)
  // The choice is made once per tap and tile - one literal add, one select - and serves both k-steps of the tap; the tap offset
  // rides in the read's immediate, biased by 128 so that it is never negative: 5.5 VALU per k-step instead of 19.
  const uint8_t* const po = act + quad * PLANE + pix0;     // the lane's pixel in plane `quad` (tile j: + j * 256, an immediate of the read)
  const uint8_t* const pom = po - 128;                     // on-board base of a tap read (act starts ACT_OFF = 256 B into LDS: never below 0)
  const uint8_t* const zcell = act + quad * PLANE + ZERO_OFF;   // first all-zero cell of the lane's plane
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
  // SPLIT: the same loop over three chunks per tap - chunk c = tap c / 3, pass c % 3 = (W_hi, X_hi), (W_hi, X_lo), (W_lo, X_hi).
  // ONE_BY_ONE: the head's 1x1 convolution as the centre tap alone (SPLIT only: its three chunks ride the same stream).
  // wbase: this convolution's first chunk; wnext: the first chunk of the convolution after it (chunk c + 4 is issued at chunk
  // c's barrier; past the end of the run the stream re-reads bytes behind the head's fragments into slots nobody reads).
  auto conv_run = [&](f32x4 (&acc)[NTW][MT], int slot0, const uint8_t* wbase, const uint8_t* wnext, auto nch_tag, auto one_tag) {
    constexpr int NCH = decltype(nch_tag)::value;          // weight chunks of this convolution
    constexpr bool ONE_BY_ONE = decltype(one_tag)::value;
    constexpr int NKS = NCH * CHUNK_KS;
    bf16x8 a[2][MT], b[2][NTW];
    int slot = slot0;
    auto load_a = [&](int ksl, const uint8_t* wsl, bf16x8 (&fa)[MT]) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = lds_read_frag(wsl + (ksl * MT + mt) * WFRAG_BYTES);
    };
    const uint8_t* sel[NTW];       // per tile j: the base the tap's reads go through (on the board: pom; else a zero cell minus the immediate)
    auto load_b = [&](int ks, bf16x8 (&fb)[NTW]) {
      const int chunk = ks >> 1, half = ks & 1;
      const int tap = ONE_BY_ONE ? 4 : (SPLIT ? chunk / 3 : chunk);
      const int lo_planes = (SPLIT && chunk % 3 == 1) ? 8 * PLANE : 0;     // pass 1 reads the activations' low parts
      const int tap_off = ((tap / 3 - 1) * BW + (tap % 3 - 1)) * 16;
      if (half == 0 && (!SPLIT || chunk % 3 == 0)) {       // a new tap
        int zs = (pix0 + tap_off) & 0xF0;       // slot residue of the cell this tap reads on the board (the same for every tile j)
        asm volatile("" : "+v"(zs));            // keeps the selects here: hoisted out of the block loop, a convolution's 36 bases cost 36 VGPRs (spills)
        const uint8_t* const zc = zcell + zs;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          sel[j] = ((tap_ok[j] >> tap) & 1u) ? pom : zc - (tap_off + 128 + j * 256);
      }
#pragma unroll
      for (int j = 0; j < NTW; ++j) fb[j] = lds_read_frag(sel[j] + (tap_off + 128 + j * 256 + half * 4 * PLANE + lo_planes));
    };
    load_a(0, wlane + slot * CHUNK_BYTES, a[0]);
    load_b(0, b[0]);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int cur = ks & 1, c = ks / CHUNK_KS, ksl = ks % CHUNK_KS;
      if (ksl == CHUNK_KS - 1) {            // open chunk c + 1 (the next convolution's first chunk when c is the last)
        if constexpr (DBG == 0 || DBG == 7) wait_vm<PIECES * (NRING - 3)>();      // chunk g + 1 has landed; g + 2 and g + 3 (2 pieces each) may still be in flight
        if (c == NCH - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (DBG != 2) { __builtin_amdgcn_s_barrier(); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG != 3)      // chunk c + 4 into the slot chunk c - 1 has left
          issue_chunk(c + (NRING - 1) < NCH ? wbase + (c + (NRING - 1)) * CHUNK_BYTES : wnext + (c + (NRING - 1) - NCH) * CHUNK_BYTES, slot == 0 ? NRING - 1 : slot - 1);
        slot = slot == NRING - 1 ? 0 : slot + 1;                 // ring slot of chunk c + 1
      } else {
        __builtin_amdgcn_sched_barrier(0);
      }
#if X_SWP
      if constexpr (DBG != 4) {
        if (ks + 1 < NKS) {
          load_a((ks + 1) % CHUNK_KS, wlane + slot * CHUNK_BYTES, a[cur ^ 1]);
          load_b(ks + 1, b[cur ^ 1]);
        }
      }
#else
      if (ks > 0) {      // (at the chunk's second k-step `slot` already names the next chunk's slot)
        const int slot_cur = (ksl == CHUNK_KS - 1) ? (slot == 0 ? NRING - 1 : slot - 1) : slot;
        load_a(ks % CHUNK_KS, wlane + slot_cur * CHUNK_BYTES, a[cur]); load_b(ks, b[cur]);
      }
#endif
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
        if (X_SWP && ks + 1 < NKS) {
          // issue order inside the k-step: 2 MFMAs and one A fragment of the next k-step (x4: its first MFMAs need all
          // four), then 2 MFMAs, the address arithmetic and the read of one B fragment (x4)
#if X_SCHED == 0
          constexpr int MF_A = NTW >= 4 ? 2 : 1, MF_B = (NTW * MT - 4 * MF_A) / NTW;
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
#elif X_SCHED == 1
          // one read per MFMA from the start of the k-step: A fragments first, then B (each B with its two address VALU)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#elif X_SCHED == 2
          // the B fragments first (their first MFMA row needs b0 and all four a), one read per MFMA
#pragma unroll
          for (int i = 0; i < NTW + 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#endif
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return slot;
  };
  auto conv3x3 = [&](f32x4 (&acc)[NTW][MT], int slot0, const uint8_t* wbase, const uint8_t* wnext) {
    return conv_run(acc, slot0, wbase, wnext, std::integral_constant<int, CPC>{}, std::false_type{});
  };

  X_ST(0);
  if constexpr (DBG == 6) { if (s[0][0][0] == 12345.678f) v_out[0] = s[1][1][1] + s[2][2][2] + s[3][3][3]; return; }   // timing: stem only
  int slot = 0;
  for (int blk = 0; blk < depth; ++blk) {
    const float* affine = prm + CH + PRM_BLK(blk) * 3 * CH;        // a1[64] b1[64] c1[64]
    if (blk == 0) barrier_lds();                          // the stem's reads of the planes (later blocks: the convolution's last barrier)
    // t = relu(a1 * s + b1) -> act
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(affine + mt * 16 + quad * 4);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(affine + CH + mt * 16 + quad * 4);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        if (!X_NOGUARD && !((real_m >> j) & 1u)) continue;
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = X_FMA ? fmaxf(__builtin_fmaf(a1[r], s[j][mt][r], b1[r]), 0.0f) : fmaxf(a1[r] * s[j][mt][r] + b1[r], 0.0f);
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
    if (blk == 0) { if constexpr (DBG == 0 || DBG == 7) wait_vm<PIECES * (NRING - 3)>(); }      // chunk 0 (chunks 1 and 2 stay in flight)
    barrier_lds();                                        // barrier C: planes visible (and, block 0, chunk 0 landed for every wave)
    if (blk == 0) { if constexpr (DBG != 3) issue_chunk(conv0_w + (NRING - 2) * CHUNK_BYTES, NRING - 2); }  // the stem is done with slot 3 (slot 4: chunk 4, at the first chunk barrier)
    X_ST(1);
    const uint8_t* const w1 = np.blocks + blk * block_stride + 3 * CH * sizeof(float);
    const uint8_t* const w2 = w1 + CPC * CHUNK_BYTES;
    const uint8_t* const wn = blk + 1 < depth ? w2 + CPC * CHUNK_BYTES + 3 * CH * sizeof(float) : np.head_w;
    slot = conv3x3(u, slot, w1, w2);
    X_ST(2);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        if (!X_NOGUARD && !((real_m >> j) & 1u)) continue;
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaxf(u[j][mt][r], 0.0f);
        store4(j, mt * 2, t);
      }
    barrier_lds();
    X_ST(3);
    // s = s + conv2(u)
    slot = conv3x3(s, slot, w2, wn);
    X_ST(4);
  }

  if constexpr (DBG == 7) { if (s[0][0][0] == 12345.678f) v_out[0] = s[1][1][1] + s[2][2][2] + s[3][3][3]; return; }   // timing: stem + trunk
  // ---- heads: h = relu(conv1x1(s) + bh), 64 rows = 32 value + 32 policy channels -------------------------------------
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
      if (X_NOGUARD || ((real_m >> j) & 1u)) store4(j, mt * 2, s[j][mt]);
  f32x4 hacc[NTW][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 bh = *reinterpret_cast<const f32x4*>(prm + PRM_HEAD + mt * 16 + quad * 4);
#pragma unroll
    for (int j = 0; j < NTW; ++j) hacc[j][mt] = bh;
  }
  if constexpr (SPLIT) {              // the head's three chunks ride the weight stream like a convolution's
    barrier_lds();                    // the stream's planes are visible
    slot = conv_run(hacc, slot, np.head_w, np.head_w, std::integral_constant<int, HEADCH>{}, std::true_type{});
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
  if constexpr (!SPLIT) {
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
  }
  X_ST(5);
  if constexpr (DBG == 8) { if (hacc[0][0][0] == 12345.678f) v_out[0] = hacc[1][1][1] + hacc[2][2][2]; return; }   // timing: ... + head 1x1 conv
  // (the value head's FC operands are requested now - the head accumulators are about to die - and land during the policy FC)
  // value fc1 on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32, the boards as the 16 columns; round 4 - the VALU form, one
  // hidden unit per thread, was 4.3 us of a lone wavefront per SIMD): this wave's output tiles wave, wave + 4, ... of 16 units, the
  // weights in A-fragment order [tile][k group of 16][lane][4] (element j of lane l = W[16 tile + (l & 15)][16 group + 4 j + (l >> 4)])
  constexpr int FT = 4;               // output tiles per wave: v_hidden <= 256
  const int ntile = Hd >> 4;
  f32x4 fa[FT][2], fb1[FT];
#pragma unroll
  for (int i = 0; i < FT; ++i) {
    const int t = wave + NWV * i, tt = t < ntile ? t : 0;
    fa[i][0] = reinterpret_cast<const f32x4*>(np.v_fc1_w)[(tt * 2 + 0) * 64 + lane];
    fa[i][1] = reinterpret_cast<const f32x4*>(np.v_fc1_w)[(tt * 2 + 1) * 64 + lane];
    fb1[i] = *reinterpret_cast<const f32x4*>(np.v_fc1_b + tt * 16 + quad * 4);
  }
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
  X_ST(6);
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
  X_ST(7);
  if constexpr (DBG == 9) { if (part[tid] == 12345.678f) v_out[0] = vpool[tid & 63]; return; }   // timing: ... + policy FC partials and the value pool
  if (tid < TBW * M) {
    const int b = tid / M, m = tid % M;
    float a = pib;
#pragma unroll
    for (int w = 0; w < NWV; ++w) a += part[(w * 16 + m) * 16 + b];
    logits[b * (MAXP1 + MAXM) + MAXP1 + m] = a;
  }
  {                          // value fc1: hidden[unit][board] = relu(W1 pooled + b1), 8 exact-fp32 MFMAs per tile of 16 units
#pragma unroll
    for (int i = 0; i < FT; ++i) {
      const int t = wave + NWV * i;
      if (t < ntile) {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float bq = col < TBW ? vpool[col * HC + g * 16 + j * 4 + quad] : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][g][j], bq, acc, 0, 0, 0);
          }
        if (col < TBW) {
#pragma unroll
          for (int r = 0; r < 4; ++r) vh[col * 256 + t * 16 + quad * 4 + r] = fmaxf(acc[r] + fb1[i][r], 0.0f);
        }
      }
    }
  }
  barrier_lds();
  X_ST(8);
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
  X_ST(9);
  if constexpr (DBG == 10) { if (logits[tid & 63] == 12345.678f) v_out[0] = 1.0f; return; }   // timing: ... + the value FCs
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
      if constexpr (PIPE) {
        const uint32_t sl = pio->slot[b];
        if (sl != 0xFFFFFFFFu) {
          const uint32_t gk = is_v ? pio->v_first + idx : idx;
          __hip_atomic_store(pio->res + static_cast<size_t>(sl) * pio->stride + gk,
                             (static_cast<unsigned long long>(pio->seq[b]) << 32) | __float_as_uint(pr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (pio->l0)
            __hip_atomic_store(pio->l0 + static_cast<size_t>(pio->l0_entry[b]) * pio->stride + gk,
                               (static_cast<unsigned long long>(pipe_io_tag(pio->key[b], gk)) << 32) | __float_as_uint(pr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        if (is_v) v_out[static_cast<size_t>(out_row) * P1 + idx] = pr;
        else pi_out[static_cast<size_t>(out_row) * M + idx] = pr;
      }
    }
  }
#if X_STAMP
  X_ST(10);
  if (tid == 0) for (int i = 0; i < 12; ++i) g_stamp_out[static_cast<size_t>(tile_index) * 16 + i] = st_acc[i];
#endif
}

template <class TG, int CIN, int MAXP1, int MAXM>
__global__ __launch_bounds__(NTH, 2) void k_leafnet_c4(NetDesc nd, NetPtrs np, const float* __restrict__ canon,
                                                        float* __restrict__ v_out, float* __restrict__ pi_out, uint32_t batch,
                                                        const uint32_t* __restrict__ rows, const uint32_t* __restrict__ row_count) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_c4[];
  tile<TG, CIN, MAXP1, MAXM>(nd, np, canon, v_out, pi_out, batch, rows, row_count, blockIdx.x, lds_c4);
}
}  // namespace c4

}  // namespace azmi_net_dev

// the net object behind the C ABI (leafnet.hip owns it; engine.hip reads nd / np / lds_bytes for the fused launch)
struct azmi_net_c4_view {
  azmi_net_dev::NetDesc nd;
  azmi_net_dev::NetPtrs np;
  size_t lds_bytes;
  int x3;                 // the bf16x3 tier (split bf16 operands, Tile<.., SPLIT>): one workgroup per CU; only the pipeline runs it through a view
};
// fills `out` when `net` is a Connect4-family net on the matrix cores (bf16, or bf16x3: out->x3 - the lock-step engine's fused
// launch takes the bf16 tile only and checks the flag); returns 0 otherwise
extern "C" int azmi_net_c4_view_get(const struct azmi_net* net, azmi_net_c4_view* out);
// allocates the per-stream scratch a forward of up to `max_rows` rows on `stream` needs (a forward allocates it on first use,
// which a stream capture does not allow); 0 = ok
extern "C" int azmi_net_reserve_stream(struct azmi_net* net, void* stream, uint32_t max_rows);
