// Gate micro-benchmark for the weight-stationary leaf-net conveyor: ONE wavefront per SIMD keeps the
// 72 A-fragments (288 registers: 256 accumulator registers + 32 vector registers) of one 3x3 64->64 convolution in its
// registers for the whole launch and streams n-tiles of 16 pixels through it: per k-step ONE ds_read_b128 (the B fragment) and four
// v_mfma_f32_16x16x32_bf16 - no weight DMA, no chunk barrier, no A-fragment reads.  Four such waves per workgroup (one per SIMD),
// each with its own layer and its own LDS rings, the shape a conveyor workgroup has; 3x3 taps through a ring with guard cells and
// zero cells, relu + bf16 + ds_write epilogue into the next layer's ring in the loop.
// Output: shader cycles per n-tile per wave (s_memtime around the whole stream) against the 72 x 16 = 1152 cycles of bare MFMA
// issue, and the launch's TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/conv_stationary.hip -o /tmp/cs && /tmp/cs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int RING_NT = 16;                 // n-tiles of a wave's input ring
constexpr int RING_PIX = RING_NT * 16;      // 256 pixel slots
constexpr int GUARD = 8;                    // guard cells at both ends (copies of the other end's pixels: a tap read never wraps)
constexpr int PLANE = (GUARD + RING_PIX + GUARD + 16) * 16;   // + 16 all-zero cells: 4608 B = 18 x 256
constexpr int ZERO_OFF = (GUARD + RING_PIX + GUARD) * 16;
constexpr int RING_BYTES = 8 * PLANE;       // 36,864 per wave
constexpr int BW = 7, BH = 6, PIX = 42;
static_assert(PLANE % 256 == 0, "conflict-free B-fragment reads");

__device__ __forceinline__ void mfma_a(f32x4& acc, const u32x4& wa, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x4& acc, const u32x4& wv, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wv), "v"(b));
}

// VARIANT 0: the compiler's own register allocation and schedule (builtin MFMAs, weights in a plain array)
// VARIANT 2: weights pinned in accumulator registers ("a" operands of asm MFMAs), B fragments PD k-steps ahead, epilogue behind the k-loop
// VARIANT 3: ... and the epilogue of n-tile i interleaved with the matrix stream of n-tile i + 1
template <int VARIANT, int PD>
__global__ __launch_bounds__(256, 1) void k_conv(const uint8_t* __restrict__ w, const unsigned long long* __restrict__ okmask, uint32_t ntiles,
                                                  unsigned long long* stamps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, quad = lane >> 4;
  uint8_t* const in = lds + wave * RING_BYTES;              // this wave's input ring
  uint8_t* const out = lds + ((wave + 1) & 3) * RING_BYTES;  // ... and the ring it writes (the next layer's input: here the next wave's, values do not matter)
  for (int i = tid * 16; i < 4 * RING_BYTES; i += 256 * 16) {
    const bool zero = (i % PLANE) >= ZERO_OFF;
    *reinterpret_cast<uint4*>(lds + i) = zero ? uint4{0, 0, 0, 0} : uint4{0x3c003c00u + ((unsigned)i * 2654435761u >> 28), 0x3f803c00u, 0x3c80bc00u, 0xbf003e00u};
  }
  if (tid < 128) {      // tap masks [8 positions][16 columns]
    const int pos = tid >> 4, c = tid & 15, gp = pos * 16 + c, p = gp % PIX, h = p / BW, x = p % BW;
    uint32_t m = 0;
    for (int tap = 0; tap < 9; ++tap) {
      const int hh = h + tap / 3 - 1, xx = x + tap % 3 - 1;
      if (gp < 3 * PIX && hh >= 0 && hh < BH && xx >= 0 && xx < BW) m |= 1u << tap;
    }
    reinterpret_cast<uint32_t*>(lds + 4 * RING_BYTES)[tid] = m;
  }
  __syncthreads();
  const f32x4 bias = {0.01f * col, -0.02f, 0.03f, 0.0f};
  const uint32_t in_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(in)), out_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(out));
  (void)in_lds; (void)out_lds;
  const uint32_t lanebase = quad * PLANE + GUARD * 16 + col * 16 - 128;        // the lane's pixel of n-tile 0 in plane `quad`, minus the immediates' bias
  const uint32_t outbase = (quad >> 1) * PLANE + GUARD * 16 + col * 16 + (quad & 1) * 8;
  uint32_t zb[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int off = (tap / 3 - 1) * BW + (tap % 3 - 1);
    zb[tap] = quad * PLANE + ZERO_OFF + ((col + off) & 15) * 16 - (off * 16 + 128);
  }
  unsigned long long c0 = 0;
  if constexpr (VARIANT == 0) {
    bf16x8 W[18][4];
#pragma unroll
    for (int ks = 0; ks < 18; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) W[ks][mt] = *reinterpret_cast<const bf16x8*>(w + ((wave * 18 + ks) * 4 + mt) * 1024 + lane * 16);
    c0 = __builtin_amdgcn_s_memtime();
    for (uint32_t nt = 0; nt < ntiles; ++nt) {
      const unsigned long long* om = okmask + (nt & 7) * 8;
      const uint32_t pb = lanebase + (nt & 15) * 256;
      f32x4 acc[4] = {bias, bias, bias, bias};
      uint32_t addr[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) addr[tap] = tap == 4 ? pb : ((om[tap < 4 ? tap : tap - 1] >> lane) & 1ull ? pb : zb[tap]);
      bf16x8 b[2];
      b[0] = *reinterpret_cast<const bf16x8*>(in + addr[0] + (-8 * 16 + 128));
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) {
        if (ks + 1 < 18) {
          const int tap = (ks + 1) >> 1, off = (tap / 3 - 1) * BW + (tap % 3 - 1);
          b[(ks + 1) & 1] = *reinterpret_cast<const bf16x8*>(in + addr[tap] + (off * 16 + 128 + ((ks + 1) & 1) * 4 * PLANE));
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[ks][mt], b[ks & 1], acc[mt], 0, 0, 0);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = static_cast<__bf16>(fmaxf(acc[mt][r], 0.0f));
        *reinterpret_cast<bf16x4*>(out + outbase + (nt & 15) * 256 + mt * 2 * PLANE) = o;
      }
    }
  } else {
    // the layer's 72 A fragments [k-step][m tile][lane][16 B]: k-steps 0-15 in accumulator registers, 16-17 in vector registers
    u32x4 WA[16][4], WV[2][4];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        u32x4 t = *reinterpret_cast<const u32x4*>(w + ((wave * 18 + ks) * 4 + mt) * 1024 + lane * 16);
        asm volatile("" : "=a"(WA[ks][mt]) : "0"(t));
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        u32x4 t = *reinterpret_cast<const u32x4*>(w + ((wave * 18 + 16 + ks) * 4 + mt) * 1024 + lane * 16);
        asm volatile("" : "=v"(WV[ks][mt]) : "0"(t));
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    c0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // B fragment of k-step ks of the n-tile whose tap bases are addr[]
    auto read_b = [&](const uint32_t (&addr)[9], int ks) -> bf16x8 {
      const int tap = ks >> 1, off = (tap / 3 - 1) * BW + (tap % 3 - 1);
      return *reinterpret_cast<const bf16x8*>(in + addr[tap] + (off * 16 + 128 + (ks & 1) * 4 * PLANE));
    };
    // per-lane tap masks (bit tap: the neighbour is on the board) of the 8 n-tile positions of a 3-board group: a table in LDS,
    // one ds_read_b32 per n-tile (in order with the fragment reads: counted waits stay counted - a scalar load would need lgkmcnt(0))
    const uint32_t* const mtab = reinterpret_cast<const uint32_t*>(lds + 4 * RING_BYTES) + col;
    uint32_t tm_next = mtab[0];
    auto taps = [&](uint32_t nt, uint32_t tm, uint32_t (&addr)[9]) {
      const uint32_t pb = lanebase + (nt & 15) * 256;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap == 4) { addr[tap] = pb; continue; }
        const uint32_t sel = static_cast<uint32_t>(__builtin_amdgcn_sbfe(tm, tap, 1));     // 0 / all ones
        addr[tap] = (pb & sel) | (zb[tap] & ~sel);
      }
    };
    auto epilogue_part = [&](const f32x4 (&a)[4], uint32_t nt, int mt) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) { float x; asm("v_max_f32 %0, 0, %1" : "=v"(x) : "v"(a[mt][r] * bias[r] + bias[(r + 1) & 3])); o[r] = static_cast<__bf16>(x); }
      *reinterpret_cast<bf16x4*>(out + outbase + (nt & 15) * 256 + mt * 2 * PLANE) = o;
    };
    uint32_t addr[9], addr_n[9];
    taps(0, tm_next, addr);
    bf16x8 b[PD + 1];
#pragma unroll
    for (int i = 0; i < PD; ++i) b[i] = read_b(addr, i);
    // one n-tile: its 72 MFMAs into `acc`, the epilogue of the n-tile before (accumulators `accp`) in between (VARIANT 3)
    auto body = [&](uint32_t nt, f32x4 (&acc)[4], f32x4 (&accp)[4]) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = bias;
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 1) tm_next = mtab[((nt + 1) & 7) * 16];
        if (ks == 12) taps(nt + 1, tm_next, addr_n);
        // the read PD k-steps ahead (into the next n-tile near the end of this one)
        if (ks + PD < 18) b[(ks + PD) % (PD + 1)] = read_b(addr, ks + PD);
        else b[(ks + PD) % (PD + 1)] = read_b(addr_n, ks + PD - 18);
        const bf16x8 bc = b[ks % (PD + 1)];
        if (ks < 16) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) mfma_a(acc[mt], WA[ks][mt], bc);
        } else {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) mfma_v(acc[mt], WV[ks - 16][mt], bc);
        }
        if constexpr (VARIANT == 3) {
          if (ks >= 2 && ks < 6) epilogue_part(accp, nt - 1, ks - 2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (VARIANT != 3) {
        asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) epilogue_part(acc, nt, mt);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) addr[tap] = addr_n[tap];
      // rotate the prefetched fragments so that buffer (ks % (PD + 1)) holds k-step ks of the next n-tile again: 18 % (PD + 1) places
      if constexpr (18 % (PD + 1) != 0) {
        bf16x8 t[PD + 1];
#pragma unroll
        for (int i = 0; i <= PD; ++i) t[i] = b[(i + 18) % (PD + 1)];
#pragma unroll
        for (int i = 0; i <= PD; ++i) b[i] = t[i];
      }
    };
    // VARIANT 4: every one of the 72 gaps between MFMAs carries at most two other instructions, in a fixed order (a lone wave issues in
    // order: VALU bunched behind four MFMAs starts when the fourth has issued and idles the matrix pipe meanwhile):
    //   gap 4 ks + 0: the B-fragment read of k-step ks + PD           gap 4 ks + 1: epilogue, value ks: a * x + b
    //   gap 4 ks + 2: epilogue, value ks: relu (+ the pair's bf16 pack)  gap 4 ks + 3: a tap base of the next n-tile / a store of the epilogue
    float ev[16];        // the epilogue's values in flight
    uint32_t pk[8];
    uint32_t pb_n = 0;
    auto gap = [&](uint32_t nt, const f32x4 (&accp)[4], int ks, int g) {
      if (g == 0) {
        if (ks + PD < 18) b[(ks + PD) % (PD + 1)] = read_b(addr, ks + PD);
        else b[(ks + PD) % (PD + 1)] = read_b(addr_n, ks + PD - 18);
      } else if (g == 1) {
        if (ks < 16) ev[ks] = accp[ks >> 2][ks & 3] * bias[ks & 3] + bias[(ks + 1) & 3];
      } else if (g == 2) {
        if (ks < 16) {
          asm("v_max_f32 %0, 0, %1" : "=v"(ev[ks]) : "v"(ev[ks]));
          if (ks & 1) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[ks >> 1]) : "v"(ev[ks - 1]), "v"(ev[ks]));
        }
      } else {
        constexpr int tap_of[18] = {-1, -2, 0, -3, 1, 2, 3, -3, 5, 6, 7, -3, 8, -4, -4, -3, -4, -4};
        const int what = tap_of[ks];
        if (what == -1) tm_next = mtab[((nt + 1) & 7) * 16];
        else if (what == -2) { pb_n = lanebase + ((nt + 1) & 15) * 256; addr_n[4] = pb_n; }
        else if (what == -3) {
          const int mt = ks >> 2;
          *reinterpret_cast<uint2*>(out + outbase + ((nt - 1) & 15) * 256 + mt * 2 * PLANE) = uint2{pk[mt * 2], pk[mt * 2 + 1]};
        } else if (what >= 0) {
          uint32_t sel;
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(tm_next), "i"(what));
          asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(addr_n[what]) : "v"(sel), "v"(pb_n), "v"(zb[what]));
        }
      }
    };
    auto body4 = [&](uint32_t nt, f32x4 (&acc)[4], f32x4 (&accp)[4]) {
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) {
        const bf16x8 bc = b[ks % (PD + 1)];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if (ks == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc[mt]) : "a"(WA[0][mt]), "v"(bc), "v"(bias));
          else if (ks < 16) mfma_a(acc[mt], WA[ks][mt], bc);
          else mfma_v(acc[mt], WV[ks - 16][mt], bc);
          gap(nt, accp, ks, mt);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) addr[tap] = addr_n[tap];
      if constexpr (18 % (PD + 1) != 0) {
        bf16x8 t[PD + 1];
#pragma unroll
        for (int i = 0; i <= PD; ++i) t[i] = b[(i + 18) % (PD + 1)];
#pragma unroll
        for (int i = 0; i <= PD; ++i) b[i] = t[i];
      }
    };
    f32x4 accA[4] = {bias, bias, bias, bias}, accB[4] = {bias, bias, bias, bias};
    for (uint32_t nt = 0; nt < ntiles; nt += 2) {
      if constexpr (VARIANT == 4) { body4(nt, accA, accB); body4(nt + 1, accB, accA); }
      else { body(nt, accA, accB); body(nt + 1, accB, accA); }
    }
    if constexpr (VARIANT == 3) {
      asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) epilogue_part(accB, ntiles - 1, mt);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) stamps[blockIdx.x * 4 + wave] = c1 - c0;
  if (sink && tid == 0) sink[0] = static_cast<float>(out[lane]);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VARIANT, int PD>
void run(const uint8_t* w, const unsigned long long* okmask, uint32_t wgs, uint32_t ntiles, const char* what) {
  const size_t ldsb = 4 * RING_BYTES + 512;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv<VARIANT, PD>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
  unsigned long long* st; CK(hipMalloc(&st, wgs * 4 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) k_conv<VARIANT, PD><<<wgs, 256, ldsb>>>(w, okmask, ntiles, st, nullptr);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k_conv<VARIANT, PD><<<wgs, 256, ldsb>>>(w, okmask, ntiles, st, nullptr);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(wgs * 4);
  CK(hipMemcpy(h.data(), st, wgs * 4 * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double cyc = double(h[h.size() / 2]) / ntiles;
  const double flops = double(wgs) * 4 * ntiles * 72 * 16384.0;
  printf("%-44s wgs %4u: %6.0f cycles per n-tile (median wave; 1152 = bare MFMA issue -> %5.1f %%), %8.1f us per launch, %6.0f TFLOP/s\n",
         what, wgs, cyc, 100.0 * 1152.0 / cyc, ms * 1e3 / reps, flops / (ms * 1e-3 / reps) * 1e-12);
  CK(hipFree(st));
}

int main() {
  std::vector<uint16_t> hw(size_t(4) * 72 * 512);
  for (auto& x : hw) x = static_cast<uint16_t>(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15));
  uint8_t* w; CK(hipMalloc(&w, hw.size() * 2)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  // lane masks "tap t of the lane's pixel is on the board" per position of an n-tile in its 3-board group of 8 n-tiles
  std::vector<unsigned long long> om(64, 0);
  for (int pos = 0; pos < 8; ++pos)
    for (int t8 = 0; t8 < 8; ++t8) {
      const int tap = t8 < 4 ? t8 : t8 + 1;
      for (int lane = 0; lane < 64; ++lane) {
        const int gp = pos * 16 + (lane & 15), p = gp % PIX, h = p / BW, x = p % BW;
        const int hh = h + tap / 3 - 1, xx = x + tap % 3 - 1;
        if (gp < 3 * PIX && hh >= 0 && hh < BH && xx >= 0 && xx < BW) om[pos * 8 + t8] |= 1ull << lane;
      }
    }
  unsigned long long* dom; CK(hipMalloc(&dom, 64 * 8)); CK(hipMemcpy(dom, om.data(), 64 * 8, hipMemcpyHostToDevice));
  for (uint32_t wgs : {1u, 256u}) {
    run<0, 1>(w, dom, wgs, 4096, "compiler-allocated weights");
    run<2, 2>(w, dom, wgs, 4096, "weights in AGPRs, reads 2 ahead");
    run<2, 3>(w, dom, wgs, 4096, "weights in AGPRs, reads 3 ahead");
    run<3, 2>(w, dom, wgs, 4096, "... + interleaved epilogue, reads 2 ahead");
    run<3, 3>(w, dom, wgs, 4096, "... + interleaved epilogue, reads 3 ahead");
    run<4, 2>(w, dom, wgs, 4096, "every gap <= 2 instructions, reads 2 ahead");
    run<4, 3>(w, dom, wgs, 4096, "every gap <= 2 instructions, reads 3 ahead");
    run<4, 5>(w, dom, wgs, 4096, "every gap <= 2 instructions, reads 5 ahead");
  }
  return 0;
}
