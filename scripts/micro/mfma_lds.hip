// Calibration microbenchmark (not part of the product): what one wave per SIMD (or two) sustains on gfx950 when every
// k-step is 8 x ds_read_b128 (4 A + 4 B fragments) + 16 x v_mfma_f32_16x16x32_bf16, software-pipelined one k-step ahead.
// Variants: V=0 MFMA only, V=1 + LDS reads, V=2 + 24 VALU per k-step, V=3 + s_barrier every 3 k-steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int V, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void k(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid * 16; i < 65536; i += WAVES * 64 * 16) *reinterpret_cast<float4*>(lds + i) = make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f);
  __syncthreads();
  f32x4 acc[4][4];
  for (int j = 0; j < 4; ++j) for (int m = 0; m < 4; ++m) acc[j][m] = f32x4{0, 0, 0, 0};
  bf16x8 a[2][4], b[2][4];
  int off = lane * 16;
  for (int m = 0; m < 4; ++m) { a[0][m] = *reinterpret_cast<bf16x8*>(lds + off + m * 1024); b[0][m] = *reinterpret_cast<bf16x8*>(lds + 8192 + off + m * 1024); }
  int x = lane;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const int cur = ks & 1;
      if (V >= 3 && ks % 3 == 2) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
      if (V >= 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          a[cur ^ 1][m] = *reinterpret_cast<bf16x8*>(lds + off + (ks * 4 + m) * 1024);
          if (V >= 2) { int y = x; asm volatile("" : "+v"(y)); y = (y + 0x70); int z = (y & 0xF0) | 0x1000; y = (lane & 1) ? y : z; x = x ^ (y & 1); }
          b[cur ^ 1][m] = *reinterpret_cast<bf16x8*>(lds + 32768 + off + ((ks * 4 + m) & 15) * 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[cur][m], b[cur][j], acc[j][m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int m = 0; m < 4; ++m) s += acc[j][m][0] + acc[j][m][3];
  out[blockIdx.x * blockDim.x + tid] = s + x;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int WAVES>
void run(const char* name, int blocks, int iters) {
  float* out; long long* cyc;
  hipMalloc(&out, blocks * WAVES * 64 * 4); hipMalloc(&cyc, blocks * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V, WAVES><<<blocks, WAVES * 64, 65536>>>(out, cyc, iters);
  hipEventRecord(e0);
  k<V, WAVES><<<blocks, WAVES * 64, 65536>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto c : h) avg += c; avg /= blocks;
  const double mf = 6.0 * 16 * iters;
  printf("%-28s blocks=%4d waves/blk=%d: %.1f us, %.1f memtime-ticks per MFMA (100MHz ticks x24 = %.1f cyc@2.4GHz), %.1f cyc/MFMA by wall@2.4GHz\n", name, blocks, WAVES, ms * 1e3,
         avg / mf, avg / mf * 24, ms * 1e-3 * 2.4e9 / mf);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int it = 2000;
  run<0, 4>("mfma only", 256, it); run<1, 4>("mfma+lds", 256, it); run<2, 4>("mfma+lds+valu", 256, it); run<3, 4>("mfma+lds+valu+barrier", 256, it);
  run<0, 4>("mfma only 2wg/cu", 512, it); run<1, 4>("mfma+lds 2wg/cu", 512, it); run<2, 4>("mfma+lds+valu 2wg/cu", 512, it); run<3, 4>("mfma+lds+valu+barrier 2wg", 512, it);
  run<1, 8>("mfma+lds 8 waves", 256, it); run<2, 8>("mfma+lds+valu 8 waves", 256, it); run<3, 8>("mfma+lds+valu+barrier 8w", 256, it);
  return 0;
}
