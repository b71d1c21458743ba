// Timing-only harness of the Connect4 leaf-net tile (csrc/leafnet_c4.h): the tile's DBG variants (wrong results, by design)
// at a given row count, to see where a tile's latency goes.   build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/c4_tile_timing.hip -o /tmp/c4t && /tmp/c4t 714 4096
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include "../alphazero-pybind11_amd/csrc/leafnet_c4.h"
using namespace azmi_net_dev;

template <class TG, int DBG>
__global__ __launch_bounds__(c4::NTH, 2) void k_dbg(NetDesc nd, NetPtrs np, const float* canon, float* v, float* pi, uint32_t batch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_dbg[];
  c4::tile<TG, 4, 4, 16, DBG>(nd, np, canon, v, pi, batch, nullptr, nullptr, blockIdx.x, lds_dbg);
}
// the same tile with the shader clock (s_memtime) and the 100 MHz wall clock (s_memrealtime) stamped around it by one lane:
// in-kernel clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).  The stamps go
// to a buffer nothing else reads.
template <class TG, int DBG>
__global__ __launch_bounds__(c4::NTH, 2) void k_clk(NetDesc nd, NetPtrs np, const float* canon, float* v, float* pi, uint32_t batch, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_dbg[];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  c4::tile<TG, 4, 4, 16, DBG>(nd, np, canon, v, pi, batch, nullptr, nullptr, blockIdx.x, lds_dbg);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}
template <class TG, int DBG>
void clock_of(const NetDesc& nd, const NetPtrs& np, const float* canon, float* v, float* pi, uint32_t batch, const char* what) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_clk<TG, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, TG::LDS_BYTES);
  const uint32_t tiles = (batch + TG::TBW - 1) / TG::TBW;
  unsigned long long* st; hipMalloc(&st, tiles * 16); hipMemset(st, 0, tiles * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // two seconds of back-to-back launches, then the stamps of the last one
  int launches = 0;
  hipEventRecord(e0);
  for (;;) {
    for (int i = 0; i < 2000; ++i) k_clk<TG, DBG><<<tiles, c4::NTH, TG::LDS_BYTES>>>(nd, np, canon, v, pi, batch, st);
    launches += 2000;
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (ms > 2000.0f) { printf("    (%d launches, %.1f us each)\n", launches, ms * 1e3f / launches); break; }
  }
  std::vector<unsigned long long> h(2 * tiles);
  hipMemcpy(h.data(), st, tiles * 16, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (uint32_t i = 0; i < tiles; ++i) if (h[2 * i + 1]) ghz.push_back(0.1 * double(h[2 * i]) / double(h[2 * i + 1]));
  std::sort(ghz.begin(), ghz.end());
  std::vector<double> cyc; for (uint32_t i = 0; i < tiles; ++i) cyc.push_back(double(h[2 * i]));
  std::sort(cyc.begin(), cyc.end());
  printf("  in-kernel clock, %s: median %.2f GHz (min %.2f, max %.2f) over %zu workgroups; median tile = %.0f shader cycles\n", what, ghz[ghz.size() / 2], ghz.front(), ghz.back(), ghz.size(), cyc[cyc.size() / 2]);
  hipFree(st);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class TG, int DBG>
float run(const NetDesc& nd, const NetPtrs& np, const float* canon, float* v, float* pi, uint32_t batch, int reps) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dbg<TG, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, TG::LDS_BYTES));
  const uint32_t tiles = (batch + TG::TBW - 1) / TG::TBW;
  for (int i = 0; i < 5; ++i) k_dbg<TG, DBG><<<tiles, c4::NTH, TG::LDS_BYTES>>>(nd, np, canon, v, pi, batch);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) k_dbg<TG, DBG><<<tiles, c4::NTH, TG::LDS_BYTES>>>(nd, np, canon, v, pi, batch);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

template <class TG>
void sweep(const NetDesc& nd, const NetPtrs& np, uint32_t batch) {
  const int M = 7, P1 = 3;
    float *canon, *v, *pi;
    CK(hipMalloc(&canon, size_t(batch) * 4 * 42 * 4));
    { std::vector<float> hc(size_t(batch) * 4 * 42); for (auto& x : hc) x = float(rand() & 1); CK(hipMemcpy(canon, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&v, size_t(batch) * P1 * 4)); CK(hipMalloc(&pi, size_t(batch) * M * 4));
    const int reps = 200;
    printf("rows %u, tiles of %d boards\n", batch, TG::TBW);
    printf("  0 full                      %7.1f us\n", run<TG, 0>(nd, np, canon, v, pi, batch, reps));
    printf("  1 no DMA waits              %7.1f us\n", run<TG, 1>(nd, np, canon, v, pi, batch, reps));
    printf("  2 no DMA waits, no barriers %7.1f us\n", run<TG, 2>(nd, np, canon, v, pi, batch, reps));
    printf("  3 no weight DMA             %7.1f us\n", run<TG, 3>(nd, np, canon, v, pi, batch, reps));
    printf("  4 no fragment reads         %7.1f us\n", run<TG, 4>(nd, np, canon, v, pi, batch, reps));
    printf("  5 no MFMAs                  %7.1f us\n", run<TG, 5>(nd, np, canon, v, pi, batch, reps));
    printf("  6 stem only                 %7.1f us\n", run<TG, 6>(nd, np, canon, v, pi, batch, reps));
    printf("  7 stem + trunk              %7.1f us\n", run<TG, 7>(nd, np, canon, v, pi, batch, reps));
    printf("  8 ... + head 1x1 conv       %7.1f us\n", run<TG, 8>(nd, np, canon, v, pi, batch, reps));
    printf("  9 ... + policy FC, v pool   %7.1f us\n", run<TG, 9>(nd, np, canon, v, pi, batch, reps));
    printf(" 10 ... + value FCs           %7.1f us\n", run<TG, 10>(nd, np, canon, v, pi, batch, reps));
    if (getenv("C4T_NO_CLOCK")) { CK(hipFree(canon)); CK(hipFree(v)); CK(hipFree(pi)); return; }
    clock_of<TG, 0>(nd, np, canon, v, pi, batch, "full tile");
    clock_of<TG, 4>(nd, np, canon, v, pi, batch, "matrix stream alone (no fragment reads)");
    clock_of<TG, 5>(nd, np, canon, v, pi, batch, "fragment reads alone (no MFMAs)");
    CK(hipFree(canon)); CK(hipFree(v)); CK(hipFree(pi));
}

int main(int argc, char** argv) {
  const int depth = 6, Hd = 256, M = 7, P1 = 3;
  NetDesc nd{4, 6, 7, depth, M, 2, Hd};
  const size_t wconv = 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
  size_t n = wsmall + CH * 4 + depth * (3 * CH * 4 + 2 * wconv) + wsmall + CH * 4 + (size_t(Hd) * HC + Hd) * 4 + (size_t(P1) * Hd + P1) * 4 + 42 * 2 * WFRAG_BYTES + M * 4;
  std::vector<uint16_t> host(n / 2 + 8);
  srand(1);
  // random data (zero or constant operands let the chip clock higher: MI355X_MICROARCH.md DVFS give-back): bf16 values of
  // either sign around 2^-5 .. 2^-4, which are also harmless as halves of the fp32 parameters
  for (auto& x : host) x = uint16_t((rand() & 0x8000) | 0x3d00 | (rand() & 0xff));
  uint8_t* blob; CK(hipMalloc(&blob, n + 64)); CK(hipMemcpy(blob, host.data(), n, hipMemcpyHostToDevice));
  NetPtrs np; const uint8_t* p = blob;
  np.stem_w = p; p += wsmall; np.stem_b = (const float*)p; p += CH * 4;
  np.blocks = p; p += depth * (3 * CH * 4 + 2 * wconv);
  np.head_w = p; p += wsmall; np.head_b = (const float*)p; p += CH * 4;
  np.v_fc1_w = (const float*)p; p += size_t(Hd) * HC * 4; np.v_fc1_b = (const float*)p; p += Hd * 4;
  np.v_fc2_w = (const float*)p; p += size_t(P1) * Hd * 4; np.v_fc2_b = (const float*)p; p += P1 * 4;
  np.pi_fc_w = p; p += 42 * 2 * WFRAG_BYTES; np.pi_fc_b = (const float*)p;
  for (int a = 1; a < argc; ++a) {
    const uint32_t batch = atoi(argv[a]);
    sweep<c4::TileBig>(nd, np, batch);
    sweep<c4::TileSmall>(nd, np, batch);
  }
  return 0;
}
