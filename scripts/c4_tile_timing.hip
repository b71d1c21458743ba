// Timing-only harness of the Connect4 leaf-net tile (csrc/leafnet_c4.h): the tile's DBG variants (wrong results, by design)
// at a given row count, to see where a tile's latency goes.   build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/c4_tile_timing.hip -o /tmp/c4t && /tmp/c4t 714 4096
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../alphazero-pybind11_amd/csrc/leafnet_c4.h"
using namespace azmi_net_dev;

template <class TG, int DBG>
__global__ __launch_bounds__(c4::NTH, 2) void k_dbg(NetDesc nd, NetPtrs np, const float* canon, float* v, float* pi, uint32_t batch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_dbg[];
  c4::tile<TG, 4, 4, 16, DBG>(nd, np, canon, v, pi, batch, nullptr, nullptr, blockIdx.x, lds_dbg);
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
    CK(hipMalloc(&canon, size_t(batch) * 4 * 42 * 4)); CK(hipMemset(canon, 0, size_t(batch) * 4 * 42 * 4));
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
    CK(hipFree(canon)); CK(hipFree(v)); CK(hipFree(pi));
}

int main(int argc, char** argv) {
  const int depth = 6, Hd = 256, M = 7, P1 = 3;
  NetDesc nd{4, 6, 7, depth, M, 2, Hd};
  const size_t wconv = 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
  size_t n = wsmall + CH * 4 + depth * (3 * CH * 4 + 2 * wconv) + wsmall + CH * 4 + (size_t(Hd) * HC + Hd) * 4 + (size_t(P1) * Hd + P1) * 4 + 42 * 2 * WFRAG_BYTES + M * 4;
  std::vector<uint16_t> host(n / 2 + 8);
  srand(1);
  for (auto& x : host) x = 0x3c00 + (rand() & 0xff);   // small positive bf16 / harmless fp32 halves
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
