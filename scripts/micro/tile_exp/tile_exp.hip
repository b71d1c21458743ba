// Experiment harness of round 6 for the Connect4 tile: one build per knob set (-DX_...), sustained throughput of the full tile at
// several row counts (many workgroup rounds per launch, so co-resident workgroups de-phase as in the persistent kernel) and the
// latency of a one-round launch.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DX_MINB=2 ... tile_exp.hip -o /tmp/tx && /tmp/tx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "c4x.h"
#ifndef X_MINB
#define X_MINB 2
#endif
using namespace azmi_net_dev;
template <class TG>
__global__ __launch_bounds__(c4::NTH, X_MINB) void k_x(NetDesc nd, NetPtrs np, const float* canon, float* v, float* pi, uint32_t batch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_dbg[];
  c4::tile<TG, 4, 4, 16, 0>(nd, np, canon, v, pi, batch, nullptr, nullptr, blockIdx.x, lds_dbg);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <class TG>
float run(const NetDesc& nd, const NetPtrs& np, const float* canon, float* v, float* pi, uint32_t batch, int reps) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_x<TG>), hipFuncAttributeMaxDynamicSharedMemorySize, TG::LDS_BYTES));
  const uint32_t tiles = (batch + TG::TBW - 1) / TG::TBW;
  for (int i = 0; i < 3; ++i) k_x<TG><<<tiles, c4::NTH, TG::LDS_BYTES>>>(nd, np, canon, v, pi, batch);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) k_x<TG><<<tiles, c4::NTH, TG::LDS_BYTES>>>(nd, np, canon, v, pi, batch);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}
int main(int argc, char** argv) {
  const int depth = 6, Hd = 256, M = 7, P1 = 3;
  NetDesc nd{4, 6, 7, depth, M, 2, Hd};
  const size_t wconv = 18 * MT * WFRAG_BYTES, wsmall = 2 * MT * WFRAG_BYTES;
  size_t n = wsmall + CH * 4 + depth * (3 * CH * 4 + 2 * wconv) + wsmall + CH * 4 + (size_t(Hd) * HC + Hd) * 4 + (size_t(P1) * Hd + P1) * 4 + 42 * 2 * WFRAG_BYTES + M * 4;
  std::vector<uint16_t> host(n / 2 + 8);
  srand(1);
  for (auto& x : host) x = uint16_t((rand() & 0x8000) | 0x3d00 | (rand() & 0xff));
  uint8_t* blob; CK(hipMalloc(&blob, n + 65536)); CK(hipMemcpy(blob, host.data(), n, hipMemcpyHostToDevice));
  NetPtrs np; const uint8_t* p = blob;
  np.stem_w = p; p += wsmall; np.stem_b = (const float*)p; p += CH * 4;
  np.blocks = p; p += depth * (3 * CH * 4 + 2 * wconv);
  np.head_w = p; p += wsmall; np.head_b = (const float*)p; p += CH * 4;
  np.v_fc1_w = (const float*)p; p += size_t(Hd) * HC * 4; np.v_fc1_b = (const float*)p; p += Hd * 4;
  np.v_fc2_w = (const float*)p; p += size_t(P1) * Hd * 4; np.v_fc2_b = (const float*)p; p += P1 * 4;
  np.pi_fc_w = p; p += 42 * 2 * WFRAG_BYTES; np.pi_fc_b = (const float*)p;
  const uint32_t maxb = 18432;
  float *canon, *v, *pi;
  CK(hipMalloc(&canon, size_t(maxb) * 4 * 42 * 4));
  { std::vector<float> hc(size_t(maxb) * 4 * 42); for (auto& x : hc) x = float(rand() & 1); CK(hipMemcpy(canon, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&v, size_t(maxb) * P1 * 4)); CK(hipMalloc(&pi, size_t(maxb) * M * 4));
  printf("knobs: MINB %d NRING %d SCHED %d M0 %d PRMFAKE %d | LDS big %d small %d\n", X_MINB, X_NRING, X_SCHED, X_M0, X_PRMFAKE, (X_NRING >= 5 ? c4::TileBig::LDS_BYTES : 0), c4::TileSmall::LDS_BYTES);
  for (uint32_t rows : {714u, 3072u, 18432u}) {
    const int reps = rows > 4000 ? 60 : 200;
#if X_MINB < 3 && X_NRING >= 5
    { const float us = run<c4::TileBig>(nd, np, canon, v, pi, rows, reps); printf("  big   rows %5u: %8.1f us  %6.2f M boards/s  %.3f of peak\n", rows, us, rows / us, rows * 37.7e6 / us / 1e6 / 2.5e6); }
#endif
    { const float us = run<c4::TileSmall>(nd, np, canon, v, pi, rows, reps); printf("  small rows %5u: %8.1f us  %6.2f M boards/s  %.3f of peak\n", rows, us, rows / us, rows * 37.7e6 / us / 1e6 / 2.5e6); }
  }
  return 0;
}
