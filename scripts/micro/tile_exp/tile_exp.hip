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
#ifndef X_DBG
#define X_DBG 0            // 7: stem + trunk only (the gate of the 8-wave tile: its heads are not written)
#endif
using namespace azmi_net_dev;
#if X_NWV == 8
using XBig = c4::Tile<6, 2>; using XSmall = c4::Tile<3, 1>;
#else
using XBig = c4::TileBig; using XSmall = c4::TileSmall;
#endif
template <class TG>
__global__ __launch_bounds__(c4::NTH, X_MINB * (c4::NTH / 256)) void k_x(NetDesc nd, NetPtrs np, const float* canon, float* v, float* pi, uint32_t batch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_dbg[];
  c4::tile<TG, 4, 4, 16, X_DBG>(nd, np, canon, v, pi, batch, nullptr, nullptr, blockIdx.x, lds_dbg);
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
#if X_STAMP
  unsigned long long* stamp_dev; CK(hipMalloc(&stamp_dev, size_t(maxb) * 16 * 8)); CK(hipMemset(stamp_dev, 0, size_t(maxb) * 16 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c4::g_stamp_out), &stamp_dev, sizeof(stamp_dev)));
#endif
  printf("knobs: MINB %d NRING %d SCHED %d M0 %d PRMFAKE %d NWV %d SWP %d DBG %d | LDS big %d small %d\n", X_MINB, X_NRING, X_SCHED, X_M0, X_PRMFAKE, X_NWV, X_SWP, X_DBG, (X_NRING >= 5 ? XBig::LDS_BYTES : 0), XSmall::LDS_BYTES);
  for (uint32_t rows : {714u, 3072u, 18432u}) {
    const int reps = rows > 4000 ? 60 : 200;
#if X_MINB < 3 && X_NRING >= 5
    { const float us = run<XBig>(nd, np, canon, v, pi, rows, reps); printf("  big   rows %5u: %8.1f us  %6.2f M boards/s  %.3f of peak\n", rows, us, rows / us, rows * 37.7e6 / us / 1e6 / 2.5e6); }
#endif
    { const float us = run<XSmall>(nd, np, canon, v, pi, rows, reps); printf("  small rows %5u: %8.1f us  %6.2f M boards/s  %.3f of peak\n", rows, us, rows / us, rows * 37.7e6 / us / 1e6 / 2.5e6); }
  }
#if X_STAMP
  {   // phase cycles of the LAST launch (18432 rows, the small tile: the last one run), averaged over its tiles
    const char* names[11] = {"stem", "epilogue 1 (+barrier) x6", "conv1 x6", "epilogue 2 (+barrier) x6", "conv2 x6", "head 1x1 conv (+store, waits)", "value / policy planes", "policy FC + pool", "logits + value fc1", "value fc2", "softmax + stores"};
    for (int which = 0; which < 2; ++which) {
      const uint32_t rows = 714;
      if (which == 0) run<XBig>(nd, np, canon, v, pi, rows, 3); else run<XSmall>(nd, np, canon, v, pi, rows, 3);
      const uint32_t tiles = (rows + (which == 0 ? XBig::TBW : XSmall::TBW) - 1) / (which == 0 ? XBig::TBW : XSmall::TBW);
      std::vector<unsigned long long> h(size_t(tiles) * 16);
      CK(hipMemcpy(h.data(), stamp_dev, h.size() * 8, hipMemcpyDeviceToHost));
      printf("  phase cycles per tile (wave 0, %s tile, %u tiles alone on their CUs):\n", which == 0 ? "big" : "small", tiles);
      double tot = 0;
      for (int i = 0; i < 11; ++i) { double a = 0; for (uint32_t t = 0; t < tiles; ++t) a += double(h[size_t(t) * 16 + i]); a /= tiles; tot += a; printf("    %-34s %9.0f cycles  %6.2f us\n", names[i], a, a / 2400.0); }
      printf("    %-34s %9.0f cycles  %6.2f us\n", "sum", tot, tot / 2400.0);
    }
  }
#endif
  return 0;
}
