// Timing harness of the spatial-policy-head tile (leafnet_sp.h copy spx.h with early exits, -DX_DBG=n): where a Tawlbwrdd / StarGambit
// tile's time goes.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DX_DBG=0 sp_exp.hip -o /tmp/spx && /tmp/spx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "spx.h"
using namespace azmi_net_dev;
template <class G>
__global__ __launch_bounds__(sp::NTH, 2) void k_x(sp::SpDesc nd, sp::SpPtrs np, const float* canon, float* vpool, float* ppool, float* pi, uint32_t batch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_x[];
  sp::tile<G>(nd, np, canon, vpool, ppool, pi, batch, nullptr, nullptr, blockIdx.x, lds_x);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <class G>
void run(const char* name, int c_in, int moves, int pol_ch, int num_global, std::initializer_list<uint32_t> rows_list) {
  sp::SpDesc nd{c_in, G::H, G::W, 4, moves, 2, 256, 2, pol_ch, num_global, 64};
  const int nch = sp::stream_chunks(4, c_in, 0);
  std::vector<uint16_t> host(size_t(nch + 8) * sp::CHUNK_BYTES / 2);
  srand(1);
  for (auto& x : host) x = uint16_t((rand() & 0x8000) | 0x3d00 | (rand() & 0xff));
  uint8_t* stream; CK(hipMalloc(&stream, host.size() * 2)); CK(hipMemcpy(stream, host.data(), host.size() * 2, hipMemcpyHostToDevice));
  float* prm; CK(hipMalloc(&prm, 8192 * 4)); CK(hipMemcpy(prm, host.data(), 8192 * 4, hipMemcpyHostToDevice));
  sp::SpPtrs np{}; np.stream = stream; np.prm = prm;
  const uint32_t maxb = 8192;
  float *canon, *vpool, *ppool, *pi;
  CK(hipMalloc(&canon, size_t(maxb) * c_in * G::PIX * 4));
  { std::vector<float> hc(size_t(maxb) * c_in * G::PIX); for (auto& x : hc) x = float(rand() & 1); CK(hipMemcpy(canon, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&vpool, size_t(maxb) * 64 * 4)); CK(hipMalloc(&ppool, size_t(maxb) * 64 * 4)); CK(hipMalloc(&pi, size_t(maxb) * moves * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_x<G>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  for (uint32_t rows : rows_list) {
    const uint32_t tiles = (rows + G::TBW - 1) / G::TBW;
    const int reps = rows > 2048 ? 50 : 200;
    for (int i = 0; i < 3; ++i) k_x<G><<<tiles, sp::NTH, G::LDS_BYTES>>>(nd, np, canon, vpool, ppool, pi, rows);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_x<G><<<tiles, sp::NTH, G::LDS_BYTES>>>(nd, np, canon, vpool, ppool, pi, rows);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-10s dbg %d rows %5u (%4u tiles): %8.1f us\n", name, X_DBG, rows, tiles, ms * 1e3f / reps);
  }
}
int main() {
  run<sp::Geo11>("tawlbwrdd", 7, 2662, 22, 0, {256u, 512u, 1024u, 8192u});
  run<sp::Geo13>("stargambit", 36, 1709, 10, 19, {128u, 256u, 512u, 4096u});
  return 0;
}
