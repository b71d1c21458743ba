// A stand-in for the leaf-net tile's weight stream: every workgroup reads the same 1 MB buffer (L2-resident after the first pass) over and
// over for `ms` milliseconds - what the tree kernel's round trips cost beside that traffic (scripts/l2_pressure.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/micro/libl2stream.so scripts/micro/l2_stream.hip
#include <hip/hip_runtime.h>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ __launch_bounds__(256) void k_l2_stream(const uint8_t* buf, uint32_t bytes, unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(buf), 0, bytes, 0x00020000);
  u32x4 acc = {0, 0, 0, 0};
  unsigned long long n = 0;
  uint32_t off = (blockIdx.x * 4096u + threadIdx.x * 16u) % bytes;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {      // 8 x 4 KB per workgroup in flight, device-scope loads (served by L2, not the CU's L1)
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
      acc += v;
      off += 4096u; if (off >= bytes) off -= bytes;
    }
    n += 8;
  }
  if (acc[0] == 0xFFFFFFFFu && acc[1] == 1u) out[1] = acc[2];
  if (threadIdx.x == 0) atomicAdd(out, n * 4096ull);
}
extern "C" int l2_stream_launch(const void* buf, uint32_t bytes, uint32_t wgs, double ms, void* out, void* stream) {
  k_l2_stream<<<wgs, 256, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const uint8_t*>(buf), bytes, static_cast<unsigned long long>(ms * 1e5), static_cast<unsigned long long*>(out));
  return static_cast<int>(hipGetLastError());
}
