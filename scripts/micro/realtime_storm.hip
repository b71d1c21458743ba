// How long does one s_memrealtime take when hundreds of wavefronts read it in a poll loop (as the pipeline's idle workgroups do)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/realtime_storm scripts/micro/realtime_storm.hip && /tmp/realtime_storm
// Each workgroup's wave 0 makes N reads, s_sleep(SLEEP) between them, and times every read with the SQ's own cycle counter
// (s_memtime); optionally (HOT=1) it also loads one hot line device-scope, as the pollers do.  Prints, over all workgroups: the mean and
// the longest read, and how many reads took longer than 10 us / 1 ms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_storm(unsigned long long* out, const unsigned* hot, int n, int sleep_on, int hot_on) {
  if (threadIdx.x >= 64) { __syncthreads(); return; }
  unsigned long long sum = 0, mx = 0, over10 = 0, over1k = 0, acc = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r = __builtin_amdgcn_s_memrealtime();
    unsigned h = 0;
    if (hot_on) h = __hip_atomic_load(hot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" :: "s"(r), "v"(h));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long d = c1 - c0;
    sum += d; if (d > mx) mx = d; if (d > 1000) ++over10; if (d > 100000) ++over1k;      // s_memtime: 100 MHz on gfx9 (10 ns units)
    acc += r + h;
    if (sleep_on) __builtin_amdgcn_s_sleep(16);
  }
  if (threadIdx.x == 0) { out[blockIdx.x * 8 + 0] = sum; out[blockIdx.x * 8 + 1] = mx; out[blockIdx.x * 8 + 2] = over10; out[blockIdx.x * 8 + 3] = over1k; out[blockIdx.x * 8 + 4] = acc; }
  __syncthreads();
}
int main() {
  const int n = getenv("N") ? atoi(getenv("N")) : 200000;
  for (int hot_on = 0; hot_on < 2; ++hot_on)
    for (int wgs : {8, 64, 512, 1024}) {
      unsigned long long* d; unsigned* hot;
      hipMalloc(&d, sizeof(unsigned long long) * 8 * wgs); hipMalloc(&hot, 256); hipMemset(hot, 0, 256);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      k_storm<<<wgs, 256>>>(d, hot, n, 1, hot_on);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(8 * wgs);
      hipMemcpy(h.data(), d, sizeof(unsigned long long) * 8 * wgs, hipMemcpyDeviceToHost);
      unsigned long long sum = 0, mx = 0, o10 = 0, o1k = 0;
      for (int w = 0; w < wgs; ++w) { sum += h[w * 8]; if (h[w * 8 + 1] > mx) mx = h[w * 8 + 1]; o10 += h[w * 8 + 2]; o1k += h[w * 8 + 3]; }
      printf("%4d workgroups, hot line %d: %d reads each in %.1f ms; a read: mean %.0f ticks, longest %llu ticks (s_memtime units), > 1000 ticks: %llu, > 100000 ticks: %llu\n",
             wgs, hot_on, n, ms, double(sum) / (double(n) * wgs), mx, o10, o1k);
      hipFree(d); hipFree(hot);
    }
  return 0;
}
