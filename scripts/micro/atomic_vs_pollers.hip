// Does a returning atomic on a word starve while hundreds of wavefronts poll the same word with device-scope loads?
// (round 5: the pipeline's tree wavefronts were seen to stand still for seconds inside a pass while ~500 idle net workgroups polled the
// request ring's tail word - the word the tree side draws its tickets from.)
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/atomic_vs_pollers.bin scripts/micro/atomic_vs_pollers.hip
// Workgroup 0: M returning atomic adds on X, each timed with s_memtime; workgroups 1..P: poll X with sc1 loads (s_sleep(SLEEP) between
// polls) until workgroup 0 sets a stop word.  Prints the mean / longest atomic, and how many took > 10 us / > 1 ms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned* x, unsigned* stop, unsigned long long* out, int m, int sleep_units, int same_line) {
  if (threadIdx.x >= 64) return;
  if (blockIdx.x == 0) {
    unsigned long long sum = 0, mx = 0, o10 = 0, o1k = 0; unsigned acc = 0;
    for (int i = 0; i < m; ++i) {
      const unsigned long long c0 = __builtin_amdgcn_s_memtime();
      unsigned r = 0;
      if (threadIdx.x == 0) r = atomicAdd(x, 1u);
      asm volatile("s_waitcnt vmcnt(0)" :: "v"(r));
      const unsigned long long d = __builtin_amdgcn_s_memtime() - c0;
      sum += d; if (d > mx) mx = d; if (d > 1000) ++o10; if (d > 100000) ++o1k; acc += r;
      __builtin_amdgcn_s_sleep(64);
    }
    if (threadIdx.x == 0) { out[0] = sum; out[1] = mx; out[2] = o10; out[3] = o1k; out[4] = acc; __hip_atomic_store(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return;
  }
  const unsigned* px = same_line ? x : x + 64;      // (the same word, or a word 256 bytes away)
  unsigned acc = 0;
  for (;;) {
    unsigned v = 0, s = 0;
    if (threadIdx.x == 0) { v = __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s = __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    acc += v;
    if (__builtin_amdgcn_readfirstlane(s)) break;
    if (sleep_units) __builtin_amdgcn_s_sleep(16);
  }
  if (threadIdx.x == 0 && acc == 0xFFFFFFFFu) out[8] = acc;
}
int main() {
  const int m = getenv("M") ? atoi(getenv("M")) : 20000;
  unsigned* x; unsigned* stop; unsigned long long* out;
  hipMalloc(&x, 1024); hipMalloc(&stop, 256); hipMalloc(&out, 256);
  for (int same_line = 1; same_line >= 0; --same_line)
    for (int sleep_units : {1, 0})
      for (int pollers : {0, 64, 504, 1016}) {
        hipMemset(x, 0, 1024); hipMemset(stop, 0, 256); hipMemset(out, 0, 256);
        k<<<1 + pollers, 256>>>(x, stop, out, m, sleep_units, same_line);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("%4d pollers of %s, %s: %d atomics: mean %.0f ticks, longest %llu ticks, > 1000 ticks: %llu, > 100000 ticks: %llu\n", pollers,
               same_line ? "the atomic's word" : "a word 256 B away", sleep_units ? "s_sleep(16) between polls" : "no sleep", m, double(h[0]) / m, h[1], h[2], h[3]);
        fflush(stdout);
      }
  return 0;
}
