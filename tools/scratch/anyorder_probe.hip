// Does a kernel launched with hipExtAnyOrderLaunch overlap its predecessor on the same stream?
//   hipcc --offload-arch=gfx950 -O3 -o anyorder_probe anyorder_probe.hip && ./anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long cycles, unsigned long long* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = wall_clock64();
}
int main() {
  unsigned long long *a, *b;
  CK(hipMalloc(&a, 4096 * 8)); CK(hipMalloc(&b, 4096 * 8));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long long cyc = 5000;      // wall_clock64 ticks at 100 MHz: 50 us
  for (int flags = 0; flags < 2; ++flags)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, cyc, a);
      hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, cyc, b);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 10LL, a);     // an ordinary launch behind both
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long ha, hb; CK(hipMemcpy(&ha, a + 1, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, b + 1, 8, hipMemcpyDeviceToHost));
      printf("flags %d rep %d: two 50 us kernels took %.1f us; end(b) - end(a) = %.1f us\n", flags, rep, ms * 1e3, ((double)hb - (double)ha) / 100.0);
    }
  return 0;
}
