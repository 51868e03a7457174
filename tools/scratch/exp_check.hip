// chomp::exp (chomp_math.h) against the device library's exp, bit for bit (scratch check):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/exp_check tools/scratch/exp_check.hip && /tmp/exp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../include/chomp_mi355x.h"
#include "../../chomp_amd/csrc/chomp_math.h"
__global__ void k(const double* x, unsigned long long* bad, double* first, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = chomp::exp(x[i]), b = ::exp(x[i]);
  if (__double_as_longlong(a) != __double_as_longlong(b) && !(a != a && b != b)) {
    if (atomicAdd(bad, 1ull) == 0) { first[0] = x[i]; first[1] = a; first[2] = b; }
  }
}
int main() {
  const int n = 1 << 24;
  double* h = new double[n];
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    h[i] = i < n / 2 ? -760.0 + 1480.0 * u : -40.0 + 80.0 * u;     // incl. overflow / subnormal ends
  }
  h[0] = 0.0; h[1] = 1100.0; h[2] = -1100.0; h[3] = 709.9; h[4] = -745.2; h[5] = -708.4;
  double *d, *f; unsigned long long* b;
  hipMalloc(&d, n * 8); hipMalloc(&f, 24); hipMalloc(&b, 8);
  hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice); hipMemset(b, 0, 8); hipMemset(f, 0, 24);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, b, f, n);
  unsigned long long nb; double ff[3];
  hipMemcpy(&nb, b, 8, hipMemcpyDeviceToHost); hipMemcpy(ff, f, 24, hipMemcpyDeviceToHost);
  printf("exp: %llu of %d differ", nb, n);
  if (nb) printf(" (first: x = %.17g: %.17g vs %.17g)", ff[0], ff[1], ff[2]);
  printf("\n");
  return nb != 0;
}
