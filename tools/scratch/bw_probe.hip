// Write-bandwidth probe for the Stage-E store pattern (scratch; not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE> __device__ inline void st(double* p, d2 v) {
  if (MODE == 0) *(d2*)p = v;
  else if (MODE == 1) __builtin_nontemporal_store(v, (d2*)p);
  else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

template <int MODE> __global__ void k_linear(double* out, size_t n2) {   // n2 = number of d2
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n2; i += stride) st<MODE>(out + 2 * i, d2{1.0, 2.0});
}

// the Stage-E pattern: thread owns 2 consecutive k, walks nz rows (pitch in doubles)
template <int MODE, int ROT> __global__ void k_rows(double* out, size_t nk, int nz, size_t pitch) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= nk) return;
  int e0 = ROT ? (int)((blockIdx.x * ROT) % nz) : 0;
  for (int j = 0; j < nz; ++j) {
    int e = e0 + j; if (e >= nz) e -= nz;
    st<MODE>(out + (size_t)e * pitch + i, d2{(double)e, 2.0});
  }
}

// rows split over blockIdx.y (each block walks nz/gy rows)
template <int MODE> __global__ void k_rows_y(double* out, size_t nk, int nz, size_t pitch) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= nk) return;
  const int per = nz / gridDim.y, e0 = blockIdx.y * per;
  for (int e = e0; e < e0 + per; ++e) st<MODE>(out + (size_t)e * pitch + i, d2{(double)e, 2.0});
}


// thread owns U chunks of 2 k (chunks blockDim*2 apart), walks the rows writing U stores per row
template <int MODE, int U> __global__ void k_rows_u(double* out, size_t nk, int nz, int rot) {
  const size_t base = (size_t)blockIdx.x * blockDim.x * 2 * U + threadIdx.x * 2;
  int e = (int)((blockIdx.x * (unsigned)rot) % (unsigned)nz);
  for (int j = 0; j < nz; ++j) {
    double* o = out + (size_t)e * nk + base;
#pragma unroll
    for (int u = 0; u < U; ++u) st<MODE>(o + (size_t)u * blockDim.x * 2, d2{(double)e, 2.0});
    if (++e == nz) e = 0;
  }
}
// linear in (row, k): grid (nk / (2 * block), nz / per); per-k operands (16 B per k) from a table
template <int MODE> __global__ void k_tab(const double* __restrict__ tbl, double* out, size_t nk, int per) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const d2 a = *(const d2*)(tbl + 2 * i), b = *(const d2*)(tbl + 2 * i + 2);
  for (int e = blockIdx.y * per; e < (int)(blockIdx.y + 1) * per; ++e) {
    const double c = (double)e;
    st<MODE>(out + (size_t)e * nk + i, d2{fma(a.x, c, a.y), fma(b.x, c, b.y)});
  }
}

// as k_tab, with NF dependent fp64 FMAs per sample (the arithmetic density of Stage E)
template <int NF> __global__ void k_tab_fma(const double* __restrict__ tbl, double* out, size_t nk, int per) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const d2 a = *(const d2*)(tbl + 2 * i), b = *(const d2*)(tbl + 2 * i + 2);
  for (int e = blockIdx.y * per; e < (int)(blockIdx.y + 1) * per; ++e) {
    double x = (double)e + a.x, y = (double)e + b.x;
#pragma unroll
    for (int q = 0; q < NF; ++q) { x = fma(x, a.y + 0.5, 1e-3); y = fma(y, b.y + 0.5, 1e-3); }
    st<2>(out + (size_t)e * nk + i, d2{x, y});
  }
}
__global__ void k_copy(const double* __restrict__ in, double* __restrict__ out, size_t n2) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n2; i += stride) *(d2*)(out + 2 * i) = *(const d2*)(in + 2 * i);
}
__global__ void k_read(const double* __restrict__ in, double* __restrict__ out, size_t n2) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  d2 acc = {0, 0};
  for (; i < n2; i += stride) acc += *(const d2*)(in + 2 * i);
  if (acc.x == 1.2345) out[0] = acc.y;
}

template <class F> float timeit(F f, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t nk = 1 << 20; const int nz = 64;
  const size_t padmax = 4096;
  double *out, *in;
  CK(hipMalloc(&out, (nk + padmax) * nz * sizeof(double)));
  CK(hipMalloc(&in, nk * nz * sizeof(double)));
  CK(hipMemset(in, 0, nk * nz * sizeof(double)));
  const double bytes = (double)nk * nz * 8;
  auto rep = [&](const char* name, float ms, double b) { printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, b / ms / 1e6); };
  rep("hipMemsetAsync", timeit([&] { hipMemsetAsync(out, 0, nk * nz * 8, 0); }), bytes);
  rep("copy d2d kernel (r+w bytes)", timeit([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, in, out, nk * nz / 2); }), 2 * bytes);
  rep("hipMemcpyAsync d2d (r+w bytes)", timeit([&] { hipMemcpyAsync(out, in, nk * nz * 8, hipMemcpyDeviceToDevice, 0); }), 2 * bytes);
  rep("read-only kernel", timeit([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, in, out, nk * nz / 2); }), bytes);
  for (int g : {2048, 8192, 32768, 131072}) {
    char nm[96];
    snprintf(nm, 96, "linear plain grid=%d", g); rep(nm, timeit([&] { hipLaunchKernelGGL(k_linear<0>, dim3(g), dim3(256), 0, 0, out, nk * nz / 2); }), bytes);
    snprintf(nm, 96, "linear nt    grid=%d", g); rep(nm, timeit([&] { hipLaunchKernelGGL(k_linear<1>, dim3(g), dim3(256), 0, 0, out, nk * nz / 2); }), bytes);
    snprintf(nm, 96, "linear sc1   grid=%d", g); rep(nm, timeit([&] { hipLaunchKernelGGL(k_linear<2>, dim3(g), dim3(256), 0, 0, out, nk * nz / 2); }), bytes);
    snprintf(nm, 96, "linear sc0sc1 grid=%d", g); rep(nm, timeit([&] { hipLaunchKernelGGL(k_linear<3>, dim3(g), dim3(256), 0, 0, out, nk * nz / 2); }), bytes);
  }

  {
    double* tbl; CK(hipMalloc(&tbl, nk * 2 * sizeof(double))); CK(hipMemset(tbl, 0, nk * 2 * sizeof(double)));
    char nm[96];
    for (int bs : {128, 256, 512})
      for (int per : {1, 2, 4, 8, 16}) {
        const unsigned gx = (unsigned)(nk / 2 / bs);
        snprintf(nm, 96, "tab sc1 per=%d block=%d", per, bs); rep(nm, timeit([&] { hipLaunchKernelGGL(k_tab<2>, dim3(gx, nz / per), dim3(bs), 0, 0, tbl, out, nk, per); }), bytes);
        snprintf(nm, 96, "tab plain per=%d block=%d", per, bs); rep(nm, timeit([&] { hipLaunchKernelGGL(k_tab<0>, dim3(gx, nz / per), dim3(bs), 0, 0, tbl, out, nk, per); }), bytes);
        snprintf(nm, 96, "tab nt per=%d block=%d", per, bs); rep(nm, timeit([&] { hipLaunchKernelGGL(k_tab<1>, dim3(gx, nz / per), dim3(bs), 0, 0, tbl, out, nk, per); }), bytes);
      }
    for (int bs : {64, 256})
      for (int rot : {0, 1}) {
        snprintf(nm, 96, "rows_u sc1 U=2 rot=%d block=%d", rot, bs); rep(nm, timeit([&] { hipLaunchKernelGGL((k_rows_u<2, 2>), dim3(nk / 2 / bs / 2), dim3(bs), 0, 0, out, nk, nz, rot); }), bytes);
        snprintf(nm, 96, "rows_u sc1 U=4 rot=%d block=%d", rot, bs); rep(nm, timeit([&] { hipLaunchKernelGGL((k_rows_u<2, 4>), dim3(nk / 2 / bs / 4), dim3(bs), 0, 0, out, nk, nz, rot); }), bytes);
        snprintf(nm, 96, "rows_u sc1 U=8 rot=%d block=%d", rot, bs); rep(nm, timeit([&] { hipLaunchKernelGGL((k_rows_u<2, 8>), dim3(nk / 2 / bs / 8), dim3(bs), 0, 0, out, nk, nz, rot); }), bytes);
      }

    {   // sustained behaviour: 30 batches of 10 launches of the row-major pattern (per = 2)
      const unsigned gx = (unsigned)(nk / 2 / 256);
      hipEvent_t ev[31];
      for (int i = 0; i < 31; ++i) hipEventCreate(&ev[i]);
      hipDeviceSynchronize();
      hipEventRecord(ev[0], 0);
      for (int b = 0; b < 30; ++b) {
        for (int r = 0; r < 10; ++r)
          hipLaunchKernelGGL(k_tab<2>, dim3(gx, nz / 2), dim3(256), 0, 0, tbl, out, nk, 2);
        hipEventRecord(ev[b + 1], 0);
      }
      hipDeviceSynchronize();
      printf("sustained tab sc1 per=2, us per launch in batches of 10:");
      for (int b = 0; b < 30; ++b) { float ms; hipEventElapsedTime(&ms, ev[b], ev[b + 1]); printf(" %.0f", ms * 100.0f); }
      printf("\n");
    }

    {   // sustained, with Stage E's arithmetic density
      const unsigned gx = (unsigned)(nk / 2 / 256);
      hipEvent_t ev[31];
      for (int i = 0; i < 31; ++i) hipEventCreate(&ev[i]);
      hipDeviceSynchronize();
      hipEventRecord(ev[0], 0);
      for (int b = 0; b < 30; ++b) {
        for (int r = 0; r < 10; ++r)
          hipLaunchKernelGGL(k_tab_fma<14>, dim3(gx, nz / 2), dim3(256), 0, 0, tbl, out, nk, 2);
        hipEventRecord(ev[b + 1], 0);
      }
      hipDeviceSynchronize();
      printf("sustained tab sc1 per=2 + 14 FMA/sample, us per launch in batches of 10:");
      for (int b = 0; b < 30; ++b) { float ms; hipEventElapsedTime(&ms, ev[b], ev[b + 1]); printf(" %.0f", ms * 100.0f); }
      printf("\n");
    }
    for (int gy : {16, 32, 64}) {
      snprintf(nm, 96, "rows sc1 gy=%d block=256", gy); rep(nm, timeit([&] { hipLaunchKernelGGL(k_rows_y<2>, dim3(nk / 512, gy), dim3(256), 0, 0, out, nk, nz, nk); }), bytes);
    }
  }
  return 0;
}
