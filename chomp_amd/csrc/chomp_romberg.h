// chomp_romberg.h -- wavefront / workgroup Romberg quadrature for gfx950.
//
// The reference evaluates every tabulated integral with scipy.integrate.romberg
// (call sites: cosmology.py:634-639, halo.py:692-698, 909-915, 951-957, 976-982,
// 1018-1024, 1065-1071, 1277-1282, kernel.py:455-460, 614-620, 699-704, 825-830,
// correlation.py:253-259, 371-377).  Its stopping rule is part of the numbers it
// produces (several halo-model integrands are discontinuous and stop "early"), so
// the rule is reproduced here exactly; what changes is the execution shape:
//
//   * one integral (or NF integrals sharing their nodes) per GROUP of NW
//     wavefronts; the 2^(i-1) new mid-points of level i are strided over the
//     64*NW lanes, partial sums reduced with __shfl_xor butterflies (+ one LDS
//     exchange when NW > 1);
//   * the Richardson row lives one entry per lane (lane k holds R[i][k]), so the
//     extrapolation is 2 register values per lane instead of a 21-entry array;
//   * every lane of the group ends up with the same sums, so the stopping test
//     is wave-uniform and needs no broadcast.
//
// Nodes follow SciPy's formula lox + h*j with h = (b-a)/2^(i-1), lox = a + h/2.
#pragma once

#include <hip/hip_runtime.h>

namespace chomp {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over the group.  NW == 1: the group is one wavefront (any number of groups
// per block, no barrier).  NW > 1: the group is the whole block (blockDim.x ==
// 64*NW); `red` points to 2*NW doubles of LDS, `flip` alternates the two halves so
// one barrier per call is enough.
template <int NW>
__device__ __forceinline__ double group_sum(double v, double* red, int& flip) {
  v = wave_sum(v);
  if constexpr (NW == 1) {
    return v;
  } else {
    double* r = red + flip * NW;
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += r[i];
    flip ^= 1;
    return t;
  }
}

template <int NF>
struct RombergOut {
  double value[NF];
  int level[NF];
};

// F: void operator()(double x, double (&out)[NF]) const
template <int NW, int NF, class F>
__device__ __forceinline__ RombergOut<NF> romberg_group(const F& f, double a, double b,
                                                        double tol, double rtol,
                                                        int divmax, double* red) {
  constexpr int NT = 64 * NW;
  const int lane = threadIdx.x & 63;
  const int gt = (NW == 1) ? lane : (int)threadIdx.x;
  int flip = 0;
  const double intrange = b - a;

  double ordsum[NF], last[NF], result[NF];
  bool done[NF];
  RombergOut<NF> out;

  // T_0: the two end points (lanes 0 and 1 of the group)
  {
    double v[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) v[q] = 0.0;
    if (gt < 2) f(gt == 0 ? a : b, v);
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      ordsum[q] = 0.5 * group_sum<NW>(v[q], red, flip);
      result[q] = intrange * ordsum[q];
      last[q] = (lane == 0) ? result[q] : 0.0;   // lane k holds R[i-1][k]
      done[q] = false;
      out.value[q] = result[q];
      out.level[q] = 0;
    }
  }

  long n = 1;
  for (int i = 1; i <= divmax; ++i) {
    n *= 2;
    const long numtosum = n / 2;
    const double h = intrange / (double)numtosum;
    const double lox = a + 0.5 * h;
    double part[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) part[q] = 0.0;
    for (long j = gt; j < numtosum; j += NT) {
      double v[NF];
      f(lox + h * (double)j, v);
#pragma unroll
      for (int q = 0; q < NF; ++q) part[q] += v[q];
    }
    bool all_done = true;
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const double s = group_sum<NW>(part[q], red, flip);
      if (done[q]) continue;
      ordsum[q] += s;
      double cur = intrange * ordsum[q] / (double)n;     // R[i][0]
      double mine = (lane == 0) ? cur : 0.0;
      double p4 = 1.0;
      for (int k = 0; k < i; ++k) {                      // Richardson
        p4 *= 4.0;
        const double lastk = __shfl(last[q], k, 64);
        cur = (p4 * cur - lastk) / (p4 - 1.0);
        if (lane == k + 1) mine = cur;
      }
      const double lastresult = __shfl(last[q], i - 1, 64);
      const double err = fabs(cur - lastresult);
      result[q] = cur;
      out.value[q] = cur;
      out.level[q] = i;
      if (err < tol || err < rtol * fabs(cur)) done[q] = true;
      last[q] = mine;
      all_done = all_done && done[q];
    }
    if (all_done) break;
  }
  return out;
}

// Single-integrand convenience wrapper: F is double operator()(double).
template <class F>
struct Scalar1 {
  const F& f;
  __device__ __forceinline__ void operator()(double x, double (&out)[1]) const {
    out[0] = f(x);
  }
};

template <int NW, class F>
__device__ __forceinline__ double romberg1(const F& f, double a, double b, double tol,
                                           double rtol, int divmax, double* red,
                                           int* level = nullptr) {
  Scalar1<F> w{f};
  RombergOut<1> r = romberg_group<NW, 1>(w, a, b, tol, rtol, divmax, red);
  if (level) *level = r.level[0];
  return r.value[0];
}

// Fixed-node Gauss-Legendre over `npanel` equal panels of [a, b], 16 nodes each,
// strided over the group (used where the reference's Romberg converges to ~1e-8
// on a smooth integrand and the value, not the stopping level, is what matters:
// mass_function.py:227-241, 539-545).  xw: 16 abscissae then 16 weights (LDS or
// global).
template <int NW, class F>
__device__ __forceinline__ double gauss_panels(const F& f, double a, double b,
                                               int npanel, const double* xw,
                                               double* red, int& flip) {
  constexpr int NT = 64 * NW;
  const int gt = (NW == 1) ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  const double w = (b - a) / (double)npanel;
  double part = 0.0;
  for (int idx = gt; idx < 16 * npanel; idx += NT) {
    const int p = idx >> 4, q = idx & 15;
    const double mid = a + w * ((double)p + 0.5);
    part += xw[16 + q] * f(mid + 0.5 * w * xw[q]);
  }
  return 0.5 * w * group_sum<NW>(part, red, flip);
}

}  // namespace chomp
