// chomp_math.h -- fp64 building blocks shared by every kernel of the hot path:
// special functions (Si/Ci, J0, J2), not-a-knot cubic splines, the Eisenstein-Hu
// linear spectrum, mass-function / HOD / NFW pieces.  Everything here is plain
// arithmetic on doubles (CHOMP_HD = __host__ __device__) so the same code can be
// checked on the CPU by tests/hostcheck without a GPU; the product only ever runs
// it inside HIP kernels.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CHOMP_HD __host__ __device__ __forceinline__
#else
#define CHOMP_HD inline
#endif

#if defined(__HIPCC__)
#define CHOMP_DEVICE_CONST __device__ const
#endif
#include "special_tables.h"

namespace chomp {

constexpr double kPi = 3.14159265358979323846;
constexpr double kHalfPi = 1.57079632679489661923;
constexpr double kE = 2.71828182845904523536;

// a b + C for a compile-time constant C, the constant read from a scalar register pair.  The
// compiler's own form of a Horner step is v_fmac_f64 with the constant as the (tied) destination:
// gfx950's VOP3 takes no 64-bit literal, so that constant is first put into a VGPR pair by two
// v_mov_b32 -- each a full-rate issue slot like the FMA itself, i.e. a polynomial step costs
// three vector instructions, and 20 of the 75 of a sigma(R) table node were such moves.
// v_fma_f64 with the addend in SGPRs is one; the two s_mov_b32 that fill them go to the scalar
// unit, which issues beside the vector pipeline.  Same operation, same bits.
// (The hazard recogniser does not look inside an asm statement: a and b must not be the direct
//  result of a transcendental instruction -- v_rcp / v_sqrt / v_rsq_f64 need a wait state before a
//  non-transcendental reader on gfx940+ -- and d must not be read by a DPP instruction right
//  behind it.  Every use here takes products and polynomial partial sums.)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fma_k(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}
#else
CHOMP_HD double fma_k(double a, double b, double c) { return fma(a, b, c); }
#endif

// exp for every unqualified call inside this namespace.  Device code: the ROCm device library's
// algorithm and coefficients (n = rint(x log2 e), Cody-Waite reduction by ln 2 in two FMAs, the
// degree-11 polynomial, ldexp), operation for operation -- the same bits -- with the polynomial's
// addends read from scalar registers (fma_k): 9 of its 11 steps cost the library's inlined code
// three vector instructions each.  The library's two range clamps are left out: v_ldexp_f64
// overflows to inf and underflows to 0 by itself.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double exp(double x) {
  const double n = rint(x * 1.4426950408889634);                       // 0x3ff71547652b82fe
  double r = fma(-6.93147180559945286e-01, n, x);                      // ln 2: 0x3fe62e42fefa39ef
  r = fma(-2.31904681384629956e-17, n, r);                             //       0x3c7abc9e3b39803f
  double p = fma(r, __longlong_as_double(0x3e5ade156a5dcb37LL), __longlong_as_double(0x3e928af3fca7ab0cLL));
  p = fma_k(r, p, __longlong_as_double(0x3ec71dee623fde64LL));
  p = fma_k(r, p, __longlong_as_double(0x3efa01997c89e6b0LL));
  p = fma_k(r, p, __longlong_as_double(0x3f2a01a014761f6eLL));
  p = fma_k(r, p, __longlong_as_double(0x3f56c16c1852b7b0LL));
  p = fma_k(r, p, __longlong_as_double(0x3f81111111122322LL));
  p = fma_k(r, p, __longlong_as_double(0x3fa55555555502a1LL));
  p = fma_k(r, p, __longlong_as_double(0x3fc5555555555511LL));
  p = fma_k(r, p, __longlong_as_double(0x3fe000000000000bLL));
  p = fma(r, p, 1.0);
  p = fma(r, p, 1.0);
  return ldexp(p, (int)n);
}
#else
CHOMP_HD double exp(double x) { return ::exp(x); }
#endif

// ---------------------------------------------------------------------------
// Special-function tables as one POD block: uploaded once per context, staged in
// LDS by the kernels that need them.
// ---------------------------------------------------------------------------
// sin x and cos x together, |error| <~ 1 ulp for |x| up to ~1e9 (absolute error grows
// like 1e-33 |x| beyond): Cody-Waite reduction by pi/2 in two FMA steps, then the
// fdlibm kernel polynomials on [-pi/4, pi/4].  ~55 instructions against ~160 of the
// library routine with its Payne-Hanek branch; the sigma(R) and NFW integrands spend
// most of their time here.
// ---------------------------------------------------------------------------
// The leading coefficients of the two polynomials below, for a caller with a loop around its
// sincos: a Horner chain's first step has two constants and only one of them can come from
// scalar registers, the other costs two v_mov_b32 per evaluation -- unless it sits in vector
// registers the compiler cannot re-materialise (the empty asm), from outside the loop.
struct SinCosLead {
  double s0, c0;
};
CHOMP_HD SinCosLead sincos_lead() {
  SinCosLead L{1.58969099521155010221e-10, -1.13596475577881948265e-11};
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(L.s0), "+v"(L.c0));
#endif
  return L;
}
// (the reduced argument's sine and cosine and the quadrant)
CHOMP_HD unsigned sincos_reduced(double x, double* sp, double* cp,
                                 const SinCosLead lead = SinCosLead{1.58969099521155010221e-10,
                                                                    -1.13596475577881948265e-11}) {
  const double n = rint(x * 6.36619772367581382433e-01);
  double r = fma(-n, 1.57079632679489655800e+00, x);
  r = fma(-n, 6.12323399573676603587e-17, r);
  // quadrant = n mod 4 from the low mantissa bits of n + 1.5 * 2^52 (no integer overflow)
  const double shifted = n + 6755399441055744.0;
  unsigned long long bits;
  __builtin_memcpy(&bits, &shifted, sizeof bits);
  const double z = r * r;
  double ps = lead.s0;
  ps = fma_k(ps, z, -2.50507602534068634195e-08);
  ps = fma_k(ps, z, 2.75573137070700676789e-06);
  ps = fma_k(ps, z, -1.98412698298579493134e-04);
  ps = fma_k(ps, z, 8.33333333332248946124e-03);
  ps = fma_k(ps, z, -1.66666666666666324348e-01);
  *sp = fma(r * z, ps, r);
  double pc = lead.c0;
  pc = fma_k(pc, z, 2.08757232129817482790e-09);
  pc = fma_k(pc, z, -2.75573143513906633035e-07);
  pc = fma_k(pc, z, 2.48015872894767294178e-05);
  pc = fma_k(pc, z, -1.38888888888741095749e-03);
  pc = fma_k(pc, z, 4.16666666666666019037e-02);
  *cp = fma(z * z, pc, fma(-0.5, z, 1.0));
  return (unsigned)bits;
}
CHOMP_HD void fast_sincos(double x, double* sp, double* cp) {
  double s, c;
  const unsigned q = sincos_reduced(x, &s, &c);
  const bool swap = (q & 1u) != 0;
  double so = swap ? c : s, co = swap ? s : c;
  if (q & 2u) so = -so;
  if ((q + 1u) & 2u) co = -co;
  *sp = so;
  *cp = co;
}
// +-(sin x - x cos x), the numerator of the top-hat window, for callers that square it: the sign
// (the quadrant's second bit) is not worked out -- in an odd quadrant (sin, cos) = +-(c, -s) of
// the reduced argument, in an even one +-(s, c), and negating both inputs of the FMA negates its
// result exactly.  Nine vector instructions fewer per node than through fast_sincos.
CHOMP_HD double tophat_numer_pm(double x, const SinCosLead lead = SinCosLead{1.58969099521155010221e-10,
                                                                           -1.13596475577881948265e-11}) {
  double s, c;
  const unsigned q = sincos_reduced(x, &s, &c, lead);
  const double even = fma(-x, c, s), odd = fma(x, s, c);
  return (q & 1u) ? odd : even;
}
// +-(sin x, cos x) with ONE unknown sign for both (as above, for even functions of the pair)
CHOMP_HD void fast_sincos_pm(double x, double* sp, double* cp) {
  double s, c;
  const unsigned q = sincos_reduced(x, &s, &c);
  const bool swap = (q & 1u) != 0;
  *sp = swap ? c : s;
  *cp = swap ? -s : c;
}

// ---------------------------------------------------------------------------
struct SiCiTab {
  double si_ser[CHOMP_SICI_NSER];
  double ci_ser[CHOMP_SICI_NSER];
  // (rows padded by one double: lanes in different rows then hit different LDS banks --
  //  16-double rows put rows of equal parity on the same banks, a 4-way conflict)
  double f[CHOMP_FG_NINT][CHOMP_FG_NCOEF + 1];
  double g[CHOMP_FG_NINT][CHOMP_FG_NCOEF + 1];
};
struct BesselTab {
  double cheb[8][CHOMP_J0_NCOEF];   // J0 and J2 have the same table shape
  double p[CHOMP_J0_NHANKEL];
  double xq[CHOMP_J0_NHANKEL];
};
static_assert(CHOMP_J0_NCOEF == CHOMP_J2_NCOEF, "table shape");
static_assert(CHOMP_J0_NHANKEL == CHOMP_J2_NHANKEL, "table shape");

inline void fill_tables(SiCiTab* s, BesselTab* j0, BesselTab* j2) {
  for (int i = 0; i < CHOMP_SICI_NSER; ++i) {
    s->si_ser[i] = CHOMP_SI_SER[i];
    s->ci_ser[i] = CHOMP_CI_SER[i];
  }
  for (int j = 0; j < CHOMP_FG_NINT; ++j) {
    for (int i = 0; i < CHOMP_FG_NCOEF; ++i) {
      s->f[j][i] = CHOMP_AUX_F[j][i];
      s->g[j][i] = CHOMP_AUX_G[j][i];
    }
    s->f[j][CHOMP_FG_NCOEF] = 0.0;
    s->g[j][CHOMP_FG_NCOEF] = 0.0;
  }
  for (int j = 0; j < 8; ++j)
    for (int i = 0; i < CHOMP_J0_NCOEF; ++i) {
      j0->cheb[j][i] = CHOMP_J0_CHEB[j][i];
      j2->cheb[j][i] = CHOMP_J2_CHEB[j][i];
    }
  for (int i = 0; i < CHOMP_J0_NHANKEL; ++i) {
    j0->p[i] = CHOMP_J0_P[i];
    j0->xq[i] = CHOMP_J0_XQ[i];
    j2->p[i] = CHOMP_J2_P[i];
    j2->xq[i] = CHOMP_J2_XQ[i];
  }
}

// Clenshaw sum of c[0..n-1] in Chebyshev polynomials of t in [-1, 1].
template <int N>
CHOMP_HD double cheb_eval(const double* c, double t) {
  double b1 = 0.0, b2 = 0.0;
  const double t2 = 2.0 * t;
#pragma unroll
  for (int k = N - 1; k >= 1; --k) {
    const double b0 = fma(t2, b1, c[k] - b2);
    b2 = b1;
    b1 = b0;
  }
  return fma(t, b1, c[0] - b2);
}

// Si(x), Ci(x) for x > 0, given s = sin x and c = cos x (only used when x >= 4).
// Replaces scipy.special.sici at halo.py:578-579.
CHOMP_HD void sici_sc(double x, double s, double c, const SiCiTab& T, double* si,
                      double* ci) {
  if (x < 4.0) {
    const double x2 = x * x;
    double ps = T.si_ser[CHOMP_SICI_NSER - 1], pc = T.ci_ser[CHOMP_SICI_NSER - 1];
#pragma unroll
    for (int k = CHOMP_SICI_NSER - 2; k >= 0; --k) {
      ps = fma(ps, x2, T.si_ser[k]);
      pc = fma(pc, x2, T.ci_ser[k]);
    }
    *si = x * ps;
    *ci = CHOMP_EULER_GAMMA + log(x) + x2 * pc;
  } else {
    const double u = 4.0 / x;
    int j = (int)(8.0 * u);
    j = j > 7 ? 7 : j;
    const double t = 16.0 * u - (double)(2 * j + 1);
    const double F = cheb_eval<CHOMP_FG_NCOEF>(T.f[j], t);
    const double G = cheb_eval<CHOMP_FG_NCOEF>(T.g[j], t);
    // (x f(x) = F, x^2 g(x) = G; with 1 / x = u / 4 at hand the two further divisions -- ten
    //  instructions each in fp64 -- are multiplications)
    const double f = F * (0.25 * u), g = G * ((0.0625 * u) * u);
    *si = kHalfPi - f * c - g * s;
    *ci = f * s - g * c;
  }
}

// Same with ln x supplied by the caller (saves the log in the series branch).
CHOMP_HD void sici_sc_ln(double x, double ln_x, double s, double c, const SiCiTab& T,
                         double* si, double* ci) {
  if (x < 4.0) {
    const double x2 = x * x;
    double ps = T.si_ser[CHOMP_SICI_NSER - 1], pc = T.ci_ser[CHOMP_SICI_NSER - 1];
#pragma unroll
    for (int k = CHOMP_SICI_NSER - 2; k >= 0; --k) {
      ps = fma(ps, x2, T.si_ser[k]);
      pc = fma(pc, x2, T.ci_ser[k]);
    }
    *si = x * ps;
    *ci = CHOMP_EULER_GAMMA + ln_x + x2 * pc;
  } else {
    const double u = 4.0 / x;
    int j = (int)(8.0 * u);
    j = j > 7 ? 7 : j;
    const double t = 16.0 * u - (double)(2 * j + 1);
    const double F = cheb_eval<CHOMP_FG_NCOEF>(T.f[j], t);
    const double G = cheb_eval<CHOMP_FG_NCOEF>(T.g[j], t);
    // (x f(x) = F, x^2 g(x) = G; with 1 / x = u / 4 at hand the two further divisions -- ten
    //  instructions each in fp64 -- are multiplications)
    const double f = F * (0.25 * u), g = G * ((0.0625 * u) * u);
    *si = kHalfPi - f * c - g * s;
    *ci = f * s - g * c;
  }
}

CHOMP_HD void sici(double x, const SiCiTab& T, double* si, double* ci) {
  double s = 0.0, c = 1.0;
  if (x >= 4.0) fast_sincos(x, &s, &c);
  sici_sc(x, s, c, T, si, ci);
}

// J_n(x), n = 0 or 2 (ORDER selects the Hankel phase), x >= 0.
// Replaces scipy.special.j0 / jn(2, .) at kernel.py:712, 839.
template <int ORDER>
CHOMP_HD double bessel_j(double x, const BesselTab& T) {
  x = fabs(x);
  if (x < 32.0) {
    int j = (int)(0.25 * x);
    j = j > 7 ? 7 : j;
    const double t = 0.5 * x - (double)(2 * j + 1);
    return cheb_eval<CHOMP_J0_NCOEF>(T.cheb[j], t);
  }
  const double r = 32.0 / x;
  const double t = 2.0 * r * r - 1.0;
  const double P = cheb_eval<CHOMP_J0_NHANKEL>(T.p, t);
  const double Q = cheb_eval<CHOMP_J0_NHANKEL>(T.xq, t) / x;
  const double chi = x - (0.5 * ORDER + 0.25) * kPi;
  return sqrt(2.0 / (kPi * x)) * (P * cos(chi) - Q * sin(chi));
}

// ---------------------------------------------------------------------------
// Not-a-knot cubic interpolating spline == FITPACK InterpolatedUnivariateSpline
// (k=3) used at ~25 reference sites (e.g. mass_function.py:217-220,
// halo.py:918, 985, kernel.py:311, 646).  Piecewise-polynomial form:
//   s(x) = c[4i] + c[4i+1] d + c[4i+2] d^2 + c[4i+3] d^3,  d = x - x[i],
// interval i = 0..n-2; outside the knots the end pieces extrapolate, as FITPACK
// (ext=0) does.
// ---------------------------------------------------------------------------
// Serial build (one thread).  `work` needs 2n doubles.  n >= 4.
CHOMP_HD void spline_build(const double* x, const double* y, int n, double* c,
                           double* work) {
  double* diag = work;      // modified diagonal
  double* s = work + n;     // rhs -> slopes
  const double h0 = x[1] - x[0], h1 = x[2] - x[1];
  const double d0 = (y[1] - y[0]) / h0, d1 = (y[2] - y[1]) / h1;
  // row 0 (not-a-knot): h1 s0 + (h0+h1) s1 = rhs
  double b0 = h1, c0 = h0 + h1;
  double r0 = ((3.0 * h0 + 2.0 * h1) * h1 * d0 + h0 * h0 * d1) / (h0 + h1);
  diag[0] = b0;
  s[0] = r0;
  double cprev = c0;   // super-diagonal of the previous row
  for (int i = 1; i < n - 1; ++i) {
    const double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
    const double dl = (y[i] - y[i - 1]) / hl, dr = (y[i + 1] - y[i]) / hr;
    const double a = hr;                   // sub-diagonal
    const double b = 2.0 * (hl + hr);      // diagonal
    const double cc = hl;                  // super-diagonal
    const double r = 3.0 * (hr * dl + hl * dr);
    const double m = a / diag[i - 1];
    diag[i] = b - m * cprev;
    s[i] = r - m * s[i - 1];
    cprev = cc;
  }
  {
    const int i = n - 1;
    const double hl = x[n - 1] - x[n - 2], hll = x[n - 2] - x[n - 3];
    const double dl = (y[n - 1] - y[n - 2]) / hl, dll = (y[n - 2] - y[n - 3]) / hll;
    const double a = hl + hll;             // sub-diagonal
    const double b = hll;                  // diagonal
    const double r = (hl * hl * dll + (2.0 * hll + 3.0 * hl) * hll * dl) / (hll + hl);
    const double m = a / diag[i - 1];
    diag[i] = b - m * cprev;
    s[i] = r - m * s[i - 1];
  }
  // back substitution; super-diagonals: row 0 -> c0, row i -> x[i]-x[i-1]
  s[n - 1] = s[n - 1] / diag[n - 1];
  for (int i = n - 2; i >= 0; --i) {
    const double sup = (i == 0) ? c0 : (x[i] - x[i - 1]);
    s[i] = (s[i] - sup * s[i + 1]) / diag[i];
  }
  for (int i = 0; i < n - 1; ++i) {
    const double h = x[i + 1] - x[i];
    const double d = (y[i + 1] - y[i]) / h;
    c[4 * i + 0] = y[i];
    c[4 * i + 1] = s[i];
    c[4 * i + 2] = (3.0 * d - 2.0 * s[i] - s[i + 1]) / h;
    c[4 * i + 3] = (s[i] + s[i + 1] - 2.0 * d) / (h * h);
  }
}

// --- The same spline, built in parallel -------------------------------------
// Row i of the not-a-knot slope system  A s = r  (see spline_build): sub-, main and
// super-diagonal and right-hand side.
CHOMP_HD void spline_row(const double* x, const double* y, int n, int i, double* a,
                         double* b, double* c, double* r) {
  if (i == 0) {
    const double h0 = x[1] - x[0], h1 = x[2] - x[1];
    const double d0 = (y[1] - y[0]) / h0, d1 = (y[2] - y[1]) / h1;
    *a = 0.0; *b = h1; *c = h0 + h1;
    *r = ((3.0 * h0 + 2.0 * h1) * h1 * d0 + h0 * h0 * d1) / (h0 + h1);
  } else if (i == n - 1) {
    const double hl = x[n - 1] - x[n - 2], hll = x[n - 2] - x[n - 3];
    const double dl = (y[n - 1] - y[n - 2]) / hl, dll = (y[n - 2] - y[n - 3]) / hll;
    *a = hl + hll; *b = hll; *c = 0.0;
    *r = (hl * hl * dll + (2.0 * hll + 3.0 * hl) * hll * dl) / (hll + hl);
  } else {
    const double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
    const double dl = (y[i] - y[i - 1]) / hl, dr = (y[i + 1] - y[i]) / hr;
    *a = hr; *b = 2.0 * (hl + hr); *c = hl;
    *r = 3.0 * (hr * dl + hl * dr);
  }
}

// One parallel-cyclic-reduction step of stride s for row i: reads (a,b,c,d)[in],
// writes [out].  After ceil(log2 n) steps x_i = d_i / b_i.
CHOMP_HD void pcr_step(int n, int i, int s, const double* ai, const double* bi,
                       const double* ci, const double* di, double* ao, double* bo,
                       double* co, double* dout) {
  const int lo = i - s, hi = i + s;
  const double al = lo >= 0 ? -ai[i] / bi[lo] : 0.0;
  const double ga = hi < n ? -ci[i] / bi[hi] : 0.0;
  double b = bi[i], d = di[i], a = 0.0, c = 0.0;
  if (lo >= 0) { b += al * ci[lo]; d += al * di[lo]; a = al * ai[lo]; }
  if (hi < n) { b += ga * ai[hi]; d += ga * di[hi]; c = ga * ci[hi]; }
  ao[i] = a; bo[i] = b; co[i] = c; dout[i] = d;
}

// pp coefficients of interval i from the slopes s.
CHOMP_HD void spline_coef(const double* x, const double* y, const double* s, int i,
                          double* c) {
  const double h = x[i + 1] - x[i];
  const double d = (y[i + 1] - y[i]) / h;
  c[4 * i + 0] = y[i];
  c[4 * i + 1] = s[i];
  c[4 * i + 2] = (3.0 * d - 2.0 * s[i] - s[i + 1]) / h;
  c[4 * i + 3] = (s[i] + s[i + 1] - 2.0 * d) / (h * h);
}

CHOMP_HD double pp_poly(const double* c, int i, double d) {
  const double* q = c + 4 * i;
  return fma(fma(fma(q[3], d, q[2]), d, q[1]), d, q[0]);
}

// Evaluate on arbitrary (increasing) knots: binary search for the interval.
CHOMP_HD double spline_eval(const double* x, const double* c, int n, double xv) {
  int lo = 0, hi = n - 2;   // interval index range
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (xv >= x[mid]) lo = mid; else hi = mid - 1;
  }
  return pp_poly(c, lo, xv - x[lo]);
}

// Evaluate on uniform knots x_i = x0 + i*dx.
CHOMP_HD double spline_eval_uniform(double x0, double dx, const double* c, int n,
                                    double xv) {
  int i = (int)floor((xv - x0) / dx);
  i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
  return pp_poly(c, i, xv - (x0 + dx * (double)i));
}

// ---------------------------------------------------------------------------
// Quintic interpolating spline (FITPACK InterpolatedUnivariateSpline(k=5)) and its
// first two derivatives at one point: what HaloFit needs for n_eff and C
// (halo.py:1289-1294).  Knots as FITPACK's fpcurf chooses them for s = 0, odd k:
// 6-fold end knots and the data sites x[3] .. x[n-4] inside.
// work: (n + 6) + 11 n + n doubles.  n >= 7.
// ---------------------------------------------------------------------------
// B-spline basis of degree p (<= 5) at x in knot span m: N[r] = B_{m-p+r,p}(x).
CHOMP_HD void bspline_basis(const double* t, int m, int p, double x, double* N) {
  double left[6], right[6];
  N[0] = 1.0;
  for (int j = 1; j <= p; ++j) {
    left[j] = x - t[m + 1 - j];
    right[j] = t[m + j] - x;
    double saved = 0.0;
    for (int r = 0; r < j; ++r) {
      const double temp = N[r] / (right[r + 1] + left[j - r]);
      N[r] = saved + right[r + 1] * temp;
      saved = left[j - r] * temp;
    }
    N[j] = saved;
  }
}

CHOMP_HD void quintic_derivs(const double* x, const double* y, int n, double xq,
                             double* work, double* d1, double* d2) {
  double* t = work;               // [n + 6]
  double* ab = t + (n + 6);       // [n][11] band storage, column j at j - i + 5
  double* c = ab + 11 * n;        // [n]
  for (int i = 0; i < 6; ++i) { t[i] = x[0]; t[n + i] = x[n - 1]; }
  for (int i = 0; i < n - 6; ++i) t[6 + i] = x[3 + i];
  for (int i = 0; i < 11 * n; ++i) ab[i] = 0.0;
  double N[6];
  int m = 5;
  for (int i = 0; i < n; ++i) {
    while (m < n - 1 && x[i] >= t[m + 1]) ++m;
    bspline_basis(t, m, 5, x[i], N);
    for (int r = 0; r < 6; ++r) {
      const int j = m - 5 + r;
      ab[11 * i + (j - i + 5)] = N[r];
    }
    c[i] = y[i];
  }
  // banded elimination without pivoting (the collocation matrix is totally positive)
  for (int col = 0; col < n; ++col) {
    const double piv = ab[11 * col + 5];
    const int rmax = col + 5 < n - 1 ? col + 5 : n - 1;
    for (int r = col + 1; r <= rmax; ++r) {
      const double a = ab[11 * r + (col - r + 5)];
      if (a == 0.0) continue;
      const double f = a / piv;
      const int jmax = col + 5 < n - 1 ? col + 5 : n - 1;
      for (int j = col; j <= jmax; ++j) {
        const int kr = j - r + 5;
        if (kr >= 0 && kr <= 10) ab[11 * r + kr] -= f * ab[11 * col + (j - col + 5)];
      }
      c[r] -= f * c[col];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double sum = c[i];
    const int jmax = i + 5 < n - 1 ? i + 5 : n - 1;
    for (int j = i + 1; j <= jmax; ++j) sum -= ab[11 * i + (j - i + 5)] * c[j];
    c[i] = sum / ab[11 * i + 5];
  }
  // derivatives at xq
  m = 5;
  while (m < n - 1 && xq >= t[m + 1]) ++m;
  double c1[6], c2[5];
  for (int r = 1; r <= 5; ++r) {          // j = m-5+r = m-4 .. m
    const int j = m - 5 + r;
    c1[r] = 5.0 * (c[j] - c[j - 1]) / (t[j + 5] - t[j]);
  }
  for (int r = 2; r <= 5; ++r) {          // j = m-3 .. m
    const int j = m - 5 + r;
    c2[r - 1] = 4.0 * (c1[r] - c1[r - 1]) / (t[j + 4] - t[j]);
  }
  bspline_basis(t, m, 4, xq, N);
  double s1 = 0.0;
  for (int r = 0; r < 5; ++r) s1 += c1[r + 1] * N[r];
  bspline_basis(t, m, 3, xq, N);
  double s2 = 0.0;
  for (int r = 0; r < 4; ++r) s2 += c2[r + 1] * N[r];
  *d1 = s1;
  *d2 = s2;
}

// numpy.linspace(a, b, n)[i]
// (numpy forms arange(n)*step + start with two roundings: no FMA contraction here,
// so the knot grids are bit-identical to the reference's.)
CHOMP_HD double linspace_at(double a, double b, int n, int i) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  if (i == n - 1) return b;
  const double step = (b - a) / (double)(n - 1);
#if defined(__clang__)
  const double prod = (double)i * step;
#else
  volatile double prod = (double)i * step;
#endif
  return prod + a;
}

// ---------------------------------------------------------------------------
// Per-epoch state (one cosmology.SingleEpoch + MassFunction + Halo).
// ---------------------------------------------------------------------------
struct Epoch {
  // inputs (cosmo_dict, redshift)
  double om0, ob0, ol0, or0, tcmb, h, sigma8, ns, z;
  // derived background (cosmology.py:39-119, 375-447)
  double H0, delta_H, growth_norm, growth, sigma_norm, chi;
  double E0z, omega_m_z, omega_l_z, delta_c, delta_v, rho_bar;
  // Eisenstein-Hu constants (cosmology.py:460-466)
  double eh_theta, eh_s, eh_alpha, eh_omh;
  // Delta^2(k) = amp * exp((3+n) (ln k - ln H0)) * T(k)^2
  double amp, ln_H0;
  double k_min, k_max;
  // ln k_min, ln k_max (sigma_r's limits where the range is the table's) and the uniform ln k
  // grid of the interpolated integrand, [ln(k_min / 100), ln(100 k_max)] in kGTabN steps: what
  // every sigma(R) integral started with two logarithms and three divisions for
  double ln_k_min, ln_k_max, gtab_xlo, gtab_dx, gtab_inv_dx, pad_k;
  int flat, open, closed, mf_kind;
  // mass function (mass_function.py)
  double stq, st_a, mf_delta_v;
  double ln_mass_min, ln_mass_max, nu_min, nu_max, m_star, f_norm, bias_norm;
  double t_alpha, t_beta, t_gamma, t_phi, t_eta;     // Tinker f(nu)
  double tb_A, tb_a, tb_C, tb_dca;                   // Tinker bias constants
  int n_search, cosmo_slot;        // cosmo_slot: index of this epoch's cosmology-only tables
  // halo profile (halo.py:71-83, 873-902)
  double c0, beta, prof_delta_v, ln_rv_const, ln_c_const;
  // HOD (hod.py:156-186)
  double hod_log_M_min, hod_sigma, hod_log_M_0, hod_log_M_1p, hod_alpha;
  double hod_first_zero, hod_second_zero, hod_safe_norm, hod_M0, hod_M1p;
  double ln_nu_lo_first, ln_nu_lo_second;
  double n_bar_over_rho_bar, n_bar;
  // HaloFit (halo.py:1261-1319)
  double hf_f1, hf_f2, hf_f3, hf_k_s, hf_n_eff, hf_C, hf_a_n, hf_b_n, hf_c_n,
      hf_gamma_n, hf_alpha_n, hf_beta_n, hf_mu_n, hf_nu_n;
  double ln_st_a, ln_t_beta;       // logs used by mf_node
  // E&H transfer function with baryon wiggles (SingleEpoch(with_bao=True),
  // cosmology.py:474-538): flag and the k-independent constants
  int with_bao, pad_bao;
  double bao_hs, bao_q_scale, bao_ksilk_h, bao_alpha_b, bao_beta_b, bao_alpha_c, bao_beta_c,
      bao_beta_node, bao_s, bao_ObO, bao_OcO;
};
static_assert(sizeof(Epoch) % 16 == 0, "Epoch must keep LDS carve-ups 16-byte aligned");

// cosmology.py:165-178
CHOMP_HD double E0_of(double om0, double ol0, double or0, double z) {
  const double a = 1.0 / (1.0 + z);
  return ol0 + om0 / (a * a * a) + or0 / (a * a * a * a);
}

// cosmology.py:215-231 (always returned by growth_factor_eval, :326)
CHOMP_HD double growth_approx(double om0, double ol0, double a) {
  const double om = om0 / (a * a * a);
  const double denom = ol0 + om;
  const double Omega_m = om / denom;
  const double Omega_L = ol0 / denom;
  const double coeff = 5.0 * Omega_m / (2.0 / a);
  const double term1 = Omega_m * (4.0 / 7.0);
  const double term3 = (1.0 + 0.5 * Omega_m) * (1.0 + Omega_L / 70.0);
  return coeff / (term1 - Omega_L + term3);
}

CHOMP_HD void bao_constants(Epoch& e);
constexpr int kGTabIntervals = 8192;      // (= kGTabN of chomp_mass_kernels.h)
CHOMP_HD void epoch_k_range(Epoch& e) {
  e.ln_k_min = log(e.k_min);
  e.ln_k_max = log(e.k_max);
  e.gtab_xlo = log(e.k_min / 100.0);
  e.gtab_dx = (log(e.k_max * 100.0) - e.gtab_xlo) / (double)kGTabIntervals;
  e.gtab_inv_dx = 1.0 / e.gtab_dx;
  e.pad_k = 0.0;
}

// SingleEpoch.__init__ minus the two integrals (chi, sigma_norm).
CHOMP_HD void epoch_background(Epoch& e, double cosmo_precision, double k_min,
                               double k_max, int with_bao = 0) {
  if (e.z < 0.0) e.z = 0.0;
  e.H0 = 100.0 / (2.998 * 100000.0);
  e.ln_H0 = log(e.H0);
  const double tot = e.om0 + e.ol0 + e.or0;
  e.flat = (tot <= 1.0 + cosmo_precision && tot >= 1.0 - cosmo_precision) ? 1 : 0;
  e.open = (tot <= 1.0 - cosmo_precision) ? 1 : 0;
  e.closed = (tot > 1.0 + cosmo_precision) ? 1 : 0;
  e.k_min = k_min;
  e.k_max = k_max;
  epoch_k_range(e);
  e.delta_H = 1.94e-5 * pow(e.om0, -0.785 - 0.05 * log(e.om0)) *
              exp(-0.95 * (e.ns - 1.0) - 0.169 * (e.ns - 1.0) * (e.ns - 1.0));
  e.growth_norm = growth_approx(e.om0, e.ol0, 1.0);
  e.growth = growth_approx(e.om0, e.ol0, 1.0 / (1.0 + e.z)) / e.growth_norm;
  e.E0z = E0_of(e.om0, e.ol0, e.or0, e.z);
  const double opz = 1.0 + e.z;
  e.omega_m_z = e.om0 * (opz * opz * opz) / e.E0z;
  e.omega_l_z = e.ol0 / e.E0z;
  double dc = 0.15 * pow(12.0 * kPi, 2.0 / 3.0);      // cosmology.py:393-407
  double dv = 178.0;                                  // :409-423
  if (e.open) {
    dc *= pow(e.omega_m_z, 0.0185);
    dv /= pow(e.omega_m_z, 0.7);
  }
  if (e.flat && e.om0 < 1.0001) {
    dc *= pow(e.omega_m_z, 0.0055);
    dv /= pow(e.omega_m_z, 0.55);
  }
  e.delta_c = dc;
  e.delta_v = dv / e.growth;
  e.rho_bar = (1.879 / (1.989) * (3.086 * 3.086 * 3.086) * 1e10 * e.E0z) *
              e.omega_m_z;                            // :437-447
  // Eisenstein-Hu constants; (Omb2)**(3/4) is **0 under Python 2 -> sqrt(11).
  const double Omh2 = e.om0 * e.h * e.h;
  const double ratio = e.ob0 / e.om0;
  e.eh_theta = e.tcmb / 2.7;
  e.eh_s = 44.5 * log(9.83 / Omh2) / sqrt(1.0 + 10.0 * 1.0);
  e.eh_alpha = 1.0 - 0.328 * log(431.0 * Omh2) * ratio +
               0.38 * log(22.3 * Omh2) * ratio * ratio;
  e.eh_omh = e.om0 * e.h;
  e.sigma_norm = 1.0;
  e.amp = e.delta_H * e.delta_H / e.h * (e.growth * e.growth);
  e.with_bao = with_bao ? 1 : 0;
  e.pad_bao = 0;
  if (e.with_bao) bao_constants(e);
}

// What the cosmology-only integrals (sigma node table, sigma_8, ln S(R)) need of an epoch
// record: the transfer-function constants, n_s, H0 and the k limits -- with amp = 1 and
// sigma_norm = 1, so that SigmaIntegrandT integrates the bare shape (k/H0)^(3+n) T^2 W^2.
// A third of epoch_background's dependent instruction chain (no delta_H, delta_c, delta_v,
// rho_bar, growth): it sits at the head of every block of k_sigma_nodes.
CHOMP_HD void epoch_shape_only(Epoch& e, double k_min, double k_max, int with_bao) {
  e.H0 = 100.0 / (2.998 * 100000.0);
  e.ln_H0 = log(e.H0);
  e.k_min = k_min;
  e.k_max = k_max;
  // (no epoch_k_range: nothing that integrates the bare shape goes through sigma2_block)
  const double Omh2 = e.om0 * e.h * e.h;
  const double ratio = e.ob0 / e.om0;
  e.eh_theta = e.tcmb / 2.7;
  e.eh_s = 44.5 * log(9.83 / Omh2) / sqrt(1.0 + 10.0 * 1.0);
  e.eh_alpha = 1.0 - 0.328 * log(431.0 * Omh2) * ratio +
               0.38 * log(22.3 * Omh2) * ratio * ratio;
  e.eh_omh = e.om0 * e.h;
  e.sigma_norm = 1.0;
  e.amp = 1.0;
  e.with_bao = with_bao ? 1 : 0;
  e.pad_bao = 0;
  if (e.with_bao) bao_constants(e);
}

// Constants of the wiggle transfer function (cosmology.py:484-527).
CHOMP_HD void bao_constants(Epoch& e) {
  const double theta = e.tcmb / 2.7;
  const double Ob = e.ob0, Om = e.om0, Oc = Om - Ob, h = e.h;
  const double Obh2 = Ob * h * h, Oh2 = Om * h * h, ObO = Ob / Om;
  const double th2 = theta * theta, th4 = th2 * th2;
  const double zeq = 2.5e4 * Oh2 / th4;
  const double keq = 7.46e-2 * Oh2 / th2;
  double b1 = 0.313 * pow(Oh2, -0.419) * (1.0 + 0.607 * pow(Oh2, 0.674));
  double b2 = 0.238 * pow(Oh2, 0.223);
  const double zd = 1291.0 * (pow(Oh2, 0.251) / (1.0 + 0.659 * pow(Oh2, 0.828))) *
                    (1.0 + b1 * pow(Obh2, b2));
  const double Req = 31.5 * Obh2 / th4 * (1000.0 / zeq);
  const double Rd = 31.5 * Obh2 / th4 * (1000.0 / zd);
  const double s = (2.0 / (3.0 * keq)) * sqrt(6.0 / Req) *
                   log((sqrt(1.0 + Rd) + sqrt(Rd + Req)) / (1.0 + sqrt(Req)));
  const double kSilk = 1.6 * pow(Obh2, 0.52) * pow(Oh2, 0.73) * (1.0 + pow(10.4 * Oh2, -0.95));
  const double y = (1.0 + zeq) / (1.0 + zd);
  const double G = y * (-6.0 * sqrt(1.0 + y) +
                        (2.0 + 3.0 * y) * log((sqrt(1.0 + y) + 1.0) / (sqrt(1.0 + y) - 1.0)));
  e.bao_alpha_b = 2.07 * keq * s * pow(1.0 + Rd, -3.0 / 4.0) * G;
  e.bao_beta_b = 0.5 + ObO + (3.0 - 2.0 * ObO) * sqrt((17.2 * Oh2) * (17.2 * Oh2) + 1.0);
  const double a1 = pow(46.9 * Oh2, 0.670) * (1.0 + pow(32.1 * Oh2, -0.532));
  const double a2 = pow(12.0 * Oh2, 0.424) * (1.0 + pow(45.0 * Oh2, -0.582));
  e.bao_alpha_c = pow(a1, -ObO) * pow(a2, -(ObO * ObO * ObO));
  b1 = 0.944 / (1.0 + pow(458.0 * Oh2, -0.708));
  b2 = pow(0.395 * Oh2, -0.0266);
  e.bao_beta_c = 1.0 / (1.0 + b1 * (pow(Oc / Om, b2) - 1.0));
  e.bao_beta_node = 8.41 * pow(Oh2, 0.435);
  e.bao_s = s;
  e.bao_hs = h * s;                       // ks = k h s
  e.bao_q_scale = h / (13.41 * keq);      // q = k h / (13.41 keq)
  e.bao_ksilk_h = h / kSilk;              // k h / kSilk
  e.bao_ObO = ObO;
  e.bao_OcO = Oc / Om;
}

// cosmology.py:474-538.
CHOMP_HD double eh_bao_transfer(const Epoch& e, double k) {
  const double ks = k * e.bao_hs;
  const double q = k * e.bao_q_scale;
  const double q2 = q * q;
  const double c386 = 386.0 / (1.0 + 69.9 * pow(q, 1.08));
  const double ks54 = ks / 5.4, f = 1.0 / (1.0 + (ks54 * ks54) * (ks54 * ks54));
  const double Lc = log(kE + 1.8 * e.bao_beta_c * q);
  const double T_c1 = Lc / (Lc + (14.2 + c386) * q2);
  const double T_ca = Lc / (Lc + (14.2 / e.bao_alpha_c + c386) * q2);
  const double Tc = f * T_c1 + (1.0 - f) * T_ca;
  const double L1 = log(kE + 1.8 * q);
  const double T_11 = L1 / (L1 + (14.2 + c386) * q2);
  const double bn = e.bao_beta_node / ks;
  const double stilde = e.bao_s / cbrt(1.0 + bn * bn * bn);
  const double ks52 = ks / 5.2;
  const double Tb1 = T_11 / (1.0 + ks52 * ks52);
  const double bb = e.bao_beta_b / ks;
  const double Tb2 = (e.bao_alpha_b / (1.0 + bb * bb * bb)) * exp(-pow(k * e.bao_ksilk_h, 1.4));
  const double x = k * stilde;                          // numpy.sinc(x / pi) = sin x / x
  const double sinc = x == 0.0 ? 1.0 : sin(x) / x;
  return e.bao_ObO * (sinc * (Tb1 + Tb2)) + e.bao_OcO * Tc;
}

// ln x for a normal positive x to about 1 ulp (|error| < 3e-16 |ln x| + 2e-16) in ~35
// instructions: x = 2^e m, m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m-1)/(m+1)).
// The library logarithm costs ~100; Stage E takes four logarithms per pair of k.
CHOMP_HD double fast_log(double x) {
  int e;
  double m = frexp(x, &e);                               // [0.5, 1)
  if (m < 0.70710678118654752440) { m += m; --e; }
  const double s = (m - 1.0) / (m + 1.0);                // |s| <= 0.1716
  const double z = s * s;
  double p = 1.0 / 23.0;
  p = fma_k(p, z, 1.0 / 21.0);
  p = fma_k(p, z, 1.0 / 19.0);
  p = fma_k(p, z, 1.0 / 17.0);
  p = fma_k(p, z, 1.0 / 15.0);
  p = fma_k(p, z, 1.0 / 13.0);
  p = fma_k(p, z, 1.0 / 11.0);
  p = fma_k(p, z, 1.0 / 9.0);
  p = fma_k(p, z, 1.0 / 7.0);
  p = fma_k(p, z, 1.0 / 5.0);
  p = fma_k(p, z, 1.0 / 3.0);
  const double s2 = s + s;
  const double lm = fma(s2 * z, p, s2);
  const double de = (double)e;
  // ln 2 split so that e * hi is exact
  return fma(de, 6.93147180369123816490e-01, fma(de, 1.90821492927058770002e-10, lm));
}

// cosmology.py:449-472, arranged with two divisions instead of four and the ~35-instruction
// logarithm above (the library's is ~95): q as one quotient, T = L0 D / (L0 D + N q^2) with
// D = 1 + 62.5 q, N = 14.2 D + 731 -- the reference's L0 / (L0 + C0 q^2), C0 = 14.2 + 731 / D, to
// ~3e-16.  One form for the sigma(R) node tables, linear_power and the streaming Stage E.
CHOMP_HD double eh_transfer(const Epoch& e, double k) {
  const double t = 1.0 + 0.43 * k * e.eh_s;
  const double t2 = t * t, t4 = t2 * t2;
  const double q = k * e.eh_theta * t4 / (e.eh_omh * fma(e.eh_alpha, t4, 1.0 - e.eh_alpha));
  const double L0 = fast_log(2.0 * kE + 1.8 * q);
  const double D = fma(62.5, q, 1.0);
  const double N = fma(14.2, D, 731.0);
  const double LD = L0 * D;
  return LD / fma(N, q * q, LD);
}

// Stage E: 2 pi^2 (k/H0)^(3+n) T(k)^2 / k^3 -- linear_power(k) without its amplitude
// (cosmology.py:449-472, 574-600), the transfer function as eh_transfer arranges it.
template <bool BAO>
CHOMP_HD double power_shape_t(const Epoch& e, double ln_k, double k) {
  if (BAO) {
    const double Tb = eh_bao_transfer(e, k);
    return 2.0 * kPi * kPi * exp(fma(e.ns, ln_k, -(3.0 + e.ns) * e.ln_H0)) * Tb * Tb;
  }
  const double T = eh_transfer(e, k);
  return 2.0 * kPi * kPi * exp(fma(e.ns, ln_k, -(3.0 + e.ns) * e.ln_H0)) * T * T;
}
// (transfer function chosen at run time: callers off the streaming path)
CHOMP_HD double power_shape(const Epoch& e, double ln_k, double k) {
  return e.with_bao ? power_shape_t<true>(e, ln_k, k) : power_shape_t<false>(e, ln_k, k);
}

// SingleEpoch.transfer_function (cosmology.py:556-572).
CHOMP_HD double transfer_function(const Epoch& e, double k) {
  return e.with_bao ? eh_bao_transfer(e, k) : eh_transfer(e, k);
}
// The same with the choice made at compile time: the sigma(R) kernels are instantiated
// once per transfer function, so that the no-wiggle instance carries no trace (registers,
// code) of the other.
template <bool BAO>
CHOMP_HD double transfer_t(const Epoch& e, double k) {
  return BAO ? eh_bao_transfer(e, k) : eh_transfer(e, k);
}

// Delta^2(k) = k^3 P(k)/(2 pi^2), cosmology.py:574-587, from ln k.  BAO: the transfer
// function, fixed at compile time in device code (see transfer_t).
template <bool BAO>
CHOMP_HD double delta_k_ln_t(const Epoch& e, double ln_k, double k) {
  const double T = transfer_t<BAO>(e, k);
  return e.amp * e.sigma_norm * e.sigma_norm * exp((3.0 + e.ns) * (ln_k - e.ln_H0)) *
         T * T;
}

// linear_power(k), cosmology.py:589-600
template <bool BAO>
CHOMP_HD double linear_power_t(const Epoch& e, double k) {
  if (!(k > 1e-16)) return 1e-16;
  const double lk = log(k);
  return 2.0 * kPi * kPi * delta_k_ln_t<BAO>(e, lk, k) / (k * k * k);
}

// Run-time choice of the transfer function (host-side checks; device code uses the _t forms).
CHOMP_HD double delta_k_ln(const Epoch& e, double ln_k, double k) {
  return e.with_bao ? delta_k_ln_t<true>(e, ln_k, k) : delta_k_ln_t<false>(e, ln_k, k);
}
CHOMP_HD double linear_power(const Epoch& e, double k) {
  return e.with_bao ? linear_power_t<true>(e, k) : linear_power_t<false>(e, k);
}

// sigma_r limits, cosmology.py:611-632
CHOMP_HD void sigma_limits(const Epoch& e, double scale, double* ln_lo, double* ln_hi) {
  double k_min = e.k_min, k_max = e.k_max;
  const double need_min = 1.0 / scale / 10.0;
  const double need_max = 1.0 / scale * 14.0662;
  if (need_min <= k_min && need_min > e.k_min / 100.0) k_min = need_min;
  else if (need_min <= k_min && need_min <= e.k_min / 100.0) k_min = e.k_min / 100.0;
  if (need_max >= k_max && need_max < e.k_max * 100.0) k_max = need_max;
  else if (need_max >= k_max && need_max >= e.k_max * 100.0) k_max = e.k_max * 100.0;
  *ln_lo = log(k_min);
  *ln_hi = log(k_max);
}

// Integrand of sigma^2(R) over ln k divided by 2 pi^2 (cosmology.py:644-660):
// dk P W^2 k^2 / (2 pi^2) = Delta^2(k) W(kR)^2.
template <bool BAO>
struct SigmaIntegrandT {
  const Epoch* e;
  double scale;
  CHOMP_HD double operator()(double ln_k) const {
    const double k = exp(ln_k);
    const double kR = scale * k;
    double s, c;
    fast_sincos_pm(kR, &s, &c);                   // (W is squared)
    const double kR2 = kR * kR;
    const double W = 3.0 * (s / (kR2 * kR) - c / kR2);
    const double T = transfer_t<BAO>(*e, k);
    return e->amp * e->sigma_norm * e->sigma_norm * exp((3.0 + e->ns) * (ln_k - e->ln_H0)) *
           T * T * W * W;
  }
};
typedef SigmaIntegrandT<false> SigmaIntegrand;

// HaloFit sigma^2(R) with a Gaussian filter, halo.py:1321-1323.
template <bool BAO>
struct HalofitSigmaIntegrand {
  const Epoch* e;
  double R;
  CHOMP_HD double operator()(double ln_k) const {
    const double k = exp(ln_k);
    return delta_k_ln_t<BAO>(*e, ln_k, k) * exp(-k * k * R * R);
  }
};

// E(z) = c/H(z), cosmology.py:153-163
struct EIntegrand {
  double om0, ol0, or0, H0;
  CHOMP_HD double operator()(double z) const {
    return 1.0 / (H0 * sqrt(E0_of(om0, ol0, or0, z)));
  }
};

// (cbrt for the reference's x ** (1 / 3): the two differ by ~ln(x) 2e-17 relative, a third of
//  the instructions)
CHOMP_HD double scale_of_mass(const Epoch& e, double mass) {     // cosmology.py:671
  return cbrt(3.0 * mass / (4.0 * kPi * e.rho_bar));
}

// ---------------------------------------------------------------------------
// Mass function f(nu), b(nu): Sheth-Tormen (mass_function.py:243-255, 290-302)
// and Tinker10 (:494-530).
// ---------------------------------------------------------------------------
CHOMP_HD double f_nu(const Epoch& e, double nu) {
  if (e.mf_kind == 0) {
    const double np_ = nu * e.st_a;
    return e.f_norm * (1.0 + pow(np_, -1.0 * e.stq)) * sqrt(np_) * exp(-0.5 * np_) / nu;
  }
  const double sq = sqrt(nu);
  return e.t_alpha * (1.0 + pow(e.t_beta * sq, -2.0 * e.t_phi)) * pow(nu, e.t_eta) *
         exp(-e.t_gamma * nu / 2.0) / sq;
}

CHOMP_HD double bias_nu(const Epoch& e, double nu) {
  if (e.mf_kind == 0) {
    const double np_ = nu * e.st_a;
    return e.bias_norm * (1.0 + (np_ - 1.0) / e.delta_c +
                          2.0 * e.stq / (e.delta_c * (1.0 + pow(np_, e.stq))));
  }
  const double sq = sqrt(nu);
  const double sa = pow(sq, e.tb_a);
  return e.bias_norm * (1.0 - e.tb_A * sa / (sa + e.tb_dca) + 0.183 * pow(sq, 1.5) +
                        e.tb_C * pow(sq, 2.4));
}

// nu f(nu) and b(nu) at a Romberg node, where ln nu is the integration variable
// itself: every power of nu becomes one exp of a multiple of ln nu (fp64 pow costs
// ~5 exps on the device).  Same quantities as f_nu / bias_nu to ~1e-15 relative.
CHOMP_HD void mf_node(const Epoch& e, double nu, double ln_nu, bool want_bias, double* nu_f,
                      double* bias) {
  if (e.mf_kind == 0) {
    const double np_ = nu * e.st_a;
    const double ln_np = ln_nu + e.ln_st_a;
    const double e1 = exp(-e.stq * ln_np);                       // nu'^-q
    *nu_f = e.f_norm * (1.0 + e1) * exp(0.5 * ln_np - 0.5 * np_);   // nu f(nu)
    if (want_bias)
      *bias = e.bias_norm * (1.0 + (np_ - 1.0) / e.delta_c +
                             2.0 * e.stq / (e.delta_c * (1.0 + 1.0 / e1)));
  } else {
    const double e1 = exp(-2.0 * e.t_phi * (e.ln_t_beta + 0.5 * ln_nu));   // (beta sqrt nu)^-2phi
    *nu_f = e.t_alpha * (1.0 + e1) *
            exp((e.t_eta + 0.5) * ln_nu - 0.5 * e.t_gamma * nu);   // nu * nu^eta e^(-g nu/2) / sqrt nu
    if (want_bias) {
      const double sa = exp(0.5 * e.tb_a * ln_nu);
      *bias = e.bias_norm * (1.0 - e.tb_A * sa / (sa + e.tb_dca) + 0.183 * exp(0.75 * ln_nu) +
                             e.tb_C * exp(1.2 * ln_nu));
    }
  }
}

// Tinker bias constants from delta_v (mass_function.py:521-528).
CHOMP_HD void tinker_bias_constants(Epoch& e) {
  const double y = log10(e.mf_delta_v);
  const double ex = exp(-pow(4.0 / y, 4.0));
  e.tb_A = 1.0 + 0.24 * y * ex;
  e.tb_a = 0.44 * y - 0.88;
  e.tb_C = 0.019 + 0.107 * y + 0.19 * ex;
  e.tb_dca = pow(e.delta_c, e.tb_a);
}

// ---------------------------------------------------------------------------
// Zheng07 HOD moments (hod.py:189-230).
// ---------------------------------------------------------------------------
CHOMP_HD double zheng_central(const Epoch& e, double mass) {
  const double lm = log10(mass);
  if (e.hod_sigma <= 0.0) return lm > e.hod_log_M_min ? 1.0 : 0.0;
  return 0.5 * (1.0 + erf((lm - e.hod_log_M_min) / e.hod_sigma));
}
CHOMP_HD double zheng_satellite(const Epoch& e, double mass) {
  const double diff = mass - e.hod_M0;
  if (!(diff > 0.0)) return 0.0;
  return zheng_central(e, mass) * pow(diff / e.hod_M1p, e.hod_alpha);
}
// HOD moments at a node where ln M is already known (no log10, no pow for alpha = 1).
CHOMP_HD void zheng_node(const Epoch& e, double mass, double ln_mass, double* n_first,
                         double* n_second) {
  const double lm = ln_mass * 0.43429448190325182765;            // log10 M
  double nc;
  if (e.hod_sigma <= 0.0) nc = lm > e.hod_log_M_min ? 1.0 : 0.0;
  else nc = 0.5 * (1.0 + erf((lm - e.hod_log_M_min) / e.hod_sigma));
  const double diff = mass - e.hod_M0;
  double ns = 0.0;
  if (diff > 0.0) {
    const double r = diff / e.hod_M1p;
    ns = nc * (e.hod_alpha == 1.0 ? r : exp(e.hod_alpha * log(r)));
  }
  *n_first = nc + ns;
  *n_second = (2.0 + ns) * ns;
}

CHOMP_HD double zheng_first(const Epoch& e, double mass) {
  return zheng_central(e, mass) + zheng_satellite(e, mass);
}
CHOMP_HD double zheng_second(const Epoch& e, double mass) {
  const double ns = zheng_satellite(e, mass);
  return (2.0 + ns) * ns;
}

// ---------------------------------------------------------------------------
// NFW profile transform y(k, M), halo.py:561-585.  The reference evaluates
// ln r_v and ln c through splines of exactly linear functions of ln M
// (halo.py:839-855, 873-902); they are evaluated in closed form here.
// ---------------------------------------------------------------------------
// HaloExclusion._mass_window (halo.py:1223-1233): transform of the window that removes
// halo pairs closer than two virial radii, kR = 2 k r_v.
CHOMP_HD double exclusion_window(const SiCiTab& T, double kR) {
  double s, c, si, ci;
  fast_sincos(kR, &s, &c);
  sici_sc(kR, s, c, T, &si, &ci);
  return (kR * c + kR * kR * kR * ci + (2.0 - kR * kR) * s) / (3.0 * kR);
}

CHOMP_HD double y_nfw(const Epoch& e, const SiCiTab& T, double ln_k, double ln_mass) {
  const double ln_c = e.ln_c_const + e.beta * ln_mass;
  const double ln_rv = (e.ln_rv_const + ln_mass) * (1.0 / 3.0);
  const double con = exp(ln_c);
  const double cp = 1.0 + con;
  const double z = exp(ln_k + ln_rv - ln_c);
  const double cz = con * z;
  double sz, cz_c, scz, ccz;
  fast_sincos(z, &sz, &cz_c);
  fast_sincos(cz, &scz, &ccz);
  // sin/cos of (1+c) z by angle addition
  const double s_cp = sz * ccz + cz_c * scz;
  const double c_cp = cz_c * ccz - sz * scz;
  double si_z, ci_z, si_cz, ci_cz;
  sici_sc(z, sz, cz_c, T, &si_z, &ci_z);
  sici_sc(cp * z, s_cp, c_cp, T, &si_cz, &ci_cz);
  const double rho_km = cz_c * (ci_cz - ci_z) + sz * (si_cz - si_z) - scz / (cp * z);
  const double mass_k = log(cp) - con / cp;
  return rho_km / mass_k;
}

// The k-dependent core of y_nfw for a halo whose concentration con, ln(1+con),
// ln r_s and 1/(ln(1+c) - c/(1+c)) are already known (per-node tables).
CHOMP_HD double y_nfw_core(const SiCiTab& T, double ln_k, double ln_rs, double con,
                           double ln_cp, double inv_mass_k, double* z_out = nullptr) {
  const double ln_z = ln_k + ln_rs;
  const double z = exp(ln_z);
  if (z_out) *z_out = z;                         // k r_s
  const double cp = 1.0 + con;
  const double cz = con * z;
  double sz, cz_c, scz, ccz;
  fast_sincos(z, &sz, &cz_c);
  fast_sincos(cz, &scz, &ccz);
  const double s_cp = sz * ccz + cz_c * scz;
  const double c_cp = cz_c * ccz - sz * scz;
  double si_z, ci_z, si_cz, ci_cz;
  sici_sc_ln(z, ln_z, sz, cz_c, T, &si_z, &ci_z);
  sici_sc_ln(cp * z, ln_z + ln_cp, s_cp, c_cp, T, &si_cz, &ci_cz);
  const double rho_km = cz_c * (ci_cz - ci_z) + sz * (si_cz - si_z) - scz / (cp * z);
  return rho_km * inv_mass_k;
}

// The same from a node table that also holds r_s and 1 / ((1 + c) r_s): z = k r_s and
// 1 / ((1 + c) z) = (1 / k) / ((1 + c) r_s) are products (the exponential and the fp64 division
// above are ~70 of the transform's ~450 instructions); ln z = ln k + ln r_s is still the sum, for
// the logarithmic term of Ci.
CHOMP_HD double y_nfw_core_tab(const SiCiTab& T, double ln_k, double k, double inv_k, double ln_rs,
                               double con, double ln_cp, double inv_mass_k, double rs,
                               double inv_cprs, double* z_out = nullptr) {
  const double ln_z = ln_k + ln_rs;
  const double z = k * rs;
  if (z_out) *z_out = z;                         // k r_s
  const double cp = 1.0 + con;
  const double cz = con * z;
  double sz, cz_c, scz, ccz;
  fast_sincos(z, &sz, &cz_c);
  fast_sincos(cz, &scz, &ccz);
  const double s_cp = sz * ccz + cz_c * scz;
  const double c_cp = cz_c * ccz - sz * scz;
  double si_z, ci_z, si_cz, ci_cz;
  sici_sc_ln(z, ln_z, sz, cz_c, T, &si_z, &ci_z);
  sici_sc_ln(cp * z, ln_z + ln_cp, s_cp, c_cp, T, &si_cz, &ci_cz);
  const double rho_km = cz_c * (ci_cz - ci_z) + sz * (si_cz - si_z) - scz * (inv_k * inv_cprs);
  return rho_km * inv_mass_k;
}

// Halo constants from the profile halo_dict (halo.py:71-83, 873-902).
CHOMP_HD void halo_constants(Epoch& e, double c0_in, double beta, double delta_v_in) {
  e.c0 = c0_in / (1.0 + e.z);
  e.beta = beta;
  e.prof_delta_v = (delta_v_in == -1.0) ? e.delta_v : delta_v_in;
  e.ln_rv_const = log(3.0 / (4.0 * kPi * e.prof_delta_v * e.rho_bar));
  e.ln_c_const = log(e.c0) - e.beta * log(e.m_star);
}

}  // namespace chomp
