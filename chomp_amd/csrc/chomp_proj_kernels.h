// chomp_proj_kernels.h -- HIP kernels of the projection side (gfx950):
//
//   k_halofit_sigma / k_halofit_finalize   HaloFit._initialize_sigma_spline (halo.py:1268-1319)
//   k_proj_chi          MultiEpoch chi(z) Romberg per grid point (cosmology.py:787-794),
//                       growth table (:813-814), dNdz.normalize (kernel.py:43-54)
//   k_proj_me_splines   chi(z), z(chi), D(z) splines (cosmology.py:795-817) and the
//                       windows' chi grids (kernel.py:289-306)
//   k_proj_window       raw_window_function per chi knot: galaxy (kernel.py:382-387),
//                       convergence with its inner Romberg (:443-484)
//   k_proj_window_splines  window splines (:308-313) + Kernel.__init__ scalars and
//                       _find_z_bar (:595-639)
//   k_proj_kernel_knots K(ln k theta) knots: Romberg over chi with J0 / J2 truncated at
//                       the 8th Bessel zero (:678-712, 812-839)
//   k_proj_kernel_spline  (:641-649)
//   k_kernel_eval / k_window_eval   Kernel.kernel (:714-729), window_function (:326-340)
//   k_wtheta            Correlation.correlation (correlation.py:242-275)
//   k_cell              CorrelationFourier.correlation (correlation.py:360-392)
//
// One integral per workgroup (1 or 4 wavefronts), all splines / Bessel Chebyshev
// tables / P(k) knot splines staged in LDS, __shfl_xor reductions.
#pragma once

#include <hip/hip_runtime.h>

#include "chomp_power_kernels.h"

namespace chomp {

struct DndzDev {
  int kind, pp_n;
  double z_min, z_max, p[4], norm;
  const double* pp;          // CHOMP_DNDZ_PPOLY: breaks[pp_n + 1], then coef[pp_n][pp_order + 1]
  int pp_order, pad;
};

// Scalars of one projection set-up (host fills the inputs, kernels fill the rest).
struct ProjDev {
  // SingleEpoch(0) background of the MultiEpoch (cosmology.py:761-783)
  double om0, ol0, or0, H0, growth_norm;
  double me_z_min[3], me_z_max[3];   // 0: the kernel's MultiEpoch, 1/2: windows' copies
  DndzDev dist[2];
  int wkind[2];
  double w_z_min[2], w_z_max[2], w_chi_min[2], w_chi_max[2], w_g_chi_min[2];
  double z_min, z_max, chi_min, chi_max, ln_kt_min, ln_kt_max, j_limit, z_bar, D_zbar;
  int order, pad;
};
static_assert(sizeof(ProjDev) % 8 == 0, "ProjDev granularity");
constexpr int kProjDoubles = (int)(sizeof(ProjDev) / sizeof(double));

struct ProjLayout {
  int NC, NWp, NKT;
  int me_z[3], me_chi[3], me_growth[3], me_pp_chi[3], me_pp_z[3], me_pp_g[3];
  int w_chi[2], w_wf[2], w_pp[2];
  int k_ln, k_arr, k_pp, k_lev;
  int total;
};

inline ProjLayout make_proj_layout(int NC, int NWp, int NKT) {
  ProjLayout L;
  L.NC = NC; L.NWp = NWp; L.NKT = NKT;
  int o = 0;
  for (int m = 0; m < 3; ++m) {
    L.me_z[m] = o; o += NC;
    L.me_chi[m] = o; o += NC;
    L.me_growth[m] = o; o += NC;
    L.me_pp_chi[m] = o; o += 4 * (NC - 1);
    L.me_pp_z[m] = o; o += 4 * (NC - 1);
    L.me_pp_g[m] = o; o += 4 * (NC - 1);
  }
  for (int w = 0; w < 2; ++w) {
    L.w_chi[w] = o; o += NWp;
    L.w_wf[w] = o; o += NWp;
    L.w_pp[w] = o; o += 4 * (NWp - 1);
  }
  L.k_ln = o; o += NKT;
  L.k_arr = o; o += NKT;
  L.k_pp = o; o += 4 * (NKT - 1);
  L.k_lev = o; o += NKT;
  L.total = (o + 7) & ~7;
  return L;
}

struct ProjState {
  bool ready = false;      // full kernel set-up done
  bool me_ready = false;   // MultiEpoch 0 tables valid
  ProjLayout L;
  ProjDev host;            // host copy of the scalars (refreshed by kernel_info)
  ProjDev* d_pd = nullptr;
  ProjDev* d_pd_init = nullptr;   // the block as the host prepared it (the device fills d_pd in)
  double* d_tab = nullptr;
  size_t cap_tab = 0;      // doubles allocated at d_tab
  size_t pp_total = 0;     // doubles of tabulated redshift distributions behind the tables
  bool cov_ready = false;  // covariance table (chomp_covariance_table) valid
  double* d_cov = nullptr;
};
inline void proj_free(ProjState& p) {
  if (p.d_pd) (void)hipFree(p.d_pd);
  if (p.d_pd_init) (void)hipFree(p.d_pd_init);
  if (p.d_tab) (void)hipFree(p.d_tab);
  if (p.d_cov) (void)hipFree(p.d_cov);
  p.d_pd = nullptr;
  p.d_pd_init = nullptr;
  p.d_tab = nullptr;
  p.d_cov = nullptr;
  p.ready = false;
  p.me_ready = false;
  p.cov_ready = false;
}

// ---------------------------------------------------------------------------
// HaloFit
// ---------------------------------------------------------------------------
// grid NK, block 256: ln sigma^2(R_i), R_i = exp(linspace(ln 0.1, ln 10, NK)).
template <bool BAO>
__global__ __launch_bounds__(256) void k_halofit_sigma(chomp_config cfg, TabLayout L,
                                                       const Epoch* __restrict__ epochs,
                                                       int e, double* __restrict__ tab) {
  __shared__ Epoch E;
  __shared__ double red[romberg_scratch<4, 2>()];
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  __syncthreads();
  const int i = blockIdx.x;
  const double R = exp(linspace_at(log(0.1), log(10.0), L.NK, i));
  HalofitSigmaIntegrand<BAO> f{&E, R};
  const double s2 = romberg1<4>(f, log(cfg.k_min), log(cfg.k_max), cfg.global_precision,
                                cfg.halo_precision, cfg.divmax, red);
  if (threadIdx.x == 0) tab[(size_t)e * L.stride + L.off_hf_lns2 + i] = log(s2);
}

// First two derivatives at xq of the quintic interpolating spline through (x, y) -- what
// quintic_derivs (chomp_math.h) computes -- by one wavefront: the collocation rows are set up
// one per lane, the banded elimination runs its (row, column) updates of a pivot step in
// parallel lanes, the back substitution its five products.  Same row operations in the same
// order as the serial routine (only the 5-term sums of the back substitution are added in
// another order).  x, y: LDS or global; work: (n + 6) + 11 n + n doubles of LDS; n <= 64 rows
// per pass are handled by striding the lanes.  All 64 lanes call; barriers inside (the block
// is one wavefront).
__device__ __forceinline__ void quintic_derivs_wave(const double* x, const double* y, int n,
                                                    double xq, double* work, double* d1,
                                                    double* d2) {
  const int lane = threadIdx.x & 63;
  double* t = work;               // [n + 6]
  double* ab = t + (n + 6);       // [n][11] band storage, column j at j - i + 5
  double* c = ab + 11 * n;        // [n]
  for (int i = lane; i < 6; i += 64) { t[i] = x[0]; t[n + i] = x[n - 1]; }
  for (int i = lane; i < n - 6; i += 64) t[6 + i] = x[3 + i];
  for (int i = lane; i < 11 * n; i += 64) ab[i] = 0.0;
  __syncthreads();
  for (int i = lane; i < n; i += 64) {
    int m = 5;
    while (m < n - 1 && x[i] >= t[m + 1]) ++m;
    double N[6];
    bspline_basis(t, m, 5, x[i], N);
#pragma unroll
    for (int r = 0; r < 6; ++r) ab[11 * i + (m - 5 + r - i + 5)] = N[r];
    c[i] = y[i];
  }
  __syncthreads();
  // banded elimination without pivoting (the collocation matrix is totally positive):
  // lane = 6 (r - col - 1) + (j - col) updates element (r, j); lanes 30..34 the right-hand side
  for (int col = 0; col < n; ++col) {
    const int rr = lane < 30 ? lane / 6 : lane - 30, jj = lane < 30 ? lane % 6 : 0;
    const int r = col + 1 + rr, j = col + jj;
    const bool row_ok = lane < 35 && r < n;
    double f = 0.0;
    if (row_ok) {
      const double a = ab[11 * r + (col - r + 5)];
      f = a == 0.0 ? 0.0 : a / ab[11 * col + 5];
    }
    __syncthreads();               // (every lane has read its factor before (r, col) is cleared)
    if (row_ok && f != 0.0) {
      if (lane < 30) {
        if (j < n) ab[11 * r + (j - r + 5)] -= f * ab[11 * col + (j - col + 5)];
      } else {
        c[r] -= f * c[col];
      }
    }
    __syncthreads();
  }
  for (int i = n - 1; i >= 0; --i) {
    const int j = i + 1 + lane;
    double term = (lane < 5 && j < n) ? ab[11 * i + (j - i + 5)] * c[j] : 0.0;
    term += __shfl_xor(term, 1, 64);
    term += __shfl_xor(term, 2, 64);
    term += __shfl_xor(term, 4, 64);
    if (lane == 0) c[i] = (c[i] - term) / ab[11 * i + 5];
    __syncthreads();
  }
  if (lane == 0) {
    int m = 5;
    while (m < n - 1 && xq >= t[m + 1]) ++m;
    double c1[6], c2[5], N[6];
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
  __syncthreads();
}

// The inverse of the quintic spline's collocation matrix on HaloFit's ln R grid
// (linspace(ln 0.1, ln 10, n): halo.py:1270-1272 -- the grid, hence the matrix, depends on n
// alone), transposed: ainv_t[j * n + i] = (A^-1)[i][j].  Host, once per context (long double
// Gauss-Jordan with partial pivoting).  The B-spline coefficients of the sigma table are then
// c = A^-1 y: fifty multiply-adds per lane, where the banded elimination and back substitution
// of quintic_derivs_wave are ~150 dependent steps of one wavefront (30 of k_halofit_finalize's
// 39 us).
inline void halofit_collocation_inverse_host(int n, double* ainv_t) {
  std::vector<double> x(n), t(n + 6);
  for (int i = 0; i < n; ++i) x[i] = linspace_at(std::log(0.1), std::log(10.0), n, i);
  for (int i = 0; i < 6; ++i) { t[i] = x[0]; t[n + i] = x[n - 1]; }
  for (int i = 0; i < n - 6; ++i) t[6 + i] = x[3 + i];
  std::vector<long double> a((size_t)n * 2 * n, 0.0L);          // [A | I]
  int m = 5;
  for (int i = 0; i < n; ++i) {
    while (m < n - 1 && x[i] >= t[m + 1]) ++m;
    double N[6];
    bspline_basis(t.data(), m, 5, x[i], N);
    for (int r = 0; r < 6; ++r) a[(size_t)i * 2 * n + (m - 5 + r)] = N[r];
    a[(size_t)i * 2 * n + n + i] = 1.0L;
  }
  for (int col = 0; col < n; ++col) {
    int piv = col;
    for (int r = col + 1; r < n; ++r)
      if (fabsl(a[(size_t)r * 2 * n + col]) > fabsl(a[(size_t)piv * 2 * n + col])) piv = r;
    if (piv != col)
      for (int j = 0; j < 2 * n; ++j) std::swap(a[(size_t)col * 2 * n + j], a[(size_t)piv * 2 * n + j]);
    const long double d = a[(size_t)col * 2 * n + col];
    for (int j = 0; j < 2 * n; ++j) a[(size_t)col * 2 * n + j] /= d;
    for (int r = 0; r < n; ++r) {
      if (r == col) continue;
      const long double f = a[(size_t)r * 2 * n + col];
      if (f == 0.0L) continue;
      for (int j = 0; j < 2 * n; ++j) a[(size_t)r * 2 * n + j] -= f * a[(size_t)col * 2 * n + j];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) ainv_t[(size_t)j * n + i] = (double)a[(size_t)i * 2 * n + n + j];
}

// quintic_derivs_wave with the coefficients from the tabulated inverse (one wavefront; work:
// (n + 6) + n doubles of LDS; barriers inside).
__device__ __forceinline__ void quintic_derivs_inv(const double* x, const double* y, int n,
                                                   double xq, const double* __restrict__ ainv_t,
                                                   double* work, double* d1, double* d2) {
  const int lane = threadIdx.x & 63;
  double* t = work;               // [n + 6]
  double* c = t + (n + 6);        // [n]
  for (int i = lane; i < 6; i += 64) { t[i] = x[0]; t[n + i] = x[n - 1]; }
  for (int i = lane; i < n - 6; i += 64) t[6 + i] = x[3 + i];
  for (int i = lane; i < n; i += 64) {
    double acc = 0.0;
    for (int j = 0; j < n; ++j) acc = fma(ainv_t[(size_t)j * n + i], y[j], acc);
    c[i] = acc;
  }
  __syncthreads();
  if (lane == 0) {
    int m = 5;
    while (m < n - 1 && xq >= t[m + 1]) ++m;
    double c1[6], c2[5], N[6];
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
  __syncthreads();
}

// grid 1, block 64 (one wavefront, on LDS copies): k_sigma, n_eff, C and the Takahashi et al.
// coefficients (halo.py:1285-1317) of epoch `src` stored into epoch `dst`.  The cubic spline of
// ln R over ln sigma^2 by parallel cyclic reduction, the quintic one of ln sigma^2 over ln R
// by quintic_derivs_wave.  Dynamic LDS: 25 NK + 64 doubles.
__global__ __launch_bounds__(64) void k_halofit_finalize(TabLayout L, Epoch* __restrict__ epochs,
                                                         int dst, int src,
                                                         const double* __restrict__ tab, double f1,
                                                         double f2, double f3, double omega_l,
                                                         double w, const double* __restrict__ ainv_t) {
  extern __shared__ __align__(16) double work[];
  __shared__ double sh_d[2];
  const int n = L.NK;
  double* lns2 = work + 24 * n + 64;
  double* xr = work;             // reversed ln sigma^2 (increasing)
  double* yr = xr + n;           // reversed ln R
  double* lnR = yr + n;
  double* c = lnR + n;           // [4(n-1)]
  double* w2 = c + 4 * (n - 1);  // scratch: max(9 n, (n + 6) + 12 n)
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    lns2[i] = tab[(size_t)src * L.stride + L.off_hf_lns2 + i];
    lnR[i] = linspace_at(log(0.1), log(10.0), n, i);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    xr[i] = lns2[n - 1 - i];
    yr[i] = lnR[n - 1 - i];
  }
  __syncthreads();
  spline_build_pcr(xr, yr, n, c, w2, (int)threadIdx.x, 64, true);
  const double k_s = 1.0 / exp(spline_eval(xr, c, n, 0.0));            // halo.py:1285-1287
  if (ainv_t != nullptr)                                                        // :1289-1292
    quintic_derivs_inv(lnR, lns2, n, log(1.0 / k_s), ainv_t, w2, &sh_d[0], &sh_d[1]);
  else
    quintic_derivs_wave(lnR, lns2, n, log(1.0 / k_s), w2, &sh_d[0], &sh_d[1]);
  if (threadIdx.x != 0) return;
  const double ne = -sh_d[0] - 3.0, C = -sh_d[1];
  Epoch& E = epochs[dst];
  E.hf_f1 = f1; E.hf_f2 = f2; E.hf_f3 = f3;
  E.hf_k_s = k_s; E.hf_n_eff = ne; E.hf_C = C;
  E.hf_a_n = pow(10.0, 1.5222 + 2.8553 * ne + 2.3706 * ne * ne + 0.9903 * ne * ne * ne +
                           0.2250 * ne * ne * ne * ne + -0.6038 * C +
                           0.1749 * omega_l * (1.0 + w));
  E.hf_b_n = pow(10.0, -0.5642 + 0.5864 * ne + 0.5716 * ne * ne + -1.5474 * C +
                           0.2279 * omega_l * (1.0 + w));
  E.hf_c_n = pow(10.0, 0.3698 + 2.0404 * ne + 0.8161 * ne * ne + 0.5869 * C);
  E.hf_gamma_n = 0.1971 - 0.0843 * ne + 0.8460 * C;
  E.hf_alpha_n = fabs(6.0835 + 1.3373 * ne - 0.1959 * ne * ne + -5.5274 * C);
  E.hf_beta_n = 2.0379 - 0.7354 * ne + 0.3157 * ne * ne + 1.2490 * ne * ne * ne +
                0.3980 * ne * ne * ne * ne + -0.1682 * C;
  E.hf_mu_n = 0.0;
  E.hf_nu_n = pow(10.0, 5.2105 + 3.6902 * ne);
}

// ---------------------------------------------------------------------------
// Redshift distributions (kernel.py:26-179)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double dndz_raw(const DndzDev& d, double z) {
  if (d.kind == CHOMP_DNDZ_MAGLIM)          // z^a exp(-(z/z0)^b), p = {a, z0, b}
    return pow(z, d.p[0]) * exp(-1.0 * pow(z / d.p[1], d.p[2]));
  if (d.kind == CHOMP_DNDZ_BOXCAR) return 1.0;   // the base class (kernel.py:56-65)
  if (d.kind == CHOMP_DNDZ_PPOLY) {              // dNdzInterpolation: the caller's spline
    const double* br = d.pp;
    int lo = 0, hi = d.pp_n - 1;                 // last piece with br[i] <= z (ends extrapolate)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (z >= br[mid]) lo = mid; else hi = mid - 1;
    }
    const double* cf = br + d.pp_n + 1 + (size_t)lo * (d.pp_order + 1);
    const double t = z - br[lo];
    double v = cf[d.pp_order];
    for (int m = d.pp_order - 1; m >= 0; --m) v = fma(v, t, cf[m]);
    return v;
  }
  const double t = z - d.p[0];              // Gaussian, p = {z0, sigma_z}
  return exp(-1.0 * t * t / (2.0 * d.p[1] * d.p[1]));
}
__device__ __forceinline__ double dndz_eval(const DndzDev& d, double z) {
  return (z <= d.z_max && z >= d.z_min) ? d.norm * dndz_raw(d, z) : 0.0;
}
struct DndzRaw {
  const DndzDev* d;
  __device__ __forceinline__ double operator()(double z) const { return dndz_raw(*d, z); }
};

// grid (NC, 4), block 64.  y < 3: chi and growth of MultiEpoch y at grid point x;
// y == 3, x < 2: normalisation of distribution x.
__global__ __launch_bounds__(64) void k_proj_chi(chomp_config cfg, ProjLayout L,
                                                 ProjDev* __restrict__ pd,
                                                 double* __restrict__ tab) {
  const int i = blockIdx.x, m = blockIdx.y;
  if (m == 3) {
    if (i >= 2) return;
    __shared__ DndzDev D;
    if (threadIdx.x == 0) D = pd->dist[i];
    __syncthreads();
    DndzRaw f{&D};
    const double norm = romberg1<1>(f, D.z_min, D.z_max, cfg.global_precision,
                                    cfg.dNdz_precision, cfg.divmax, nullptr);
    if (threadIdx.x == 0) pd->dist[i].norm = 1.0 / norm;
    return;
  }
  const double z = linspace_at(pd->me_z_min[m], pd->me_z_max[m], L.NC, i);
  EIntegrand f{pd->om0, pd->ol0, pd->or0, pd->H0};
  const double chi = romberg1<1>(f, 0.0, z, cfg.global_precision, cfg.cosmo_precision,
                                 cfg.divmax, nullptr);
  if (threadIdx.x == 0) {
    tab[L.me_z[m] + i] = z;
    tab[L.me_chi[m] + i] = chi;
    tab[L.me_growth[m] + i] = growth_approx(pd->om0, pd->ol0, 1.0 / (1.0 + z)) / pd->growth_norm;
  }
}

// MultiEpoch lookups with the reference's range rules (cosmology.py:873-953).
struct MEView {
  const double *z, *chi, *pp_chi, *pp_z, *pp_g;   // z uniform, chi increasing
  int NC;
  double z_min, z_max;
  __device__ __forceinline__ double dz() const { return (z_max - z_min) / (double)(NC - 1); }
  __device__ __forceinline__ double comoving_distance(double zz) const {
    return (zz <= z_max && zz >= z_min) ? spline_eval_uniform(z_min, dz(), pp_chi, NC, zz) : 0.0;
  }
  __device__ __forceinline__ double redshift(double c) const {
    return spline_eval(chi, pp_z, NC, c);
  }
  __device__ __forceinline__ double growth_factor(double zz) const {
    return (zz <= z_max && zz >= z_min) ? spline_eval_uniform(z_min, dz(), pp_g, NC, zz) : 1.0;
  }
};
__device__ __forceinline__ MEView me_view(const ProjLayout& L, const ProjDev& pd,
                                          const double* tab, int m) {
  return MEView{tab + L.me_z[m], tab + L.me_chi[m], tab + L.me_pp_chi[m], tab + L.me_pp_z[m],
                tab + L.me_pp_g[m], L.NC, pd.me_z_min[m], pd.me_z_max[m]};
}

// Not-a-knot build of one spline per wavefront, in LDS: the lanes copy the knots in, the
// wavefronts run the parallel cyclic reduction of chomp_mass_kernels.h in lockstep and copy
// their coefficients out.  Every thread of the block calls it (block barriers); `mine`:
// this wavefront has a system.  lds: 11 n doubles per wavefront.
__device__ __forceinline__ void spline_build_staged(const double* x, const double* y, int n,
                                                    double* c, double* lds, bool mine) {
  const int lane = threadIdx.x & 63;
  double* lx = lds;
  double* ly = lx + n;
  double* lw = ly + n;            // [9n]
  if (mine)
    for (int i = lane; i < n; i += 64) { lx[i] = x[i]; ly[i] = y[i]; }
  __syncthreads();
  spline_build_pcr(lx, ly, n, c, lw, lane, 64, mine);
  __threadfence_block();
  __syncthreads();
}

// grid 3, block 192.  Splines of MultiEpoch m; blocks 1/2 then lay out window
// (m-1)'s chi grid (kernel.py:298-304, 438-441).
__global__ __launch_bounds__(192) void k_proj_me_splines(chomp_config cfg, ProjLayout L,
                                                         ProjDev* __restrict__ pd,
                                                         double* __restrict__ tab) {
  extern __shared__ __align__(16) double sm_me[];      // 3 x 11 NC doubles
  const int m = blockIdx.x, NC = L.NC;
  const int wave = threadIdx.x >> 6;
  const double* sx = tab + (wave == 1 ? L.me_chi[m] : L.me_z[m]);
  const double* sy = tab + (wave == 0 ? L.me_chi[m] : (wave == 1 ? L.me_z[m] : L.me_growth[m]));
  double* sc = tab + (wave == 0 ? L.me_pp_chi[m] : (wave == 1 ? L.me_pp_z[m] : L.me_pp_g[m]));
  spline_build_staged(sx, sy, NC, sc, sm_me + (size_t)wave * 11 * NC, true);
  if (m == 0) return;
  const int w = m - 1;
  __shared__ double lim[2];
  if (threadIdx.x == 0) {
    const MEView me = me_view(L, *pd, tab, m);
    const double wp = cfg.window_precision;
    double cmin = me.comoving_distance(pd->w_z_min[w]);
    if (cmin < wp) cmin = wp;
    const double cmax = me.comoving_distance(pd->w_z_max[w]);
    double g = me.comoving_distance(pd->dist[w].z_min);
    if (g < wp) g = wp;
    pd->w_chi_min[w] = cmin;
    pd->w_chi_max[w] = cmax;
    pd->w_g_chi_min[w] = g;
    lim[0] = cmin;
    lim[1] = cmax;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L.NWp; i += blockDim.x)
    tab[L.w_chi[w] + i] = linspace_at(lim[0], lim[1], L.NWp, i);
}

// Lensing-efficiency integrand, kernel.py:479-484.
struct LensIntegrand {
  const double *chi_knots, *pp_z;
  int NC;
  const DndzDev* d;
  double om0, ol0, or0, H0, chi0;
  __device__ __forceinline__ double operator()(double c) const {
    const double z = spline_eval(chi_knots, pp_z, NC, c);
    const double dzdchi = H0 * sqrt(E0_of(om0, ol0, or0, z));
    return dzdchi * dndz_eval(*d, z) * (c - chi0) / c;
  }
};

// grid (NWp, 2), block 64: raw window function of window y at its x-th chi knot.
__global__ __launch_bounds__(64) void k_proj_window(chomp_config cfg, ProjLayout L,
                                                    const ProjDev* __restrict__ pd,
                                                    double* __restrict__ tab) {
  extern __shared__ __align__(16) double sm[];
  __shared__ DndzDev D;
  const int i = blockIdx.x, w = blockIdx.y, m = w + 1, NC = L.NC;
  double* chi_knots = sm;
  double* pp_z = sm + NC;
  copy_doubles(chi_knots, tab + L.me_chi[m], NC);
  copy_doubles(pp_z, tab + L.me_pp_z[m], 4 * (NC - 1));
  if (threadIdx.x == 0) D = pd->dist[w];
  __syncthreads();
  const double chi = tab[L.w_chi[w] + i];
  const double z = spline_eval(chi_knots, pp_z, NC, chi);
  double val;
  if (pd->wkind[w] == CHOMP_WINDOW_GALAXY) {                       // kernel.py:382-387
    val = pd->H0 * sqrt(E0_of(pd->om0, pd->ol0, pd->or0, z)) * dndz_eval(D, z);
  } else if (pd->wkind[w] == CHOMP_WINDOW_FLAT_CONVERGENCE) {      // kernel.py:507-513
    val = 3.0 / 2.0 * pd->om0 * (pd->H0 * pd->H0 * 1907.71);
  } else if (pd->wkind[w] == CHOMP_WINDOW_CONVERGENCE_DELTA) {     // kernel.py:541-556
    const double a = 1.0 / (1.0 + pd->w_z_max[w]);                 // (the source plane's a)
    const double cmax = pd->w_chi_max[w];
    double g = chi > cmax ? 0.0 : (cmax - chi) / cmax;
    g *= pd->H0 * pd->H0 * chi;
    val = 3.0 / 2.0 * pd->om0 * g / a;
  } else {                                                         // kernel.py:443-477
    const double a = 1.0 / (1.0 + z);
    double bound = chi;
    if (bound < pd->w_g_chi_min[w]) bound = pd->w_g_chi_min[w];
    double g = 0.0;
    if (!(bound <= cfg.window_precision)) {
      LensIntegrand f{chi_knots, pp_z, NC, &D, pd->om0, pd->ol0, pd->or0, pd->H0, chi};
      g = romberg1<1>(f, bound, pd->w_chi_max[w], cfg.global_precision,
                      cfg.window_precision, cfg.divmax, nullptr);
    }
    g *= pd->H0 * pd->H0 * chi;
    val = 3.0 / 2.0 * pd->om0 * g / a;
  }
  if (threadIdx.x == 0) tab[L.w_wf[w] + i] = val;
}

// Window lookup with the range rule of kernel.py:326-340 (uniform chi knots).
struct WindowView {
  const double* pp;
  int N;
  double chi_min, chi_max;
  __device__ __forceinline__ double operator()(double c) const {
    if (!(c >= chi_min && c <= chi_max)) return 0.0;
    return spline_eval_uniform(chi_min, (chi_max - chi_min) / (double)(N - 1), pp, N, c);
  }
};

// grid 1, block 128: window splines, then Kernel.__init__ scalars + _find_z_bar.
__global__ __launch_bounds__(128) void k_proj_window_splines(chomp_config cfg, ProjLayout L,
                                                             ProjDev* __restrict__ pd,
                                                             double* __restrict__ tab) {
  extern __shared__ __align__(16) double sm_w[];       // 2 x 11 NWp doubles
  const int wave = threadIdx.x >> 6;
  spline_build_staged(tab + L.w_chi[wave], tab + L.w_wf[wave], L.NWp, tab + L.w_pp[wave],
                      sm_w + (size_t)wave * 11 * L.NWp, true);
  const MEView me = me_view(L, *pd, tab, 0);
  const WindowView wa{tab + L.w_pp[0], L.NWp, pd->w_chi_min[0], pd->w_chi_max[0]};
  const WindowView wb{tab + L.w_pp[1], L.NWp, pd->w_chi_min[1], pd->w_chi_max[1]};
  const double z_min = pd->w_z_min[0] > pd->w_z_min[1] ? pd->w_z_min[0] : pd->w_z_min[1];
  const double z_max = pd->w_z_max[0] < pd->w_z_max[1] ? pd->w_z_max[0] : pd->w_z_max[1];
  for (int i = threadIdx.x; i < L.NKT; i += blockDim.x)
    tab[L.k_ln + i] = linspace_at(pd->ln_kt_min, pd->ln_kt_max, L.NKT, i);
  // _find_z_bar (:635-639): argmax over linspace(z_min, z_max, NKT) of W_a W_b D^2 --
  // numpy.argmax: the first of equal maxima.  One candidate per thread, then thread 0 scans
  // the values in order.
  __shared__ double cand_v[128];
  for (int base = 0; base < L.NKT; base += 128) {        // (NKT <= 128 in any sensible set-up)
    const int i = base + (int)threadIdx.x;
    if (i < L.NKT && i < base + 128) {
      const double z = linspace_at(z_min, z_max, L.NKT, i);
      const double chi = me.comoving_distance(z);
      const double D = me.growth_factor(me.redshift(chi));
      cand_v[threadIdx.x] = wa(chi) * wb(chi) * D * D;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double best = base == 0 ? -INFINITY : pd->D_zbar;  // (D_zbar: scratch for the running maximum)
      double zb = base == 0 ? z_min : pd->z_bar;
      for (int q = 0; q < 128 && base + q < L.NKT; ++q)
        if (cand_v[q] > best) { best = cand_v[q]; zb = linspace_at(z_min, z_max, L.NKT, base + q); }
      pd->z_bar = zb;
      pd->D_zbar = best;
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  pd->z_min = z_min;                                               // kernel.py:595-598
  pd->z_max = z_max;
  const double c0 = me.comoving_distance(z_min);
  pd->chi_min = cfg.window_precision > c0 ? cfg.window_precision : c0;   // :610-612
  pd->chi_max = me.comoving_distance(z_max);
  pd->D_zbar = me.growth_factor(pd->z_bar);                        // correlation.py:94
}

// LDS-resident view of everything a projection integrand needs.
struct ProjLds {
  MEView me;
  WindowView wa, wb;
  const BesselTab* bess;
  // Carve `sm` and copy the tables in (all threads; barrier afterwards).
  __device__ __forceinline__ double* stage(const ProjLayout& L, const ProjDev& pd,
                                           const double* tab, double* sm) {
    const int NC = L.NC, NW = L.NWp;
    double* chi = sm;                 // [NC]
    double* pp_z = chi + NC;          // [4(NC-1)]
    double* pp_g = pp_z + 4 * (NC - 1);
    double* pa = pp_g + 4 * (NC - 1); // [4(NW-1)]
    double* pb = pa + 4 * (NW - 1);
    copy_doubles(chi, tab + L.me_chi[0], NC);
    copy_doubles(pp_z, tab + L.me_pp_z[0], 4 * (NC - 1));
    copy_doubles(pp_g, tab + L.me_pp_g[0], 4 * (NC - 1));
    copy_doubles(pa, tab + L.w_pp[0], 4 * (NW - 1));
    copy_doubles(pb, tab + L.w_pp[1], 4 * (NW - 1));
    me = MEView{nullptr, chi, nullptr, pp_z, pp_g, NC, pd.me_z_min[0], pd.me_z_max[0]};
    wa = WindowView{pa, NW, pd.w_chi_min[0], pd.w_chi_max[0]};
    wb = WindowView{pb, NW, pd.w_chi_min[1], pd.w_chi_max[1]};
    return pb + 4 * (NW - 1);
  }
  static __host__ __device__ int doubles(const ProjLayout& L) {
    return L.NC + 8 * (L.NC - 1) + 8 * (L.NWp - 1);
  }
};

// kernel.py:707-712 (J0) / 833-839 (J2)
template <int ORDER>
struct KernelIntegrand {
  const ProjLds* P;
  double ktheta;
  __device__ __forceinline__ double operator()(double chi) const {
    const double D = P->me.growth_factor(P->me.redshift(chi));
    return P->wa(chi) * P->wb(chi) * D * D * bessel_j<ORDER>(ktheta * chi, *P->bess);
  }
};

// grid NKT, block 256: one kernel knot per block.
__global__ __launch_bounds__(256) void k_proj_kernel_knots(chomp_config cfg, ProjLayout L,
                                                           const ProjDev* __restrict__ pdg,
                                                           double* __restrict__ tab,
                                                           const BesselTab* __restrict__ bess_g,
                                                           const double* __restrict__ ln_in,
                                                           double* __restrict__ out) {
  // ln_in == nullptr: the 50 knots of the spline table (Kernel._initialize_spline);
  // otherwise Kernel.raw_kernel at the caller's ln(k theta), results into `out`.
  extern __shared__ __align__(16) double sm[];
  __shared__ ProjDev pd;
  __shared__ BesselTab B;
  __shared__ double red[romberg_scratch<4, 2>()];
  copy_doubles(reinterpret_cast<double*>(&pd), reinterpret_cast<const double*>(pdg), kProjDoubles);
  copy_doubles(reinterpret_cast<double*>(&B), reinterpret_cast<const double*>(bess_g),
               (int)(sizeof(BesselTab) / sizeof(double)));
  __syncthreads();
  ProjLds P;
  P.stage(L, pd, tab, sm);
  P.bess = &B;
  __syncthreads();
  const int i = blockIdx.x;
  const double ktheta = exp(ln_in ? ln_in[i] : tab[L.k_ln + i]);
  double chi_max = pd.j_limit / ktheta;                            // kernel.py:689-691
  if (chi_max >= pd.chi_max) chi_max = pd.chi_max;
  int level = 0;
  double v;
  if (pd.order == 0) {
    KernelIntegrand<0> f{&P, ktheta};
    v = romberg1<4>(f, pd.chi_min, chi_max, cfg.global_precision, cfg.kernel_precision,
                    cfg.divmax, red, &level);
  } else {
    KernelIntegrand<2> f{&P, ktheta};
    v = romberg1<4>(f, pd.chi_min, chi_max, cfg.global_precision, cfg.kernel_precision,
                    cfg.divmax, red, &level);
  }
  if (threadIdx.x == 0) {
    if (out) {
      out[i] = v;
    } else {
      tab[L.k_arr + i] = v;
      tab[L.k_lev + i] = (double)level;
    }
  }
}

__global__ void k_proj_kernel_spline(ProjLayout L, double* __restrict__ tab) {
  extern __shared__ __align__(16) double sm_k[];       // 11 NKT doubles
  spline_build_staged(tab + L.k_ln, tab + L.k_arr, L.NKT, tab + L.k_pp, sm_k, threadIdx.x < 64);
}

// Kernel.kernel(ln_ktheta), kernel.py:714-729 (uniform ln(k theta) knots).
struct KernelView {
  const double* pp;
  int N;
  double lo, hi;
  __device__ __forceinline__ double operator()(double x) const {
    const double dx = (hi - lo) / (double)(N - 1);
    const double inv_dx = (double)(N - 1) / (hi - lo);    // (both loop-invariant for a caller)
    if (x < lo) return pp_poly(pp, 0, 0.0);
    if (x <= hi) {
      int i = (int)floor((x - lo) * inv_dx);
      i = i < 0 ? 0 : (i > N - 2 ? N - 2 : i);
      return pp_poly(pp, i, x - (lo + dx * (double)i));
    }
    return 0.0;
  }
};

__global__ void k_kernel_eval(ProjLayout L, const ProjDev* __restrict__ pd,
                              const double* __restrict__ tab, const double* __restrict__ x,
                              int n, double* __restrict__ out) {
  const KernelView K{tab + L.k_pp, L.NKT, pd->ln_kt_min, pd->ln_kt_max};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[i] = K(x[i]);
}

__global__ void k_me_eval(ProjLayout L, const ProjDev* __restrict__ pd,
                          const double* __restrict__ tab, int what,
                          const double* __restrict__ x, int n, double* __restrict__ out) {
  const MEView me = me_view(L, *pd, tab, 0);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double v = x[i];
    out[i] = what == CHOMP_ME_CHI_OF_Z ? me.comoving_distance(v)
             : what == CHOMP_ME_Z_OF_CHI ? me.redshift(v) : me.growth_factor(v);
  }
}

__global__ void k_window_eval(ProjLayout L, const ProjDev* __restrict__ pd,
                              const double* __restrict__ tab, int w,
                              const double* __restrict__ x, int n, double* __restrict__ out) {
  const WindowView W{tab + L.w_pp[w], L.NWp, pd->w_chi_min[w], pd->w_chi_max[w]};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[i] = W(x[i]);
}

// correlation.py:270-275
template <bool HF, bool BAO = false>   // HF: a HaloFit spectrum; BAO: wiggle transfer function
struct WthetaIntegrand {
  const PowerEval* P;
  const KernelView* K;
  double theta, inv_D2, ln_theta;
  __device__ __forceinline__ double operator()(double ln_k) const {
    const double k = exp(ln_k);
    return k * k / (2.0 * kPi) * P->template at_ln<HF, BAO>(ln_k, k) * inv_D2 * (*K)(ln_k + ln_theta);
  }
};

// Every theta integrates over the same ln k range, so all of them visit the same Romberg
// nodes and the theta-independent factor k^2/(2 pi) P(k)/D_z^2 of the integrand is
// tabulated once per call on the level-LT grid (level-major, as the sigma(R) and halo node
// tables): per node a theta block then only evaluates the kernel spline.
// grid ceil((2^LT + 1) / (256 kWthNodesPerThread)), block 256.
constexpr int kWthetaTabLevel = 20;     // 2^20 + 1 doubles = 8 MiB: the default divmax
constexpr int kWthNodesPerThread = 8;
template <bool HF, bool BAO>
__global__ __launch_bounds__(256) void k_wtheta_nodes(chomp_config cfg, TabLayout HL,
                                                      const Epoch* __restrict__ epochs, int e,
                                                      const double* __restrict__ htab, int which,
                                                      double k_min, double k_max, double D_z,
                                                      int LT, double* __restrict__ nodes) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  PowerEval P;
  P.stage(cfg, HL, &E, htab + (size_t)e * HL.stride, which, sm);
  __syncthreads();
  P.template finish_t<BAO>();
  const double a = log(k_min), b = log(k_max), intrange = b - a;
  // (kWthNodesPerThread nodes per thread: a block stages 5 KB of spline tables before its first
  //  node -- at one node per thread, 4097 blocks of it, that staging was most of the 33 us)
#pragma unroll 2
  for (int r = 0; r < kWthNodesPerThread; ++r) {
    const long idx = ((long)blockIdx.x * kWthNodesPerThread + r) * blockDim.x + threadIdx.x;
    if (idx > (1L << LT)) return;
    double x;
    if (idx < 2) {
      x = idx == 0 ? a : b;
    } else {                                   // the node arithmetic of chomp_romberg.h
      const unsigned m = (unsigned)(idx - 1);
      const int lev = 32 - __builtin_clz(m);
      const long j = (long)m - (1L << (lev - 1));
      const double h = ldexp(intrange, 1 - lev);
      x = (a + 0.5 * h) + h * (double)j;
    }
    const double k = exp(x);
    nodes[idx] = k * k / (2.0 * kPi) * P.template at_ln<HF, BAO>(x, k) * (1.0 / (D_z * D_z));
  }
}

// correlation.py:270-275 from the node table (levels <= LT), directly beyond
template <bool HF, bool BAO>
struct WthetaTabIntegrand {
  const double* nodes;
  int LT;
  WthetaIntegrand<HF, BAO> direct;
  __device__ __forceinline__ void operator()(double ln_k, double (&out)[1], int lev,
                                             long j) const {
    if (lev <= LT) {
      const long idx = lev == 0 ? j : 1 + (1L << (lev - 1)) + j;
      out[0] = nodes[idx] * (*direct.K)(ln_k + direct.ln_theta);
    } else {
      out[0] = direct(ln_k);
    }
  }
};

// grid n_theta, block 64 * kWthetaNW: one theta per workgroup.  The integrals of large
// theta run to 2^18..2^20 nodes (the kernel oscillates in ln k theta) and set the launch's
// duration, hence the wide group.
constexpr int kWthetaNW = 16;
template <bool HF, bool BAO>
__global__ __launch_bounds__(64 * kWthetaNW) void k_wtheta(chomp_config cfg, TabLayout HL, ProjLayout L,
                                                const Epoch* __restrict__ epochs, int e,
                                                const double* __restrict__ htab, int which,
                                                const ProjDev* __restrict__ pd,
                                                const double* __restrict__ ptab, double k_min,
                                                double k_max, double D_z,
                                                const double* __restrict__ theta,
                                                double* __restrict__ out,
                                                const double* __restrict__ nodes, int LT) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ double red[romberg_scratch<kWthetaNW, 2>()];
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  PowerEval P;
  P.stage(cfg, HL, &E, htab + (size_t)e * HL.stride, which, sm);
  double* kpp = sm + 12 * (HL.NK - 1);
  copy_doubles(kpp, ptab + L.k_pp, 4 * (L.NKT - 1));
  __syncthreads();
  P.template finish_t<BAO>();
  const KernelView K{kpp, L.NKT, pd->ln_kt_min, pd->ln_kt_max};
  const double th = theta[blockIdx.x];
  WthetaTabIntegrand<HF, BAO> f{nodes, LT, {&P, &K, th, 1.0 / (D_z * D_z), log(th)}};
  const RombergOut<1> r = romberg_group<kWthetaNW, 1>(f, log(k_min), log(k_max),
                                                      cfg.global_precision, cfg.corr_precision,
                                                      cfg.divmax, red);
  if (threadIdx.x == 0) out[blockIdx.x] = r.value[0];
}

// ---------------------------------------------------------------------------
// w(theta) without visiting the nodes.  The kernel K(ln k + ln theta) is a 50-knot cubic
// spline -- a piecewise cubic -- and the rest of the integrand is the theta-independent node
// table g_j of k_wtheta_nodes.  Over the nodes that fall into one piece of the spline the
// level sum is therefore sum_j g_j (c0 + c1 d_j + c2 d_j^2 + c3 d_j^3), d_j = x_j + ln theta -
// X_i: a combination of the MOMENTS sum g_j u_j^q (q = 0..3) of the node table over an index
// range, and those come from prefix sums that do not depend on theta.  A Romberg level of
// 2^19 nodes then costs a theta one prefix lookup per spline knot (50) instead of 2^19 spline
// evaluations -- the same sum up to rounding, SciPy's rows and stopping test unchanged.
//
// Conditioning: a moment about a far origin would be recombined with large cancellation
// ((x - X)^3 from powers of x ~ 6), so the ln k range is cut into segments about as wide as a
// spline piece (never narrower: a piece then meets at most two of them), each with its own
// origin, and the prefix sums restart at every segment.  Recombination shifts a segment's
// moments by |delta| <= two piece widths.  Segment membership is integer arithmetic on the node
// index (wth_seg_of / WthSeg::start), so that both kernels agree on it exactly.
// Measured agreement with the node-by-node kernel: tests/test_gpu_projection.py.
// ---------------------------------------------------------------------------
constexpr int kWthItems = 8;                 // nodes per thread and tile of k_wtheta_moments

// Node j of a level (x_j = a + (b - a)(j + 1/2) / n, n = 2^(lev - 1)) belongs to the segment its
// abscissa falls into, m = floor((2 j + 1) nseg / 2^lev): 0 <= x_j - O_m < the segment width.
__device__ __forceinline__ int wth_seg_of(long j, int nseg, int lev) {
  return (int)(((2 * j + 1) * nseg) >> lev);
}
// ... and the first node of segment m, the same statement solved for j (m = nseg: n):
// ceil((m 2^lev - nseg) / (2 nseg)), by a floating-point quotient made exact by three integer
// corrections (a 64-bit integer division costs a wavefront a few hundred instructions).
struct WthSeg {
  int nseg;
  double inv_q, a, seg_w;
  __device__ __forceinline__ WthSeg(int nseg_, double a_, double b_)
      : nseg(nseg_), inv_q(0.5 / (double)nseg_), a(a_), seg_w((b_ - a_) / (double)nseg_) {}
  __device__ __forceinline__ long start(int m, int lev) const {
    const long p = ((long)m << lev) - nseg, q = 2L * nseg;
    if (p <= 0) return 0;
    long c = (long)((double)p * inv_q) - 1;                // <= floor(p / q), short by 2 at most
    c += (c * q < p) ? 1 : 0;
    c += (c * q < p) ? 1 : 0;
    c += (c * q < p) ? 1 : 0;
    return c;
  }
  __device__ __forceinline__ double origin(int m) const { return a + seg_w * (double)m; }
};

// grid (nseg, LT, kWthParts), block 256: inclusive prefix sums of g_j u_j^q, q = 0..3, u_j = x_j
// - O_m, over the nodes of segment m of level blockIdx.y + 1 -- KEPT only where k_wtheta_fast
// cannot cheaply rebuild them: at every kWthItems-th node (rec: the record of node j with
// (j + 1) % kWthItems == 0 lives at index (2^(lev-1) + j + 1) / kWthItems, a heap over the levels;
// a look-up adds the <= kWthItems - 1 nodes behind the record itself) and at the last node of the
// segment (segtot[(lev nseg + m)]).  Four moments for every node were 32 MiB of writes per call
// (0.18 of the HBM rate for a scan); these are 4 MiB.  A long segment (deep levels) is cut into
// up to kWthParts parts, one block each: a part first sums the moments of the parts before it (a
// plain strided reduction of nodes already in L2) for its carry, then scans its own tiles of
// 256 * kWthItems nodes.  No block waits for another.
constexpr int kWthParts = 8;
__host__ __device__ inline size_t wth_rec_count(int LT) { return ((size_t)1 << LT) / kWthItems + 2; }
__global__ __launch_bounds__(256) void k_wtheta_moments(const double* __restrict__ nodes, int LT,
                                                        int nseg, double a, double b,
                                                        double* __restrict__ rec,
                                                        double* __restrict__ segtot) {
  __shared__ double wtot[2][4][4];
  __shared__ double ctot[4][4];
  const int m = blockIdx.x, lev = blockIdx.y + 1, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const long n = 1L << (lev - 1);
  const WthSeg G(nseg, a, b);
  const long j0 = G.start(m, lev), j1 = G.start(m + 1, lev);
  constexpr long kTile = 256L * kWthItems;
  const long len = j1 - j0;
  long parts = (len + kTile - 1) / kTile;
  parts = parts > kWthParts ? kWthParts : parts;
  if ((long)blockIdx.z >= parts) return;                    // (block-uniform; also len == 0)
  const long plen = (len + parts - 1) / parts;
  const long p0 = j0 + plen * blockIdx.z;
  const long p1 = p0 + plen < j1 ? p0 + plen : j1;
  const double h = (b - a) / (double)n, lox = a + 0.5 * h;
  const double O = G.origin(m);
  const long base = 1 + n;                                  // level-major index of j = 0
  static_assert((kWthItems & (kWthItems - 1)) == 0, "records at a power-of-two spacing");
  double carry[4] = {0.0, 0.0, 0.0, 0.0};
  if (p0 > j0) {                                            // (block-uniform)
    double r[4] = {0.0, 0.0, 0.0, 0.0};
    for (long jj = j0 + t; jj < p0; jj += 256 * 8) {       // (eight loads in flight)
      double g[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long j = jj + 256L * i;
        g[i] = j < p0 ? nodes[base + j] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const double u = (lox + h * (double)(jj + 256L * i)) - O;
        r[0] += g[i]; r[1] += g[i] * u; r[2] += g[i] * (u * u); r[3] += g[i] * (u * u * u);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double x = wave_sum(r[q]);
      if (lane == 0) ctot[w][q] = x;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) carry[q] = (ctot[0][q] + ctot[1][q]) + (ctot[2][q] + ctot[3][q]);
  }
  int buf = 0;
  for (long tile = p0; tile < p1; tile += kTile, buf ^= 1) {
    const long c0 = tile + (long)t * kWthItems;
    double v[kWthItems][4];
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < kWthItems; ++i) {
      const long j = c0 + i;
      if (j < p1) {
        const double g = nodes[base + j], u = (lox + h * (double)j) - O;
        s[0] += g; s[1] += g * u; s[2] += g * (u * u); s[3] += g * (u * u * u);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v[i][q] = s[q];
    }
    double inc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) inc[q] = s[q];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double y = __shfl_up(inc[q], off, 64);
        if (lane >= off) inc[q] += y;
      }
    }
    if (lane == 63) {
#pragma unroll
      for (int q = 0; q < 4; ++q) wtot[buf][w][q] = inc[q];
    }
    __syncthreads();
    double pre[4], next[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double before = 0.0, all = 0.0;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) {
        const double x = wtot[buf][ww][q];
        if (ww < w) before += x;
        all += x;
      }
      pre[q] = carry[q] + before + (inc[q] - s[q]);
      next[q] = carry[q] + all;
    }
    // of this thread's kWthItems consecutive nodes exactly one closes a record; the segment's
    // last node also leaves the segment's total
    const int i_rec = (int)((kWthItems - ((c0 + 1) & (kWthItems - 1))) & (kWthItems - 1));
#pragma unroll
    for (int i = 0; i < kWthItems; ++i) {
      const long j = c0 + i;
      if (j < p1 && (i == i_rec || j == j1 - 1)) {
        const double2 lo2 = make_double2(pre[0] + v[i][0], pre[1] + v[i][1]);
        const double2 hi2 = make_double2(pre[2] + v[i][2], pre[3] + v[i][3]);
        if (i == i_rec) {
          double2* o = reinterpret_cast<double2*>(rec + 4 * ((n + j + 1) / kWthItems));
          o[0] = lo2; o[1] = hi2;
        }
        if (j == j1 - 1) {
          double2* o = reinterpret_cast<double2*>(segtot + 4 * ((size_t)lev * nseg + m));
          o[0] = lo2; o[1] = hi2;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) carry[q] = next[q];
  }
}

// Moments of the whole of segment m (0 for an empty one).
__device__ __forceinline__ void wth_seg_total(const double* __restrict__ segtot, int m,
                                              const WthSeg& G, int lev, double (&T)[4]) {
  const long g0 = G.start(m, lev), g1 = G.start(m + 1, lev);
  if (g1 > g0) {
    const double2* p = reinterpret_cast<const double2*>(segtot + 4 * ((size_t)lev * G.nseg + m));
    const double2 x = p[0], y = p[1];
    T[0] = x.x; T[1] = x.y; T[2] = y.x; T[3] = y.y;
  } else {
    T[0] = T[1] = T[2] = T[3] = 0.0;
  }
}
// sum over a range of c0 + c1 d + c2 d^2 + c3 d^3, d = u + dl, from the range's moments in u.
__device__ __forceinline__ double wth_combine(const double (&D)[4], double dl, double c0, double c1,
                                              double c2, double c3) {
  const double p1 = D[1] + dl * D[0];
  const double p2 = D[2] + dl * (2.0 * D[1] + dl * D[0]);
  const double p3 = D[3] + dl * (3.0 * D[2] + dl * (3.0 * D[1] + dl * D[0]));
  return c0 * D[0] + c1 * p1 + c2 * p2 + c3 * p3;
}

// grid n_theta, block 256: one theta per workgroup, one Romberg level per wavefront and round
// (four levels a round); lane i <= NP holds knot i of the kernel spline: the first node at or
// beyond it and the prefix moments just below.  Levels <= LT only (the host keeps the
// node-by-node kernel for a divmax beyond the table).  Needs NKT <= 63.
__global__ __launch_bounds__(256) void k_wtheta_fast(chomp_config cfg, ProjLayout L,
                                                     const ProjDev* __restrict__ pd,
                                                     const double* __restrict__ ptab, double k_min,
                                                     double k_max, const double* __restrict__ theta,
                                                     double* __restrict__ out,
                                                     const double* __restrict__ nodes,
                                                     const double* __restrict__ rec,
                                                     const double* __restrict__ segtot, int LT,
                                                     int nseg) {
  __shared__ double level_sum[2][4];
  __shared__ double ctab[(kWthetaTabLevel + 1) * 32];   // the rows' weights (RombergRows2::ctab)
  romberg_weights_to_lds(ctab, LT);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double a = log(k_min), b = log(k_max);
  const double lo = pd->ln_kt_min, hi = pd->ln_kt_max;
  const int NP = L.NKT - 1;                              // spline pieces
  const double dxK = (hi - lo) / (double)NP;
  const double s = log(theta[blockIdx.x]);
  const double* pp = ptab + L.k_pp;
  const int piece = lane < NP ? lane : NP - 1;
  const double c0 = pp[4 * piece], c1 = pp[4 * piece + 1], c2 = pp[4 * piece + 2],
               c3 = pp[4 * piece + 3];
  const double k_lo = pp[0];                             // K below its range (kernel.py:723-725)
  const double X = lo + dxK * (double)lane;              // knot i (= hi for lane NP, up to rounding)
  const KernelView K{pp, L.NKT, lo, hi};
  RombergRows2 R;
  R.ctab = ctab;
  R.start(b - a, cfg.global_precision, cfg.corr_precision,
          0.5 * (nodes[0] * K(a + s) + nodes[1] * K(b + s)), 0.0, true, false);
  const WthSeg G(nseg, a, b);
  for (int g = 0; 4 * g < LT && !R.all_done(); ++g) {
    const int lv = 4 * g + 1 + wave;
    double contrib = 0.0;
    if (lv <= LT) {                                      // (wave-uniform)
      const long n = 1L << (lv - 1);
      const double h = (b - a) / (double)n, lox = a + 0.5 * h, inv_h = (double)n / (b - a);
      const long base = 1 + n;
      // B: the first node of this lane's region (piece i, or beyond the range for lane NP)
      long B = 0;
      if (lane < NP) {
        const double q = ceil((X - s - lox) * inv_h);
        B = q <= 0.0 ? 0 : (q >= (double)n ? n : (long)q);
      } else if (lane == NP) {                           // x + s == hi still inside (kernel.py:728)
        const double q = floor((hi - s - lox) * inv_h) + 1.0;
        B = q <= 0.0 ? 0 : (q >= (double)n ? n : (long)q);
      }
      // prefix moments just below B -- the record at or before node B - 1 and the nodes behind
      // it, or the nodes from the segment's start -- and the total of that segment (every load
      // issued before anything waits for it; for B = 0 valid addresses whose values are not used)
      const bool has = lane <= NP && B >= 1;
      const int mB = has ? wth_seg_of(B - 1, nseg, lv) : 0;
      const long jl = has ? B - 1 : 0;                   // the last node of the prefix
      const long js = has ? G.start(mB, lv) : 0;         // ... the first node of its segment
      const long jb = ((jl + 1) / kWthItems) * kWthItems;     // nodes < jb are in the record
      const bool from_rec = jb > js;
      const long i0 = from_rec ? jb : js;                // the nodes i0 .. jl are added here
      const double2* pR = reinterpret_cast<const double2*>(rec + 4 * (from_rec ? (n + jb) / kWthItems : 0));
      const double2* pT = reinterpret_cast<const double2*>(segtot + 4 * ((size_t)lv * nseg + mB));
      const double2 R01 = pR[0], R23 = pR[1], T01 = pT[0], T23 = pT[1];
      double gtail[kWthItems - 1];
#pragma unroll
      for (int i = 0; i < kWthItems - 1; ++i) {
        const long jn = i0 + i;
        gtail[i] = nodes[base + (jn <= jl ? jn : jl)];
      }
      double P[4] = {from_rec ? R01.x : 0.0, from_rec ? R01.y : 0.0, from_rec ? R23.x : 0.0,
                     from_rec ? R23.y : 0.0};
      {
        const double O = G.origin(mB);
#pragma unroll
        for (int i = 0; i < kWthItems - 1; ++i) {
          const long jn = i0 + i;
          const double g = jn <= jl ? gtail[i] : 0.0;
          const double u = (lox + h * (double)jn) - O;
          P[0] += g; P[1] += g * u; P[2] += g * (u * u); P[3] += g * (u * u * u);
        }
      }
      if (!has) { P[0] = P[1] = P[2] = P[3] = 0.0; }
      const long B0 = __shfl((long long)B, 0, 64);
      const int m0 = __shfl(mB, 0, 64);
      double below = 0.0;                                // plain sum of g over the segments < m0
      if (B0 >= 1) {
        for (int m = lane; m < m0; m += 64) {
          double T[4];
          wth_seg_total(segtot, m, G, lv, T);
          below += T[0];
        }
      }
      // the next knot's
      const long Bn = __shfl_down((long long)B, 1, 64);
      const int mn = __shfl_down(mB, 1, 64);
      double Pn[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) Pn[q] = __shfl_down(P[q], 1, 64);
      if (lane < NP && Bn > B) {
        if (mn == mB && B >= 1) {
          const double D[4] = {Pn[0] - P[0], Pn[1] - P[1], Pn[2] - P[2], Pn[3] - P[3]};
          contrib = wth_combine(D, G.origin(mB) + s - X, c0, c1, c2, c3);
        } else {
          // the rest of segment mB, whole segments between, the head of segment mn
          if (B >= 1) {
            const double D[4] = {T01.x - P[0], T01.y - P[1], T23.x - P[2], T23.y - P[3]};
            contrib = wth_combine(D, G.origin(mB) + s - X, c0, c1, c2, c3);
          }
          for (int m = B >= 1 ? mB + 1 : 0; m < mn; ++m) {
            double T[4];
            wth_seg_total(segtot, m, G, lv, T);
            contrib += wth_combine(T, G.origin(m) + s - X, c0, c1, c2, c3);
          }
          contrib += wth_combine(Pn, G.origin(mn) + s - X, c0, c1, c2, c3);
        }
      }
      // nodes below the kernel's range: K(lo) times the plain sum of g over [0, B_0)
      if (lane == 0) below += P[0];
      contrib += k_lo * below;
    }
    const double S = wave_sum(contrib);
    if (lane == 0) level_sum[g & 1][wave] = S;
    __syncthreads();
    for (int q = 0; q < 4 && 4 * g + 1 + q <= LT && !R.all_done(); ++q)
      R.advance(4 * g + 1 + q, level_sum[g & 1][q], 0.0);
  }
  if (threadIdx.x == 0) out[blockIdx.x] = R.value[0];
}

// ---------------------------------------------------------------------------
// Precision sweep of w(theta) (BASELINE.json configs[4]): the same
// integral with parts of the arithmetic in fp32.
//   CHOMP_PREC_F32_EVAL    integrand evaluated in fp32 from the fp64 tables, fp64 sums
//   CHOMP_PREC_F32_TABLES  spline coefficients rounded to fp32, fp64 evaluation and sums
//   CHOMP_PREC_F32_ALL     fp32 tables, fp32 evaluation, fp32 sums and extrapolation
// The fp64 mode (k_wtheta) is the product path; these exist to measure what each
// narrowing costs against the reference's numbers.
// ---------------------------------------------------------------------------
namespace f32 {

// fp32 restatement of PowerEval + KernelView (same branches, halo.py:314-320, 649-672,
// 1339-1360; kernel.py:714-729); tables are read as stored and narrowed on the fly.
struct Eval {
  const PowerEval* P;
  const KernelView* K;
  float theta, inv_D2;
  float eh_s, eh_omh, eh_alpha, eh_theta, amp2, ns, ln_H0;
  float k_s, beta_n, alpha_n, a_n, b_n, c_n, gamma_n, mu_n, nu_n, f1, f2, f3;

  __device__ __forceinline__ void init(const PowerEval* P_, const KernelView* K_, double th,
                                       double invD2) {
    P = P_; K = K_; theta = (float)th; inv_D2 = (float)invD2;
    const Epoch& E = *P->E;
    eh_s = (float)E.eh_s; eh_omh = (float)E.eh_omh; eh_alpha = (float)E.eh_alpha;
    eh_theta = (float)E.eh_theta; amp2 = (float)(E.amp * E.sigma_norm * E.sigma_norm);
    ns = (float)E.ns; ln_H0 = (float)E.ln_H0;
    k_s = (float)E.hf_k_s; beta_n = (float)E.hf_beta_n; alpha_n = (float)E.hf_alpha_n;
    a_n = (float)E.hf_a_n; b_n = (float)E.hf_b_n; c_n = (float)E.hf_c_n;
    gamma_n = (float)E.hf_gamma_n; mu_n = (float)E.hf_mu_n; nu_n = (float)E.hf_nu_n;
    f1 = (float)E.hf_f1; f2 = (float)E.hf_f2; f3 = (float)E.hf_f3;
  }
  __device__ __forceinline__ float poly(const double* c, int i, float d) const {
    const double* q = c + 4 * i;
    return fmaf(fmaf(fmaf((float)q[3], d, (float)q[2]), d, (float)q[1]), d, (float)q[0]);
  }
  __device__ __forceinline__ float spline(float x0, float dx, const double* c, int n,
                                          float xv) const {
    int i = (int)floorf((xv - x0) / dx);
    i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
    return poly(c, i, xv - (x0 + dx * (float)i));
  }
  __device__ __forceinline__ float delta_k(float lk, float k) const {
    const float t = 1.0f + 0.43f * k * eh_s;
    const float t2 = t * t;
    const float G = eh_omh * (eh_alpha + (1.0f - eh_alpha) / (t2 * t2));
    const float q = k * eh_theta / G;
    const float L0 = logf(2.0f * 2.7182818f + 1.8f * q);
    const float C0 = 14.2f + 731.0f / (1.0f + 62.5f * q);
    const float T = L0 / (L0 + C0 * q * q);
    return amp2 * expf((3.0f + ns) * (lk - ln_H0)) * T * T;
  }
  __device__ __forceinline__ float halofit(float lk, float k) const {
    const float dk = delta_k(lk, k);
    const float y = k / k_s;
    const float d2q = dk * (powf(1.0f + dk, beta_n) / (1.0f + alpha_n * dk) *
                            expf(-(y / 4.0f + y * y / 8.0f)));
    const float d2h = (a_n * powf(y, 3.0f * f1) /
                       (1.0f + b_n * powf(y, f2) + powf(c_n * f3 * y, 3.0f - gamma_n))) /
                      (1.0f + mu_n / y + nu_n / (y * y));
    return 2.0f * 9.8696044f / (k * k * k) * (d2q + d2h);
  }
  __device__ __forceinline__ float power(float lk, float k) const {
    const float x0 = (float)P->x0, dx = (float)P->dx;
    const float k_min = (float)P->k_min, k_max = (float)P->k_max;
    const float plin = 2.0f * 9.8696044f * delta_k(lk, k) / (k * k * k);
    if (P->w == CHOMP_P_LIN) return plin;
    const bool in = k >= k_min && k <= k_max;
    if (P->halofit) {
      const float pmm = halofit(lk, k);
      if (P->w == CHOMP_P_MM) return pmm;
      float ha = 0.0f, hb = 0.0f, pp = 0.0f;
      if (in) {
        ha = spline(x0, dx, P->ca, P->NK, lk);
        hb = spline(x0, dx, P->cb, P->NK, lk);
        pp = spline(x0, dx, P->cp, P->NK, lk);
      }
      return pmm * ha * hb + pp;
    }
    if (k < k_min) return plin * (float)P->c_lo;
    if (in) {
      return plin * spline(x0, dx, P->ca, P->NK, lk) * spline(x0, dx, P->cb, P->NK, lk) +
             spline(x0, dx, P->cp, P->NK, lk);
    }
    return 0.0f;
  }
  __device__ __forceinline__ float kernel(float x) const {
    const float lo = (float)K->lo, hi = (float)K->hi;
    const float dx = (hi - lo) / (float)(K->N - 1);
    if (x < lo) return poly(K->pp, 0, 0.0f);
    if (x <= hi) return spline(lo, dx, K->pp, K->N, x);
    return 0.0f;
  }
  // correlation.py:270-275
  __device__ __forceinline__ float operator()(float ln_k) const {
    const float k = expf(ln_k);
    return k * k / (2.0f * 3.14159265f) * power(ln_k, k) * inv_D2 * kernel(logf(k * theta));
  }
};

struct EvalAsDouble {          // fp32 integrand under the fp64 Romberg driver
  const Eval* f;
  __device__ __forceinline__ double operator()(double ln_k) const {
    return (double)(*f)((float)ln_k);
  }
};

// scipy.integrate.romberg in single precision, one integral per 256-thread block
// (plain level-by-level version: this path is for the accuracy sweep, not for speed).
// red: 8 floats of LDS.
template <class F>
__device__ float romberg_block(const F& f, float a, float b, float tol, float rtol, int divmax,
                               float* red) {
  const float len = b - a;
  float ordsum = 0.5f * (f(a) + f(b));
  float prev[24], cur[24];
  prev[0] = len * ordsum;
  float result = prev[0];
  int n = 1;
  for (int i = 1; i <= divmax; ++i) {
    const float h = len / (float)n;
    float part = 0.0f;
    for (int j = threadIdx.x; j < n; j += blockDim.x) part += f(a + 0.5f * h + h * (float)j);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    float tot = 0.0f;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) tot += red[wv];
    ordsum += tot;
    n *= 2;
    cur[0] = len * ordsum / (float)n;
    float p4 = 1.0f;
    for (int m = 1; m <= i; ++m) {
      p4 *= 4.0f;
      cur[m] = (p4 * cur[m - 1] - prev[m - 1]) / (p4 - 1.0f);
    }
    result = cur[i];
    const float err = fabsf(result - prev[i - 1]);
    for (int m = 0; m <= i; ++m) prev[m] = cur[m];
    if (err < tol || err < rtol * fabsf(result)) break;
  }
  return result;
}

}  // namespace f32

// grid n_theta, block 256: k_wtheta with a narrowed precision mode (MODE = CHOMP_PREC_*).
template <int MODE>
__global__ __launch_bounds__(256) void k_wtheta_mixed(chomp_config cfg, TabLayout HL, ProjLayout L,
                                                      const Epoch* __restrict__ epochs, int e,
                                                      const double* __restrict__ htab, int which,
                                                      const ProjDev* __restrict__ pd,
                                                      const double* __restrict__ ptab,
                                                      double k_min, double k_max, double D_z,
                                                      const double* __restrict__ theta,
                                                      double* __restrict__ out) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ double red[romberg_scratch<4, 2>()];
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  PowerEval P;
  P.stage(cfg, HL, &E, htab + (size_t)e * HL.stride, which, sm);
  double* kpp = sm + 12 * (HL.NK - 1);
  copy_doubles(kpp, ptab + L.k_pp, 4 * (L.NKT - 1));
  __syncthreads();
  if (MODE == CHOMP_PREC_F32_TABLES || MODE == CHOMP_PREC_F32_ALL) {
    const int n_tab = 12 * (HL.NK - 1) + 4 * (L.NKT - 1);
    for (int i = threadIdx.x; i < n_tab; i += blockDim.x) sm[i] = (double)(float)sm[i];
    __syncthreads();
  }
  P.finish();
  const KernelView K{kpp, L.NKT, pd->ln_kt_min, pd->ln_kt_max};
  double v;
  if (MODE == CHOMP_PREC_F32_TABLES) {
    if (P.halofit) {
      WthetaIntegrand<true> f{&P, &K, theta[blockIdx.x], 1.0 / (D_z * D_z), log(theta[blockIdx.x])};
      v = romberg1<4>(f, log(k_min), log(k_max), cfg.global_precision, cfg.corr_precision,
                      cfg.divmax, red);
    } else {
      WthetaIntegrand<false> f{&P, &K, theta[blockIdx.x], 1.0 / (D_z * D_z), log(theta[blockIdx.x])};
      v = romberg1<4>(f, log(k_min), log(k_max), cfg.global_precision, cfg.corr_precision,
                      cfg.divmax, red);
    }
  } else {
    f32::Eval f;
    f.init(&P, &K, theta[blockIdx.x], 1.0 / (D_z * D_z));
    if (MODE == CHOMP_PREC_F32_EVAL) {
      f32::EvalAsDouble g{&f};
      v = romberg1<4>(g, log(k_min), log(k_max), cfg.global_precision, cfg.corr_precision,
                      cfg.divmax, red);
    } else {
      v = (double)f32::romberg_block(f, (float)log(k_min), (float)log(k_max),
                                     (float)cfg.global_precision, (float)cfg.corr_precision,
                                     cfg.divmax, reinterpret_cast<float*>(red));
    }
  }
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}

// correlation.py:496-501 (Correlation3d._correlation_integrand)
template <bool BAO>
struct Xi3dIntegrand {
  const PowerEval* P;
  const BesselTab* B;
  double r;
  __device__ __forceinline__ double operator()(double ln_k) const {
    const double k = exp(ln_k);
    const double p = P->halofit ? P->template at_ln<true, BAO>(ln_k, k)
                                : P->template at_ln<false, BAO>(ln_k, k);
    return k * k / (2.0 * kPi) * p * bessel_j<0>(k * r, *B);
  }
};

// grid n_r, block 256: one separation per workgroup.
template <bool BAO>
__global__ __launch_bounds__(256) void k_xi3d(chomp_config cfg, TabLayout HL,
                                              const Epoch* __restrict__ epochs, int e,
                                              const double* __restrict__ htab, int which,
                                              const BesselTab* __restrict__ bess_g,
                                              double k_min, double k_max,
                                              const double* __restrict__ r,
                                              double* __restrict__ out) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ BesselTab B;
  __shared__ double red[romberg_scratch<4, 2>()];
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  copy_doubles(reinterpret_cast<double*>(&B), reinterpret_cast<const double*>(bess_g),
               (int)(sizeof(BesselTab) / sizeof(double)));
  PowerEval P;
  P.stage(cfg, HL, &E, htab + (size_t)e * HL.stride, which, sm);
  __syncthreads();
  P.template finish_t<BAO>();
  Xi3dIntegrand<BAO> f{&P, &B, r[blockIdx.x]};
  const double v = romberg1<4>(f, log(k_min), log(k_max), cfg.global_precision,
                               cfg.corr_precision, cfg.divmax, red);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}

// Not-a-knot cubic spline through (xk, yk) evaluated at x: one block; c: 4 (nk - 1)
// doubles, w: 2 nk doubles of scratch.
__global__ void k_spline_eval(const double* __restrict__ xk, const double* __restrict__ yk, int nk,
                              double* __restrict__ c, double* __restrict__ w,
                              const double* __restrict__ x, int n, int deriv,
                              double* __restrict__ out) {
  if (threadIdx.x == 0) spline_build(xk, yk, nk, c, w);
  __threadfence_block();
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (deriv == 0) {
      out[i] = spline_eval(xk, c, nk, x[i]);
    } else {                       // derivative of the piece that holds x
      int lo = 0, hi = nk - 2;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (x[i] >= xk[mid]) lo = mid; else hi = mid - 1;
      }
      const double* q = c + 4 * lo;
      const double d = x[i] - xk[lo];
      out[i] = fma(fma(3.0 * q[3], d, 2.0 * q[2]), d, q[1]);
    }
  }
}

// correlation.py:387-392
template <bool BAO>
struct CellIntegrand {
  const PowerEval* P;
  const ProjLds* G;
  double ell, inv_D2;
  __device__ __forceinline__ double operator()(double chi) const {
    const double D = G->me.growth_factor(G->me.redshift(chi));
    return P->template eval_t<BAO>(ell / chi) * inv_D2 * G->wa(chi) * G->wb(chi) * D * D / (chi * chi);
  }
};

// Every multipole integrates over the same chi range, so all of them visit the same Romberg
// nodes and everything in the integrand that depends on chi alone -- both windows, the growth
// factor (two spline look-ups each), 1 / chi^2, 1 / D_z^2 -- is tabulated once per call on
// the level-LT grid (level-major, as w(theta)'s k-only factor): F_j, ln chi_j and chi_j.  A
// multipole then only evaluates P(l / chi_j) per node, with ln k = ln l - ln chi_j for free.
// grid ceil((2^LT + 1) / 256), block 256; dynamic LDS ProjLds::doubles(L).
constexpr int kCellTabLevel = 16;       // 3 x (2^16 + 1) doubles = 1.5 MiB (the C_l integrals of
                                        // configs[3] / [4] stop at levels 7..15, a few at 14, 15)
__global__ __launch_bounds__(256) void k_cell_nodes(ProjLayout L, const ProjDev* __restrict__ pdg,
                                                    const double* __restrict__ ptab, double D_z,
                                                    int LT, double* __restrict__ nodes,
                                                    int* __restrict__ deep) {
  extern __shared__ __align__(16) double sm[];
  __shared__ ProjDev pd;
  if (blockIdx.x == 0 && threadIdx.x < 2) deep[threadIdx.x] = 0;   // k_cell's list: count, head
  copy_doubles(reinterpret_cast<double*>(&pd), reinterpret_cast<const double*>(pdg), kProjDoubles);
  __syncthreads();
  ProjLds G;
  G.stage(L, pd, ptab, sm);
  G.bess = nullptr;
  __syncthreads();
  const long N = (1L << LT) + 1;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N) return;
  const double a = pd.chi_min, b = pd.chi_max, intrange = b - a;
  double chi;
  if (idx < 2) {
    chi = idx == 0 ? a : b;
  } else {                                   // the node arithmetic of chomp_romberg.h
    const unsigned m = (unsigned)(idx - 1);
    const int lev = 32 - __builtin_clz(m);
    const long j = (long)m - (1L << (lev - 1));
    const double h = ldexp(intrange, 1 - lev);
    chi = (a + 0.5 * h) + h * (double)j;
  }
  const double D = G.me.growth_factor(G.me.redshift(chi));
  nodes[idx] = (1.0 / (D_z * D_z)) * G.wa(chi) * G.wb(chi) * D * D / (chi * chi);
  nodes[N + idx] = log(chi);
  nodes[2 * N + idx] = chi;
}

// P(k) of the spectrum a C_l call integrates, on a uniform grid of kPTabN intervals in ln k
// over [ln k_min, ln k_max] (one evaluation of PowerEval::at_ln per point, once per call):
// inside that range a multipole's node then reads P(l / chi) off a 6-point Lagrange stencil
// (~1e-16 relative at this spacing: the spectrum is smooth in ln k) instead of evaluating
// three knot splines and the linear spectrum -- or the HaloFit formula -- per node.  Outside
// the range: zero where the spectrum is identically zero, else evaluated directly (the rescaled
// linear spectrum / the extrapolations, halo.py:300-320).  No-wiggle spectra only.
// grid ceil((kPTabN + 1) / 256), block 256; dynamic LDS 12 (NK - 1) doubles.
constexpr int kPTabN = 8192;
template <bool HF, bool BAO>
__global__ __launch_bounds__(256) void k_cell_ptab(chomp_config cfg, TabLayout HL,
                                                   const Epoch* __restrict__ epochs, int e,
                                                   const double* __restrict__ htab, int which,
                                                   double* __restrict__ ptab_out) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  PowerEval P;
  P.stage(cfg, HL, &E, htab + (size_t)e * HL.stride, which, sm);
  __syncthreads();
  P.template finish_t<BAO>();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > kPTabN) return;
  const double x0 = log(cfg.k_min), x1 = log(cfg.k_max);
  double x = x0 + (x1 - x0) * ((double)i / (double)kPTabN);
  double k = exp(x);
  // (the end points exactly: exp(log(k_max)) may round past k_max, where the spectrum is 0)
  if (i == 0) { k = cfg.k_min; x = x0; }
  if (i == kPTabN) { k = cfg.k_max; x = x1; }
  ptab_out[i] = P.template at_ln<HF, BAO>(x, k);
}

// correlation.py:387-392 from the node table (levels <= LT), directly beyond
// DIRECT = false: the caller never asks for a level beyond the node table (k_cell below its
// hand-over level): the node-by-node route -- windows, growth, ~100 registers -- is not compiled.
template <bool HF, bool BAO, bool DIRECT = true>
struct CellTabIntegrand {
  const PowerEval* P;
  const double* nodes;
  long N;
  int LT;
  double ell, ln_ell;
  const double* ptab;      // k_cell_ptab (nullptr: evaluate the spectrum at every node)
  double px0, pdx, pinv_dx;
  CellIntegrand<BAO> direct;
  // The spectrum at ln k = lk from the table, as straight-line code: inside [ln k_min, ln k_max]
  // the 6-point stencil, shifted at the two ends so that it never leaves the table (P drops to
  // zero beyond k_max: a stencil must not straddle that); outside the range the value where it
  // is identically zero (halo.py:300-320 without extrapolation above k_max; :649-672 for the
  // HaloFit cross spectra on both sides).  *ok = false: neither -- the caller evaluates
  // PowerEval::at_ln.  ptab must not be null.
  __device__ __forceinline__ double from_table(double lk, bool* ok) const {
    const double u = (lk - px0) * pinv_dx;
    const bool in = u >= 0.0 && u <= (double)kPTabN;
    const bool zero_out = HF ? P->w != CHOMP_P_MM
                             : (u > 0.0 && !P->extrap && P->w != CHOMP_P_LIN);
    int i = (int)u;
    i = (in && i > 2) ? (i < kPTabN - 3 ? i : kPTabN - 3) : 2;
    const double t = u - (double)i;
    const double* q = ptab + i - 2;
    const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5];
    const double a = t + 2.0, b = t + 1.0, d = t - 1.0, e = t - 2.0, f = t - 3.0;
    const double ab = a * b, ef = e * f, cd = t * d;
    const double v = q0 * (b * cd * ef) * (-1.0 / 120.0) + q1 * (a * cd * ef) * (1.0 / 24.0) +
                     q2 * (ab * d * ef) * (-1.0 / 12.0) + q3 * (ab * t * ef) * (1.0 / 12.0) +
                     q4 * (ab * cd * f) * (-1.0 / 24.0) + q5 * (ab * cd * e) * (1.0 / 120.0);
    *ok = in || zero_out;
    return in ? v : 0.0;
  }
  __device__ __forceinline__ void operator()(double chi, double (&out)[1], int lev, long j) const {
    if (lev <= LT) {
      const long idx = lev == 0 ? j : 1 + (1L << (lev - 1)) + j;
      const double lk = ln_ell - nodes[N + idx];
      bool ok = false;
      double p = 0.0;
      if (ptab != nullptr) p = from_table(lk, &ok);
      if (!ok) p = P->template at_ln<HF, BAO>(lk, ell / nodes[2 * N + idx]);
      out[0] = p * nodes[idx];
    } else if constexpr (DIRECT) {
      out[0] = direct(chi);
    } else {
      out[0] = 0.0;
    }
  }
  // The same value for a node of the table that from_table answers, as straight-line code
  // (detail::fast_f); false otherwise.
  __device__ __forceinline__ bool fast(double, double (&out)[1], int lev, long j) const {
    if (ptab == nullptr || lev > LT || lev == 0) return false;     // (uniform over the batch)
    const long idx = 1 + (1L << (lev - 1)) + j;
    const double g = nodes[idx], lk = ln_ell - nodes[N + idx];
    bool ok;
    out[0] = from_table(lk, &ok) * g;
    return ok;
  }
};

// grid n_ell, block 256: one multipole per workgroup, up to Romberg level `split` (= divmax: the
// whole integral).  A multipole not converged there is listed for k_cell_deep with its Romberg
// state: deep[0] counts them, deep[2 + i] names them, state[kRombergDump * multipole].
// (Most multipoles of configs[3] / [4] stop at levels 7..9, a few dozen of the highest run to
//  14, 15: in one 256-thread block those walk 64..128 nodes per thread, each two dependent
//  reads of tables another kernel wrote -- a microsecond a piece on a CU whose L2 has not seen
//  them -- and set the launch's duration: 65 us measured against 9 us for a level-8 block.)
constexpr int kCellSplitLevel = 11;
template <bool HF, bool BAO, bool DIRECT>
// (three blocks per CU for the lean instance: 168 registers, one fewer than it would take)
__global__ __launch_bounds__(256, DIRECT ? 2 : 3) void k_cell(chomp_config cfg, TabLayout HL, ProjLayout L,
                                              const Epoch* __restrict__ epochs, int e,
                                              const double* __restrict__ htab, int which,
                                              const ProjDev* __restrict__ pdg,
                                              const double* __restrict__ ptab, double D_z,
                                              const double* __restrict__ ell,
                                              double* __restrict__ out,
                                              const double* __restrict__ nodes, int LT,
                                              const double* __restrict__ pk_tab, int split,
                                              int* __restrict__ deep, double* __restrict__ state) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ ProjDev pd;
  __shared__ double red[romberg_scratch<4, 2>()];
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  copy_doubles(reinterpret_cast<double*>(&pd), reinterpret_cast<const double*>(pdg), kProjDoubles);
  __syncthreads();
  PowerEval P;
  P.stage(cfg, HL, &E, htab + (size_t)e * HL.stride, which, sm);
  ProjLds G;
  G.stage(L, pd, ptab, sm + 12 * (HL.NK - 1));
  G.bess = nullptr;
  __syncthreads();
  P.template finish_t<BAO>();
  const double l = ell[blockIdx.x];
  const double px0 = log(cfg.k_min), pdx = (log(cfg.k_max) - px0) / (double)kPTabN;
  CellTabIntegrand<HF, BAO, DIRECT> f{&P, nodes, (1L << LT) + 1, LT, l, log(l), pk_tab, px0, pdx,
                                      1.0 / pdx, {&P, &G, l, 1.0 / (D_z * D_z)}};
  const bool hand_over = split < cfg.divmax;
  const RombergOut<1> r = romberg_group<4, 1, CellTabIntegrand<HF, BAO, DIRECT>, 4>(
      f, pd.chi_min, pd.chi_max, cfg.global_precision, cfg.corr_precision, split, red,
      hand_over ? state + (size_t)blockIdx.x * kRombergDump : nullptr);
  if (threadIdx.x == 0) {
    out[blockIdx.x] = r.value[0];
    if (hand_over && !r.converged[0]) deep[2 + atomicAdd(&deep[0], 1)] = (int)blockIdx.x;
  }
}

// k_cell4: the same integrals, FOUR multipoles to a block -- one per wavefront up to level
// kCell4Level (romberg_wave6: a multipole that stops at level 7..8, the rule, costs two to four
// nodes per lane and no block barrier), then the block's multipoles that go on are walked by all
// four wavefronts together up to `split` (RombergResume from the rows the wavefront left), and
// what has not converged there is listed for k_cell_deep exactly as k_cell lists it.  The block's
// staging (epoch record, spectrum splines, projection tables) is shared by its four multipoles,
// and the 2048 multipoles of configs[3] / [4] are 512 blocks: resident at once, where k_cell's
// 2048 blocks took three rounds.  Wavefront w of block b takes multipole b + gridDim.x w (the
// deep ones are the highest l: one to a block).  grid ceil(n_ell / 4), block 256; the lean
// instance only (split <= LT, split >= 6).
constexpr int kCell4Level = 8;     // (7: 27.6, 8: 24.3, 9: 25.2, 10: 28.7 us per configs[3] launch)
template <bool HF, bool BAO>
__global__ __launch_bounds__(256, 3) void k_cell4(chomp_config cfg, TabLayout HL, ProjLayout L,
                                              const Epoch* __restrict__ epochs, int e,
                                              const double* __restrict__ htab, int which,
                                              const ProjDev* __restrict__ pdg,
                                              const double* __restrict__ ptab, double D_z,
                                              const double* __restrict__ ell, int n_ell,
                                              double* __restrict__ out,
                                              const double* __restrict__ nodes, int LT,
                                              const double* __restrict__ pk_tab, int split,
                                              int* __restrict__ deep, double* __restrict__ state) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ ProjDev pd;
  __shared__ double red[romberg_scratch<4, 2>()];
  __shared__ double co_dump[4][kRombergDump];
  __shared__ double co_val[4];
  __shared__ int co_need[4], co_conv[4];
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  copy_doubles(reinterpret_cast<double*>(&pd), reinterpret_cast<const double*>(pdg), kProjDoubles);
  __syncthreads();
  PowerEval P;
  P.stage(cfg, HL, &E, htab + (size_t)e * HL.stride, which, sm);
  ProjLds G;
  G.stage(L, pd, ptab, sm + 12 * (HL.NK - 1));
  G.bess = nullptr;
  __syncthreads();
  P.template finish_t<BAO>();
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const int il = (int)blockIdx.x + (int)gridDim.x * wave;
  const bool have = il < n_ell;
  const double px0 = log(cfg.k_min), pdx = (log(cfg.k_max) - px0) / (double)kPTabN;
  const double a = pd.chi_min, b = pd.chi_max;
  const bool hand_over = split < cfg.divmax;
  const int top = split < kCell4Level ? split : kCell4Level;      // (the wavefronts' own levels)
  RombergOut<1> r;
  r.value[0] = 0.0; r.level[0] = 0; r.converged[0] = true;
  if (have) {
    const double l = ell[il];
    CellTabIntegrand<HF, BAO, false> f{&P, nodes, (1L << LT) + 1, LT, l, log(l), pk_tab, px0, pdx,
                                       1.0 / pdx, {&P, &G, l, 1.0 / (D_z * D_z)}};
    double fb[1];
    f(b, fb, 0, 1L);                             // (the upper end point: node 1 of level 0)
    r = romberg_wave6<1>(f, a, b, fb, cfg.global_precision, cfg.corr_precision, top, co_dump[wave]);
  }
  if (lane == 0) {
    co_need[wave] = (have && !r.converged[0] && split > top) ? 1 : 0;
    co_val[wave] = r.value[0];
    co_conv[wave] = r.converged[0] ? 1 : 0;
  }
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (!co_need[w]) continue;                   // (block-uniform)
    const int ilw = (int)blockIdx.x + (int)gridDim.x * w;
    const double l = ell[ilw];
    CellTabIntegrand<HF, BAO, false> f{&P, nodes, (1L << LT) + 1, LT, l, log(l), pk_tab, px0, pdx,
                                       1.0 / pdx, {&P, &G, l, 1.0 / (D_z * D_z)}};
    RombergResume R;
    R.load(co_dump[w], top, b - a, cfg.global_precision, cfg.corr_precision);
    int flip = 0;
    for (int i = top + 1; i <= split && !R.done; ++i) {
      const double c_il = CHOMP_ROMBERG_C[i][lane & 31];
      const long numtosum = 1L << (i - 1);
      const double h = ldexp(b - a, 1 - i);                    // ((b - a) / numtosum)
      const double lox = a + 0.5 * h;
      double part = 0.0;
      for (long j = threadIdx.x; j < numtosum; j += 256) {
        double v[1];
        f(lox + h * (double)j, v, i, j);
        part += v[0];
      }
      const double S = group_sum<4>(part, red, flip);
      R.advance(i, S, c_il);
    }
    __syncthreads();                             // (red: the last sum has been read)
    if (threadIdx.x < 64) {                      // (every wavefront holds the same rows)
      if (!R.done && hand_over) {
        double* st = state + (size_t)ilw * kRombergDump;
        if (lane < 32) st[lane] = R.Tl;
        if (lane == 0) { st[32] = R.ordsum; st[33] = R.prev; }
      }
      if (lane == 0) { co_val[w] = R.value; co_conv[w] = R.done ? 1 : 0; }
    }
  }
  __syncthreads();
  if (have && lane == 0) {
    out[il] = co_val[wave];
    if (hand_over && !co_conv[wave]) {
      // (a multipole that never entered the cooperative walk -- split <= kCell4Level -- hands
      //  on the rows its wavefront left)
      if (split <= top) {
        double* st = state + (size_t)il * kRombergDump;
        for (int q = 0; q < kRombergDump; ++q) st[q] = co_dump[wave][q];
      }
      deep[2 + atomicAdd(&deep[0], 1)] = il;
    }
  }
}

// One node by the general route (inlined: a call would put a stack -- scratch memory -- behind
// every launch of the kernel, used or not).
template <bool HF, bool BAO>
__device__ __forceinline__ double cell_node_general(const CellTabIntegrand<HF, BAO>& f, double chi,
                                                    int lev, long j) {
  double v[1];
  f(chi, v, lev, j);
  return v[0];
}

// grid <= n_ell (blocks draw multipoles from the list until it is empty), block 512: the
// levels beyond `split` of the multipoles k_cell listed.  The spectrum table (64 KiB) is staged
// in LDS -- the second of a node's two dependent reads then costs an LDS access -- and a level's
// nodes are spread over eight wavefronts, four to a thread in flight.
// dynamic LDS: 12 (NK - 1) + ProjLds::doubles(L) + kPTabN + 1 doubles.
constexpr int kCellDeepThreads = 512;
template <bool HF, bool BAO>
__global__ __launch_bounds__(kCellDeepThreads) void k_cell_deep(
    chomp_config cfg, TabLayout HL, ProjLayout L, const Epoch* __restrict__ epochs, int e,
    const double* __restrict__ htab, int which, const ProjDev* __restrict__ pdg,
    const double* __restrict__ ptab, double D_z, const double* __restrict__ ell,
    double* __restrict__ out, const double* __restrict__ nodes, int LT,
    const double* __restrict__ pk_tab, int split, int* __restrict__ deep,
    const double* __restrict__ state) {
  constexpr int NW = kCellDeepThreads / 64, NT = kCellDeepThreads, U = 4;
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ ProjDev pd;
  __shared__ double red[2 * NW];
  __shared__ int item_sh;
  const int count = deep[0];
  if ((int)blockIdx.x >= count) return;                  // (block-uniform)
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  copy_doubles(reinterpret_cast<double*>(&pd), reinterpret_cast<const double*>(pdg), kProjDoubles);
  __syncthreads();
  PowerEval P;
  P.stage(cfg, HL, &E, htab + (size_t)e * HL.stride, which, sm);
  ProjLds G;
  double* pk_lds = G.stage(L, pd, ptab, sm + 12 * (HL.NK - 1));
  G.bess = nullptr;
  if (pk_tab != nullptr) copy_doubles(pk_lds, pk_tab, kPTabN + 1);
  __syncthreads();
  P.template finish_t<BAO>();
  const double px0 = log(cfg.k_min), pdx = (log(cfg.k_max) - px0) / (double)kPTabN;
  const double a = pd.chi_min, b = pd.chi_max;
  int flip = 0;
  for (;;) {
    __syncthreads();                                     // (the previous item's item_sh is read)
    if (threadIdx.x == 0) item_sh = atomicAdd(&deep[1], 1);
    __syncthreads();
    if (item_sh >= count) return;                        // (block-uniform)
    const int il = deep[2 + item_sh];
    const double l = ell[il];
    const CellTabIntegrand<HF, BAO> f{&P, nodes, (1L << LT) + 1, LT, l, log(l),
                                      pk_tab != nullptr ? pk_lds : nullptr, px0, pdx, 1.0 / pdx,
                                      {&P, &G, l, 1.0 / (D_z * D_z)}};
    RombergResume R;
    R.load(state + (size_t)il * kRombergDump, split, b - a, cfg.global_precision,
           cfg.corr_precision);
    for (int lv = split + 1; lv <= cfg.divmax && !R.done; ++lv) {
      const double c_il = CHOMP_ROMBERG_C[lv][threadIdx.x & 31];   // (in flight behind the nodes)
      const long numtosum = 1L << (lv - 1);
      const double h = ldexp(b - a, 1 - lv), lox = a + 0.5 * h;   // ((b - a) / numtosum)
      double part = 0.0;
      long j = threadIdx.x;
      for (; j + (U - 1) * (long)NT < numtosum; j += U * (long)NT) {
        double v[U][1];
        bool ok = true;
#pragma unroll
        for (int u = 0; u < U; ++u)
          ok = f.fast(lox + h * (double)(j + u * (long)NT), v[u], lv, j + u * (long)NT) && ok;
        if (!ok) {
          for (int u = 0; u < U; ++u)
            v[u][0] = cell_node_general<HF, BAO>(f, lox + h * (double)(j + u * (long)NT), lv,
                                                 j + u * (long)NT);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) part += v[u][0];
      }
      for (; j < numtosum; j += NT) {
        double v[1];
        if (!f.fast(lox + h * (double)j, v, lv, j))
          v[0] = cell_node_general<HF, BAO>(f, lox + h * (double)j, lv, j);
        part += v[0];
      }
      R.advance(lv, group_sum<NW>(part, red, flip), c_il);
    }
    if (threadIdx.x == 0) out[il] = R.value;
  }
}

// ---------------------------------------------------------------------------
// Gaussian covariance of w(theta): Covariance(corr, corr), covariance.py:361-543
// ---------------------------------------------------------------------------
// Table layout of the covariance state: ln K knots | projected spectrum | its spline |
// scalars {ln_K_min, ln_K_max, D_z, chi_peak}.
struct CovLayout {
  int N, ln_K, proj, pp, lev, scal, work, total;
};
inline CovLayout make_cov_layout(int N) {
  CovLayout C;
  C.N = N;
  int o = 0;
  C.ln_K = o; o += N;
  C.proj = o; o += N;
  C.pp = o; o += 4 * (N - 1);
  C.lev = o; o += N;
  C.scal = o; o += 8;
  C.work = o; o += 2 * N;
  C.total = (o + 7) & ~7;
  return C;
}

// covariance.py:545-552 with kernel.py:1066-1071 (_halo_a_integrand)
template <bool BAO>
struct CovProjIntegrand {
  const PowerEval* P;
  const ProjLds* G;
  double K, norm;
  __device__ __forceinline__ double operator()(double chi) const {
    const double D = G->me.growth_factor(G->me.redshift(chi));
    // Where the lower limit is chi = K / k_max (the large-K knots), the end point's wavenumber
    // K / (K / k_max) is k_max or the double above it, by the luck of two roundings -- and halo
    // spectra are zero above k_max (halo.py:649-672): half an end-point value more or less in
    // every trapezoid sum, four Romberg levels and 1e-6 of the knot.  The reference's own knots
    // (G12) all sit on the k_max side; a wavenumber within 2^-50 above k_max is k_max here.
    double k = K / chi;
    if (k > P->k_max && k <= P->k_max * (1.0 + 8.9e-16)) k = P->k_max;
    return norm * P->template eval_t<BAO>(k) * (G->wa(chi) * G->wb(chi) * D * D / (chi * chi));
  }
};

// grid N (= kernel_npoints), block 256: one ln K knot of Covariance._halo_a_spline per
// workgroup (covariance.py:455-543, the matching_corrs branch).
template <bool BAO>
__global__ __launch_bounds__(256) void k_cov_proj_knots(chomp_config cfg, TabLayout HL,
                                                        ProjLayout L, CovLayout C,
                                                        const Epoch* __restrict__ epochs, int e,
                                                        const double* __restrict__ htab, int which,
                                                        const ProjDev* __restrict__ pdg,
                                                        const double* __restrict__ ptab,
                                                        double D_z, double* __restrict__ ctab) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ ProjDev pd;
  __shared__ double red[romberg_scratch<4, 2>()];
  __shared__ double chi_peak_s;
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  copy_doubles(reinterpret_cast<double*>(&pd), reinterpret_cast<const double*>(pdg), kProjDoubles);
  __syncthreads();
  PowerEval P;
  P.stage(cfg, HL, &E, htab + (size_t)e * HL.stride, which, sm);
  ProjLds G;
  G.stage(L, pd, ptab, sm + 12 * (HL.NK - 1));
  G.bess = nullptr;
  if (threadIdx.x == 0)                                  // :466: a growth factor used as z
    chi_peak_s = me_view(L, pd, ptab, 0).comoving_distance(D_z);
  __syncthreads();
  P.template finish_t<BAO>();
  const int i = blockIdx.x;
  // pd.chi_min / chi_max are Covariance._chi_min_a / _chi_max_a (:132-141 = kernel.py:610-612)
  const double ln_K_min = log(cfg.k_min * pd.chi_min), ln_K_max = log(cfg.k_max * pd.chi_max);
  const double ln_K = linspace_at(ln_K_min, ln_K_max, C.N, i);
  const double K = exp(ln_K);
  double chi_min = K / cfg.k_max, chi_max = K / cfg.k_min;
  if (chi_min < pd.chi_min) chi_min = pd.chi_min;
  if (chi_max > pd.chi_max) chi_max = pd.chi_max;
  CovProjIntegrand<BAO> f{&P, &G, K, 1.0};
  const double norm_int = f(chi_peak_s);
  f.norm = norm_int > 0.0 ? 1.0 / norm_int : 1.0;
  int level = 0;
  const double v = romberg1<4>(f, chi_min, chi_max, cfg.global_precision, cfg.corr_precision,
                               cfg.divmax, red, &level);
  if (threadIdx.x == 0) {
    ctab[C.ln_K + i] = ln_K;
    ctab[C.proj + i] = v / f.norm;
    ctab[C.lev + i] = (double)level;
    if (i == 0) {
      ctab[C.scal + 0] = ln_K_min;
      ctab[C.scal + 1] = ln_K_max;
      ctab[C.scal + 2] = D_z;
      ctab[C.scal + 3] = chi_peak_s;
    }
  }
}

__global__ void k_cov_spline(CovLayout C, double* __restrict__ ctab) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    spline_build(ctab + C.ln_K, ctab + C.proj, C.N, ctab + C.pp, ctab + C.work);
}

// covariance.py:397-453 (_covariance_G_integrand, matching_corrs: the second two-point
// term repeats the first)
struct CovGIntegrand {
  const double *xk, *pp;
  int N;
  const BesselTab* B;
  double theta_a, theta_b, inv_D2, poiss_a, poiss_b, norm;
  __device__ __forceinline__ double operator()(double ln_K) const {
    const double K = exp(ln_K);
    const double s = spline_eval(xk, pp, N, log(K));
    const double Pa = s * inv_D2, Pb = s * inv_D2;
    const double t1 = Pa * Pb + Pa * poiss_b + Pb * poiss_a;
    return K * K * norm * (t1 + t1) * bessel_j<0>(K * theta_a, *B) * bessel_j<0>(K * theta_b, *B);
  }
};

// grid n pairs, block 256: Covariance.covariance_G(theta_a, theta_b) (covariance.py:361-395).
__global__ __launch_bounds__(256) void k_cov_gaussian(chomp_config cfg, CovLayout C,
                                                      const double* __restrict__ ctab,
                                                      const BesselTab* __restrict__ bess_g,
                                                      double j0_limit, double area,
                                                      double poiss_a, double poiss_b,
                                                      const double* __restrict__ theta_a,
                                                      const double* __restrict__ theta_b,
                                                      double* __restrict__ out,
                                                      double* __restrict__ levels) {
  extern __shared__ __align__(16) double sm[];
  __shared__ BesselTab B;
  __shared__ double red[romberg_scratch<4, 2>()];
  double* xk = sm;
  double* pp = sm + C.N;
  copy_doubles(xk, ctab + C.ln_K, C.N);
  copy_doubles(pp, ctab + C.pp, 4 * (C.N - 1));
  copy_doubles(reinterpret_cast<double*>(&B), reinterpret_cast<const double*>(bess_g),
               (int)(sizeof(BesselTab) / sizeof(double)));
  __syncthreads();
  const double ln_K_min = ctab[C.scal + 0], ln_K_hi = ctab[C.scal + 1], D_z = ctab[C.scal + 2];
  const double ta = theta_a[blockIdx.x], tb = theta_b[blockIdx.x];
  double ln_K_max = log(fmax(j0_limit / ta, j0_limit / tb));
  double v = 0.0;
  int level = 0;
  if (ln_K_max > ln_K_hi) ln_K_max = ln_K_hi;
  if (ln_K_max > ln_K_min) {
    CovGIntegrand f{xk, pp, C.N, &B, 0.0, 0.0, 1.0 / (D_z * D_z), poiss_a, poiss_b, 1.0};
    const double norm = 1.0 / f(0.0);                    // :390
    f.theta_a = ta;
    f.theta_b = tb;
    f.norm = norm;
    v = romberg1<4>(f, ln_K_min, ln_K_max, cfg.global_precision, cfg.corr_precision,
                    cfg.divmax, red, &level) / (norm * 2.0 * kPi * area);
  }
  if (threadIdx.x == 0) {
    out[blockIdx.x] = v;
    if (levels) levels[blockIdx.x] = (double)level;
  }
}

}  // namespace chomp
