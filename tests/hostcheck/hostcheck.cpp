// TEST-ONLY harness: compiles the fp64 building blocks of chomp_amd/csrc/chomp_math.h
// for the CPU (g++) so the special functions, the not-a-knot spline and the
// integrand pieces can be checked against SciPy / the oracle without a GPU.
// It is never loaded by the chomp_amd package and is not a fallback of anything.
#include <cstring>
#include "../../chomp_amd/csrc/chomp_math.h"

using namespace chomp;
static SiCiTab g_sici;
static BesselTab g_j0, g_j2;
static bool g_init = false;
static void init() {
  if (!g_init) { fill_tables(&g_sici, &g_j0, &g_j2); g_init = true; }
}

extern "C" {
void hc_sici(const double* x, int n, double* si, double* ci) {
  init();
  for (int i = 0; i < n; ++i) sici(x[i], g_sici, &si[i], &ci[i]);
}
void hc_bessel(int order, const double* x, int n, double* out) {
  init();
  for (int i = 0; i < n; ++i)
    out[i] = order == 0 ? bessel_j<0>(x[i], g_j0) : bessel_j<2>(x[i], g_j2);
}
void hc_spline(const double* x, const double* y, int n, const double* xe, int ne,
               double* out, int uniform) {
  double* c = new double[4 * (n - 1)];
  double* w = new double[2 * n];
  spline_build(x, y, n, c, w);
  for (int i = 0; i < ne; ++i)
    out[i] = uniform ? spline_eval_uniform(x[0], (x[n - 1] - x[0]) / (n - 1), c, n, xe[i])
                     : spline_eval(x, c, n, xe[i]);
  delete[] c; delete[] w;
}
// serial emulation of the parallel (PCR) spline build used by the kernels
void hc_spline_pcr(const double* x, const double* y, int n, const double* xe, int ne,
                   double* out) {
  double* w = new double[9 * n + 4 * n];
  double *a[2] = {w, w + 4 * n}, *b[2] = {w + n, w + 5 * n}, *c[2] = {w + 2 * n, w + 6 * n},
         *d[2] = {w + 3 * n, w + 7 * n};
  double* sl = w + 8 * n;
  double* cf = w + 9 * n;
  for (int i = 0; i < n; ++i) spline_row(x, y, n, i, &a[0][i], &b[0][i], &c[0][i], &d[0][i]);
  int cur = 0;
  for (int s = 1; s < n; s *= 2) {
    for (int i = 0; i < n; ++i)
      pcr_step(n, i, s, a[cur], b[cur], c[cur], d[cur], a[cur ^ 1], b[cur ^ 1], c[cur ^ 1], d[cur ^ 1]);
    cur ^= 1;
  }
  for (int i = 0; i < n; ++i) sl[i] = d[cur][i] / b[cur][i];
  for (int i = 0; i < n - 1; ++i) spline_coef(x, y, sl, i, cf);
  for (int i = 0; i < ne; ++i) out[i] = spline_eval(x, cf, n, xe[i]);
  delete[] w;
}
void hc_quintic(const double* x, const double* y, int n, double xq, double* d) {
  double* w = new double[(n + 6) + 11 * n + n];
  quintic_derivs(x, y, n, xq, w, &d[0], &d[1]);
  delete[] w;
}
// epoch background + linear power + sigma integrand + y_nfw + mass function + HOD
void hc_epoch(const double* cosmo, double z, double sigma_norm, Epoch* e) {
  std::memset(e, 0, sizeof(Epoch));
  e->om0 = cosmo[0]; e->ob0 = cosmo[1]; e->ol0 = cosmo[2]; e->or0 = cosmo[3];
  e->tcmb = cosmo[4]; e->h = cosmo[5]; e->sigma8 = cosmo[6]; e->ns = cosmo[7];
  e->z = z;
  epoch_background(*e, 1.48e-8, 0.001, 100.0);
  e->sigma_norm = sigma_norm;
}
void hc_epoch_bao(const double* cosmo, double z, double sigma_norm, Epoch* e) {
  hc_epoch(cosmo, z, sigma_norm, e);
  epoch_background(*e, 1.48e-8, 0.001, 100.0, 1);
  e->sigma_norm = sigma_norm;
}
void hc_transfer(const Epoch* e, const double* k, int n, double* out) {
  for (int i = 0; i < n; ++i) out[i] = transfer_function(*e, k[i]);
}
int hc_sizeof_epoch() { return (int)sizeof(Epoch); }
void hc_linear_power(const Epoch* e, const double* k, int n, double* out) {
  for (int i = 0; i < n; ++i) out[i] = linear_power(*e, k[i]);
}
void hc_fast_sincos(const double* x, int n, double* s, double* c) {
  for (int i = 0; i < n; ++i) fast_sincos(x[i], &s[i], &c[i]);
}
void hc_fast_log(const double* x, int n, double* out) {
  for (int i = 0; i < n; ++i) out[i] = fast_log(x[i]);
}
// Stage E: amp * sigma_norm^2 * power_shape(k) == linear_power(k)
void hc_power_shape(const Epoch* e, const double* k, int n, double* out) {
  for (int i = 0; i < n; ++i)
    out[i] = e->amp * e->sigma_norm * e->sigma_norm * power_shape(*e, fast_log(k[i]), k[i]);
}
void hc_sigma_integrand(const Epoch* e, double scale, const double* lnk, int n, double* out) {
  SigmaIntegrand f{e, scale};
  for (int i = 0; i < n; ++i) out[i] = f(lnk[i]);
}
void hc_scalars(const Epoch* e, double* out) {
  out[0] = e->growth; out[1] = e->omega_m_z; out[2] = e->omega_l_z; out[3] = e->delta_c;
  out[4] = e->delta_v; out[5] = e->rho_bar; out[6] = e->delta_H;
}
void hc_y_nfw(Epoch* e, double c0, double beta, double delta_v_in, double m_star,
              const double* lnk, const double* lnm, int n, double* out) {
  init();
  e->m_star = m_star;
  halo_constants(*e, c0, beta, delta_v_in);
  for (int i = 0; i < n; ++i) out[i] = y_nfw(*e, g_sici, lnk[i], lnm[i]);
}
void hc_mass_function(Epoch* e, int kind, const double* par, const double* nu, int n,
                      double* f, double* b) {
  e->mf_kind = kind;
  if (kind == 0) { e->stq = par[0]; e->st_a = par[1]; e->f_norm = par[2]; e->bias_norm = par[3]; }
  else {
    e->mf_delta_v = par[0]; e->t_alpha = par[1]; e->t_beta = par[2]; e->t_gamma = par[3];
    e->t_phi = par[4]; e->t_eta = par[5]; e->bias_norm = par[6]; e->f_norm = 1.0;
    tinker_bias_constants(*e);
  }
  for (int i = 0; i < n; ++i) { f[i] = f_nu(*e, nu[i]); b[i] = bias_nu(*e, nu[i]); }
}
void hc_zheng(Epoch* e, const double* hod, const double* mass, int n, double* n1, double* n2) {
  e->hod_log_M_min = hod[0]; e->hod_sigma = hod[1]; e->hod_log_M_0 = hod[2];
  e->hod_log_M_1p = hod[3]; e->hod_alpha = hod[4];
  e->hod_M0 = pow(10.0, hod[2]); e->hod_M1p = pow(10.0, hod[3]);
  for (int i = 0; i < n; ++i) { n1[i] = zheng_first(*e, mass[i]); n2[i] = zheng_second(*e, mass[i]); }
}
}
