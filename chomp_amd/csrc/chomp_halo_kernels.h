// chomp_halo_kernels.h -- HIP kernels of the halo-model path (gfx950).
//
// Stage K (tables, per epoch = one (cosmology, z) pair):
//   k_epoch_init    SingleEpoch.__init__ (cosmology.py:39-119) + the mass-limit
//                   search of MassFunction._set_mass_limits (mass_function.py:160-203)
//   k_nu_table      MassFunction._initialize_splines nu_m loop (mass_function.py:205-210)
//   k_mass_setup    splines, m_star, f/bias normalisation (mass_function.py:212-241,
//                   Tinker 532-564)
//   k_halo_knots    n_bar (halo.py:674-700) and the 50-knot integrals h_m, pp_mm,
//                   h_g, pp_gm, pp_gg (halo.py:904-1086)
//   k_halo_finalize normalisations + not-a-knot splines over ln k (halo.py:916-918,
//                   959-961, 983-986, 1026-1029, 1072-1075)
// Stage E (grid evaluation):
//   k_power         Halo.linear_power/power_mm/power_gm/power_gg (halo.py:266-439),
//                   HaloFit.power_* (halo.py:1325-1413)
//
// Execution shape: one integral (or a pair sharing nodes) per group of wavefronts;
// tables (per-epoch splines, Si/Ci Chebyshev coefficients, Gauss nodes) are staged
// in LDS; reductions are wavefront __shfl_xor butterflies.  No MFMA: there is no
// dense contraction anywhere on this path.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/chomp_mi355x.h"
#include "chomp_math.h"
#include "chomp_romberg.h"

namespace chomp {

// Per-epoch table block (doubles), offsets fixed by the context's point counts.
struct TabLayout {
  int NM, NK;
  int off_ln_mass, off_nu, off_nu_pp, off_lnm_pp;
  int off_knot[5], off_kpp[5];
  int off_levels, off_hf_lns2, off_misc;   // misc[0] = n_bar / rho_bar (raw integral)
  int stride;
};

inline TabLayout make_layout(int NM, int NK) {
  TabLayout L;
  L.NM = NM;
  L.NK = NK;
  int o = 0;
  L.off_ln_mass = o; o += NM;
  L.off_nu = o; o += NM;
  L.off_nu_pp = o; o += 4 * (NM - 1);
  L.off_lnm_pp = o; o += 4 * (NM - 1);
  for (int f = 0; f < 5; ++f) { L.off_knot[f] = o; o += NK; }
  for (int f = 0; f < 5; ++f) { L.off_kpp[f] = o; o += 4 * (NK - 1); }
  L.off_levels = o; o += 5 * NK;
  L.off_hf_lns2 = o; o += NK;
  L.off_misc = o; o += 8;
  L.stride = (o + 7) & ~7;
  return L;
}

// Families: index into off_knot / off_kpp.
enum { F_HM = 0, F_PPMM = 1, F_HG = 2, F_PPGM = 3, F_PPGG = 4 };

// Node tables of the halo integrals: every knot k of an epoch integrates over the
// SAME ln(nu) nodes, so everything that does not depend on k (nu f(nu), b(nu), M(nu),
// concentration, r_s, HOD moments) is tabulated once per (epoch, integration range)
// on the level-kNodeLevel Romberg grid, stored level by level so a level's nodes are
// contiguous.  Deeper levels fall back to direct evaluation.
constexpr int kNodeLevel = 10;
constexpr int kNodeCount = (1 << kNodeLevel) + 1;
constexpr int kNodeFields = 7;   // wA, wB, ln_rs, con, ln_cp, inv_mass_k, flag
constexpr int kNodeStride = kNodeFields * kNodeCount + 8;   // doubles per (epoch, group);
                                                            // tail: integration limits a, b
__host__ __device__ inline int node_index(int lev, long j) {
  return lev == 0 ? (int)j : 1 + (1 << (lev - 1)) + (int)j;
}

constexpr int kKnotScratch = romberg_scratch<4, 2>();   // LDS doubles of a knot block
constexpr int kSearchJ = 2048;        // candidates per walking direction
constexpr int kEpochDoubles = (int)(sizeof(Epoch) / sizeof(double));
static_assert(sizeof(Epoch) % sizeof(double) == 0, "Epoch must be 8-byte granular");

__device__ __forceinline__ bool same_cosmology(const Epoch& a, const Epoch& b) {
  return a.om0 == b.om0 && a.ob0 == b.ob0 && a.ol0 == b.ol0 && a.or0 == b.or0 &&
         a.tcmb == b.tcmb && a.h == b.h && a.ns == b.ns;
}

// Cooperative copy of POD blocks as doubles.
__device__ __forceinline__ void copy_doubles(double* dst, const double* src, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// sigma(R) node table.  For 14.0662/k_max < R < 0.1/k_min (0.14 < R < 100 Mpc/h at
// the default limits) SingleEpoch.sigma_r integrates over the fixed range
// [ln k_min, ln k_max] (cosmology.py:611-632), so every such integral of an epoch
// visits the same ln k nodes: k_j and the R-independent factor
// (k_j/H0)^(3+n) T(k_j)^2 of Delta^2 are tabulated once per epoch on the
// level-kSigmaLevel Romberg grid (level-major order) by k_sigma_nodes.
constexpr int kSigmaLevel = 13;
constexpr int kSigmaCount = (1 << kSigmaLevel) + 1;
// Coarse table of ln S(R), S = int dlnk (k/H0)^(3+n) T^2 W(kR)^2 over sigma_r's own
// limits, used only to AIM the mass-limit search (k_epoch_init).  S(R) has kinks where
// sigma_r's integration limits switch (cosmology.py:611-632): at R = 14.0662 / k_max,
// that / 100, 0.1 / k_min and that x 100.  The table is therefore four segments between
// those radii, each uniform in ln R, and is never interpolated across a kink.
constexpr int kSGrid = 48;
// Segment s covers ln R in [b[s], b[s + 1]] with kSegN[s] points starting at kSegOff[s].
// (Scalars and if-chains rather than arrays: dynamic indexing would put them in scratch.)
struct SGrid {
  double b0, b1, b2, b3, b4;
  bool valid;
  // segment of x: lo, hi, number of points, offset of its first point
  __host__ __device__ void segment(double x, double* lo, double* hi, int* n, int* off) const {
    if (x <= b1) { *lo = b0; *hi = b1; *n = 8; *off = 0; }
    else if (x <= b2) { *lo = b1; *hi = b2; *n = 14; *off = 8; }
    else if (x <= b3) { *lo = b2; *hi = b3; *n = 18; *off = 22; }
    else { *lo = b3; *hi = b4; *n = 8; *off = 40; }
  }
  __host__ __device__ double ln_r(int i) const {
    double lo, hi;
    int n, off;
    if (i < 8) { lo = b0; hi = b1; n = 8; off = 0; }
    else if (i < 22) { lo = b1; hi = b2; n = 14; off = 8; }
    else if (i < 40) { lo = b2; hi = b3; n = 18; off = 22; }
    else { lo = b3; hi = b4; n = 8; off = 40; }
    return lo + (hi - lo) * (double)(i - off) / (double)(n - 1);
  }
};
__host__ __device__ inline SGrid make_sgrid(double k_min, double k_max) {
  SGrid g;
  const double a = log(14.0662 / k_max) - log(100.0), b = log(14.0662 / k_max),
               c = log(0.1 / k_min);
  g.b0 = a - log(20.0);                          // the walks of z <~ 1.5 end above R_a / 2
  g.b1 = a; g.b2 = b; g.b3 = c;
  g.b4 = c + log(3.0);
  g.valid = b < c;
  return g;
}
constexpr int kSigmaOffI8 = 2 * kSigmaCount, kSigmaOffLnS = 2 * kSigmaCount + 8;
constexpr int kSigmaStride = 2 * kSigmaCount + 8 + kSGrid;  // doubles per cosmology: k[], d2[],
                                                   // I8 = int dlnk d2 W(8k)^2 (sigma_8 norm.), ln S[]

// Everything here depends on the cosmology only, not on z, so it is built once per
// distinct cosmology of the batch ("slot"; the z-axis of a (k, z) grid is one slot).
// grid (ceil(kSigmaCount / 256) + 1 + kSGrid, n_slots + ceil(n_epoch / 256)), block 256;
// first[s] = an epoch that has cosmology s.  For y < n_slots: after the node-table
// blocks, one x-block does the sigma_8 integral (cosmology.py:118-119) and kSGrid blocks
// the coarse ln S(R) table, all by direct evaluation.  The rows y >= n_slots fill in the
// closed-form part of every epoch record (SingleEpoch.__init__ minus its two integrals),
// one epoch per thread, while the cosmology-only integrals run.
__global__ __launch_bounds__(256) void k_sigma_nodes(chomp_config cfg,
                                                     const chomp_cosmo* __restrict__ cosmo,
                                                     const double* __restrict__ zin,
                                                     const int* __restrict__ first,
                                                     const int* __restrict__ slots, int n_slots,
                                                     int n_epoch, Epoch* __restrict__ epochs,
                                                     double* __restrict__ snodes) {
  __shared__ Epoch E;
  __shared__ double red[romberg_scratch<4, 1>()];
  if ((int)blockIdx.y >= n_slots) {
    const int e = ((int)blockIdx.y - n_slots) * 256 + (int)threadIdx.x;
    if (blockIdx.x != 0 || e >= n_epoch) return;
    Epoch B;
    double* q = reinterpret_cast<double*>(&B);
    for (int i = 0; i < kEpochDoubles; ++i) q[i] = 0.0;
    const chomp_cosmo c = cosmo[e];
    B.om0 = c.omega_m0; B.ob0 = c.omega_b0; B.ol0 = c.omega_l0; B.or0 = c.omega_r0;
    B.tcmb = c.cmb_temp; B.h = c.h; B.sigma8 = c.sigma_8; B.ns = c.n_scalar;
    B.z = zin[e];
    epoch_background(B, cfg.cosmo_precision, cfg.k_min, cfg.k_max);
    B.cosmo_slot = slots[e];
    epochs[e] = B;
    return;
  }
  const int slot = blockIdx.y, e = first[slot];
  if (threadIdx.x == 0) {
    const chomp_cosmo c = cosmo[e];
    E.om0 = c.omega_m0; E.ob0 = c.omega_b0; E.ol0 = c.omega_l0; E.or0 = c.omega_r0;
    E.tcmb = c.cmb_temp; E.h = c.h; E.sigma8 = c.sigma_8; E.ns = c.n_scalar;
    E.z = zin[e];
    epoch_background(E, cfg.cosmo_precision, cfg.k_min, cfg.k_max);
  }
  __syncthreads();
  double* n = snodes + (size_t)slot * kSigmaStride;
  const int nb = (int)gridDim.x - 1 - kSGrid;     // node-table blocks
  if ((int)blockIdx.x >= nb) {
    const int i = (int)blockIdx.x - nb - 1;       // -1: sigma_8 block, else ln S point
    const double R = i < 0 ? 8.0 : exp(make_sgrid(cfg.k_min, cfg.k_max).ln_r(i));
    double lo, hi;
    sigma_limits(E, R, &lo, &hi);
    SigmaIntegrand f{&E, R};                      // sigma_norm = 1: amp * integral
    // (the ln S points only aim the search: 1e-5 and at most 2^12 panels are plenty)
    const double s2 = romberg1<4>(f, lo, hi, cfg.global_precision,
                                  i < 0 ? cfg.cosmo_precision : 1e-5,
                                  i < 0 || cfg.divmax < 12 ? cfg.divmax : 12, red);
    if (threadIdx.x == 0) {
      if (i < 0) n[kSigmaOffI8] = s2 / E.amp;
      else n[kSigmaOffLnS + i] = log(s2 / E.amp);
    }
    return;
  }
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= kSigmaCount) return;
  const double a = log(cfg.k_min), b = log(cfg.k_max);
  double x;
  if (idx < 2) {
    x = idx == 0 ? a : b;
  } else {
    const int m = idx - 1;
    const int lev = 32 - __builtin_clz((unsigned)m);
    const long j = m - (1 << (lev - 1));
    const double h = (b - a) / (double)(1L << (lev - 1));
    x = (a + 0.5 * h) + h * (double)j;
  }
  const double k = exp(x);
  const double T = eh_transfer(E, k);
  n[idx] = k;
  n[kSigmaCount + idx] = exp((3.0 + E.ns) * (x - E.ln_H0)) * T * T;
}

// Delta^2(k) W(kR)^2 / (amp sigma_norm^2) from the table (levels <= kSigmaLevel),
// direct evaluation beyond.
struct SigmaTabIntegrand {
  const Epoch* e;
  const double* node;      // this epoch's table
  double scale, inv_amp;
  __device__ __forceinline__ void operator()(double ln_k, double (&out)[1], int lev,
                                             long j) const {
    if (lev <= kSigmaLevel) {
      const int idx = node_index(lev, j);
      const double kR = scale * node[idx];
      double s, c;
      fast_sincos(kR, &s, &c);
      const double W = 3.0 * (s - kR * c) / (kR * kR * kR);
      out[0] = node[kSigmaCount + idx] * W * W;
    } else {
      SigmaIntegrand f{e, scale};
      out[0] = f(ln_k) * inv_amp;
    }
  }
};

// sigma^2(R) = int dlnk Delta^2 W^2 (cosmology.py:602-642) with the whole group on
// one Romberg integral; `rtol` is cosmo_precision for reference-exact values.
// UNROLL > 1 overlaps the table loads of several nodes (worth it where few blocks share a
// CU, as in k_epoch_init; with many resident blocks the extra registers cost more).
template <int NW, int UNROLL = 1>
__device__ __forceinline__ double sigma2_block(const Epoch& E, const double* snode, double R,
                                               const chomp_config& cfg, double rtol,
                                               double* red) {
  double lo, hi;
  sigma_limits(E, R, &lo, &hi);
  const double need_min = 1.0 / R / 10.0, need_max = 1.0 / R * 14.0662;
  const double amp2 = E.amp * E.sigma_norm * E.sigma_norm;
  if (need_min > E.k_min && need_max < E.k_max) {          // fixed range: table path
    SigmaTabIntegrand f{&E, snode, R, 1.0 / amp2};
    const RombergOut<1> r = romberg_group<NW, 1, SigmaTabIntegrand, UNROLL>(f, lo, hi, cfg.global_precision, rtol,
                                                 cfg.divmax, red);
    return amp2 * r.value[0];
  }
  SigmaIntegrand f{&E, R};
  return romberg1<NW>(f, lo, hi, cfg.global_precision, rtol, cfg.divmax, red);
}

// Not-a-knot spline build by parallel cyclic reduction.  Every thread of the block
// must call it (it contains barriers); the threads with 0 <= tid < tps of an
// `active` caller work on one system, rows strided by tps, so several systems can
// be built in lockstep by disjoint thread ranges.  w: 9 n doubles of LDS per system.
__device__ __forceinline__ void spline_build_pcr(const double* x, const double* y, int n,
                                                 double* c, double* w, int tid, int tps,
                                                 bool active) {
  double *a0 = w, *b0 = w + n, *c0 = w + 2 * n, *d0 = w + 3 * n;
  double *a1 = w + 4 * n, *b1 = w + 5 * n, *c1 = w + 6 * n, *d1 = w + 7 * n;
  double* sl = w + 8 * n;
  if (active)
    for (int i = tid; i < n; i += tps) spline_row(x, y, n, i, &a0[i], &b0[i], &c0[i], &d0[i]);
  __syncthreads();
  for (int s = 1; s < n; s *= 2) {
    if (active)
      for (int i = tid; i < n; i += tps) pcr_step(n, i, s, a0, b0, c0, d0, a1, b1, c1, d1);
    __syncthreads();
    double* t;
    t = a0; a0 = a1; a1 = t;
    t = b0; b0 = b1; b1 = t;
    t = c0; c0 = c1; c1 = t;
    t = d0; d0 = d1; d1 = t;
  }
  if (active)
    for (int i = tid; i < n; i += tps) sl[i] = d0[i] / b0[i];
  __syncthreads();
  if (active)
    for (int i = tid; i < n - 1; i += tps) spline_coef(x, y, sl, i, c);
  __syncthreads();
}

// nu(M) = (delta_c / sigma(R(M)))^2, cosmology.py:662-699.
template <int NW, int UNROLL = 1>
__device__ __forceinline__ double nu_of_mass_block(const Epoch& E, const double* snode,
                                                   double mass, const chomp_config& cfg,
                                                   double rtol, double* red) {
  const double s2 = sigma2_block<NW, UNROLL>(E, snode, scale_of_mass(E, mass), cfg, rtol, red);
  const double sq = E.delta_c / sqrt(s2);
  return sq * sq;
}

// ---------------------------------------------------------------------------
// Mass-limit search (mass_function.py:160-203).  The reference walks mass_min (from 1e9)
// and mass_max (from 1e16) in 5 % steps until nu(M) lands in 0.1(1 +- 0.05) /
// 50(1 +- 0.05), one sigma(R) Romberg per step (57 steps at z=0, 306 at z=1.5).  nu(M) is
// monotone, so the step the walk stops at is the first index j of the (bit-identical,
// host-tabulated) candidate sequence M_0 * 1.05^(-+j) whose nu passes a one-sided
// threshold test.
//
// k_epoch_init AIMS with the coarse ln S(R) table of the cosmology (nu to ~1e-4) and
// CERTIFIES with Romberg integrals at the four candidates around the crossing -- eight
// independent integrals per epoch, one per block -- so the decision rests on the same
// integrals the reference would have evaluated there.  Whatever cannot be certified
// (estimate too close to a threshold, candidate outside the table, disagreement) falls
// back to the bracketing secant search on exact integrals.
// ---------------------------------------------------------------------------
constexpr int kInitNW = 4;       // wavefronts per k_epoch_init block

struct SideThresholds {
  double thr_lo, thr_hi;
  const double *down, *up;      // candidate masses; [0] is the starting mass of both
};
__device__ __forceinline__ SideThresholds side_thresholds(int side, const double* cand) {
  // cand: [0] min-side divide, [1] min-side multiply, [2] max-side multiply,
  //       [3] max-side divide; index 0 of each is the starting mass.
  SideThresholds t;
  t.thr_lo = (side == 0 ? 0.1 : 50.0) * (1.0 - 0.05);
  t.thr_hi = (side == 0 ? 0.1 : 50.0) * (1.0 + 0.05);
  t.down = cand + (side == 0 ? 0 : 3) * kSearchJ;
  t.up = cand + (side == 0 ? 1 : 2) * kSearchJ;
  return t;
}

// nu(M) at a probe: a looser Romberg tolerance first; a probe that lands within
// kAmbiguous (in ln nu) of a band edge is redone at the reference's tolerance, so every
// comparison that decides the stopping step is either clear of the edge or exact.
template <int NW>
__device__ __forceinline__ double nu_probe(const Epoch& E, const double* snode, double m,
                                           const chomp_config& cfg, double thr_lo,
                                           double thr_hi, double* red) {
  // (a cosmo_precision looser than the probe tolerance is used as it is: the reference's
  // decision rests on exactly that integral)
  const double kAmbiguous = 2e-5;
  const double rtol_probe = cfg.cosmo_precision > 1e-6 ? cfg.cosmo_precision : 1e-6;
  double nu = nu_of_mass_block<NW, 4>(E, snode, m, cfg, rtol_probe, red);
  const double edge = fmin(fabs(log(nu / thr_lo)), fabs(log(nu / thr_hi)));
  if (edge < kAmbiguous && cfg.cosmo_precision < rtol_probe)
    nu = nu_of_mass_block<NW, 4>(E, snode, m, cfg, cfg.cosmo_precision, red);
  return nu;
}

// The bracketing secant search on exact integrals (whole block).  Returns the mass the
// reference's walk stops at; *n_eval counts the sigma integrals.  Seeds (from probes that
// did not certify the estimate): seed_dir != 0 fixes the walking direction; candidate
// seed_jl is known to FAIL the threshold test (nu_l: its nu, exact or estimated -- it only
// steers the secant); seed_jh >= 0 is known to PASS it with exact nu_h.
template <int NW>
__device__ double search_side_exact(const Epoch& E, const double* snode, int side,
                                    const chomp_config& cfg, const double* cand, double* red,
                                    int* n_eval, int seed_dir = 0, int seed_jl = 0,
                                    double nu_l = 0.0, int seed_jh = -1, double nu_h = 0.0) {
  const SideThresholds T = side_thresholds(side, cand);
  double mass = T.down[0];
  int dir = seed_dir;
  double nu0 = nu_l;
  if (seed_dir == 0) {
    nu0 = nu_probe<NW>(E, snode, T.down[0], cfg, T.thr_lo, T.thr_hi, red);
    ++*n_eval;
    if (T.thr_hi < nu0) dir = -1; else if (T.thr_lo > nu0) dir = +1;
  }
  if (dir != 0) {
    const double* tab = dir < 0 ? T.down : T.up;
    const double thr = dir < 0 ? T.thr_hi : T.thr_lo;
    int jl = seed_dir == 0 ? 0 : seed_jl, jh = seed_dir == 0 ? -1 : seed_jh;
    double tl = dir < 0 ? log(nu0 / thr) : log(thr / nu0);
    double th = jh < 0 ? 0.0 : (dir < 0 ? log(nu_h / thr) : log(thr / nu_h));
    double d = (side == 0 ? 0.25 : 0.7) * 0.04879;     // guess of d ln(nu) per step
    for (int it = 0; it < 4 * kSearchJ; ++it) {
      int jp;
      if (jh < 0) {
        double want = ceil(tl / d);
        if (!(want >= 1.0)) want = 1.0;
        if (want > 256.0) want = 256.0;
        jp = jl + (int)want;
        if (jp > kSearchJ - 1) jp = kSearchJ - 1;
      } else {
        if (jh == jl + 1) break;
        const double dd = (tl - th) / (double)(jh - jl);
        double want = ceil(tl / dd);
        if (!(want >= 1.0)) want = 1.0;
        if (want > (double)(jh - jl - 1)) want = (double)(jh - jl - 1);
        jp = jl + (int)want;
      }
      const double nu = nu_probe<NW>(E, snode, tab[jp], cfg, T.thr_lo, T.thr_hi, red);
      ++*n_eval;
      const bool pred = dir < 0 ? !(thr < nu) : !(thr > nu);
      const double tp = dir < 0 ? log(nu / thr) : log(thr / nu);
      if (pred) {
        jh = jp;
        th = tp;
      } else {
        if (jh < 0) {
          const double dn = (tl - tp) / (double)(jp - jl);
          d = dn > 1e-6 ? dn : 1e-6;
        }
        jl = jp;
        tl = tp;
        if (jh < 0 && jp == kSearchJ - 1) { jh = jp; break; }   // table exhausted
      }
    }
    if (jh < 0) jh = jl;
    mass = tab[jh];
  }
  return mass;
}

// ln S at ln R = x from the coarse table (6-point Lagrange inside one segment); NaN
// outside the table.
__device__ __forceinline__ double ln_s_estimate(const SGrid& G, const double* lns, double x) {
  if (!G.valid || !(x >= G.b0 && x <= G.b4)) return NAN;
  double lo, hi;
  int n, off;
  G.segment(x, &lo, &hi, &n, &off);
  const double u = (x - lo) / (hi - lo) * (double)(n - 1);
  int i0 = (int)floor(u) - 2;
  i0 = i0 < 0 ? 0 : (i0 > n - 6 ? n - 6 : i0);
  const double* f = lns + off + i0;
  const double t = u - (double)i0;               // position among nodes 0..5
  const double t0 = t, t1 = t - 1.0, t2 = t - 2.0, t3 = t - 3.0, t4 = t - 4.0, t5 = t - 5.0;
  return f[0] * (t1 * t2 * t3 * t4 * t5) * (-1.0 / 120.0) +
         f[1] * (t0 * t2 * t3 * t4 * t5) * (1.0 / 24.0) +
         f[2] * (t0 * t1 * t3 * t4 * t5) * (-1.0 / 12.0) +
         f[3] * (t0 * t1 * t2 * t4 * t5) * (1.0 / 12.0) +
         f[4] * (t0 * t1 * t2 * t3 * t5) * (-1.0 / 24.0) +
         f[5] * (t0 * t1 * t2 * t3 * t4) * (1.0 / 120.0);
}

// Where the estimate says the walk of `side` stops: dir (0: the starting mass already
// passes), and the first index j of the candidate table whose estimate passes.  ok = false
// when the estimate cannot be trusted to within one candidate.  The whole block calls it
// (every candidate is tried at once); sh: one int of LDS.
struct SidePlan {
  bool ok;
  int dir, j;
  double nu_start;              // estimate at the starting mass
  bool at_edge;                 // the starting mass sits on a band edge: dir is a guess, the
                                // probes are candidates 0..3 (j = 2) and 0 decides exactly
};
__device__ __forceinline__ SidePlan plan_side(const Epoch& E, const double* lns, int side,
                                              const double* cand, int* sh) {
  const SideThresholds T = side_thresholds(side, cand);
  const SGrid G = make_sgrid(E.k_min, E.k_max);
  const double margin = 1e-2;             // estimate error ~1e-4; candidates are >= 0.4 % apart
  SidePlan P{false, 0, 0, 0.0, false};
  // ln nu = ln_nu_c - ln S(ln R); the candidates are M_0 * 1.05^(-+j), so ln R moves by
  // ln(1.05) / 3 per step (to the estimate's accuracy)
  const double ln_nu_c = log(E.delta_c * E.delta_c / (E.amp * E.sigma_norm * E.sigma_norm));
  const double x0 = log(scale_of_mass(E, T.down[0]));
  const double step = 0.016263388 /* ln(1.05) / 3 */;
  const double nu0 = exp(ln_nu_c - ln_s_estimate(G, lns, x0));
  P.nu_start = nu0;
  if (!(nu0 == nu0)) return P;
  if (nu0 > T.thr_hi * (1.0 + margin)) P.dir = -1;
  else if (nu0 < T.thr_lo * (1.0 - margin)) P.dir = +1;
  else if (nu0 >= T.thr_lo * (1.0 + margin) && nu0 <= T.thr_hi * (1.0 - margin)) {
    P.ok = true;                          // inside the band with room to spare
    return P;
  } else {                                // on an edge: probe the start and its neighbours
    P.ok = true;
    P.at_edge = true;
    P.dir = nu0 > 0.5 * (T.thr_lo + T.thr_hi) ? -1 : +1;
    P.j = 2;
    return P;
  }
  const double ln_thr = log(P.dir < 0 ? T.thr_hi : T.thr_lo);
  __syncthreads();
  if (threadIdx.x == 0) *sh = kSearchJ;
  __syncthreads();
  // nu is monotone along the table, so the first passing index is the minimum over the
  // passing ones; a candidate outside the ln S table counts as passing, and is rejected
  // below if it turns out to be the first.
  int mine = kSearchJ;
  for (int j = 1 + (int)threadIdx.x; j < kSearchJ; j += blockDim.x) {
    const double ln_nu = ln_nu_c - ln_s_estimate(G, lns, x0 + (double)(P.dir * j) * step);
    const bool pass = !(ln_nu == ln_nu) || (P.dir < 0 ? !(ln_thr < ln_nu) : !(ln_thr > ln_nu));
    if (pass) { mine = j; break; }               // (ascending j: the first is the smallest)
  }
  if (mine < kSearchJ) atomicMin(sh, mine);
  __syncthreads();
  const int j = *sh;
  if (j >= kSearchJ) return P;                   // nothing passes: exact search
  const double lsj = ln_s_estimate(G, lns, x0 + (double)(P.dir * j) * step);
  if (!(lsj == lsj)) return P;                   // the walk leaves the ln S table
  P.j = j;
  P.ok = true;
  return P;
}

// grid (n_epoch, 2 * kProbes), block 64 * kInitNW.  blockIdx.y = kProbes * side + p
// certifies candidate j - 2 + p of side 0 (mass_min) / 1 (mass_max); role kProbes also
// does the comoving distance (or only that, with fixed mass limits).  The last block of
// an epoch to finish combines the results (count[e], reset by it).  epochs[e] holds the closed-form part of
// the record (k_sigma_nodes) on entry and the complete record on exit.
constexpr int kProbes = 4;
constexpr int kProbeStride = 24;   // doubles per epoch: nu[2][kProbes], chi, pad[3], plan[2][4]
__global__ __launch_bounds__(64 * kInitNW) void k_epoch_init(
    chomp_config cfg, Epoch* __restrict__ epochs, double* __restrict__ search,
    const double* __restrict__ cand, const double* __restrict__ snodes,
    double* __restrict__ probe, int* __restrict__ count) {
  __shared__ Epoch E;
  __shared__ double red[romberg_scratch<kInitNW, 1>()];
  __shared__ double lns[kSGrid];   // the cosmology's coarse ln S(R) table
  __shared__ int last, sh_j;
  const int e = blockIdx.x, role = blockIdx.y;
  const bool chi_role = role == kProbes;
  const int side = role / kProbes, p = role % kProbes;
  const bool fixed = cfg.mass_min > 0.0 && cfg.mass_max > 0.0;     // mass_function.py:163-170
  if (fixed && !chi_role) return;
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  __syncthreads();
  const double* snode = snodes + (size_t)E.cosmo_slot * kSigmaStride;
  copy_doubles(lns, snode + kSigmaOffLnS, kSGrid);
  if (threadIdx.x == 0) {
    // sigma_8 normalisation, cosmology.py:118-119: sigma_r(8)^2 = amp * I8 with the
    // cosmology-only integral I8 from k_sigma_nodes
    E.sigma_norm = E.sigma8 * E.growth / sqrt(E.amp * snode[kSigmaOffI8]);
  }
  __syncthreads();
  double* pr = probe + (size_t)e * kProbeStride;
  if (chi_role) {                  // comoving distance, cosmology.py:106-110
    EIntegrand f{E.om0, E.ol0, E.or0, E.H0};
    const double chi = romberg1<kInitNW>(f, 0.0, E.z, cfg.global_precision,
                                         cfg.cosmo_precision, cfg.divmax, red);
    __syncthreads();
    if (threadIdx.x == 0) { E.chi = chi; pr[2 * kProbes] = chi; }
    __syncthreads();
    if (fixed) {
      if (threadIdx.x == 0) {
        search[(e * 2 + 0) * 2 + 0] = log(cfg.mass_min);
        search[(e * 2 + 0) * 2 + 1] = 0.0;
        search[(e * 2 + 1) * 2 + 0] = log(cfg.mass_max);
        search[(e * 2 + 1) * 2 + 1] = 0.0;
      }
      __syncthreads();
      copy_doubles(reinterpret_cast<double*>(&epochs[e]), reinterpret_cast<const double*>(&E),
                   kEpochDoubles);
      return;
    }
  }
  {
    // ---- this block's probe: candidate j - 2 + p of its side
    const SidePlan plan = plan_side(E, lns, side, cand, &sh_j);
    const SideThresholds T = side_thresholds(side, cand);
    double nu_mine = NAN;
    if (plan.ok && plan.dir != 0) {
      const int c = plan.j - 2 + p;
      // (away from an edge candidate 0 fails by the margin of the estimate)
      if ((c > 0 || (c == 0 && plan.at_edge)) && c < kSearchJ) {
        const double* tab = plan.dir < 0 ? T.down : T.up;
        nu_mine = nu_probe<kInitNW>(E, snode, tab[c], cfg, T.thr_lo, T.thr_hi, red);
      }
    }
    if (threadIdx.x == 0) {
      pr[role] = nu_mine;
      if (p == 0) {
        double* pl = pr + 2 * kProbes + 4 + 4 * side;
        pl[0] = plan.ok ? (plan.at_edge ? 2.0 : 1.0) : 0.0; pl[1] = (double)plan.dir;
        pl[2] = (double)plan.j; pl[3] = plan.nu_start;
      }
    }
  }
  if (threadIdx.x == 0) {
    __threadfence();               // results visible before the arrival is counted
    last = atomicAdd(&count[e], 1) == 2 * kProbes - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!last) return;
  // ---- last block of the epoch: certify both sides, exact search where that fails
  __threadfence();
  auto peek = [](const double* q) {
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  for (int sd = 0; sd < 2; ++sd) {
    __syncthreads();
    const double* pl = pr + 2 * kProbes + 4 + 4 * sd;
    const double mode = peek(pl);
    const bool ok = mode != 0.0, at_edge = mode == 2.0;
    int dir = (int)peek(pl + 1);
    const int j = (int)peek(pl + 2);
    double nu_start = peek(pl + 3);
    const SideThresholds S = side_thresholds(sd, cand);
    double mass = S.down[0];
    int n_eval = 0;
    bool certified = ok;
    int seed_dir = 0, seed_jl = 0, seed_jh = -1;
    double nu_l = 0.0, nu_h = 0.0;
    bool walk = ok && dir != 0;
    if (at_edge) {                 // the exact nu of the starting mass decides the direction
      nu_start = peek(pr + kProbes * sd);
      const int dir_exact = S.thr_hi < nu_start ? -1 : (S.thr_lo > nu_start ? +1 : 0);
      n_eval = 1;
      if (!(nu_start == nu_start)) { certified = false; walk = false; n_eval = 0; }
      else if (dir_exact == 0) walk = false;                     // stays at the start: done
      else if (dir_exact != dir) {                               // guessed the other way
        certified = false; walk = false;
        seed_dir = dir_exact; seed_jl = 0; nu_l = nu_start;
      }
    }
    if (walk) {
      const double* tab = dir < 0 ? S.down : S.up;
      const double thr = dir < 0 ? S.thr_hi : S.thr_lo;
      // status of candidates j - 2 .. j + 1: 0 fails, 1 passes, -1 unknown
      int st[kProbes];
      double nu[kProbes];
#pragma unroll
      for (int q = 0; q < kProbes; ++q) {
        const int c = j - 2 + q;
        nu[q] = peek(pr + kProbes * sd + q);
        if (c <= 0) st[q] = 0;     // (at an edge: candidate 0 fails exactly, see above)
        else if (!(nu[q] == nu[q])) st[q] = -1;
        else { st[q] = (dir < 0 ? !(thr < nu[q]) : !(thr > nu[q])) ? 1 : 0; ++n_eval; }
      }
      certified = false;
      int first_pass = -1, before = -1;          // status of the candidate before it
      double nu_first = 0.0;
#pragma unroll
      for (int q = kProbes - 1; q >= 0; --q)
        if (st[q] == 1) { first_pass = q; nu_first = nu[q]; before = q > 0 ? st[q - 1] : -1; }
      if (first_pass > 0 && before == 0) {                       // fails at c - 1, passes at c
        certified = true;
        mass = tab[j - 2 + first_pass];
      } else if (first_pass == 0 && j - 2 == 1) {                // passes at 1, 0 fails
        certified = true;
        mass = tab[1];
      }
      if (!certified) {            // the estimate was off by more than the probes cover:
        seed_dir = dir;            // the exact search starts from what they established
        seed_jl = 0; nu_l = nu_start;
#pragma unroll
        for (int q = 0; q < kProbes; ++q)
          if (st[q] == 0 && j - 2 + q > 0) { seed_jl = j - 2 + q; nu_l = nu[q]; }
        if (first_pass >= 0 && j - 2 + first_pass > seed_jl) {
          seed_jh = j - 2 + first_pass; nu_h = nu_first;
        }
      }
    }
    if (!certified)                // block-uniform: every thread read the same values
      mass = search_side_exact<kInitNW>(E, snode, sd, cfg, cand, red, &n_eval, seed_dir, seed_jl,
                                        nu_l, seed_jh, nu_h);
    if (threadIdx.x == 0) {
      search[(e * 2 + sd) * 2 + 0] = log(mass);
      search[(e * 2 + sd) * 2 + 1] = (double)n_eval;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    E.chi = peek(pr + 2 * kProbes);
    count[e] = 0;
  }
  __syncthreads();
  copy_doubles(reinterpret_cast<double*>(&epochs[e]), reinterpret_cast<const double*>(&E),
               kEpochDoubles);
}

// ---------------------------------------------------------------------------
// k_nu_table: grid (NM, n_epoch), block 256: nu_i = nu_m(exp(ln_mass_i)).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nu_table(chomp_config cfg, TabLayout L,
                                                  const Epoch* __restrict__ epochs,
                                                  const double* __restrict__ search,
                                                  const double* __restrict__ snodes,
                                                  double* __restrict__ tab) {
  __shared__ Epoch E;
  __shared__ double red[romberg_scratch<4, 2>()];
  const int i = blockIdx.x, e = blockIdx.y;
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  __syncthreads();
  const double ln_lo = search[(e * 2 + 0) * 2], ln_hi = search[(e * 2 + 1) * 2];
  const double lnm = linspace_at(ln_lo, ln_hi, L.NM, i);
  const double nu = nu_of_mass_block<4>(E, snodes + (size_t)E.cosmo_slot * kSigmaStride, exp(lnm), cfg,
                                        cfg.cosmo_precision, red);
  if (threadIdx.x == 0) {
    double* t = tab + (size_t)e * L.stride;
    t[L.off_ln_mass + i] = lnm;
    t[L.off_nu + i] = nu;
  }
}

// Tinker10 parameter splines (mass_function.py:450-470): x[9] then 5 x 32 pp
// coefficients (alpha, beta, gamma, phi, eta), built on the host at context
// creation with the same spline_build.
struct TinkerTab {
  double x[9];
  double c[5][32];
};

struct FnuLn {          // f(nu) d nu = f(e^t) e^t dt
  const Epoch* e;
  __device__ __forceinline__ double operator()(double t) const {
    const double nu = exp(t);
    return f_nu(*e, nu) * nu;
  }
};
struct FnuLin {         // mass_function.py:227-231, as the reference integrates it
  const Epoch* e;
  __device__ __forceinline__ double operator()(double nu) const { return f_nu(*e, nu); }
};
struct FnuBiasLin {     // mass_function.py:235-240
  const Epoch* e;
  __device__ __forceinline__ double operator()(double nu) const {
    return f_nu(*e, nu) * bias_nu(*e, nu);
  }
};
constexpr int kNormLiteralDivmax = 18;
struct FnuBiasLn {
  const Epoch* e;
  __device__ __forceinline__ double operator()(double t) const {
    const double nu = exp(t);
    return f_nu(*e, nu) * bias_nu(*e, nu) * nu;
  }
};

// ---------------------------------------------------------------------------
// k_mass_setup: grid n_epoch, block 256.  Dynamic LDS: see carve-up below.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mass_setup(
    chomp_config cfg, TabLayout L, Epoch* __restrict__ epochs,
    const double* __restrict__ search, double* __restrict__ tab,
    const chomp_halo_par* __restrict__ par, int mf_kind,
    const TinkerTab* __restrict__ tinker, const double* __restrict__ gl16) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  const int NM = L.NM;
  double* x_lnm = sm;                    // [NM]
  double* y_nu = x_lnm + NM;             // [NM]
  double* c_nu = y_nu + NM;              // [4(NM-1)]
  double* c_lnm = c_nu + 4 * (NM - 1);   // [4(NM-1)]
  double* work = c_lnm + 4 * (NM - 1);   // [18 NM]
  double* gl = work + 18 * NM;           // [32]
  double* red = gl + 32;                 // [romberg_scratch<4, 1>()]
  const int e = blockIdx.x;
  double* t = tab + (size_t)e * L.stride;
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  copy_doubles(x_lnm, t + L.off_ln_mass, NM);
  copy_doubles(y_nu, t + L.off_nu, NM);
  copy_doubles(gl, gl16, 32);
  __syncthreads();
  {   // nu(ln M) on threads 0..127, ln M(nu) on threads 128..255, in lockstep
    const int sys = threadIdx.x >> 7, tid = threadIdx.x & 127;
    spline_build_pcr(sys == 0 ? x_lnm : y_nu, sys == 0 ? y_nu : x_lnm, NM,
                     sys == 0 ? c_nu : c_lnm, work + sys * 9 * NM, tid, 128, true);
  }
  if (threadIdx.x == 0) {
    const chomp_halo_par hp = par[e];
    E.ln_mass_min = search[(e * 2 + 0) * 2];
    E.ln_mass_max = search[(e * 2 + 1) * 2];
    E.n_search = (int)(search[(e * 2 + 0) * 2 + 1] + search[(e * 2 + 1) * 2 + 1]);
    E.nu_min = 1.001 * y_nu[0];                       // mass_function.py:212-213
    E.nu_max = 0.999 * y_nu[NM - 1];
    E.m_star = exp(spline_eval(y_nu, c_lnm, NM, 1.0));   // :223
    E.stq = hp.stq;
    E.st_a = hp.st_little_a;
    E.mf_delta_v = (hp.delta_v == -1.0) ? E.delta_v : hp.delta_v;
    E.mf_kind = mf_kind;
    E.f_norm = 1.0;
    E.bias_norm = 1.0;
    E.ln_st_a = log(hp.st_little_a);
    E.ln_t_beta = 0.0;
    if (mf_kind == CHOMP_MF_TINKER) {                 // mass_function.py:547-564
      const double ld = log(E.mf_delta_v);
      const double opz = 1.0 + E.z;
      E.t_alpha = spline_eval(tinker->x, tinker->c[0], 9, ld);
      E.t_beta = spline_eval(tinker->x, tinker->c[1], 9, ld) * pow(opz, 0.20);
      E.t_gamma = spline_eval(tinker->x, tinker->c[2], 9, ld) * pow(opz, -0.01);
      E.t_phi = spline_eval(tinker->x, tinker->c[3], 9, ld) * pow(opz, -0.08);
      E.t_eta = spline_eval(tinker->x, tinker->c[4], 9, ld) * pow(opz, 0.27);
      E.ln_t_beta = log(E.t_beta);
      tinker_bias_constants(E);
    }
  }
  __syncthreads();
  // Normalisations (mass_function.py:225-241; Tinker: bias only, :532-545).  The
  // reference integrates in linear nu with Romberg to rtol 1.48e-8 (8193 nodes);
  // the integrand is analytic, so 8 x 16 Gauss-Legendre nodes in ln nu give the
  // same number to ~4e-11.
  // With a shallow divmax the reference's Romberg cannot converge and returns its last
  // row; that (not the true integral) is then the reference's number, so the literal
  // Romberg in linear nu is run instead (at most 2^17 cheap nodes).
  int flip = 0;
  const bool literal = cfg.divmax < kNormLiteralDivmax;
  const double a = log(E.nu_min), b = log(E.nu_max);
  if (mf_kind == CHOMP_MF_ST) {
    double norm;
    if (literal) {
      FnuLin f{&E};
      norm = romberg1<4>(f, E.nu_min, E.nu_max, cfg.global_precision, cfg.mass_precision,
                         cfg.divmax, red);
    } else {
      FnuLn f{&E};
      norm = gauss_panels<4>(f, a, b, 8, gl, red, flip);
    }
    __syncthreads();
    if (threadIdx.x == 0) E.f_norm = 1.0 / norm;
    __syncthreads();
  }
  {
    double norm;
    if (literal) {
      FnuBiasLin f{&E};
      norm = romberg1<4>(f, E.nu_min, E.nu_max, cfg.global_precision, cfg.mass_precision,
                         cfg.divmax, red);
    } else {
      FnuBiasLn f{&E};
      norm = gauss_panels<4>(f, a, b, 8, gl, red, flip);
    }
    __syncthreads();
    if (threadIdx.x == 0) E.bias_norm = 1.0 / norm;
    __syncthreads();
  }
  copy_doubles(reinterpret_cast<double*>(&epochs[e]), reinterpret_cast<const double*>(&E),
               kEpochDoubles);
  copy_doubles(t + L.off_nu_pp, c_nu, 4 * (NM - 1));
  copy_doubles(t + L.off_lnm_pp, c_lnm, 4 * (NM - 1));
}

// ---------------------------------------------------------------------------
// Halo integrands over ln nu (halo.py:702-707, 922-927, 964-969, 989-994,
// 1032-1041, 1078-1086).  The reference multiplies each integrand by a constant
// `norm` and divides it out again; it cancels in the relative stopping test and is
// omitted.
// ---------------------------------------------------------------------------
struct HaloCtx {
  const Epoch* e;
  const SiCiTab* sici;
  const double* nu_knots;    // [NM]  knots of ln M(nu)
  const double* lnm_pp;      // [4(NM-1)]
  int NM;
  double ln_k;
  bool exclusion;            // HaloExclusion: the 2-halo integrands carry the mass window
  __device__ __forceinline__ double window(double lnm) const {
    if (!exclusion) return 1.0;
    const double ln_rv = (e->ln_rv_const + lnm) * (1.0 / 3.0);
    return exclusion_window(*sici, 2.0 * exp(ln_k + ln_rv));
  }
};

struct IntegrandMM {       // out[0] = h_m, out[1] = pp_mm (x rho_bar)
  HaloCtx c;
  __device__ __forceinline__ void operator()(double ln_nu, double (&out)[2]) const {
    const double nu = exp(ln_nu);
    const double lnm = spline_eval(c.nu_knots, c.lnm_pp, c.NM, nu);
    const double y = y_nfw(*c.e, *c.sici, c.ln_k, lnm);
    double nf, b;
    mf_node(*c.e, nu, ln_nu, true, &nf, &b);
    out[0] = nf * b * y * c.window(lnm);
    out[1] = nf * exp(lnm) * y * y;
  }
};

struct IntegrandGM {       // out[0] = h_g, out[1] = pp_gm
  HaloCtx c;
  bool want_hg;            // false: h_g has converged, skip its bias factor
  __device__ __forceinline__ void operator()(double ln_nu, double (&out)[2]) const {
    const double nu = exp(ln_nu);
    const double lnm = spline_eval(c.nu_knots, c.lnm_pp, c.NM, nu);
    const double mass = exp(lnm);
    const double y = y_nfw(*c.e, *c.sici, c.ln_k, lnm);
    double nf, b = 0.0, n1, n2;
    mf_node(*c.e, nu, ln_nu, want_hg, &nf, &b);
    zheng_node(*c.e, mass, lnm, &n1, &n2);
    out[0] = nf * b * y * n1 / mass * (want_hg ? c.window(lnm) : 1.0);
    out[1] = (n1 < 1.0) ? nf * n1 * y : nf * n1 * y * y;
  }
};

struct IntegrandGG {       // out[0] = pp_gg
  HaloCtx c;
  __device__ __forceinline__ void operator()(double ln_nu, double (&out)[1]) const {
    const double nu = exp(ln_nu);
    const double lnm = spline_eval(c.nu_knots, c.lnm_pp, c.NM, nu);
    const double mass = exp(lnm);
    const double y = y_nfw(*c.e, *c.sici, c.ln_k, lnm);
    double nf, b, n1, n2;
    mf_node(*c.e, nu, ln_nu, false, &nf, &b);
    zheng_node(*c.e, mass, lnm, &n1, &n2);
    out[0] = (n2 < 1.0) ? nf * n2 * y / mass : nf * n2 * y * y / mass;
  }
};

struct IntegrandNbar {     // halo.py:702-707
  HaloCtx c;
  __device__ __forceinline__ double operator()(double ln_nu) const {
    const double nu = exp(ln_nu);
    const double lnm = spline_eval(c.nu_knots, c.lnm_pp, c.NM, nu);
    const double mass = exp(lnm);
    double nf, b, n1, n2;
    mf_node(*c.e, nu, ln_nu, false, &nf, &b);
    zheng_node(*c.e, mass, lnm, &n1, &n2);
    return nf * n1 / mass;
  }
};

// HOD-derived constants (hod.py:172-186) are computed on the host (erfinv) and
// passed in; the lower limits of the HOD integrals follow halo.py:935-939,
// 1002-1006.
struct HodDev {
  double log_M_min, sigma, log_M_0, log_M_1p, alpha;
  double first_zero, second_zero, safe_norm;
};

// Halo-profile and HOD constants of one epoch (Halo.__init__, halo.py:71-88; the
// lower limits of the HOD integrals, halo.py:935-939, 1002-1006).  nu_pp: pp
// coefficients of nu(ln M) on the uniform ln M grid starting at lnm0.
__device__ __forceinline__ void apply_halo_hod(Epoch& E, const chomp_halo_par& hp,
                                               const HodDev& h, const double* nu_pp,
                                               double lnm0, int NM) {
  halo_constants(E, hp.c0, hp.beta, hp.delta_v);
  E.hod_log_M_min = h.log_M_min; E.hod_sigma = h.sigma; E.hod_log_M_0 = h.log_M_0;
  E.hod_log_M_1p = h.log_M_1p; E.hod_alpha = h.alpha;
  E.hod_first_zero = h.first_zero; E.hod_second_zero = h.second_zero;
  E.hod_safe_norm = h.safe_norm;
  E.hod_M0 = pow(10.0, h.log_M_0);
  E.hod_M1p = pow(10.0, h.log_M_1p);
  const double dlnm = (E.ln_mass_max - E.ln_mass_min) / (double)(NM - 1);
  double nu1 = E.nu_min, nu2 = E.nu_min;
  if (h.first_zero > -1.0 && h.first_zero > exp(E.ln_mass_min))
    nu1 = spline_eval_uniform(lnm0, dlnm, nu_pp, NM, log(h.first_zero));
  if (h.second_zero > -1.0 && h.second_zero > exp(E.ln_mass_min))
    nu2 = spline_eval_uniform(lnm0, dlnm, nu_pp, NM, log(h.second_zero));
  E.ln_nu_lo_first = log(nu1);
  E.ln_nu_lo_second = log(nu2);
}

// Lower limit of group g's integrals: 0: nu_min; 1: nu(first_moment_zero); 2:
// nu(second_moment_zero) (halo.py:909-911, 935-939, 1002-1006).
__device__ __forceinline__ double group_lower(const Epoch& E, int group) {
  return group == 0 ? log(E.nu_min) : (group == 1 ? E.ln_nu_lo_first : E.ln_nu_lo_second);
}

// Stage what every halo-integral block needs into LDS and derive the epoch's
// halo/HOD constants there (all threads call; ends with a barrier).
struct HaloLds {
  double *nu_knots, *lnm_pp, *nu_pp, *rest;
  __device__ __forceinline__ void stage(const TabLayout& L, Epoch& E, SiCiTab& S,
                                        const Epoch* epochs, int e, const double* t,
                                        const chomp_halo_par* profile, const HodDev* hod,
                                        const SiCiTab* sici_g, double* sm) {
    const int NM = L.NM;
    nu_knots = sm;
    lnm_pp = nu_knots + NM;
    nu_pp = lnm_pp + 4 * (NM - 1);
    rest = nu_pp + 4 * (NM - 1);
    copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
                 kEpochDoubles);
    copy_doubles(reinterpret_cast<double*>(&S), reinterpret_cast<const double*>(sici_g),
                 (int)(sizeof(SiCiTab) / sizeof(double)));
    copy_doubles(nu_knots, t + L.off_nu, NM);
    copy_doubles(lnm_pp, t + L.off_lnm_pp, 4 * (NM - 1));
    copy_doubles(nu_pp, t + L.off_nu_pp, 4 * (NM - 1));
    __syncthreads();
    if (threadIdx.x == 0) apply_halo_hod(E, profile[e], hod[e], nu_pp, t[L.off_ln_mass], NM);
    __syncthreads();
  }
};

// ---------------------------------------------------------------------------
// k_halo_nodes: grid (ceil(kNodeCount / 256) + 1, n_epoch, n_groups), block 256: one
// node of the (epoch, group) table per thread; the extra x-block of z == 0 does the
// epoch's n_bar integral (halo.py:674-700).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halo_nodes(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs,
    const double* __restrict__ tab, const chomp_halo_par* __restrict__ profile,
    const HodDev* __restrict__ hod, const SiCiTab* __restrict__ sici_g,
    double* __restrict__ nodes, double* __restrict__ tab_out, int g0, int g1, int g2) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ SiCiTab S;
  const int e = blockIdx.y;
  const int group = blockIdx.z == 0 ? g0 : (blockIdx.z == 1 ? g1 : g2);
  const bool nbar_block = blockIdx.x == gridDim.x - 1;
  if (nbar_block && blockIdx.z != 0) return;
  if (!nbar_block && (group < 0 || group > 2)) return;
  HaloLds H;
  H.stage(L, E, S, epochs, e, tab + (size_t)e * L.stride, profile, hod, sici_g, sm);
  if (nbar_block) {
    HaloCtx c{&E, &S, H.nu_knots, H.lnm_pp, L.NM, 0.0, false};
    IntegrandNbar f{c};
    const double v = romberg1<4>(f, E.ln_nu_lo_first, log(E.nu_max), cfg.global_precision,
                                 cfg.halo_precision, cfg.divmax, H.rest);
    if (threadIdx.x == 0) tab_out[(size_t)e * L.stride + L.off_misc] = v;
    return;
  }
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= kNodeCount) return;
  const double a = group_lower(E, group), b = log(E.nu_max);
  if (idx == 0) {
    double* hdr = nodes + ((size_t)e * 3 + group) * kNodeStride + kNodeFields * kNodeCount;
    hdr[0] = a;
    hdr[1] = b;
  }
  double x;
  if (idx < 2) {
    x = idx == 0 ? a : b;
  } else {
    const int m = idx - 1;
    const int lev = 32 - __builtin_clz((unsigned)m);      // floor(log2 m) + 1
    const long j = m - (1 << (lev - 1));
    const double h = (b - a) / (double)(1L << (lev - 1));
    x = (a + 0.5 * h) + h * (double)j;
  }
  const double nu = exp(x);
  const double lnm = spline_eval(H.nu_knots, H.lnm_pp, L.NM, nu);
  const double mass = exp(lnm);
  double nf, bias = 0.0;
  mf_node(E, nu, x, group != 2, &nf, &bias);
  const double ln_c = E.ln_c_const + E.beta * lnm;
  const double ln_rv = (E.ln_rv_const + lnm) * (1.0 / 3.0);
  const double con = exp(ln_c);
  const double cp = 1.0 + con;
  const double ln_cp = log(cp);
  double wA, wB, flag = 0.0;
  if (group == 0) {
    wA = nf * bias;
    wB = nf * mass;
  } else {
    double n1, n2;
    zheng_node(E, mass, lnm, &n1, &n2);
    if (group == 1) {
      wA = nf * bias * n1 / mass;
      wB = nf * n1;
      flag = n1 < 1.0 ? 1.0 : 0.0;
    } else {
      wA = 0.0;
      wB = nf * n2 / mass;
      flag = n2 < 1.0 ? 1.0 : 0.0;
    }
  }
  double* n = nodes + ((size_t)e * 3 + group) * kNodeStride + idx;
  n[0 * kNodeCount] = wA;
  n[1 * kNodeCount] = wB;
  n[2 * kNodeCount] = ln_rv - ln_c;
  n[3 * kNodeCount] = con;
  n[4 * kNodeCount] = ln_cp;
  n[5 * kNodeCount] = 1.0 / (ln_cp - con / cp);
  n[6 * kNodeCount] = flag;
}

// out[0] = wA y, out[1] = wB (flag ? y : y^2) from the node table; group 2 uses
// only out[1].  Valid for levels <= kNodeLevel.
struct NodeIntegrand {
  const SiCiTab* sici;
  const double* node;     // this (epoch, group)'s table
  double ln_k;
  bool exclusion;         // HaloExclusion (halo.py:1208-1233): window on the 2-halo term
  __device__ __forceinline__ void operator()(double, double (&out)[2], int lev, long j) const {
    const double* n = node + node_index(lev, j);
    double z;             // k r_s; k * 2 r_v = 2 c z
    const double y = y_nfw_core(*sici, ln_k, n[2 * kNodeCount], n[3 * kNodeCount],
                                n[4 * kNodeCount], n[5 * kNodeCount], &z);
    out[0] = n[0] * y;
    if (exclusion) out[0] *= exclusion_window(*sici, 2.0 * n[3 * kNodeCount] * z);
    out[1] = n[kNodeCount] * (n[6 * kNodeCount] != 0.0 ? y : y * y);
  }
};

__device__ __forceinline__ int group_fa(int group) { return group == 0 ? F_HM : F_HG; }
__device__ __forceinline__ int group_fb(int group) {
  return group == 0 ? F_PPMM : (group == 1 ? F_PPGM : F_PPGG);
}
constexpr double kPendingLevel = -1.0;   // levels-table marker: needs the deep pass
constexpr unsigned kMaskExclusion = 1u << 8;   // bit of the kernels' family mask: HaloExclusion

// ---------------------------------------------------------------------------
// k_halo_knots: grid (NK, n_epoch, n_groups), block 256 (4 wavefronts per integral
// pair): knot ln k_i of group groups[blockIdx.z] (0: h_m + pp_mm, 1: h_g + pp_gm,
// 2: pp_gg), Romberg levels <= kNodeLevel from the node table; integrals not
// converged by then are marked for k_halo_knots_deep.  A block only stages the Si/Ci
// tables: everything else it needs is in the (epoch, group) node table.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halo_knots(
    chomp_config cfg, TabLayout L, double* __restrict__ tab,
    const SiCiTab* __restrict__ sici_g, const double* __restrict__ nodes, int g0, int g1,
    int g2, unsigned mask) {
  __shared__ SiCiTab S;
  __shared__ double red[kKnotScratch];
  const int NK = L.NK;
  const int ik = blockIdx.x, e = blockIdx.y;
  const int group = blockIdx.z == 0 ? g0 : (blockIdx.z == 1 ? g1 : g2);
  if (group < 0 || group > 2) return;
  copy_doubles(reinterpret_cast<double*>(&S), reinterpret_cast<const double*>(sici_g),
               (int)(sizeof(SiCiTab) / sizeof(double)));
  const double* node = nodes + ((size_t)e * 3 + group) * kNodeStride;
  const double a = node[kNodeFields * kNodeCount], b = node[kNodeFields * kNodeCount + 1];
  __syncthreads();
  double* t = tab + (size_t)e * L.stride;
  const double ln_k = linspace_at(log(cfg.k_min), log(cfg.k_max), NK, ik);   // halo.py:52-54
  NodeIntegrand f{&S, node, ln_k, (mask & kMaskExclusion) != 0};
  const int dmax = cfg.divmax < kNodeLevel ? cfg.divmax : kNodeLevel;
  const RombergOut<2> r = romberg_group<4, 2>(f, a, b, cfg.global_precision,
                                              cfg.halo_precision, dmax, red);
  if (threadIdx.x == 0) {
    double* lev = t + L.off_levels;
    const int fa = group_fa(group), fb = group_fb(group);
    const bool more = cfg.divmax > kNodeLevel;
    if (group != 2 && (mask & (1u << fa))) {
      t[L.off_knot[fa] + ik] = r.value[0];
      lev[fa * NK + ik] = (!r.converged[0] && more) ? kPendingLevel : (double)r.level[0];
    }
    if (mask & (1u << fb)) {
      t[L.off_knot[fb] + ik] = r.value[1];
      lev[fb * NK + ik] = (!r.converged[1] && more) ? kPendingLevel : (double)r.level[1];
    }
  }
}

// ---------------------------------------------------------------------------
// k_halo_knots_deep: same grid as k_halo_knots (without the n_bar column).  A block
// whose knot is not marked pending exits at once; otherwise it redoes the Romberg
// integral by direct evaluation up to divmax (the discontinuous HOD integrands run
// to 2^18..2^20 nodes, halo.py:1038-1041, 1084-1086) and stores the pending
// families.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halo_knots_deep(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs,
    double* __restrict__ tab, const chomp_halo_par* __restrict__ profile,
    const HodDev* __restrict__ hod, const SiCiTab* __restrict__ sici_g, int g0, int g1,
    int g2, unsigned mask) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ SiCiTab S;
  const int NK = L.NK;
  const int ik = blockIdx.x, e = blockIdx.y;
  const int group = blockIdx.z == 0 ? g0 : (blockIdx.z == 1 ? g1 : g2);
  if (group < 0 || group > 2) return;
  double* t = tab + (size_t)e * L.stride;
  double* lev = t + L.off_levels;
  const int fa = group_fa(group), fb = group_fb(group);
  const bool pa = group != 2 && (mask & (1u << fa)) && lev[fa * NK + ik] == kPendingLevel;
  const bool pb = (mask & (1u << fb)) && lev[fb * NK + ik] == kPendingLevel;
  if (!pa && !pb) return;
  HaloLds H;
  H.stage(L, E, S, epochs, e, t, profile, hod, sici_g, sm);
  double* red = H.rest;
  const double ln_nu_max = log(E.nu_max);
  HaloCtx c{&E, &S, H.nu_knots, H.lnm_pp, L.NM,
            linspace_at(log(cfg.k_min), log(cfg.k_max), NK, ik), (mask & kMaskExclusion) != 0};
  double va = 0.0, vb = 0.0;
  int la = 0, lb = 0;
  if (group == 0) {
    IntegrandMM f{c};
    const RombergOut<2> r = romberg_group<4, 2>(f, group_lower(E, 0), ln_nu_max,
                                                cfg.global_precision, cfg.halo_precision,
                                                cfg.divmax, red);
    va = r.value[0]; vb = r.value[1]; la = r.level[0]; lb = r.level[1];
  } else if (group == 1) {
    IntegrandGM f{c, pa};
    const RombergOut<2> r = romberg_group<4, 2>(f, group_lower(E, 1), ln_nu_max,
                                                cfg.global_precision, cfg.halo_precision,
                                                cfg.divmax, red);
    va = r.value[0]; vb = r.value[1]; la = r.level[0]; lb = r.level[1];
  } else {
    IntegrandGG f{c};
    const RombergOut<1> r = romberg_group<4, 1>(f, group_lower(E, 2), ln_nu_max,
                                                cfg.global_precision, cfg.halo_precision,
                                                cfg.divmax, red);
    vb = r.value[0]; lb = r.level[0];
  }
  if (threadIdx.x == 0) {
    if (pa) { t[L.off_knot[fa] + ik] = va; lev[fa * NK + ik] = (double)la; }
    if (pb) { t[L.off_knot[fb] + ik] = vb; lev[fb * NK + ik] = (double)lb; }
  }
}

// ---------------------------------------------------------------------------
// k_halo_finalize: grid n_epoch, block 384 (6 wavefronts).  Wavefront f < 5
// normalises family f and builds its not-a-knot spline over ln k (the five builds
// run in lockstep, parallel cyclic reduction); lane 0 of wavefront 5 writes the
// epoch's halo/HOD constants and n_bar back.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(384) void k_halo_finalize(
    chomp_config cfg, TabLayout L, Epoch* __restrict__ epochs, double* __restrict__ tab,
    const chomp_halo_par* __restrict__ profile, const HodDev* __restrict__ hod,
    unsigned fam_mask) {
  extern __shared__ __align__(16) double sm[];
  const int NK = L.NK;
  double* xk = sm;                      // [NK]
  double* yk = xk + NK;                 // [5][NK]
  double* work = yk + 5 * NK;           // [5][9 NK]
  const int e = blockIdx.x;
  double* t = tab + (size_t)e * L.stride;
  const int f = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double nbr = t[L.off_misc];                    // n_bar / rho_bar
  const double rho_bar = epochs[e].rho_bar;            // unchanged by the write-back below
  for (int i = threadIdx.x; i < NK; i += blockDim.x)
    xk[i] = linspace_at(log(cfg.k_min), log(cfg.k_max), NK, i);
  const bool active = f < 5 && ((fam_mask >> f) & 1u);
  if (active) {
    const double n_bar = nbr * rho_bar;
    double scale = 1.0;
    if (f == F_PPMM) scale = 1.0 / rho_bar;                       // halo.py:983
    else if (f == F_HG) scale = 1.0 / nbr;                        // :959
    else if (f == F_PPGM) scale = 1.0 / n_bar;                    // :1072
    else if (f == F_PPGG) scale = rho_bar / (n_bar * n_bar);      // :1026
    for (int i = lane; i < NK; i += 64) {
      const double v = t[L.off_knot[f] + i] * scale;
      yk[f * NK + i] = v;
      t[L.off_knot[f] + i] = v;
    }
  }
  if (f == 5 && lane == 1) {
    // Stage-E record: amplitude of Delta^2 and "same cosmology as the previous epoch"
    const Epoch& E = epochs[e];
    t[L.off_misc + 1] = E.amp * E.sigma_norm * E.sigma_norm;
    t[L.off_misc + 2] = (e > 0 && same_cosmology(E, epochs[e - 1])) ? 1.0 : 0.0;
  }
  if (f == 5 && lane == 0) {
    Epoch E = epochs[e];
    apply_halo_hod(E, profile[e], hod[e], t + L.off_nu_pp, t[L.off_ln_mass], L.NM);
    E.n_bar_over_rho_bar = nbr;                        // halo.py:692-700
    E.n_bar = nbr * E.rho_bar;
    epochs[e] = E;
  }
  __syncthreads();
  const int fs = f < 5 ? f : 0;
  spline_build_pcr(xk, yk + fs * NK, NK, t + L.off_kpp[fs], work + fs * 9 * NK, lane, 64,
                   active);
}

// ---------------------------------------------------------------------------
// Stage E.  grid (ceil(nk / (256*KPT)), n_epoch), block 256.  Each block stages
// its epoch's scalars and the pp-coefficients of the (up to 3) knot splines it
// needs in LDS, then streams k -> P with coalesced 8-byte accesses.
// ---------------------------------------------------------------------------
struct PowerFam { int fa, fb, fp; };
__device__ __forceinline__ PowerFam power_families(int w) {
  if (w == CHOMP_P_GM) return PowerFam{F_HG, F_HM, F_PPGM};
  if (w == CHOMP_P_GG) return PowerFam{F_HG, F_HG, F_PPGG};
  return PowerFam{F_HM, F_HM, F_PPMM};
}

// Above k_max with Halo(extrapolate=True) (halo.py:300-312, 341-367, 405-431); x[] are
// the epoch's misc[3..7] written by k_power_extrap.
__device__ __forceinline__ double power_tail(const Epoch& E, const double* x, int w, double kv,
                                             double k_max) {
  if (w == CHOMP_P_MM) return linear_power(E, kv) * x[0];
  const double* vs = w == CHOMP_P_GM ? x + 1 : x + 3;          // value at k_max, log-slope
  return pow(kv / k_max, vs[1]) * vs[0];
}

__device__ __forceinline__ double halofit_mm(const Epoch& E, double k) {
  // halo.py:1339-1360
  const double lk = log(k);
  const double dk = delta_k_ln(E, lk, k);
  const double y = k / E.hf_k_s;
  const double d2q = dk * (pow(1.0 + dk, E.hf_beta_n) / (1.0 + E.hf_alpha_n * dk) *
                           exp(-(y / 4.0 + y * y / 8.0)));
  const double d2h = (E.hf_a_n * pow(y, 3.0 * E.hf_f1) /
                      (1.0 + E.hf_b_n * pow(y, E.hf_f2) +
                       pow(E.hf_c_n * E.hf_f3 * y, 3.0 - E.hf_gamma_n))) /
                     (1.0 + E.hf_mu_n / y + E.hf_nu_n / (y * y));
  return 2.0 * kPi * kPi / (k * k * k) * (d2q + d2h);
}

// P(k) of one epoch from tables staged in LDS: shared by k_power and by the
// projection integrands (correlation.py:270-275, 387-392 call halo.power_*).
struct PowerEval {
  const Epoch* E;
  const double *ca, *cb, *cp;     // pp coefficients: h_a, h_b, 1-halo term
  int NK, w;
  bool halofit, extrap;
  const double* tail;             // misc[3..7] of the epoch (k_power_extrap)
  double x0, dx, k_min, k_max, c_lo;

  // Stage the coefficient sets of spectrum `which` of epoch table `t` into `sm`
  // (needs 12 (NK-1) doubles) and set the evaluator up.  All threads call it;
  // __syncthreads() must follow before use.
  __device__ __forceinline__ void stage(const chomp_config& cfg, const TabLayout& L,
                                        const Epoch* Els, const double* t, int which,
                                        double* sm) {
    E = Els;
    NK = L.NK;
    halofit = (which & CHOMP_P_HALOFIT) != 0;
    extrap = (which & CHOMP_P_EXTRAPOLATE) != 0 && !halofit;   // HaloFit ignores it
    tail = t + L.off_misc + 3;
    w = which & 15;
    int fa = F_HM, fb = F_HM, fp = F_PPMM;
    if (w == CHOMP_P_GM) { fa = F_HG; fb = F_HM; fp = F_PPGM; }
    else if (w == CHOMP_P_GG) { fa = F_HG; fb = F_HG; fp = F_PPGG; }
    double* a = sm;
    double* b = a + 4 * (NK - 1);
    double* p = b + 4 * (NK - 1);
    if (needs_tables()) {
      copy_doubles(a, t + L.off_kpp[fa], 4 * (NK - 1));
      copy_doubles(b, t + L.off_kpp[fb], 4 * (NK - 1));
      copy_doubles(p, t + L.off_kpp[fp], 4 * (NK - 1));
    }
    ca = a; cb = b; cp = p;
    k_min = cfg.k_min;
    k_max = cfg.k_max;
    x0 = log(cfg.k_min);
    dx = (log(cfg.k_max) - x0) / (double)(NK - 1);
    c_lo = 0.0;
  }
  __device__ __forceinline__ bool needs_tables() const {
    return w != CHOMP_P_LIN && !(halofit && w == CHOMP_P_MM);
  }
  // after the barrier: k < k_min constant (halo.py:314-317)
  __device__ __forceinline__ void finish() {
    if (w != CHOMP_P_LIN && !halofit) {
      const double ha = pp_poly(ca, 0, 0.0), hb = pp_poly(cb, 0, 0.0), p0 = pp_poly(cp, 0, 0.0);
      c_lo = ha * hb + p0 / linear_power(*E, k_min);
    }
  }
  __device__ __forceinline__ double operator()(double kv) const {
    if (w == CHOMP_P_LIN) return linear_power(*E, kv);
    if (halofit) {
      const double pmm = halofit_mm(*E, kv);
      if (w == CHOMP_P_MM) return pmm;
      double ha = 0.0, hb = 0.0, pp = 0.0;               // halo.py:649-672 range rule
      if (kv >= k_min && kv <= k_max) {
        const double lk = log(kv);
        ha = spline_eval_uniform(x0, dx, ca, NK, lk);
        hb = spline_eval_uniform(x0, dx, cb, NK, lk);
        pp = spline_eval_uniform(x0, dx, cp, NK, lk);
      }
      return pmm * ha * hb + pp;
    }
    if (kv < k_min) return linear_power(*E, kv) * c_lo;
    if (extrap ? kv < k_max : kv <= k_max) {
      const double lk = log(kv);
      const double ha = spline_eval_uniform(x0, dx, ca, NK, lk);
      const double hb = spline_eval_uniform(x0, dx, cb, NK, lk);
      const double pp = spline_eval_uniform(x0, dx, cp, NK, lk);
      const double plin = 2.0 * kPi * kPi * delta_k_ln(*E, lk, kv) / (kv * kv * kv);
      return plin * ha * hb + pp;
    }
    if (extrap) return power_tail(*E, tail, w, kv, k_max);
    return 0.0;                                           // k > k_max (or NaN)
  }
};

// Halo(extrapolate=True): the constants of the continuation above k_max, from the knot
// tables of spectrum w (halo.py:300-312: misc[3]; :341-352 / :405-416: value at k_max and
// mean log-slope over knots -7..-1 into misc[4,5] (gm) / misc[6,7] (gg)).  grid n, block 64.
__global__ void k_power_extrap(chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs,
                               double* __restrict__ tab, int w, int epoch0) {
  __shared__ double lv[6], lx[6];
  const int e = epoch0 + blockIdx.x;
  const Epoch& E = epochs[e];
  double* t = tab + (size_t)e * L.stride;
  const PowerFam F = power_families(w);
  const int NK = L.NK;
  const double* ka = t + L.off_knot[F.fa];
  const double* kb = t + L.off_knot[F.fb];
  const double* kp = t + L.off_knot[F.fp];
  const int i = threadIdx.x;
  if (i < 6) {
    const int j = NK - 7 + i;
    const double x = linspace_at(log(cfg.k_min), log(cfg.k_max), NK, j);
    lx[i] = x;
    lv[i] = log(linear_power(E, exp(x)) * ka[j] * kb[j] + kp[j]);
  }
  __syncthreads();
  if (i == 0) {
    const double plin = linear_power(E, cfg.k_max);
    const double ha = ka[NK - 1], hb = kb[NK - 1], pp = kp[NK - 1];
    if (w == CHOMP_P_MM) {
      t[L.off_misc + 3] = ha * hb + pp / plin;
    } else {
      double slope = 0.0;
      for (int m = 0; m < 5; ++m) slope += (lv[m + 1] - lv[m]) / (lx[m + 1] - lx[m]);
      double* vs = t + L.off_misc + (w == CHOMP_P_GM ? 4 : 6);
      vs[0] = plin * ha * hb + pp;
      vs[1] = slope / 5.0;
    }
  }
}

__global__ __launch_bounds__(256) void k_power(chomp_config cfg, TabLayout L,
                                               const Epoch* __restrict__ epochs,
                                               const double* __restrict__ tab, int which,
                                               int epoch0, const double* __restrict__ k,
                                               size_t nk, double* __restrict__ out) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  const int e = epoch0 + blockIdx.y;
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  PowerEval P;
  P.stage(cfg, L, &E, tab + (size_t)e * L.stride, which, sm);
  __syncthreads();
  P.finish();
  double* o = out + (size_t)blockIdx.y * nk;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk;
       i += (size_t)gridDim.x * blockDim.x)
    o[i] = P(k[i]);
}

// ---------------------------------------------------------------------------
// k_power_grid (+ k_power_grid_lanes): Stage E over a whole (k, epoch) grid, built to
// run at HBM speed on large grids.  grid ceil(nk / 512), block 256; a thread owns two
// consecutive k (16-byte loads / stores) and walks the epochs:
//   * ln k, the knot interval and the Eisenstein-Hu shape (k/H0)^(3+n) T(k)^2 / k^3
//     are computed once per k and re-used for every epoch that shares the previous
//     epoch's cosmology (the z-axis of a (k, z) grid): per (k, z) sample that leaves
//     three cubic evaluations and one multiply;
//   * when all 128 k of a wavefront fall into one knot interval (the usual case for
//     a sorted grid) the 12 spline coefficients of an epoch are wave-uniform and
//     come through the scalar cache; any other wavefront takes the per-lane path.
// Algorithmic traffic: 8 B read per k + 8 B written per (k, epoch) sample.
// ---------------------------------------------------------------------------
// P(k) of epoch table t for one k on the per-lane path (any k, any interval).
__device__ __forceinline__ double power_lane(const chomp_config& cfg, const TabLayout& L,
                                             const Epoch& E, const double* t, int fa, int fb,
                                             int fp, int w, bool extrap, double kv) {
  if (w == CHOMP_P_LIN) return linear_power(E, kv);
  const double x0 = log(cfg.k_min);
  const double dx = (log(cfg.k_max) - x0) / (double)(L.NK - 1);
  if (kv < cfg.k_min) {
    const double c_lo = t[L.off_kpp[fa]] * t[L.off_kpp[fb]] +
                        t[L.off_kpp[fp]] / linear_power(E, cfg.k_min);
    return linear_power(E, kv) * c_lo;
  }
  if (extrap ? kv < cfg.k_max : kv <= cfg.k_max) {
    const double lk = log(kv);
    const double ha = spline_eval_uniform(x0, dx, t + L.off_kpp[fa], L.NK, lk);
    const double hb = spline_eval_uniform(x0, dx, t + L.off_kpp[fb], L.NK, lk);
    const double pp = spline_eval_uniform(x0, dx, t + L.off_kpp[fp], L.NK, lk);
    return 2.0 * kPi * kPi * delta_k_ln(E, lk, kv) / (kv * kv * kv) * ha * hb + pp;
  }
  if (extrap) return power_tail(E, t + L.off_misc + 3, w, kv, cfg.k_max);
  return 0.0;
}

// What one wavefront knows about its 128 k (two per lane).
struct KLanes {
  size_t i0;
  bool have0, have1, vec, in0, in1;
  double k0, k1, lk0, lk1;
  int idx0, idx1;
};

__device__ __forceinline__ KLanes load_k_lanes(const chomp_config& cfg, int NK,
                                               const double* __restrict__ k, size_t nk,
                                               size_t thread_index) {
  KLanes s;
  s.i0 = 2 * thread_index;
  s.have0 = s.i0 < nk;
  s.have1 = s.i0 + 1 < nk;
  s.vec = s.have1 && ((nk & 1) == 0);            // rows stay 16-byte aligned
  s.k0 = 1.0;
  s.k1 = 1.0;
  if (s.vec) {
    const double2 kk = *reinterpret_cast<const double2*>(k + s.i0);
    s.k0 = kk.x; s.k1 = kk.y;
  } else {
    if (s.have0) s.k0 = k[s.i0];
    if (s.have1) s.k1 = k[s.i0 + 1];
  }
  const double x0 = log(cfg.k_min);
  const double dx = (log(cfg.k_max) - x0) / (double)(NK - 1);
  const double inv_dx = 1.0 / dx;
  s.lk0 = fast_log(s.k0);
  s.lk1 = fast_log(s.k1);
  s.idx0 = (int)floor((s.lk0 - x0) * inv_dx);
  s.idx1 = (int)floor((s.lk1 - x0) * inv_dx);
  s.idx0 = s.idx0 < 0 ? 0 : (s.idx0 > NK - 2 ? NK - 2 : s.idx0);
  s.idx1 = s.idx1 < 0 ? 0 : (s.idx1 > NK - 2 ? NK - 2 : s.idx1);
  s.in0 = s.k0 >= cfg.k_min && s.k0 <= cfg.k_max;
  s.in1 = s.k1 >= cfg.k_min && s.k1 <= cfg.k_max;
  return s;
}

// 16-byte write-through store (sc1).  Issued from inline asm, so the compiler's hazard
// recogniser does not see a VMEM store: a store of more than 64 bits must not be
// followed directly by a VALU write of its data registers, hence the trailing s_nop.
__device__ __forceinline__ void store_wt16(double* p, double r0, double r1) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d rr = {r0, r1};
  v4f bits = __builtin_bit_cast(v4f, rr);
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(bits) : "memory");
}



// Wavefronts (groups of 128 consecutive k) that cannot take a streaming path are
// collected in a compact list for k_power_grid_lanes.  slow[0..1] are two counters used
// by alternate launches (`parity`): a launch appends through slow[parity] and clears
// slow[parity ^ 1] for the next one, so no separate memset is needed (launches of one
// context are stream-ordered).  slow[2...] is the list.
__device__ __forceinline__ void slow_list_begin(int* slow, int parity) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) slow[parity ^ 1] = 0;
}
__device__ __forceinline__ void slow_list_append(int* slow, int parity, int k_group) {
  const int at = atomicAdd(&slow[parity], 1);
  slow[2 + at] = k_group;
}

// The row-walking streaming pass (epochs of different cosmologies, or small grids).
// grid (ceil(nk / 512), ceil(n_epoch / epochs_per_y)), block 256.  A wavefront whose k
// do not qualify for the fast path only enters itself in the slow list.
__global__ __launch_bounds__(256) void k_power_grid(chomp_config cfg, TabLayout L,
                                                    const Epoch* __restrict__ epochs,
                                                    const double* __restrict__ tab, int w,
                                                    int epoch0, int n_epoch, int epochs_per_y,
                                                    int rot,
                                                    const double* __restrict__ k, size_t nk,
                                                    double* __restrict__ out,
                                                    int* __restrict__ slow, int parity) {
  slow_list_begin(slow, parity);
  const int k_group = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
  const PowerFam F = power_families(w);
  const int NK = L.NK;
  const KLanes s = load_k_lanes(cfg, NK, k, nk, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
  const int idxu = __builtin_amdgcn_readfirstlane(s.idx0);
  // Fast path: every k of the wavefront present, in range and in knot interval idxu or
  // idxu + 1 (a sorted grid straddles at most one knot per wavefront once nk >~ 6500;
  // indices are clamped to NK - 2, so idxu + 1 never runs past the last piece).
  const bool fast = __all(s.have0 && s.have1 && s.in0 && s.in1 &&
                          (s.idx0 == idxu || s.idx0 == idxu + 1) &&
                          (s.idx1 == idxu || s.idx1 == idxu + 1)) &&
                    w != CHOMP_P_LIN;
  if (!fast) {
    if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && (size_t)k_group * 128 < nk)
      slow_list_append(slow, parity, k_group);
    return;
  }
  const int q_lo = blockIdx.y * epochs_per_y;
  int q_hi = q_lo + epochs_per_y;
  if (q_hi > n_epoch) q_hi = n_epoch;
  const double x0 = log(cfg.k_min);
  const double dx = (log(cfg.k_max) - x0) / (double)(NK - 1);
  const bool two = !__all(s.idx0 == idxu && s.idx1 == idxu);  // wave-uniform
  const bool s0 = s.idx0 != idxu, s1 = s.idx1 != idxu;         // lane uses the upper interval
  const double xa = x0 + dx * (double)idxu, xb = x0 + dx * (double)(idxu + 1);
  const double d0 = s.lk0 - (s0 ? xb : xa), d1 = s.lk1 - (s1 ? xb : xa);
  const double k0 = s.k0, k1 = s.k1;
  const int oa = L.off_kpp[F.fa] + 4 * idxu, ob = L.off_kpp[F.fb] + 4 * idxu,
            op = L.off_kpp[F.fp] + 4 * idxu;
  double shape0 = 0.0, shape1 = 0.0;             // 2 pi^2 (k/H0)^(3+n) T^2 / k^3
  // Blocks start their walk at different rows (rows of a large grid are a power-of-two
  // stride apart: in lockstep every wavefront would hit the same HBM channels).
  const int cnt = q_hi - q_lo;
  int q = q_lo + (int)((blockIdx.x * (unsigned)rot) % (unsigned)cnt);
  // wave-uniform operands of one epoch: through the scalar cache, fetched one epoch
  // ahead of their use so that the loop never waits on a scalar load
  struct Row { double A, flag, a0, a1, a2, a3, b0, b1, b2, b3, p0, p1, p2, p3; };
  auto fetch = [&](int qq) {
    const double* t = tab + (size_t)(epoch0 + qq) * L.stride;
    return Row{t[L.off_misc + 1], t[L.off_misc + 2],
               t[oa], t[oa + 1], t[oa + 2], t[oa + 3],
               t[ob], t[ob + 1], t[ob + 2], t[ob + 3],
               t[op], t[op + 1], t[op + 2], t[op + 3]};
  };
  Row nxt = fetch(q);
  for (int j = 0; j < cnt; ++j) {
    const Row c = nxt;
    const int qc = q;
    ++q;
    if (q == q_hi) q = q_lo;
    nxt = fetch(q);                              // (one harmless re-fetch on the last trip)
    const bool same = j > 0 && qc > q_lo && c.flag != 0.0;
    if (!same) {
      const Epoch& E = epochs[epoch0 + qc];
      shape0 = power_shape(E, s.lk0, k0);
      shape1 = power_shape(E, s.lk1, k1);
    }
    double ha0 = fma(fma(fma(c.a3, d0, c.a2), d0, c.a1), d0, c.a0);
    double hb0 = fma(fma(fma(c.b3, d0, c.b2), d0, c.b1), d0, c.b0);
    double pp0 = fma(fma(fma(c.p3, d0, c.p2), d0, c.p1), d0, c.p0);
    double ha1 = fma(fma(fma(c.a3, d1, c.a2), d1, c.a1), d1, c.a0);
    double hb1 = fma(fma(fma(c.b3, d1, c.b2), d1, c.b1), d1, c.b0);
    double pp1 = fma(fma(fma(c.p3, d1, c.p2), d1, c.p1), d1, c.p0);
    if (two) {       // the wavefront straddles a knot: lanes above it use the next piece
      const double* t = tab + (size_t)(epoch0 + qc) * L.stride;
      const double A0 = t[oa + 4], A1 = t[oa + 5], A2 = t[oa + 6], A3 = t[oa + 7];
      const double B0 = t[ob + 4], B1 = t[ob + 5], B2 = t[ob + 6], B3 = t[ob + 7];
      const double P0 = t[op + 4], P1 = t[op + 5], P2 = t[op + 6], P3 = t[op + 7];
      if (s0) {
        ha0 = fma(fma(fma(A3, d0, A2), d0, A1), d0, A0);
        hb0 = fma(fma(fma(B3, d0, B2), d0, B1), d0, B0);
        pp0 = fma(fma(fma(P3, d0, P2), d0, P1), d0, P0);
      }
      if (s1) {
        ha1 = fma(fma(fma(A3, d1, A2), d1, A1), d1, A0);
        hb1 = fma(fma(fma(B3, d1, B2), d1, B1), d1, B0);
        pp1 = fma(fma(fma(P3, d1, P2), d1, P1), d1, P0);
      }
    }
    const double r0 = fma(c.A * shape0, ha0 * hb0, pp0);
    const double r1 = fma(c.A * shape1, ha1 * hb1, pp1);
    double* o = out + (size_t)qc * nk + s.i0;
    // Streamed once, never re-read by this launch: 16-byte write-through (sc1) stores
    // (plain stores leave ~0.5 GB of dirty lines for the end-of-kernel release to
    // write back; MI355X_MICROARCH.md rows "boundary" / "publish-large").
    if (s.vec) {
      store_wt16(o, r0, r1);
    } else {
      __builtin_nontemporal_store(r0, o);
      __builtin_nontemporal_store(r1, o + 1);
    }
  }
}

// ---- large grids of one cosmology: k_power_prep + k_power_stream ------------------
// HBM likes the output written in address order by short-lived wavefronts (a kernel
// whose threads each walk all rows of a 2^20 x 64 grid reaches ~4.5 TB/s of stores, the
// same stores issued row-major by (k chunk, few rows) blocks ~6.5 TB/s).  So everything
// that depends on k alone is tabulated once by k_power_prep -- 16 B per k: the offset of
// ln k in its knot interval and the Eisenstein-Hu shape 2 pi^2 (k/H0)^(3+n) T^2 / k^3 --
// and k_power_stream, launched row-major over (k chunk, PER rows), re-reads that table
// from L2 (a k chunk always lands on the same XCD: gridDim.x is a multiple of 8) and
// does 3 cubics + 1 multiply-add per sample.
constexpr int kWaveIdxMask = 0xffff, kWaveSlow = 1 << 16, kWaveTwo = 1 << 17;

// grid roundup8(ceil(nk / 512)), block 256.  ktab[2 i] = ln k_i - x_idx, ktab[2 i + 1] = shape_i,
// negative when k_i lies in the upper one of the wavefront's two knot intervals;
// winfo[g] = lowest knot interval of k group g | kWaveTwo | kWaveSlow.
__global__ __launch_bounds__(256) void k_power_prep(chomp_config cfg, TabLayout L,
                                                    const Epoch* __restrict__ epochs,
                                                    int e_shape, int w,
                                                    const double* __restrict__ k, size_t nk,
                                                    double* __restrict__ ktab,
                                                    int* __restrict__ winfo,
                                                    int* __restrict__ slow, int parity) {
  slow_list_begin(slow, parity);
  const int k_group = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int NK = L.NK;
  const size_t ti = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const KLanes s = load_k_lanes(cfg, NK, k, nk, ti);
  const int idxu = __builtin_amdgcn_readfirstlane(s.idx0);
  const bool fast = __all(s.have0 && s.have1 && s.in0 && s.in1 &&
                          (s.idx0 == idxu || s.idx0 == idxu + 1) &&
                          (s.idx1 == idxu || s.idx1 == idxu + 1)) &&
                    w != CHOMP_P_LIN;
  const bool two = !__all(s.idx0 == idxu && s.idx1 == idxu);
  if ((threadIdx.x & 63) == 0) {
    winfo[k_group] = idxu | (two ? kWaveTwo : 0) | (fast ? 0 : kWaveSlow);
    if (!fast && (size_t)k_group * 128 < nk) slow_list_append(slow, parity, k_group);
  }
  if (!fast) return;
  const Epoch& E = epochs[e_shape];
  const double x0 = log(cfg.k_min);
  const double dx = (log(cfg.k_max) - x0) / (double)(NK - 1);
  const bool s0 = s.idx0 != idxu, s1 = s.idx1 != idxu;
  const double xa = x0 + dx * (double)idxu, xb = x0 + dx * (double)(idxu + 1);
  const double sh0 = power_shape(E, s.lk0, s.k0), sh1 = power_shape(E, s.lk1, s.k1);
  double4 v;
  v.x = s.lk0 - (s0 ? xb : xa); v.y = s0 ? -sh0 : sh0;
  v.z = s.lk1 - (s1 ? xb : xa); v.w = s1 ? -sh1 : sh1;
  *reinterpret_cast<double4*>(ktab + 4 * ti) = v;
}

// grid (roundup8(ceil(nk / 512)), ceil(n_epoch / PER)), block 256; blockIdx.x fastest =
// row-major over the output.  nk even, out 16-byte aligned; winfo covers every k group
// of the (padded) grid, groups past nk are marked slow.  n_epoch is a multiple of PER
// (the host picks PER accordingly).
template <int PER>
__global__ __launch_bounds__(256) void k_power_stream(TabLayout L, const double* __restrict__ tab,
                                                      int w, int epoch0,
                                                      const double* __restrict__ ktab,
                                                      const int* __restrict__ winfo, size_t nk,
                                                      double* __restrict__ out) {
  const int k_group = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int info = winfo[k_group];
  const size_t ti = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double4 v = *reinterpret_cast<const double4*>(ktab + 4 * ti);   // (padded: in bounds)
  if (info & kWaveSlow) return;
  const int idxu = info & kWaveIdxMask;
  const bool two = (info & kWaveTwo) != 0;
  const PowerFam F = power_families(w);
  const int oa = L.off_kpp[F.fa] + 4 * idxu, ob = L.off_kpp[F.fb] + 4 * idxu,
            op = L.off_kpp[F.fp] + 4 * idxu;
  const int q_lo = blockIdx.y * PER;
  // wave-uniform operands of the PER rows: all fetched through the scalar cache before
  // the first use
  struct Row { double A, a0, a1, a2, a3, b0, b1, b2, b3, p0, p1, p2, p3; };
  Row r[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const double* t = tab + (size_t)(epoch0 + q_lo + j) * L.stride;
    r[j] = Row{t[L.off_misc + 1], t[oa], t[oa + 1], t[oa + 2], t[oa + 3],
               t[ob], t[ob + 1], t[ob + 2], t[ob + 3], t[op], t[op + 1], t[op + 2], t[op + 3]};
  }
  const double d0 = v.x, d1 = v.z;
  const bool s0 = v.y < 0.0, s1 = v.w < 0.0;
  const double shape0 = fabs(v.y), shape1 = fabs(v.w);
  double* o = out + (size_t)q_lo * nk + 2 * ti;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const Row& c = r[j];
    double ha0 = fma(fma(fma(c.a3, d0, c.a2), d0, c.a1), d0, c.a0);
    double hb0 = fma(fma(fma(c.b3, d0, c.b2), d0, c.b1), d0, c.b0);
    double pp0 = fma(fma(fma(c.p3, d0, c.p2), d0, c.p1), d0, c.p0);
    double ha1 = fma(fma(fma(c.a3, d1, c.a2), d1, c.a1), d1, c.a0);
    double hb1 = fma(fma(fma(c.b3, d1, c.b2), d1, c.b1), d1, c.b0);
    double pp1 = fma(fma(fma(c.p3, d1, c.p2), d1, c.p1), d1, c.p0);
    if (two) {       // the wavefront straddles a knot: lanes above it use the next piece
      const double* t = tab + (size_t)(epoch0 + q_lo + j) * L.stride;
      const double A0 = t[oa + 4], A1 = t[oa + 5], A2 = t[oa + 6], A3 = t[oa + 7];
      const double B0 = t[ob + 4], B1 = t[ob + 5], B2 = t[ob + 6], B3 = t[ob + 7];
      const double P0 = t[op + 4], P1 = t[op + 5], P2 = t[op + 6], P3 = t[op + 7];
      if (s0) {
        ha0 = fma(fma(fma(A3, d0, A2), d0, A1), d0, A0);
        hb0 = fma(fma(fma(B3, d0, B2), d0, B1), d0, B0);
        pp0 = fma(fma(fma(P3, d0, P2), d0, P1), d0, P0);
      }
      if (s1) {
        ha1 = fma(fma(fma(A3, d1, A2), d1, A1), d1, A0);
        hb1 = fma(fma(fma(B3, d1, B2), d1, B1), d1, B0);
        pp1 = fma(fma(fma(P3, d1, P2), d1, P1), d1, P0);
      }
    }
    const double r0 = fma(c.A * shape0, ha0 * hb0, pp0);
    const double r1 = fma(c.A * shape1, ha1 * hb1, pp1);
    // write-through: the output is never re-read by this launch, and the k table must
    // stay in L2 next to it
    store_wt16(o, r0, r1);
    o += nk;
  }
}

// The per-lane pass: any k, any knot interval, any order.  1-D grid; every wavefront
// walks work items (listed k group, chunk of epochs).  The chunk length adapts to the
// length of the list: few listed groups -> short chunks over many wavefronts (the
// per-epoch coefficient loads of this path are a dependent chain), a fully listed grid
// -> one item per group.  In-range k still re-use the Eisenstein-Hu shape across epochs
// of one cosmology; k outside [k_min, k_max] take the full formula (halo.py:314-320).
__global__ __launch_bounds__(256) void k_power_grid_lanes(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs,
    const double* __restrict__ tab, int w, bool extrap, int epoch0, int n_epoch,
    const double* __restrict__ k, size_t nk, double* __restrict__ out,
    const int* __restrict__ slow, int parity) {
  const int count = slow[parity];
  if (count == 0) return;
  const PowerFam F = power_families(w);
  const int NK = L.NK;
  const double x0 = log(cfg.k_min);
  const double dx = (log(cfg.k_max) - x0) / (double)(NK - 1);
  const long n_waves = (long)gridDim.x * 4;
  int chunks = (int)(n_waves / count);
  chunks = chunks < 1 ? 1 : (chunks > n_epoch ? n_epoch : chunks);
  const int epochs_per_item = (n_epoch + chunks - 1) / chunks;
  chunks = (n_epoch + epochs_per_item - 1) / epochs_per_item;
  const long n_items = (long)count * chunks;
  const int lane = threadIdx.x & 63;
  for (long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6); item < n_items; item += n_waves) {
    const int k_group = slow[2 + (int)(item / chunks)], c = (int)(item % chunks);
    const int q_lo = c * epochs_per_item;
    int q_hi = q_lo + epochs_per_item;
    if (q_hi > n_epoch) q_hi = n_epoch;
    const KLanes s = load_k_lanes(cfg, NK, k, nk, (size_t)k_group * 64 + lane);
    const double e0 = s.lk0 - (x0 + dx * (double)s.idx0), e1 = s.lk1 - (x0 + dx * (double)s.idx1);
    double sh0 = 0.0, sh1 = 0.0;
    for (int q = q_lo; q < q_hi; ++q) {
      const int e = epoch0 + q;
      const Epoch& E = epochs[e];
      const double* t = tab + (size_t)e * L.stride;
      double* o = out + (size_t)q * nk + s.i0;
      const bool same = q > q_lo && t[L.off_misc + 2] != 0.0;
      const double A = t[L.off_misc + 1];
      if (!same && w != CHOMP_P_LIN) {
        sh0 = power_shape(E, s.lk0, s.k0);
        sh1 = power_shape(E, s.lk1, s.k1);
      }
      if (s.have0) {
        double r;
        if (s.in0 && w != CHOMP_P_LIN) {
          const double ha = pp_poly(t + L.off_kpp[F.fa], s.idx0, e0);
          const double hb = pp_poly(t + L.off_kpp[F.fb], s.idx0, e0);
          const double pp = pp_poly(t + L.off_kpp[F.fp], s.idx0, e0);
          r = fma(A * sh0, ha * hb, pp);
        } else {
          r = power_lane(cfg, L, E, t, F.fa, F.fb, F.fp, w, extrap, s.k0);
        }
        o[0] = r;
      }
      if (s.have1) {
        double r;
        if (s.in1 && w != CHOMP_P_LIN) {
          const double ha = pp_poly(t + L.off_kpp[F.fa], s.idx1, e1);
          const double hb = pp_poly(t + L.off_kpp[F.fb], s.idx1, e1);
          const double pp = pp_poly(t + L.off_kpp[F.fp], s.idx1, e1);
          r = fma(A * sh1, ha * hb, pp);
        } else {
          r = power_lane(cfg, L, E, t, F.fa, F.fb, F.fp, w, extrap, s.k1);
        }
        o[1] = r;
      }
    }
  }
}

// sigma_r at arbitrary scales (SingleEpoch.sigma_r): grid n, block 256.
__global__ __launch_bounds__(256) void k_sigma_r(chomp_config cfg,
                                                 const Epoch* __restrict__ epochs, int e,
                                                 const double* __restrict__ scale,
                                                 const double* __restrict__ snodes,
                                                 double* __restrict__ out) {
  __shared__ Epoch E;
  __shared__ double red[romberg_scratch<4, 2>()];
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  __syncthreads();
  const double s2 = sigma2_block<4>(E, snodes + (size_t)E.cosmo_slot * kSigmaStride, scale[blockIdx.x], cfg,
                                    cfg.cosmo_precision, red);
  if (threadIdx.x == 0) out[blockIdx.x] = sqrt(s2);
}

// Halo.y (NFW) at (ln k, M) pairs.
__global__ void k_y_nfw(const Epoch* __restrict__ epochs, int e,
                        const SiCiTab* __restrict__ sici_g, const double* __restrict__ ln_k,
                        const double* __restrict__ mass, int n, double* __restrict__ out) {
  __shared__ Epoch E;
  __shared__ SiCiTab S;
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  copy_doubles(reinterpret_cast<double*>(&S), reinterpret_cast<const double*>(sici_g),
               (int)(sizeof(SiCiTab) / sizeof(double)));
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = y_nfw(E, S, ln_k[i], log(mass[i]));
}

// Element-wise accessors of one epoch's tables (the public lookup methods of
// MassFunction / HOD / Halo: mass_function.py:243-346, hod.py:189-230,
// halo.py:441-463).
__global__ void k_eval(TabLayout L, const Epoch* __restrict__ epochs, int e,
                       const double* __restrict__ tab, int what,
                       const double* __restrict__ x, int n, double* __restrict__ out) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  const int NM = L.NM;
  double* nu_knots = sm;
  double* lnm_pp = nu_knots + NM;
  double* nu_pp = lnm_pp + 4 * (NM - 1);
  const double* t = tab + (size_t)e * L.stride;
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  copy_doubles(nu_knots, t + L.off_nu, NM);
  copy_doubles(lnm_pp, t + L.off_lnm_pp, 4 * (NM - 1));
  copy_doubles(nu_pp, t + L.off_nu_pp, 4 * (NM - 1));
  __syncthreads();
  const double dlnm = (E.ln_mass_max - E.ln_mass_min) / (double)(NM - 1);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double v = x[i];
    double r = 0.0;
    switch (what) {
      case CHOMP_EV_NU_OF_MASS: r = spline_eval_uniform(E.ln_mass_min, dlnm, nu_pp, NM, log(v)); break;
      case CHOMP_EV_LN_MASS_OF_NU: r = spline_eval(nu_knots, lnm_pp, NM, v); break;
      case CHOMP_EV_F_NU: r = f_nu(E, v); break;
      case CHOMP_EV_BIAS_NU: r = bias_nu(E, v); break;
      case CHOMP_EV_HOD_FIRST: r = zheng_first(E, v); break;
      case CHOMP_EV_HOD_SECOND: r = zheng_second(E, v); break;
      case CHOMP_EV_HOD_CENTRAL: r = zheng_central(E, v); break;
      case CHOMP_EV_HOD_SATELLITE: r = zheng_satellite(E, v); break;
      case CHOMP_EV_VIRIAL_RADIUS: r = exp((E.ln_rv_const + log(v)) * (1.0 / 3.0)); break;
      case CHOMP_EV_CONCENTRATION: r = exp(E.ln_c_const + E.beta * log(v)); break;
      case CHOMP_EV_DELTA_K: r = delta_k_ln(E, log(v), v); break;
      default: r = 0.0;
    }
    out[i] = r;
  }
}

}  // namespace chomp
