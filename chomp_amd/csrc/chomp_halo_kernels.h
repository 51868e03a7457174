// chomp_halo_kernels.h -- Stage K, second half (gfx950): the halo-model knot tables.
//
//   k_halo_nodes    k-independent factors of the halo integrands on the Romberg node grid,
//                   and n_bar (halo.py:674-700)
//   k_halo_knots    the 50-knot integrals h_m, pp_mm, h_g, pp_gm, pp_gg from the node tables
//                   (halo.py:904-1086; HaloExclusion: halo.py:1201-1233)
//   k_halo_knots_deep  the knots whose Romberg runs beyond the node tables (direct evaluation)
//   k_halo_finalize normalisations + not-a-knot splines over ln k (halo.py:916-918,
//                   959-961, 983-986, 1026-1029, 1072-1075)
#pragma once

#include "chomp_mass_kernels.h"

namespace chomp {

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

struct IntegrandHodStat {  // halo.py:745-750 (bias), 786-790 (m_eff), 833-838 (f_sat)
  HaloCtx c;
  int kind;                // 0: nu <N> f b / M, 1: nu <N> f, 2: nu N_sat f / M
  __device__ __forceinline__ double operator()(double ln_nu) const {
    const double nu = exp(ln_nu);
    const double lnm = spline_eval(c.nu_knots, c.lnm_pp, c.NM, nu);
    const double mass = exp(lnm);
    double nf, b = 1.0, n1, n2;
    mf_node(*c.e, nu, ln_nu, kind == 0, &nf, &b);
    if (kind == 2) return nf * zheng_satellite(*c.e, mass) / mass;
    zheng_node(*c.e, mass, lnm, &n1, &n2);
    return kind == 0 ? nf * b * n1 / mass : nf * n1;
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

// Halo.calculate_bias / calculate_m_eff / calculate_f_sat (halo.py:709-838): grid (3, n),
// block 256, out[3 e + kind]; needs n_bar (chomp_halo_setup).
__global__ __launch_bounds__(256) void k_hod_stats(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs, int epoch0,
    const double* __restrict__ tab, const chomp_halo_par* __restrict__ profile,
    const HodDev* __restrict__ hod, const SiCiTab* __restrict__ sici_g,
    double* __restrict__ out) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ SiCiTab S;
  const int kind = blockIdx.x, e = epoch0 + blockIdx.y;
  HaloLds H;
  H.stage(L, E, S, epochs, e, tab + (size_t)e * L.stride, profile, hod, sici_g, sm);
  HaloCtx c{&E, &S, H.nu_knots, H.lnm_pp, L.NM, 0.0, false};
  IntegrandHodStat f{c, kind};
  const double lo = kind == 2 ? E.ln_nu_lo_second : E.ln_nu_lo_first;
  const double v = romberg1<4>(f, lo, log(E.nu_max), cfg.global_precision, cfg.halo_precision,
                               cfg.divmax, H.rest);
  if (threadIdx.x == 0) out[3 * blockIdx.y + kind] = v / E.n_bar_over_rho_bar;
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
constexpr int kKnotNW = 2;
__global__ __launch_bounds__(64 * kKnotNW) void k_halo_knots(
    chomp_config cfg, TabLayout L, double* __restrict__ tab,
    const SiCiTab* __restrict__ sici_g, const double* __restrict__ nodes, int g0, int g1,
    int g2, unsigned mask, int* __restrict__ pending) {
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
  const RombergOut<2> r = romberg_group<kKnotNW, 2>(f, a, b, cfg.global_precision,
                                                    cfg.halo_precision, dmax, red);
  if (threadIdx.x == 0) {
    double* lev = t + L.off_levels;
    const int fa = group_fa(group), fb = group_fb(group);
    const bool more = cfg.divmax > kNodeLevel;
    if (group != 2 && (mask & (1u << fa))) {
      t[L.off_knot[fa] + ik] = r.value[0];
      lev[fa * NK + ik] = (!r.converged[0] && more) ? kPendingLevel : (double)r.level[0];
    }
    bool any = false;
    if (group != 2 && (mask & (1u << fa))) any = any || (!r.converged[0] && more);
    if (mask & (1u << fb)) {
      t[L.off_knot[fb] + ik] = r.value[1];
      lev[fb * NK + ik] = (!r.converged[1] && more) ? kPendingLevel : (double)r.level[1];
      any = any || (!r.converged[1] && more);
    }
    // work list of k_halo_knots_deep: [0] items, [1] next item to hand out, [2...] items
    if (any) pending[2 + atomicAdd(&pending[0], 1)] = (int)((blockIdx.z * gridDim.y + e) * NK + ik);
  }
}

// ---------------------------------------------------------------------------
// k_halo_knots_deep: the knots k_halo_knots listed as not converged at the depth of the
// node tables, redone by direct evaluation up to divmax (the discontinuous HOD integrands
// run to 2^18..2^20 nodes, halo.py:1038-1041, 1084-1086).  1-D grid of any size; blocks
// draw items from the list until it is empty (an empty list costs one load per block).
// k_halo_finalize clears the list.
// ---------------------------------------------------------------------------
// NW wavefronts per integral: 4 when the list can be long (a batch of epochs keeps the chip
// busy with one block per knot), 8 for a single epoch (at most 150 knots: the deep
// integrals set the duration; 16 would cap the registers at 128 and spill).
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_halo_knots_deep(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs,
    double* __restrict__ tab, const chomp_halo_par* __restrict__ profile,
    const HodDev* __restrict__ hod, const SiCiTab* __restrict__ sici_g, int g0, int g1,
    int g2, unsigned mask, int n_epoch, int* __restrict__ pending) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ SiCiTab S;
  __shared__ int item_sh;
  const int NK = L.NK;
  const int count = pending[0];
  if (count == 0) return;          // nothing listed: no traffic on the queue head
  for (;;) {
  __syncthreads();                 // (previous item done with E, S, sm)
  if (threadIdx.x == 0) item_sh = atomicAdd(&pending[1], 1);
  __syncthreads();
  if (item_sh >= count) return;    // block-uniform
  const int item = pending[2 + item_sh];
  const int ik = item % NK, e = (item / NK) % n_epoch, zg = item / (NK * n_epoch);
  const int group = zg == 0 ? g0 : (zg == 1 ? g1 : g2);
  if (group < 0 || group > 2) continue;
  double* t = tab + (size_t)e * L.stride;
  double* lev = t + L.off_levels;
  const int fa = group_fa(group), fb = group_fb(group);
  const bool pa = group != 2 && (mask & (1u << fa)) && lev[fa * NK + ik] == kPendingLevel;
  const bool pb = (mask & (1u << fb)) && lev[fb * NK + ik] == kPendingLevel;
  if (!pa && !pb) continue;
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
    const RombergOut<2> r = romberg_group<NW, 2>(f, group_lower(E, 0), ln_nu_max,
                                                cfg.global_precision, cfg.halo_precision,
                                                cfg.divmax, red);
    va = r.value[0]; vb = r.value[1]; la = r.level[0]; lb = r.level[1];
  } else if (group == 1) {
    IntegrandGM f{c, pa};
    const RombergOut<2> r = romberg_group<NW, 2>(f, group_lower(E, 1), ln_nu_max,
                                                cfg.global_precision, cfg.halo_precision,
                                                cfg.divmax, red);
    va = r.value[0]; vb = r.value[1]; la = r.level[0]; lb = r.level[1];
  } else {
    IntegrandGG f{c};
    const RombergOut<1> r = romberg_group<NW, 1>(f, group_lower(E, 2), ln_nu_max,
                                                cfg.global_precision, cfg.halo_precision,
                                                cfg.divmax, red);
    vb = r.value[0]; lb = r.level[0];
  }
  if (threadIdx.x == 0) {
    if (pa) { t[L.off_knot[fa] + ik] = va; lev[fa * NK + ik] = (double)la; }
    if (pb) { t[L.off_knot[fb] + ik] = vb; lev[fb * NK + ik] = (double)lb; }
  }
  }   // next item
}

// ---------------------------------------------------------------------------
// The deep knots of a FEW epochs (a single Halo object: at most 150 knots): with one block
// per knot the handful of integrals that run to 2^20 nodes would set the duration on a
// mostly idle chip, so each Romberg level is spread over the whole grid instead.
//   k_halo_deep_head     one block per listed knot: levels <= kDeepHead by direct
//                        evaluation; knots that stop there are final, the others leave
//                        their Romberg state (chomp_romberg.h `dump`) in `state`
//   k_halo_deep_level    level i: the 2^(i-1) new nodes of every unfinished knot in chunks
//                        of kDeepChunk nodes over all blocks; partial sums per chunk
//   k_halo_deep_advance  level i: one wavefront per knot adds the chunk sums in order and
//                        does the row / stopping test of scipy.integrate.romberg
// The host queues head, then (level, advance) for every level up to divmax; finished knots
// cost a flag test.  Same nodes, same rows and stopping rule as k_halo_knots_deep -- only
// the order in which a level's node values are added differs.
// ---------------------------------------------------------------------------
constexpr int kDeepHead = 11;                 // levels of the head pass (2049 nodes)
constexpr int kDeepChunk = 2048;              // nodes per work item of a level pass
// per knot: 2 x kRombergDump state, a, b, done[2], level[2], value[2]
constexpr int kDeepState = 2 * kRombergDump + 8;
constexpr int kDsA = 2 * kRombergDump, kDsB = kDsA + 1, kDsDone = kDsA + 2, kDsLevel = kDsA + 4,
              kDsValue = kDsA + 6;

struct DeepItem {
  int ik, e, group, fa, fb;
  bool pa, pb;
};
__device__ __forceinline__ DeepItem deep_item(const TabLayout& L, const int* pending, int p,
                                              int n_epoch, int g0, int g1, int g2,
                                              unsigned mask, const double* tab) {
  DeepItem d;
  const int item = pending[2 + p], NK = L.NK;
  d.ik = item % NK;
  d.e = (item / NK) % n_epoch;
  const int zg = item / (NK * n_epoch);
  d.group = zg == 0 ? g0 : (zg == 1 ? g1 : g2);
  d.fa = group_fa(d.group);
  d.fb = group_fb(d.group);
  const double* lev = tab + (size_t)d.e * L.stride + L.off_levels;
  d.pa = d.group != 2 && (mask & (1u << d.fa)) && lev[d.fa * NK + d.ik] == kPendingLevel;
  d.pb = (mask & (1u << d.fb)) && lev[d.fb * NK + d.ik] == kPendingLevel;
  return d;
}

// One node of a knot's integrand pair (group 2 has one integrand: out[0] = 0).
__device__ __forceinline__ void deep_eval(int group, const HaloCtx& c, bool pa, double x,
                                          double (&out)[2]) {
  if (group == 0) {
    IntegrandMM f{c};
    f(x, out);
  } else if (group == 1) {
    IntegrandGM f{c, pa};
    f(x, out);
  } else {
    IntegrandGG f{c};
    double o1[1];
    f(x, o1);
    out[0] = 0.0;
    out[1] = o1[0];
  }
}

// grid any (blocks stride over the list), block 256.
__global__ __launch_bounds__(256) void k_halo_deep_head(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs, double* __restrict__ tab,
    const chomp_halo_par* __restrict__ profile, const HodDev* __restrict__ hod,
    const SiCiTab* __restrict__ sici_g, int g0, int g1, int g2, unsigned mask, int n_epoch,
    const int* __restrict__ pending, double* __restrict__ state) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ SiCiTab S;
  const int NK = L.NK;
  const int count = pending[0];
  for (int p = blockIdx.x; p < count; p += gridDim.x) {
    __syncthreads();               // (previous knot done with E, S, sm)
    const DeepItem d = deep_item(L, pending, p, n_epoch, g0, g1, g2, mask, tab);
    double* st = state + (size_t)p * kDeepState;
    if (d.group < 0 || d.group > 2 || (!d.pa && !d.pb)) {
      if (threadIdx.x == 0) { st[kDsDone] = 1.0; st[kDsDone + 1] = 1.0; }
      continue;
    }
    double* t = tab + (size_t)d.e * L.stride;
    HaloLds H;
    H.stage(L, E, S, epochs, d.e, t, profile, hod, sici_g, sm);
    double* red = H.rest;
    const double a = group_lower(E, d.group), b = log(E.nu_max);
    HaloCtx c{&E, &S, H.nu_knots, H.lnm_pp, L.NM,
              linspace_at(log(cfg.k_min), log(cfg.k_max), NK, d.ik), (mask & kMaskExclusion) != 0};
    const int top = cfg.divmax < kDeepHead ? cfg.divmax : kDeepHead;
    double va, vb;
    int la, lb;
    bool ca, cb;
    if (d.group == 0) {
      IntegrandMM f{c};
      const RombergOut<2> r = romberg_group<4, 2>(f, a, b, cfg.global_precision,
                                                  cfg.halo_precision, top, red, st);
      va = r.value[0]; vb = r.value[1]; la = r.level[0]; lb = r.level[1];
      ca = r.converged[0]; cb = r.converged[1];
    } else if (d.group == 1) {
      IntegrandGM f{c, d.pa};
      const RombergOut<2> r = romberg_group<4, 2>(f, a, b, cfg.global_precision,
                                                  cfg.halo_precision, top, red, st);
      va = r.value[0]; vb = r.value[1]; la = r.level[0]; lb = r.level[1];
      ca = r.converged[0]; cb = r.converged[1];
    } else {
      IntegrandGG f{c};
      const RombergOut<1> r = romberg_group<4, 1>(f, a, b, cfg.global_precision,
                                                  cfg.halo_precision, top, red,
                                                  st + kRombergDump);
      va = 0.0; vb = r.value[0]; la = 0; lb = r.level[0];
      ca = true; cb = r.converged[0];
    }
    if (threadIdx.x == 0) {
      const bool last = cfg.divmax <= kDeepHead;       // scipy returns the last row then
      const bool fin_a = !d.pa || ca || last, fin_b = !d.pb || cb || last;
      double* lev = t + L.off_levels;
      if (d.pa && fin_a) { t[L.off_knot[d.fa] + d.ik] = va; lev[d.fa * NK + d.ik] = (double)la; }
      if (d.pb && fin_b) { t[L.off_knot[d.fb] + d.ik] = vb; lev[d.fb * NK + d.ik] = (double)lb; }
      st[kDsA] = a;
      st[kDsB] = b;
      st[kDsDone] = fin_a ? 1.0 : 0.0;
      st[kDsDone + 1] = fin_b ? 1.0 : 0.0;
    }
  }
}

// grid any, block 256: level `lev` (> kDeepHead) of every unfinished knot.
__global__ __launch_bounds__(256) void k_halo_deep_level(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs,
    const double* __restrict__ tab, const chomp_halo_par* __restrict__ profile,
    const HodDev* __restrict__ hod, const SiCiTab* __restrict__ sici_g, int g0, int g1, int g2,
    unsigned mask, int n_epoch, const int* __restrict__ pending,
    const double* __restrict__ state, double* __restrict__ part, int pstride, int lev) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ SiCiTab S;
  const int NK = L.NK;
  const int count = pending[0];
  const long numtosum = 1L << (lev - 1);
  const int nchunk = (int)(numtosum / kDeepChunk);     // lev > kDeepHead: >= 1
  const long total = (long)count * nchunk;
  const long per = (total + gridDim.x - 1) / gridDim.x;   // consecutive items: one knot's
  long w = (long)blockIdx.x * per;                        // staging serves several chunks
  const long w_end = w + per < total ? w + per : total;
  int staged = -1;
  HaloLds H;
  DeepItem d;
  int flip = 0;
  for (; w < w_end; ++w) {
    const int p = (int)(w / nchunk), ch = (int)(w % nchunk);
    const double* st = state + (size_t)p * kDeepState;
    const bool da = st[kDsDone] != 0.0, db = st[kDsDone + 1] != 0.0;
    if (da && db) { w += nchunk - 1 - ch; continue; }   // block-uniform: skip the knot
    if (p != staged) {
      __syncthreads();
      d = deep_item(L, pending, p, n_epoch, g0, g1, g2, mask, tab);
      H.stage(L, E, S, epochs, d.e, tab + (size_t)d.e * L.stride, profile, hod, sici_g, sm);
      staged = p;
      flip = 0;
    }
    const double a = st[kDsA], b = st[kDsB];
    HaloCtx c{&E, &S, H.nu_knots, H.lnm_pp, L.NM,
              linspace_at(log(cfg.k_min), log(cfg.k_max), NK, d.ik), (mask & kMaskExclusion) != 0};
    const double h = (b - a) / (double)numtosum;
    const double lox = a + 0.5 * h;
    double s0 = 0.0, s1 = 0.0;
    const long j0 = (long)ch * kDeepChunk + threadIdx.x;
#pragma unroll 2
    for (int i = 0; i < kDeepChunk / 256; ++i) {
      double o[2];
      deep_eval(d.group, c, d.pa && !da, lox + h * (double)(j0 + 256L * i), o);
      s0 += o[0];
      s1 += o[1];
    }
    const double t0 = group_sum<4>(s0, H.rest, flip);
    const double t1 = group_sum<4>(s1, H.rest, flip);
    if (threadIdx.x == 0) {
      double* out = part + ((size_t)p * pstride + ch) * 2;   // pstride: chunks of level divmax
      out[0] = t0;
      out[1] = t1;
    }
  }
}

// grid ceil(max knots / 4), block 256: one wavefront per listed knot.
__global__ __launch_bounds__(256) void k_halo_deep_advance(
    chomp_config cfg, TabLayout L, double* __restrict__ tab, int g0, int g1, int g2,
    unsigned mask, int n_epoch, const int* __restrict__ pending, double* __restrict__ state,
    const double* __restrict__ part, int pstride, int lev) {
  const int count = pending[0];
  const int p = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (p >= count) return;
  double* st = state + (size_t)p * kDeepState;
  const bool done_a = st[kDsDone] != 0.0, done_b = st[kDsDone + 1] != 0.0;
  if (done_a && done_b) return;
  const DeepItem d = deep_item(L, pending, p, n_epoch, g0, g1, g2, mask, tab);
  const int nchunk = (int)((1L << (lev - 1)) / kDeepChunk);
  const double range = st[kDsB] - st[kDsA];
  const double n = (double)(1L << lev);
  const double c_il = CHOMP_ROMBERG_C[lev][lane & 31];
  const bool last = lev >= cfg.divmax;
  double* t = tab + (size_t)d.e * L.stride;
  double* levs = t + L.off_levels;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (q == 0 ? done_a : done_b) continue;             // wave-uniform
    double s = 0.0;                // chunk sums in chunk order: lane-strided, then the lanes
    for (int ch = lane; ch < nchunk; ch += 64) s += part[((size_t)p * pstride + ch) * 2 + q];
    const double S = wave_sum(s);
    double* sq = st + q * kRombergDump;
    const double ordsum = sq[32] + S;
    const double Ti = range * ordsum / n;
    double Tl = lane < 32 ? sq[lane] : 0.0;
    if (lane == lev) Tl = Ti;
    const double cur = wave_sum(lane < 32 ? c_il * Tl : 0.0);
    const double err = fabs(cur - sq[33]);
    const bool conv = err < cfg.global_precision || err < cfg.halo_precision * fabs(cur);
    if (lane == lev) sq[lane] = Ti;
    if (lane == 0) {
      sq[32] = ordsum;
      sq[33] = cur;
      if (conv || last) {
        const int fam = q == 0 ? d.fa : d.fb;
        t[L.off_knot[fam] + d.ik] = cur;
        levs[fam * L.NK + d.ik] = (double)lev;
        st[kDsDone + q] = 1.0;
      }
    }
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
    unsigned fam_mask, int* __restrict__ pending) {
  extern __shared__ __align__(16) double sm[];
  const int NK = L.NK;
  if (blockIdx.x == 0 && threadIdx.x == 0) { pending[0] = 0; pending[1] = 0; }
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

}  // namespace chomp
