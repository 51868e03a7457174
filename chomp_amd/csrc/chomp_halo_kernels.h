// chomp_halo_kernels.h -- Stage K, second half (gfx950): the halo-model knot tables.
//
//   k_nu_mass       the nu table of the mass function; its per-epoch tail builds the mass
//                   function's splines and normalisations and (chomp_stage_k) the node tables
//   k_halo_nodes    k-independent factors of the halo integrands on the Romberg node grid
//                   (the same node tables, for a halo set-up on its own)
//   k_halo_knots    the 50-knot integrals h_m, pp_mm, h_g, pp_gm, pp_gg from the node tables
//                   (halo.py:904-1086; HaloExclusion: halo.py:1201-1233), and n_bar
//                   (halo.py:674-700)
//   k_halo_knots_fast  the knots whose Romberg runs beyond the node tables; the block that
//                   completes an epoch goes on with its normalisations and not-a-knot
//                   splines over ln k (halo.py:916-918, 959-961, 983-986, 1026-1029, 1072-1075)
#pragma once

#include <type_traits>

#include "chomp_mass_kernels.h"

namespace chomp {

// ---------------------------------------------------------------------------
// Halo integrands over ln nu (halo.py:702-707, 922-927, 964-969, 989-994,
// 1032-1041, 1078-1086).  The reference multiplies each integrand by a constant
// `norm` and divides it out again; it cancels in the relative stopping test and is
// omitted.
// ---------------------------------------------------------------------------
constexpr double kPendingLevel = -1.0;   // levels-table marker: needs the deep pass
// The work lists of the knots that run beyond the node tables (ints; one buffer per context):
//   [0] items listed at the front, [1] next item to hand out, [2] items listed at the back
//       (the list is drawn front first: k_halo_knots puts the knots of the highest k, whose
//       Romberg runs deepest, at the front, so that a launch ends with short knots)
//                                                                      (k_halo_knots_fast)
//   [4] items handed on to the literal evaluation, [5] next of those (k_halo_knots_literal)
//   [kPendingDraw + r] next item of round r of k_halo_knots_fast (a round works off as many
//       listed knots as the sample buffer has slots; one round unless the batch is huge)
//   [kPendingEvCount + r], [kPendingEvDraw + r] knots of round r that the lean instance of
//       k_halo_knots_fast handed to the one that can evaluate the integrand (see EVAL there):
//       how many, and the next to hand out
//   [kPendingHead ...] the items; the second list follows at pending_literal_base(), the list
//       positions of the knots handed to the evaluating instance at pending_eval_base() (a
//       round's at the round's first slot: it cannot hand on more knots than it has slots).
constexpr int kPendingDraw = 8;
constexpr int kPendingRounds = 32;
constexpr int kPendingEvCount = kPendingDraw + kPendingRounds;
constexpr int kPendingEvDraw = kPendingEvCount + kPendingRounds;
constexpr int kPendingHead = kPendingEvDraw + kPendingRounds;
__host__ __device__ inline size_t pending_literal_base(size_t n_epoch, int NK) {
  return (size_t)kPendingHead + 3 * n_epoch * (size_t)NK;
}
__host__ __device__ inline size_t pending_eval_base(size_t n_epoch, int NK) {
  return (size_t)kPendingHead + 2 * 3 * n_epoch * (size_t)NK;
}
__host__ __device__ inline size_t pending_ints(size_t n_epoch, int NK) {
  return (size_t)kPendingHead + 3 * 3 * n_epoch * (size_t)NK;
}
constexpr int kItemOpenA = 1 << 29, kItemOpenB = 1 << 30;   // flags of a listed item: see k_halo_knots
constexpr int kItemIndex = kItemOpenA - 1;                  // ... and the mask of its index
constexpr unsigned kMaskExclusion = 1u << 8;   // bit of the kernels' family mask: HaloExclusion
constexpr unsigned kMaskDeepNodes = 1u << 9;   // ... the node tables hold level kNodeTabLevel too

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

// The same by one wavefront (all 64 lanes call; lane 0 writes E, which lives in LDS): the
// logarithms, powers and spline look-ups that apply_halo_hod does one after the other are
// independent of each other in three stages, so each stage is ONE call with a different
// argument per lane -- the serial version is ~1500 dependent instructions on the critical
// path of every halo set-up.  Same operations on the same arguments: identical numbers.
__device__ __forceinline__ void apply_halo_hod_wave(Epoch& E, const chomp_halo_par& hp,
                                                    const HodDev& h, const double* nu_pp,
                                                    double lnm0, int NM) {
  const int lane = threadIdx.x & 63;
  const double c0 = hp.c0 / (1.0 + E.z);
  const double dv = (hp.delta_v == -1.0) ? E.delta_v : hp.delta_v;
  // stage A: five logarithms, two powers of ten, one exponential
  const double la = lane == 0 ? 3.0 / (4.0 * kPi * dv * E.rho_bar)
                  : lane == 1 ? c0 : lane == 2 ? E.m_star
                  : lane == 3 ? h.first_zero : h.second_zero;
  const double lg = log(lane < 5 ? la : 1.0);
  const double pw = pow(10.0, lane == 0 ? h.log_M_0 : h.log_M_1p);
  const double mass_min = exp(E.ln_mass_min);
  const double ln_a0 = readlane_d(lg, 0), ln_c0 = readlane_d(lg, 1), ln_mstar = readlane_d(lg, 2);
  // stage B: nu at the two zeros of the HOD moments (lanes 0, 1)
  const double dlnm = (E.ln_mass_max - E.ln_mass_min) / (double)(NM - 1);
  const double zero = lane == 0 ? h.first_zero : h.second_zero;
  const double ln_zero = lane == 0 ? readlane_d(lg, 3) : readlane_d(lg, 4);
  double nu = E.nu_min;
  if (zero > -1.0 && zero > mass_min) nu = spline_eval_uniform(lnm0, dlnm, nu_pp, NM, ln_zero);
  // stage C: their logarithms
  const double ln_nu = log(nu);
  const double ln_nu1 = readlane_d(ln_nu, 0), ln_nu2 = readlane_d(ln_nu, 1);
  const double M0 = readlane_d(pw, 0), M1p = readlane_d(pw, 1);
  if (lane == 0) {
    E.c0 = c0;
    E.beta = hp.beta;
    E.prof_delta_v = dv;
    E.ln_rv_const = ln_a0;
    E.ln_c_const = ln_c0 - E.beta * ln_mstar;
    E.hod_log_M_min = h.log_M_min; E.hod_sigma = h.sigma; E.hod_log_M_0 = h.log_M_0;
    E.hod_log_M_1p = h.log_M_1p; E.hod_alpha = h.alpha;
    E.hod_first_zero = h.first_zero; E.hod_second_zero = h.second_zero;
    E.hod_safe_norm = h.safe_norm;
    E.hod_M0 = M0;
    E.hod_M1p = M1p;
    E.ln_nu_lo_first = ln_nu1;
    E.ln_nu_lo_second = ln_nu2;
  }
}

// Lower limit of group g's integrals: 0: nu_min; 1: nu(first_moment_zero); 2:
// nu(second_moment_zero) (halo.py:909-911, 935-939, 1002-1006).
__device__ __forceinline__ double group_lower(const Epoch& E, int group) {
  return group == 0 ? log(E.nu_min) : (group == 1 ? E.ln_nu_lo_first : E.ln_nu_lo_second);
}

// Stage what every halo-integral block needs into LDS (all threads call; ends with a
// barrier).  The epoch record already carries its halo / HOD constants: the node-table stage
// of a halo set-up writes them (halo_nodes_block).
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
    (void)profile; (void)hod;
    __syncthreads();
  }
};

// Fields of node x = ln nu of group `group` (see kNodeFields).
__device__ __forceinline__ void halo_node_fields(const Epoch& E, const double* nu_knots,
                                                 const double* lnm_pp, int NM, int group,
                                                 double x, double (&f)[kNodeFields]) {
  const double nu = exp(x);
  const double lnm = spline_eval(nu_knots, lnm_pp, NM, nu);
  const double mass = exp(lnm);
  double nf, bias = 0.0;
  mf_node(E, nu, x, group != 2, &nf, &bias);
  const double ln_c = E.ln_c_const + E.beta * lnm;
  const double ln_rv = (E.ln_rv_const + lnm) * (1.0 / 3.0);
  const double con = exp(ln_c);
  const double cp = 1.0 + con;
  const double ln_cp = log(cp);
  // state: the discrete state of the integrand at the node (halo_eval_coded's code): bit 0 the
  // occupation below one (which power of y, halo.py:1038-1041, 1084-1086), bit 1 satellites
  // on, bit 2 a step-function central occupation above its threshold
  double wA, wB;
  int state = 0;
  if (group == 0) {
    wA = nf * bias;
    wB = nf * mass;
  } else {
    double n1, n2;
    zheng_node(E, mass, lnm, &n1, &n2);
    state = (mass - E.hod_M0 > 0.0) ? 2 : 0;
    if (E.hod_sigma <= 0.0 && lnm * 0.43429448190325182765 > E.hod_log_M_min) state |= 4;
    if (group == 1) {
      wA = nf * bias * n1 / mass;
      wB = nf * n1;
      state |= n1 < 1.0 ? 1 : 0;
    } else {
      wA = 0.0;
      wB = nf * n2 / mass;
      state |= n2 < 1.0 ? 1 : 0;
    }
  }
  f[0] = wA; f[1] = wB; f[2] = ln_rv - ln_c; f[3] = con; f[4] = ln_cp;
  f[5] = 1.0 / (ln_cp - con / cp); f[6] = (double)state;
  const double rs = exp(f[2]);
  f[7] = rs; f[8] = 1.0 / (cp * rs);
}

// The integrand pair of a knot at one node, from the node's fields (out[0] = wA y,
// out[1] = wB (flag ? y : y^2), flag = bit 0 of the state; group 2 uses only out[1]).
// kk: the knot's ln k, k, 1 / k.
struct KnotK {
  double ln_k, k, inv_k;
  __device__ __forceinline__ explicit KnotK(double ln_k_) : ln_k(ln_k_), k(exp(ln_k_)) { inv_k = 1.0 / k; }
  // the same for a knot that is the whole wavefront's: the three numbers in scalar registers
  __device__ __forceinline__ static KnotK uniform(double ln_k_) {
    KnotK r(ln_k_);
    r.ln_k = readlane_d(r.ln_k, 0);      // (v_readfirstlane-like: every lane holds the same value)
    r.k = readlane_d(r.k, 0);
    r.inv_k = readlane_d(r.inv_k, 0);
    return r;
  }
};
__device__ __forceinline__ void node_pair(const SiCiTab& S, const KnotK& kk, bool exclusion,
                                          double wA, double wB, double ln_rs, double con,
                                          double ln_cp, double inv_mass_k, double state,
                                          double rs, double inv_cprs, double (&out)[2]) {
  double z;             // k r_s; k * 2 r_v = 2 c z
  const double y = y_nfw_core_tab(S, kk.ln_k, kk.k, kk.inv_k, ln_rs, con, ln_cp, inv_mass_k, rs,
                                  inv_cprs, &z);
  out[0] = wA * y;
  if (exclusion) out[0] *= exclusion_window(S, 2.0 * con * z);
  out[1] = wB * (((int)state & 1) ? y : y * y);
}

// The node table of one (epoch, group) by a whole block: every node of the level-kNodeLevel
// grid over [group_lower, ln nu_max], the limits behind it, and -- so that a knot's first
// Romberg round fills a wavefront exactly (k_halo_knots) -- the integrand pair of every
// knot at the upper end point, endp[2 ik + {0, 1}].  E: the complete epoch record (LDS).
__device__ __forceinline__ void halo_nodes_block(const chomp_config& cfg, const TabLayout& L,
                                                 const Epoch& E, const SiCiTab& S,
                                                 const double* nu_knots, const double* lnm_pp,
                                                 int group, bool exclusion,
                                                 double* __restrict__ node,
                                                 double* __restrict__ endp, bool deep,
                                                 int chunk = 0, int n_chunks = 1) {
  const double a = group_lower(E, group), b = log(E.nu_max);
  // (chunk c of n: nodes c, c + n, c + 2 n, ... dealt to the threads; knots likewise)
  // deep: level kNodeTabLevel as well (the coarse samples of k_halo_knots_fast)
  const int count = deep ? kNodeCount : kNodeBase;
  for (int idx = chunk + n_chunks * (int)threadIdx.x; idx < count;
       idx += n_chunks * (int)blockDim.x) {
    double x;
    if (idx < 2) {
      x = idx == 0 ? a : b;
    } else {
      const int m = idx - 1;
      const int lev = 32 - __builtin_clz((unsigned)m);      // floor(log2 m) + 1
      const long j = m - (1 << (lev - 1));
      const double h = ldexp(b - a, 1 - lev);
      x = (a + 0.5 * h) + h * (double)j;
    }
    double f[kNodeFields];
    halo_node_fields(E, nu_knots, lnm_pp, L.NM, group, x, f);
#pragma unroll
    for (int q = 0; q < kNodeFields; ++q) node[q * kNodeCount + idx] = f[q];
  }
  if (threadIdx.x == 0 && chunk == 0) {
    node[kNodeFields * kNodeCount] = a;
    node[kNodeFields * kNodeCount + 1] = b;
    // (the limits of the knots' ln k grid, halo.py:52-54, for k_halo_knots: two logarithms --
    //  ~190 instructions -- at the head of every one of its single-wavefront blocks otherwise)
    node[kNodeFields * kNodeCount + 2] = log(cfg.k_min);
    node[kNodeFields * kNodeCount + 3] = log(cfg.k_max);
  }
  // (the knots' end points from the TOP of the block: with six or more chunks of a level-10
  //  table the node loop leaves the last wavefront idle, and its lanes do this beside it -- a
  //  second node's fields and an NFW transform that the threads of wavefront 0 did after theirs)
  const int rt = (int)blockDim.x - 1 - (int)threadIdx.x;
  if (chunk + n_chunks * rt >= L.NK) return;
  double fb[kNodeFields];
  halo_node_fields(E, nu_knots, lnm_pp, L.NM, group, b, fb);   // (every thread: no exchange)
  for (int ik = chunk + n_chunks * rt; ik < L.NK; ik += n_chunks * (int)blockDim.x) {
    const KnotK kk(linspace_at(log(cfg.k_min), log(cfg.k_max), L.NK, ik));
    double o[2];
    node_pair(S, kk, exclusion, fb[0], fb[1], fb[2], fb[3], fb[4], fb[5], fb[6], fb[7], fb[8], o);
    endp[2 * ik] = o[0];
    endp[2 * ik + 1] = o[1];
  }
}

// What a halo set-up does to the state of an epoch before its knots are integrated: the
// halo / HOD constants into the record, the status bits of the previous build cleared, the
// completion counter of its knots armed (1: the token k_halo_knots_fast takes).
__device__ __forceinline__ void halo_epoch_begin(Epoch& E, const chomp_halo_par& hp,
                                                 const HodDev& h, const double* nu_pp,
                                                 double lnm0, int NM, unsigned* status_e,
                                                 int* npend_e, int* pending, bool first_epoch) {
  // (called by the 64 lanes of one wavefront)
  apply_halo_hod_wave(E, hp, h, nu_pp, lnm0, NM);
  if ((threadIdx.x & 63) != 0) return;
  atomicAnd(status_e, ~kStHaloBits);
  *npend_e = 1;
  // the work list of the knots: emptied once per set-up, before any knot is integrated
  // (every block of the previous set-up's k_halo_knots_fast has finished by now)
  if (first_epoch) {
    pending[0] = 0; pending[1] = 0; pending[2] = 0; pending[4] = 0; pending[5] = 0;
    for (int r = 0; r < 3 * kPendingRounds; ++r) pending[kPendingDraw + r] = 0;
  }
}

// ---------------------------------------------------------------------------
// k_halo_nodes: grid (n_epoch, n_groups, n_chunks), block 256: chomp_halo_setup on its own
// (after a chomp_mass_setup; the fused chomp_stage_k does the same in k_mass_nodes).  Chunk
// blockIdx.z of the (epoch, group) node table; every block derives the epoch's constants, the
// first one publishes them.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halo_nodes(
    chomp_config cfg, TabLayout L, Epoch* __restrict__ epochs, const double* __restrict__ tab,
    const chomp_halo_par* __restrict__ profile, const HodDev* __restrict__ hod,
    const SiCiTab* __restrict__ sici_g, double* __restrict__ nodes, double* __restrict__ endp,
    int g0, int g1, int g2, unsigned mask, unsigned* __restrict__ status,
    int* __restrict__ npend, int* __restrict__ pending) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ SiCiTab S;
  const int e = blockIdx.x;
  const int group = blockIdx.y == 0 ? g0 : (blockIdx.y == 1 ? g1 : g2);
  HaloLds H;
  H.stage(L, E, S, epochs, e, tab + (size_t)e * L.stride, profile, hod, sici_g, sm);
  __shared__ unsigned scratch_status;
  __shared__ int scratch_npend;
  if (threadIdx.x < 64) {
    // (every group's block derives the same constants; the first one publishes them)
    const bool first = blockIdx.y == 0 && blockIdx.z == 0;
    halo_epoch_begin(E, profile[e], hod[e], H.nu_pp, tab[(size_t)e * L.stride + L.off_ln_mass],
                     L.NM, first ? &status[e] : &scratch_status,
                     first ? &npend[e] : &scratch_npend, pending, first && e == 0);
  }
  __syncthreads();
  if (blockIdx.y == 0 && blockIdx.z == 0)
    copy_doubles(reinterpret_cast<double*>(&epochs[e]), reinterpret_cast<const double*>(&E),
                 kEpochDoubles);
  if (group < 0 || group > 2) return;            // n_bar only: the record is all it needs
  halo_nodes_block(cfg, L, E, S, H.nu_knots, H.lnm_pp, group, (mask & kMaskExclusion) != 0,
                   nodes + ((size_t)e * 3 + group) * kNodeStride,
                   endp + ((size_t)e * 3 + group) * 2 * L.NK, (mask & kMaskDeepNodes) != 0,
                   (int)blockIdx.z, (int)gridDim.z);
}

// ---------------------------------------------------------------------------
// k_mass_nodes: grid (n_epoch, max(n_groups, 1), n_chunks), block 256.  Once the nu table of
// the batch is complete (k_nu_table): the mass function's splines and normalisations
// (mass_setup_block) and, with do_nodes (chomp_stage_k: the halo model follows in the same
// call), chunk blockIdx.z of the node table of group groups[blockIdx.y] straight from the
// splines still in LDS.  Every block of an epoch repeats the mass function part (cheap, a
// latency chain, and the chip is otherwise idle here); the first one publishes it.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mass_nodes(
    chomp_config cfg, TabLayout L, Epoch* __restrict__ epochs, const double* __restrict__ search,
    double* __restrict__ tab, const chomp_halo_par* __restrict__ mass_par, int mf_kind,
    const TinkerTab* __restrict__ tinker, const double* __restrict__ gl16, int do_nodes,
    const chomp_halo_par* __restrict__ profile, const HodDev* __restrict__ hod,
    const SiCiTab* __restrict__ sici_g, double* __restrict__ nodes, double* __restrict__ endp,
    int g0, int g1, int g2, unsigned mask, unsigned* __restrict__ status,
    int* __restrict__ npend, int* __restrict__ pending) {
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ SiCiTab S;
  const int e = blockIdx.x;
  const bool first = blockIdx.y == 0 && blockIdx.z == 0;
  __shared__ unsigned scratch_status;
  __shared__ int scratch_npend;
  MSTAMP(0);
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  if (do_nodes)
    copy_doubles(reinterpret_cast<double*>(&S), reinterpret_cast<const double*>(sici_g),
                 (int)(sizeof(SiCiTab) / sizeof(double)));
  __syncthreads();
  MassLds M;
  M.carve(sm, L.NM);
  const int n_search = (int)(search[(e * 2 + 0) * 2 + 1] + search[(e * 2 + 1) * 2 + 1]);
  mass_setup_block(cfg, L, E, epochs, e, first, search[(e * 2 + 0) * 2], search[(e * 2 + 1) * 2],
                   n_search, tab + (size_t)e * L.stride, mass_par[e], mf_kind, tinker, gl16, M);
  MSTAMP(5);
  if (!do_nodes) return;
  if (threadIdx.x < 64)
    halo_epoch_begin(E, profile[e], hod[e], M.c_nu, M.x_lnm[0], L.NM,
                     first ? &status[e] : &scratch_status, first ? &npend[e] : &scratch_npend,
                     pending, first && e == 0);
  __syncthreads();
  MSTAMP(6);
  if (first)
    copy_doubles(reinterpret_cast<double*>(&epochs[e]), reinterpret_cast<const double*>(&E),
                 kEpochDoubles);
  const int group = blockIdx.y == 0 ? g0 : (blockIdx.y == 1 ? g1 : g2);
  if (group < 0 || group > 2) return;            // n_bar only: the record is all it needs
  halo_nodes_block(cfg, L, E, S, M.y_nu, M.c_lnm, group, (mask & kMaskExclusion) != 0,
                   nodes + ((size_t)e * 3 + group) * kNodeStride,
                   endp + ((size_t)e * 3 + group) * 2 * L.NK, (mask & kMaskDeepNodes) != 0,
                   (int)blockIdx.z, (int)gridDim.z);
  MSTAMP(7);
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
  KnotK kk;
  bool exclusion;         // HaloExclusion (halo.py:1208-1233): window on the 2-halo term
  __device__ __forceinline__ void operator()(double, double (&out)[2], int lev, long j) const {
    const int idx = node_index(lev, j);
    const double* n = node + idx;
    node_pair(*sici, kk, exclusion, n[0], n[kNodeCount], n[2 * kNodeCount], n[3 * kNodeCount],
              n[4 * kNodeCount], n[5 * kNodeCount], n[6 * kNodeCount], n[7 * kNodeCount],
              n[8 * kNodeCount], out);
  }
};

__device__ __forceinline__ int group_fa(int group) { return group == 0 ? F_HM : F_HG; }
__device__ __forceinline__ int group_fb(int group) {
  return group == 0 ? F_PPMM : (group == 1 ? F_PPGM : F_PPGG);
}

// ---------------------------------------------------------------------------
// halo_finalize_block: the end of an epoch's halo set-up by a whole block (>= 256 threads),
// once all its knots are final: normalise the families of fam_mask, build their not-a-knot
// splines over ln k (the builds run in lockstep, one wavefront each, parallel cyclic
// reduction), the Stage-E record, n_bar into the epoch record.  sm: 51 NK doubles.
// ---------------------------------------------------------------------------
__host__ __device__ inline int finalize_lds_doubles(int NK) { return 42 * NK; }
// through: the knot values were (in part) written by other blocks of this launch, with agent
// scope: they are read with agent-scope loads, and the caller needs no fence in front.
__device__ __forceinline__ void halo_finalize_block(const chomp_config& cfg, const TabLayout& L,
                                                    Epoch* __restrict__ epochs,
                                                    double* __restrict__ tab, int e,
                                                    unsigned fam_mask,
                                                    unsigned* __restrict__ status, double* sm,
                                                    bool through = false) {
  const int NK = L.NK;
  double* xk = sm;                      // [NK]
  double* yk = xk + NK;                 // [5][NK]
  double* work = yk + 5 * NK;           // [4][9 NK]
  double* t = tab + (size_t)e * L.stride;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double nbr = t[L.off_misc];                    // n_bar / rho_bar
  const double rho_bar = epochs[e].rho_bar;
  for (int i = threadIdx.x; i < NK; i += blockDim.x)
    xk[i] = linspace_at(log(cfg.k_min), log(cfg.k_max), NK, i);
  if (threadIdx.x == 0) {
    // Stage-E record: amplitude of Delta^2 and "same cosmology as the previous epoch"
    const Epoch& E = epochs[e];
    t[L.off_misc + 1] = E.amp * E.sigma_norm * E.sigma_norm;
    t[L.off_misc + 2] = (e > 0 && same_cosmology(E, epochs[e - 1])) ? 1.0 : 0.0;
    epochs[e].n_bar_over_rho_bar = nbr;                // halo.py:692-700
    epochs[e].n_bar = nbr * rho_bar;
  }
  const int rounds = (fam_mask >> 4) ? 2 : 1;          // (family 4 = pp_gg is the only one of round 1)
  for (int round = 0; round < rounds; ++round) {
    const int f = wave + 4 * round;                      // (wavefronts beyond the fourth only keep
    const bool active = wave < 4 && f < 5 && ((fam_mask >> f) & 1u);   //  the barriers company)
    if (active) {
      const double n_bar = nbr * rho_bar;
      double scale = 1.0;
      if (f == F_PPMM) scale = 1.0 / rho_bar;                       // halo.py:983
      else if (f == F_HG) scale = 1.0 / nbr;                        // :959
      else if (f == F_PPGM) scale = 1.0 / n_bar;                    // :1072
      else if (f == F_PPGG) scale = rho_bar / (n_bar * n_bar);      // :1026
      bool bad = false;
      for (int i = lane; i < NK; i += 64) {
        const double raw = through ? __hip_atomic_load(&t[L.off_knot[f] + i], __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT)
                                   : t[L.off_knot[f] + i];
        const double v = raw * scale;
        yk[f * NK + i] = v;
        t[L.off_knot[f] + i] = v;
        bad = bad || !(fabs(v) <= 1.79769313486231570815e308);   // NaN or infinity
      }
      if (__any(bad) && lane == 0) atomicOr(&status[e], kStNonfinite);
    }
    __syncthreads();
    const int fs = active ? f : 0;
    spline_build_pcr(xk, yk + fs * NK, NK, t + L.off_kpp[fs], work + (wave & 3) * 9 * NK, lane, 64,
                     active);
  }
  // The epoch's status word is final here (every kernel of the set-up has had its say, this
  // block its own behind the barriers above): mirrored into the pinned host words, so that a
  // status post behind the set-up puts NOTHING on the stream: the word arrives with the
  // set-up's sequence number, which is what the host waits for -- a device-to-host copy node
  // between two kernels cost the stream ~12 us, and so did an event record alone
  // (tools/scratch/c4_gap.py).
  if (L.h_status != nullptr && threadIdx.x == 0) {
    const unsigned w = __hip_atomic_load(&status[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&L.h_status[e], ((unsigned long long)L.h_seq << 32) | (unsigned long long)w,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------------------
// The knots k_halo_knots listed as not converged at the depth of the node tables.  The
// discontinuous HOD integrands (halo.py:1038-1041, 1084-1086) make the reference's Romberg
// run to 2^18..2^20 nodes or exhaust divmax, and the value it stops at is part of the answer
// (DESIGN.md section 2), so scipy's rows and stopping test are kept.  What is NOT kept is the
// evaluation of every node:
//
// k_halo_knots_fast.  A Romberg level only enters through the SUM of the integrand over
// its 2^(L-1) new mid-points.  Away from a handful of break points -- the jump where <N> or
// <N(N-1)> crosses 1, the kink where the satellites switch on at M_0, a step-function central
// occupation -- the integrand is a smooth function of ln nu, sampled by the level-LC grid
// (2^LC + 1 points, one evaluation each) far more finely than it varies.  On every coarse
// interval the new nodes of a deeper level are therefore read off the degree-7 Lagrange
// interpolant through 8 neighbouring coarse samples, and since the level only needs their
// sum, the interpolant is never evaluated: sum_r p(t_r) = sum_m W_m f_(i+m) with weights
// W_m = sum_r l_m(t_r) that depend on the level alone (tabulated by deep_weights_host).
// Coarse intervals that contain a break point (the discrete state of the integrand differs
// at their ends) are evaluated node by node, with scipy's node formula, so that every node
// falls on the same side of a discontinuity as in the reference; stencils never reach across
// such an interval (they shift towards the interior of their smooth segment instead).
// A level of 2^19 nodes then costs ~16 k multiply-adds and <= ~2 k evaluations.
// The scheme checks itself: every odd coarse sample is predicted from its even neighbours
// (the same interpolation at twice the spacing, whose error is 2^8 times larger); if that
// says the level sums could be off by more than kDeepTol relative, or if there are more
// break points than expected, the block falls back to the literal evaluation.
// ---------------------------------------------------------------------------
constexpr int kDeepCoarse = 11;                            // LC: 2049 coarse samples per knot
constexpr int kDeepThreads = 256;                          // threads per listed knot ...
constexpr int kDeepThreadsFew = 512;                       // ... and when the list cannot fill the chip
constexpr int kDeepScratch = romberg_scratch<kDeepThreadsFew / 64, 2>();
constexpr int kDeepStencil = 8;
constexpr int kDeepOffsets = kDeepStencil - 1;             // interval o..o+1 of the stencil
constexpr int kDeepWStride = kDeepOffsets * kDeepStencil;  // weights per level
constexpr int kDeepWLevels = kMaxDivmax - kDeepCoarse;     // levels whose weights a block stages
constexpr int kDeepRound = 3;                              // deep levels summed per pass
constexpr int kDeepMaxRough = 8;                           // break-point intervals
constexpr int kDeepMaxFine = 64;                           // node-by-node intervals
constexpr int kDeepKinkMargin = 16;                        // extra ones above M_0 (see kernel)
constexpr double kDeepTol = 1e-9;
constexpr int kHodCapLevel = 9;                             // see k_halo_knots (hod_cap)

// W[L - LC - 1][o][m] = sum_(r < n) l_m(o + (r + 1/2) / n), n = 2^(L - 1 - LC), with l_m the
// Lagrange basis polynomial on the stencil nodes 0..7.  Host, once per context.
inline void deep_weights_host(int LC, int top, double* w) {
  for (int L = LC + 1; L <= top; ++L) {
    const long n = 1L << (L - 1 - LC);
    double* wl = w + (size_t)(L - LC - 1) * kDeepWStride;
    for (int o = 0; o < kDeepOffsets; ++o)
      for (int m = 0; m < kDeepStencil; ++m) {
        long double den = 1.0L;
        for (int j = 0; j < kDeepStencil; ++j)
          if (j != m) den *= (long double)(m - j);
        long double sum = 0.0L, comp = 0.0L;       // Kahan in long double
        for (long r = 0; r < n; ++r) {
          const long double t = (long double)o + ((long double)r + 0.5L) / (long double)n;
          long double num = 1.0L;
          for (int j = 0; j < kDeepStencil; ++j)
            if (j != m) num *= t - (long double)j;
          const long double y = num / den - comp;
          const long double s2 = sum + y;
          comp = (s2 - sum) - y;
          sum = s2;
        }
        wl[o * kDeepStencil + m] = (double)sum;
      }
  }
}

// One node of a knot's integrand pair together with the discrete state of the integrand
// there (which branch of every `if` of halo.py:1038-1041, 1084-1086 and hod.py:189-230 is
// taken): the integrand is smooth wherever the state does not change.  Same arithmetic as
// IntegrandMM / GM / GG; group 2 has one integrand (out[0] = 0).
__device__ __forceinline__ void halo_eval_coded(int group, const HaloCtx& c, double ln_nu,
                                                double (&out)[2], int* code) {
  const Epoch& E = *c.e;
  const double nu = exp(ln_nu);
  const double lnm = spline_eval(c.nu_knots, c.lnm_pp, c.NM, nu);
  const double y = y_nfw(E, *c.sici, c.ln_k, lnm);
  double nf, b = 0.0;
  mf_node(E, nu, ln_nu, group != 2, &nf, &b);
  if (group == 0) {
    out[0] = nf * b * y * c.window(lnm);
    out[1] = nf * exp(lnm) * y * y;
    *code = 0;
    return;
  }
  const double mass = exp(lnm);
  double n1, n2;
  zheng_node(E, mass, lnm, &n1, &n2);
  int st = (mass - E.hod_M0 > 0.0) ? 2 : 0;                    // satellites on
  if (E.hod_sigma <= 0.0 && lnm * 0.43429448190325182765 > E.hod_log_M_min) st |= 4;
  if (group == 1) {
    out[0] = nf * b * y * n1 / mass * c.window(lnm);
    out[1] = (n1 < 1.0) ? nf * n1 * y : nf * n1 * y * y;
    st |= (n1 < 1.0) ? 1 : 0;
  } else {
    out[0] = 0.0;
    out[1] = (n2 < 1.0) ? nf * n2 * y / mass : nf * n2 * y * y / mass;
    st |= (n2 < 1.0) ? 1 : 0;
  }
  *code = st;
}

// The discrete state alone (halo_eval_coded's code) at a node: everything of the integrand
// that does not depend on k and decides its branch -- no NFW transform, no mass function.
__device__ __forceinline__ int halo_state_at(int group, const HaloCtx& c, double ln_nu) {
  const Epoch& E = *c.e;
  const double nu = exp(ln_nu);
  const double lnm = spline_eval(c.nu_knots, c.lnm_pp, c.NM, nu);
  const double mass = exp(lnm);
  double n1, n2;
  zheng_node(E, mass, lnm, &n1, &n2);
  int st = (mass - E.hod_M0 > 0.0) ? 2 : 0;
  if (E.hod_sigma <= 0.0 && lnm * 0.43429448190325182765 > E.hod_log_M_min) st |= 4;
  st |= ((group == 1 ? n1 : n2) < 1.0) ? 1 : 0;
  return st;
}

// The degree-7 Lagrange interpolant through samples s0 .. s0 + 7 (stencil nodes 0..7) at t, for
// the pair of sample arrays (positions through POS: the arrays' storage order).  t outside
// [0, 7] extrapolates: one-sided continuation of a smooth branch up to a break point.
// l_m(t) = c_m prod_(j < m) (t - j) prod_(j > m) (t - j): the suffix products first, the prefix
// product and the samples as the sum goes (8 doubles live, not 40).
template <class POS>
__device__ __forceinline__ void lagrange8_pair(double t, const double* F0, const double* F1,
                                               int s0, POS pos, double* o0, double* o1) {
  double suf[kDeepStencil];
  suf[kDeepStencil - 1] = 1.0;
#pragma unroll
  for (int m = kDeepStencil - 2; m >= 0; --m) suf[m] = suf[m + 1] * (t - (double)(m + 1));
  // 1 / prod_(j != m) (m - j) = (-1)^(7 - m) / (m! (7 - m)!)
  const double cm[kDeepStencil] = {-1.0 / 5040.0, 1.0 / 720.0, -1.0 / 240.0, 1.0 / 144.0,
                                   -1.0 / 144.0, 1.0 / 240.0, -1.0 / 720.0, 1.0 / 5040.0};
  double a0 = 0.0, a1 = 0.0, pre = 1.0;
#pragma unroll
  for (int m = 0; m < kDeepStencil; ++m) {
    const double l = cm[m] * (pre * suf[m]);
    const int at = pos(s0 + m);
    a0 = fma(l, F0[at], a0);
    a1 = fma(l, F1[at], a1);
    pre *= t - (double)m;
  }
  *o0 = a0;
  *o1 = a1;
}

// Rows 0..top of CHOMP_ROMBERG_C into LDS ((top + 1) * 32 doubles; all threads; a barrier must
// follow before RombergRows2::ctab is used).
__device__ __forceinline__ void romberg_weights_to_lds(double* dst, int top) {
  const double* src = &CHOMP_ROMBERG_C[0][0];
  for (int i = threadIdx.x; i < (top + 1) * 32; i += blockDim.x) dst[i] = src[i];
}

// scipy.integrate.romberg's rows and stopping test on level sums (the replay of
// chomp_romberg.h): every lane of every wavefront holds the same state.
struct RombergRows2 {
  double ordsum[2], Tl[2], prev[2], value[2], range, n, tol, rtol;
  int level[2];
  bool done[2];
  // CHOMP_ROMBERG_C's rows in LDS (romberg_weights_to_lds), or nullptr: read from the constant
  // table -- a dependent global read per row, ~1 us each on a CU that has not seen the line
  const double* ctab = nullptr;
  __device__ __forceinline__ void start(double range_, double tol_, double rtol_, double s0,
                                        double s1, bool want0, bool want1) {
    range = range_; tol = tol_; rtol = rtol_; n = 1.0;
    const int lane = threadIdx.x & 63;
    const double s[2] = {s0, s1};
    const bool want[2] = {want0, want1};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      ordsum[q] = s[q];
      value[q] = range * s[q];
      prev[q] = value[q];
      Tl[q] = lane == 0 ? value[q] : 0.0;
      level[q] = 0;
      done[q] = !want[q];
    }
  }
  __device__ __forceinline__ bool all_done() const { return done[0] && done[1]; }
  // start() and advance(1 .. LC) in one go, from the sums of levels 1..LC (ls0 / ls1[l], LDS):
  // lane l forms T_l from the running sum, the LC extrapolations R[i][i] are taken back to
  // back (they do not depend on each other -- advance()'s loop waits for a butterfly per
  // level), and only the stopping test walks through them in order.  Same operations on the
  // same operands as the loop: identical values, levels and flags.
  template <int LC>
  __device__ __forceinline__ void start_levels(double range_, double tol_, double rtol_, double e0,
                                               double e1, const double* ls0, const double* ls1,
                                               bool want0, bool want1) {
    static_assert(LC < 16, "the rows' sums run over the first sixteen lanes (wave_sum16)");
    range = range_; tol = tol_; rtol = rtol_;
    const int lane = threadIdx.x & 63;
    const double* ls[2] = {ls0, ls1};
    const double send[2] = {e0, e1};
    const bool want[2] = {want0, want1};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      double os = send[q], mine = send[q];
#pragma unroll
      for (int l = 1; l <= LC; ++l) {
        os += ls[q][l];
        if (l <= lane) mine = os;
      }
      const double T = lane <= LC ? ldexp(range * mine, -lane) : 0.0;   // (/ 2^lane)
      double cur[LC + 1];
      cur[0] = range * send[q];
#pragma unroll
      for (int i = 1; i <= LC; ++i) {
        const double c_il = ctab != nullptr ? ctab[i * 32 + (lane & 31)] : CHOMP_ROMBERG_C[i][lane & 31];
        cur[i] = wave_sum16(lane <= i ? c_il * T : 0.0);
      }
      done[q] = !want[q];
      level[q] = 0;
      value[q] = cur[0];
      prev[q] = cur[0];
      int stop = LC;
      if (!done[q]) {
#pragma unroll
        for (int i = 1; i <= LC; ++i) {
          if (!done[q]) {
            const double err = fabs(cur[i] - prev[q]);
            prev[q] = cur[i];
            value[q] = cur[i];
            level[q] = i;
            if (err < tol || err < rtol * fabs(cur[i])) { done[q] = true; stop = i; }
          }
        }
      }
      // (the state advance() would have left: the sums and T_l up to the level reached)
      double upto = send[q];
#pragma unroll
      for (int l = 1; l <= LC; ++l)
        if (l <= stop) upto += ls[q][l];
      ordsum[q] = upto;
      Tl[q] = lane <= stop ? T : 0.0;
    }
    n = (double)(1 << LC);
  }
  // The state out of the registers across a phase that needs them (the node-by-node loop of a
  // deep round): every lane of every wavefront holds the same state but for Tl, which is
  // distributed over the lanes -- wavefront 0 writes, and behind a barrier everyone reads back.
  // (Tl only lives in lanes 0..31: rows <= kMaxDivmax < 32)
  static constexpr int kPark = 2 * 32 + 8;
  __device__ __forceinline__ void park(double* sh) const {
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 32) {
      sh[lane] = Tl[0];
      sh[32 + lane] = Tl[1];
      if (lane == 0) {
        sh[64] = ordsum[0]; sh[65] = ordsum[1]; sh[66] = prev[0]; sh[67] = prev[1];
        sh[68] = value[0]; sh[69] = value[1]; sh[70] = n;
      }
    }
  }
  __device__ __forceinline__ void unpark(const double* sh) {
    const int lane = threadIdx.x & 63;
    Tl[0] = lane < 32 ? sh[lane] : 0.0; Tl[1] = lane < 32 ? sh[32 + lane] : 0.0;
    ordsum[0] = sh[64]; ordsum[1] = sh[65]; prev[0] = sh[66]; prev[1] = sh[67];
    value[0] = sh[68]; value[1] = sh[69]; n = sh[70];
  }
  // level i with the sums of its new nodes
  __device__ __forceinline__ void advance(int i, double s0, double s1) {
    const int lane = threadIdx.x & 63;
    const double c_il = ctab != nullptr ? ctab[i * 32 + (lane & 31)] : CHOMP_ROMBERG_C[i][lane & 31];
    n *= 2.0;
    const double s[2] = {s0, s1};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (done[q]) continue;                                   // block-uniform
      ordsum[q] += s[q];
      const double Ti = ldexp(range * ordsum[q], -i);          // (/ n, n = 2^i: the same bits)
      if (lane == i) Tl[q] = Ti;
      const double cur = wave_sum32(c_il * Tl[q]);
      const double err = fabs(cur - prev[q]);
      prev[q] = cur;
      value[q] = cur;
      level[q] = i;
      if (err < tol || err < rtol * fabs(cur)) done[q] = true;
    }
  }
};

// Literal evaluation of one listed knot (every Romberg node): the checker of the fast path
// and its fallback.  Whole block (64 NW threads); E, S, H staged by the caller.
template <int NW>
__device__ __forceinline__ void deep_literal(const chomp_config& cfg, const HaloCtx& c,
                                             const Epoch& E, int group, bool pa, double* red,
                                             double (&val)[2], int (&lev)[2], bool (&conv)[2]) {
  const double ln_nu_max = log(E.nu_max);
  val[0] = val[1] = 0.0; lev[0] = lev[1] = 0; conv[0] = conv[1] = true;
  if (group == 0) {
    IntegrandMM f{c};
    const RombergOut<2> r = romberg_group<NW, 2>(f, group_lower(E, 0), ln_nu_max,
                                                cfg.global_precision, cfg.halo_precision,
                                                cfg.divmax, red);
    val[0] = r.value[0]; val[1] = r.value[1]; lev[0] = r.level[0]; lev[1] = r.level[1];
    conv[0] = r.converged[0]; conv[1] = r.converged[1];
  } else if (group == 1) {
    IntegrandGM f{c, pa};
    const RombergOut<2> r = romberg_group<NW, 2>(f, group_lower(E, 1), ln_nu_max,
                                                cfg.global_precision, cfg.halo_precision,
                                                cfg.divmax, red);
    val[0] = r.value[0]; val[1] = r.value[1]; lev[0] = r.level[0]; lev[1] = r.level[1];
    conv[0] = r.converged[0]; conv[1] = r.converged[1];
  } else {
    IntegrandGG f{c};
    const RombergOut<1> r = romberg_group<NW, 1>(f, group_lower(E, 2), ln_nu_max,
                                                cfg.global_precision, cfg.halo_precision,
                                                cfg.divmax, red);
    val[1] = r.value[0]; lev[1] = r.level[0]; conv[1] = r.converged[0];
  }
}

// x of coarse node q of the level-LC grid over [a, b], by scipy's formula for the level the
// node first appears in (so a coarse sample is the value the literal evaluation sees).
template <int LC>
__device__ __forceinline__ double deep_coarse_x(double a, double b, int q) {
  constexpr int NC = 1 << LC;
  if (q == 0) return a;
  if (q == NC) return b;
  const int tz = __builtin_ctz((unsigned)q);
  const int lev = LC - tz;
  const long j = (long)(((q >> tz) - 1) >> 1);
  const double h = ldexp(b - a, 1 - lev);
  return (a + 0.5 * h) + h * (double)j;
}

// ---------------------------------------------------------------------------
// The break-point structure of an (epoch, group)'s integrands along ln nu: NOTHING of it depends
// on k -- where the discrete state changes between coarse nodes, which intervals are therefore
// evaluated node by node, the smooth segments between them, and, inside a break-point interval,
// the exact abscissa at which the state changes.  It is built once per (epoch, group) by an
// extra block of k_halo_knots (deep_plan_block), beside the knots' Romberg, and the ~15 listed
// knots of the group read it instead of each deriving it again (7.5 of a knot's 33 us), and
// instead of deciding the state of every node of a break-point interval by evaluating the HOD
// there (half of a deep round's time, and the reason the kernel carried the mass spline).
//   tstar: the nodes of ALL Romberg levels LC + 1 .. divmax inside a coarse interval are the
//   interior points t = 1 .. 2^D - 1 (D = divmax - LC) of its uniform grid of 2^D parts -- level
//   LC + d holds the odd multiples of 2^(D - d).  The state is the interval's left one for
//   t < tstar and its right one from tstar on.  Found by section search with the device's own
//   halo_state_at AT THOSE NODES' ABSCISSAE (scipy's node formula of the node's level): every
//   node the search probed is classified exactly as evaluating it would; between probes
//   monotonicity is assumed (the states are thresholds of monotone functions of the mass:
//   N < 1, M > M_0), which the search verifies at every round (a probe out of order, or in a
//   third state -- two breaks in one interval -- marks the interval "evaluate node by node").
// ---------------------------------------------------------------------------
struct DeepPlan {
  double a, b;                         // the integration range (as the node table has it)
  int tstar[kDeepMaxFine];             // per node-by-node interval (where geom says so)
  int flag;                            // 0: fine; 1: too many break points; 2: too many intervals
  int n_fine, n_seg, n_rough;
  int fine[kDeepMaxFine];              // node-by-node intervals, ascending
  int seg_lo[kDeepMaxRough + 1], seg_hi[kDeepMaxRough + 1];   // smooth segments (node ranges)
  unsigned char states[kDeepMaxFine];  // states of an interval's end nodes (low / high nibble)
  unsigned char geom[kDeepMaxFine];    // 1: both neighbours are smooth segments of >= 8 intervals,
                                       //    the states differ and tstar is valid
};
static_assert(sizeof(DeepPlan) % sizeof(double) == 0, "DeepPlan is copied as doubles");
constexpr int kDeepPlanDoubles = (int)(sizeof(DeepPlan) / sizeof(double));

// LDS of the builder (static, in k_halo_knots).
template <int LC>
struct DeepPlanLds {
  DeepPlan P;
  unsigned char code[((1 << LC) + 1 + 15) & ~15];
  int rough[kDeepMaxRough];
  int n_rough;
  unsigned long long m0[4], m1[4], m2[4];    // per wavefront: probes in the left / right / a third state
};

// Whole block (64 or 256 threads).  E, nu_knots, lnm_pp: the epoch's record and ln M(nu) spline
// in LDS; nd: the (epoch, group) node table, complete (the previous launch wrote it).
template <int LC>
__device__ __forceinline__ void deep_plan_block(const Epoch& E, const double* nu_knots,
                                                const double* lnm_pp, int NM, int group,
                                                const double* __restrict__ nd, int max_rough,
                                                int max_fine, int divmax, DeepPlanLds<LC>& W,
                                                DeepPlan* __restrict__ out) {
  constexpr int NC = 1 << LC;
  const int tid = threadIdx.x, NT = blockDim.x, wv = tid >> 6, nwv = NT >> 6;
  DeepPlan& P = W.P;
  unsigned char* code = W.code;
  const double a = nd[kNodeFields * kNodeCount], b = nd[kNodeFields * kNodeCount + 1];
  // ---- the states of the coarse nodes, in position order
  for (int q = tid; q <= NC; q += NT) {
    int idx = q == NC ? 1 : 0;
    if (q != 0 && q != NC) {
      const int tz = __builtin_ctz((unsigned)q);
      idx = 1 + (1 << (LC - tz - 1)) + (q >> (tz + 1));
    }
    code[q] = (unsigned char)(int)nd[6 * kNodeCount + idx];
  }
  if (tid == 0) { W.n_rough = 0; P.a = a; P.b = b; P.flag = 0; P.n_fine = 0; P.n_seg = 0; }
  __syncthreads();
  // ---- break points: coarse intervals whose ends are in different states
  for (int i = tid; i < NC; i += NT)
    if (code[i] != code[i + 1]) {
      const int at = atomicAdd(&W.n_rough, 1);
      if (at < kDeepMaxRough) W.rough[at] = i;
    }
  __syncthreads();
  const int nr = W.n_rough;
  if (nr > max_rough) {                                        // block-uniform
    if (tid == 0) { P.flag = 1; P.n_rough = nr; }
    __syncthreads();
    copy_doubles(reinterpret_cast<double*>(out), reinterpret_cast<const double*>(&P), kDeepPlanDoubles);
    return;
  }
  if (tid == 0) {
    // node-by-node intervals: the break points, and above one where the satellites switch on
    // a margin in which (M - M_0)^alpha is still too singular to interpolate
    int nf = 0;
    bool over = false;
    // (pp_gg starts AT M_0 when that lies inside the mass range, halo.py:1002-1006: the same
    //  singular onset, with no change of state to announce it)
    // (alpha = 1: N_sat is linear in M - M_0 -- a kink, smooth on either side: no margin)
    const bool singular_onset = E.hod_alpha != 1.0;
    if (singular_onset && group == 2 && E.ln_nu_lo_second > log(E.nu_min))
      for (int d = 0; d <= kDeepKinkMargin; ++d) P.fine[nf++] = d;
    for (int x = 0; x < nr; ++x) {
      const int i = W.rough[x];
      const int span = (((code[i] ^ code[i + 1]) & 2) && singular_onset) ? kDeepKinkMargin : 0;
      for (int d = 0; d <= span && i + d < NC; ++d) {
        if (nf < kDeepMaxFine) P.fine[nf++] = i + d; else over = true;
      }
    }
    for (int x = 1; x < nf; ++x) {                             // insertion sort
      const int v = P.fine[x];
      int y = x - 1;
      while (y >= 0 && P.fine[y] > v) { P.fine[y + 1] = P.fine[y]; --y; }
      P.fine[y + 1] = v;
    }
    int nu = 0;                                                // unique
    for (int x = 0; x < nf; ++x)
      if (x == 0 || P.fine[x] != P.fine[x - 1]) P.fine[nu++] = P.fine[x];
    // smooth segments between them (node ranges); one shorter than a stencil is evaluated
    // node by node as well
    int ns = 0, lo = 0, extra = nu;
    for (int x = 0; x <= nu; ++x) {
      const int hi = x < nu ? P.fine[x] : NC;                  // last node of the segment
      if (hi - lo + 1 >= kDeepStencil) {
        if (ns <= kDeepMaxRough) { P.seg_lo[ns] = lo; P.seg_hi[ns] = hi; ++ns; } else over = true;
      } else {
        for (int i = lo; i < hi; ++i) {
          if (extra < kDeepMaxFine) P.fine[extra++] = i; else over = true;
        }
      }
      lo = hi + 1;
    }
    P.n_rough = nr;
    P.n_seg = ns;
    if (over || extra > max_fine) { P.flag = 2; P.n_fine = 0; } else P.n_fine = extra;
  }
  __syncthreads();
  if (P.flag) {                                                // block-uniform
    copy_doubles(reinterpret_cast<double*>(out), reinterpret_cast<const double*>(&P), kDeepPlanDoubles);
    return;
  }
  const int nf = P.n_fine, ns = P.n_seg;
  if (tid < nf) {
    const int i = P.fine[tid];
    const int sl = code[i] & 15, sr = code[i + 1] & 15;
    P.states[tid] = (unsigned char)(sl | (sr << 4));
    // the geometry a continuation needs: a smooth segment of >= 8 intervals ends at i, another
    // begins at i + 1, and the two end states differ
    int seg_l = -1, seg_r = -1;
    for (int x = 0; x < ns; ++x) {
      if (P.seg_hi[x] == i) seg_l = x;
      if (P.seg_lo[x] == i + 1) seg_r = x;
    }
    bool ok = i > 0 && i < NC - 1 && seg_l >= 0 && seg_r >= 0 && sl != sr;
    if (ok) ok = i - P.seg_lo[seg_l] >= 8 && P.seg_hi[seg_r] - (i + 1) >= 8;
    P.geom[tid] = ok ? 1 : 0;
    P.tstar[tid] = 0;
  }
  __syncthreads();
  // ---- where inside such an interval the state changes: NT-way section search over the
  // interval's 2^D - 1 interior nodes (see DeepPlan), one interval after the other
  HaloCtx c{&E, nullptr, nu_knots, lnm_pp, NM, 0.0, false};
  const int D = divmax - LC;                                   // (1 .. kMaxDivmax - LC)
  for (int x = 0; x < nf; ++x) {
    if (!P.geom[x]) continue;                                  // block-uniform
    const int iv = P.fine[x];
    const int sl = P.states[x] & 15, sr = P.states[x] >> 4;
    int lo = 0, hi = 1 << D;                                   // state(lo) = sl, state(hi) = sr
    bool bad = false;
    while (hi - lo > 1) {
      // probes lo + step, lo + 2 step, ... < hi
      const int span = hi - lo;
      int step = (span + NT) / (NT + 1);
      if (step < 1) step = 1;
      const int tp = lo + step * (tid + 1);
      const bool valid = tp < hi;
      int cls = -1;
      if (valid) {
        // node tp of the interval: level LC + d with 2^(D - d) the largest power of two in tp
        const int tz = __builtin_ctz((unsigned)tp);
        const int lev = LC + D - tz;                           // (tz < D: tp is interior)
        const long j = ((long)iv << (lev - 1 - LC)) + (long)(tp >> (tz + 1));
        const double h = ldexp(b - a, 1 - lev);
        const double px = (a + 0.5 * h) + h * (double)j;
        const int st = halo_state_at(group, c, px);
        cls = st == sl ? 0 : (st == sr ? 1 : 2);
      }
      const unsigned long long b0 = __ballot(cls == 0), b1 = __ballot(cls == 1), b2 = __ballot(cls == 2);
      if ((tid & 63) == 0) { W.m0[wv] = b0; W.m1[wv] = b1; W.m2[wv] = b2; }
      __syncthreads();
      // (every thread: the same scan over the wavefronts' masks)
      int first1 = -1, last0 = -1;
      bool third = false;
      for (int w2 = 0; w2 < nwv; ++w2) {
        const unsigned long long q0 = W.m0[w2], q1 = W.m1[w2], q2 = W.m2[w2];
        third = third || q2 != 0ull;
        if (q1 != 0ull && first1 < 0) first1 = 64 * w2 + __builtin_ctzll(q1);
        if (q0 != 0ull) last0 = 64 * w2 + 63 - __builtin_clzll(q0);
      }
      __syncthreads();                                         // (the masks are re-written next round)
      if (third || (first1 >= 0 && last0 > first1)) { bad = true; break; }
      const int nlo = last0 >= 0 ? lo + step * (last0 + 1) : lo;
      const int nhi = first1 >= 0 ? lo + step * (first1 + 1) : hi;
      lo = nlo;
      hi = nhi;
    }
    if (tid == 0) {
      if (bad) P.geom[x] = 0; else P.tstar[x] = hi;
    }
  }
  __syncthreads();
  copy_doubles(reinterpret_cast<double*>(out), reinterpret_cast<const double*>(&P), kDeepPlanDoubles);
}

// ---------------------------------------------------------------------------
// k_halo_knots: grid (n_epoch, ceil(NK / 4) [+ 1], n_groups), block 256 = four wavefronts,
// each with one knot ln k_i of group groups[blockIdx.z] (0: h_m + pp_mm, 1: h_g + pp_gm,
// 2: pp_gg): the pair's Romberg on the node table, levels <= kNodeLevel.  The first round
// fills the wavefront exactly (romberg_wave6: lane p on node p of the level-6 grid, the
// upper end point from the table halo_nodes_block left), so the four fifths of the knots
// that scipy stops at level 6 or 7 cost one or two evaluations per lane.  Integrals not
// converged at the depth of the node table are listed for k_halo_knots_fast (pending[];
// npend[e] counts an epoch's listed knots on top of its token).  A block only stages the
// Si/Ci tables: everything else it needs is in the (epoch, group) node table.
// With want_nbar the extra y-block of z == 0 does the epoch's n_bar integral
// (halo.py:674-700) beside the knots; with want_plan a further extra y-block of every z builds
// the break-point plan of its (epoch, group) for k_halo_knots_fast (DeepPlan).
// The epochs are the FASTEST grid axis: blocks are dispatched in linear order, a (k, z) grid's
// launch is 900 blocks for 512-768 resident ones, and with the epochs slowest the last epochs'
// deepest knots entered the chip 11-15 us into the launch (tools/dev_knots_stamps2.py).
// KNW = 1 (a batch of a few dozen epochs): four knots to a block up to level 7, then the
// block's knots that go on are walked by all four wavefronts together (below), capped at three
// wavefronts per SIMD (168 registers, 4 spills: every block of a 64-epoch launch but the last
// 128 is resident at once).  37.9 -> 29.6 us per configs[1] launch.
// KNW = 4 (a whole block per knot pair, grid y = NK [+ 1]): for a set-up of one or a few epochs,
// whose launch lasts as long as its slowest knot -- 1, 1, 2, 4, 8 NFW transforms per lane at
// levels 6..10 on one wavefront, 1, 1, 1, 1, 2 on four.
// ---------------------------------------------------------------------------
// KNW = 0: one wavefront per knot pair AND per block (64 threads): a finished knot frees its
// slot for the next block at once, instead of idling beside the one slow knot of its four
// (from ~80 epochs x 50 knots on, and for the two-group HOD set-ups).
template <int KNW>
__global__ __launch_bounds__(KNW == 0 ? 64 : 256, KNW == 1 ? 3 : 1) void k_halo_knots(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs, double* __restrict__ tab,
    const chomp_halo_par* __restrict__ profile, const HodDev* __restrict__ hod,
    const SiCiTab* __restrict__ sici_g, const double* __restrict__ nodes,
    const double* __restrict__ endp, int g0, int g1, int g2, unsigned mask, int want_nbar,
    int* __restrict__ pending, int* __restrict__ npend, unsigned* __restrict__ status,
    int hod_cap, int want_plan, int max_rough, int max_fine, DeepPlan* __restrict__ plans) {
  extern __shared__ __align__(16) double sm[];
  __shared__ SiCiTab S;
  __shared__ Epoch E;              // (the extra blocks only)
  const int NK = L.NK;
  const int e = blockIdx.x, n_epoch = (int)gridDim.x;
  const int kb = KNW == 1 ? (NK + 3) / 4 : NK;     // knot blocks
  KNSTAMP(0, __builtin_amdgcn_s_memrealtime());
  // Blocks are dispatched x (the epochs) fastest, z slowest: the long units of EVERY epoch first
  // -- the n_bar integrals, the highest k (the deepest Romberg), the HOD groups (plan slots 1, 2)
  // before the smooth one -- so that the launch ends with the knots that stop at level 6.
  const int bx = (int)gridDim.y - 1 - (int)blockIdx.y, bz = (int)gridDim.z - 1 - (int)blockIdx.z;
  if (bx >= kb) {                  // ---- the extra blocks: n_bar, the groups' break-point plans
    const int ex = bx - kb;
    const int gx = bz == 0 ? g0 : (bz == 1 ? g1 : g2);
    const bool nbar_block = want_nbar && ex == 0 && bz == 0;
    const bool plan_block = want_plan && ex == (want_nbar ? 1 : 0) && gx >= 0 && gx <= 2;
    if (!nbar_block && !plan_block) return;
    HaloLds H;
    H.stage(L, E, S, epochs, e, tab + (size_t)e * L.stride, profile, hod, sici_g, sm);
    if (nbar_block) {
      HaloCtx c{&E, &S, H.nu_knots, H.lnm_pp, L.NM, 0.0, false};
      IntegrandNbar f{c};
      const double v = romberg1<(KNW == 0 ? 1 : 4)>(f, E.ln_nu_lo_first, log(E.nu_max),
                                                    cfg.global_precision, cfg.halo_precision,
                                                    cfg.divmax, H.rest);
      if (threadIdx.x == 0) tab[(size_t)e * L.stride + L.off_misc] = v;
      return;
    }
    // (what the listed knots of (e, gx) share: see DeepPlan)
    __shared__ DeepPlanLds<kNodeTabLevel> plan_lds;
    deep_plan_block<kNodeTabLevel>(E, H.nu_knots, H.lnm_pp, L.NM, gx,
                                   nodes + ((size_t)e * 3 + gx) * kNodeStride, max_rough, max_fine,
                                   cfg.divmax, plan_lds, plans + (size_t)e * 3 + gx);
    return;
  }
  const int group = bz == 0 ? g0 : (bz == 1 ? g1 : g2);
  if (group < 0 || group > 2) return;
  copy_doubles(reinterpret_cast<double*>(&S), reinterpret_cast<const double*>(sici_g),
               (int)(sizeof(SiCiTab) / sizeof(double)));
  __syncthreads();
  constexpr bool kCoop = KNW == 1;
  // KNW = 1 with the cooperative tail (below): wavefront w of block bx takes knot bx + kb w --
  // the knots that run deep are the highest k, and this way a block holds one of them at most
  const int ik = KNW == 1 ? (kCoop ? bx + kb * (int)(threadIdx.x >> 6) : bx * 4 + (int)(threadIdx.x >> 6)) : bx;
  KNSTAMP(1, __builtin_amdgcn_s_memrealtime());
  const bool have = ik < NK;
  if (!kCoop && !have) return;     // (no barrier below unless kCoop: the wavefronts are independent)
  const double* node = nodes + ((size_t)e * 3 + group) * kNodeStride;
  const double a = node[kNodeFields * kNodeCount], b = node[kNodeFields * kNodeCount + 1];
  const double* ep = endp + ((size_t)e * 3 + group) * 2 * NK + 2 * (have ? ik : 0);
  const double fb[2] = {ep[0], ep[1]};
  double* t = tab + (size_t)e * L.stride;
  const double ln_k0 = node[kNodeFields * kNodeCount + 2], ln_k1 = node[kNodeFields * kNodeCount + 3];
  const double ln_k = linspace_at(ln_k0, ln_k1, NK, have ? ik : 0);          // halo.py:52-54
  const bool exclusion = (mask & kMaskExclusion) != 0;
  NodeIntegrand f{&S, node, KnotK::uniform(ln_k), exclusion};
  // How deep a knot walks the node table here.  The HOD groups' knots that do not converge
  // within the table are listed for k_halo_knots_fast, whose sampling launch evaluates every
  // node of the table for them anyway: what such a knot sums here beyond level hod_cap is
  // done twice (at the default HOD the pairs either stop at levels 8-9 or run to 11..20, so
  // level 10 -- half the table's nodes, on one wavefront -- is evaluated for the listed only).
  const int top = (group > 0 && (mask & kMaskDeepNodes) && hod_cap < kNodeLevel) ? hod_cap : kNodeLevel;
  const int dmax = cfg.divmax < top ? cfg.divmax : top;
  RombergOut<2> r;
  if constexpr (kCoop) {
    // Four knots to a block, one per wavefront, up to level kCoopLevel (four fifths of the
    // knots stop there: one or two NFW transforms per lane).  The tail -- the one knot in ten
    // that goes on to levels 8..10, another 2 + 4 + 8 transforms per lane on its own wavefront,
    // which is what the launch lasted -- is then walked by the WHOLE block, knot after knot
    // (RombergResume from the state the wavefront left: 1 + 1 + 2 transforms per thread).
    constexpr int kCoopLevel = 7;
    __shared__ double co_dump[4][2 * kRombergDump];
    __shared__ double co_val[4][2];
    __shared__ int co_lev[4][2], co_conv[4][2], co_need[4];
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const bool coop = dmax > kCoopLevel;
    r.value[0] = r.value[1] = 0.0;
    r.level[0] = r.level[1] = 0;
    r.converged[0] = r.converged[1] = true;
    if (have) {
      if (dmax >= 6)
        r = romberg_wave6<2>(f, a, b, fb, cfg.global_precision, cfg.halo_precision,
                             coop ? kCoopLevel : dmax, co_dump[wave]);
      else
        r = romberg_group<1, 2>(f, a, b, cfg.global_precision, cfg.halo_precision, dmax, nullptr);
    }
    if (lane == 0) {
      co_need[wave] = (have && coop && !(r.converged[0] && r.converged[1])) ? 1 : 0;
      co_val[wave][0] = r.value[0]; co_val[wave][1] = r.value[1];
      co_lev[wave][0] = r.level[0]; co_lev[wave][1] = r.level[1];
      co_conv[wave][0] = r.converged[0] ? 1 : 0; co_conv[wave][1] = r.converged[1] ? 1 : 0;
    }
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
      if (!co_need[w]) continue;                 // (block-uniform)
      const int ikw = bx + kb * w;
      NodeIntegrand fw{&S, node, KnotK::uniform(linspace_at(ln_k0, ln_k1, NK, ikw)), exclusion};
      RombergResume R[2];
      bool dn[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        R[q].load(co_dump[w] + q * kRombergDump, kCoopLevel, b - a, cfg.global_precision,
                  cfg.halo_precision);
        R[q].value = co_val[w][q];
        R[q].level = co_lev[w][q];
        dn[q] = co_conv[w][q] != 0;
      }
      bool all = dn[0] && dn[1];
      int flip = 0;
      for (int i = kCoopLevel + 1; i <= dmax && !all; ++i) {
        const double c_il = CHOMP_ROMBERG_C[i][lane & 31];
        const long numtosum = 1L << (i - 1);
        const double h = ldexp(b - a, 1 - i);                  // ((b - a) / numtosum)
        const double lox = a + 0.5 * h;
        double part[2] = {0.0, 0.0};
        for (long j = threadIdx.x; j < numtosum; j += 256) {
          double v[2];
          fw(lox + h * (double)j, v, i, j);
          part[0] += v[0];
          part[1] += v[1];
        }
        all = true;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const double Sq = group_sum<4>(part[q], sm, flip);
          if (!dn[q]) {
            R[q].advance(i, Sq, c_il);
            dn[q] = R[q].done;
          }
          all = all && dn[q];
        }
      }
      __syncthreads();                            // (sm: the last sums have been read)
      if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          co_val[w][q] = R[q].value;
          co_lev[w][q] = R[q].level;
          co_conv[w][q] = dn[q] ? 1 : 0;
        }
      }
    }
    __syncthreads();
    if (!have) return;
    r.value[0] = co_val[wave][0]; r.value[1] = co_val[wave][1];
    r.level[0] = co_lev[wave][0]; r.level[1] = co_lev[wave][1];
    r.converged[0] = co_conv[wave][0] != 0; r.converged[1] = co_conv[wave][1] != 0;
  } else if constexpr (KNW <= 1) {
    if (dmax >= 6)
      r = romberg_wave6<2>(f, a, b, fb, cfg.global_precision, cfg.halo_precision, dmax);
    else
      r = romberg_group<1, 2>(f, a, b, cfg.global_precision, cfg.halo_precision, dmax, nullptr);
  } else {
    (void)fb;
    r = romberg_group<KNW, 2>(f, a, b, cfg.global_precision, cfg.halo_precision, dmax, sm);
  }
  KNSTAMP(2, __builtin_amdgcn_s_memrealtime());
  KNSTAMP(3, (r.level[0] > r.level[1] ? r.level[0] : r.level[1]) + 100 * ik);
  if ((KNW <= 1 ? (threadIdx.x & 63) : threadIdx.x) == 0) {
    double* lev = t + L.off_levels;
    const int fa = group_fa(group), fb_ = group_fb(group);
    const bool more = cfg.divmax > top;
    if (group != 2 && (mask & (1u << fa))) {
      t[L.off_knot[fa] + ik] = r.value[0];
      lev[fa * NK + ik] = (!r.converged[0] && more) ? kPendingLevel : (double)r.level[0];
    }
    bool any = false, need_a = false, need_b = false;
    if (group != 2 && (mask & (1u << fa))) need_a = !r.converged[0] && more;
    if (mask & (1u << fb_)) {
      t[L.off_knot[fb_] + ik] = r.value[1];
      lev[fb_ * NK + ik] = (!r.converged[1] && more) ? kPendingLevel : (double)r.level[1];
      need_b = !r.converged[1] && more;
    }
    any = need_a || need_b;
    // divmax within the node tables: scipy returns the last row with an AccuracyWarning
    if (!more) {
      unsigned st = 0u;
      if (group != 2 && (mask & (1u << fa)) && !r.converged[0]) st |= kStHaloDivmax0 << fa;
      if ((mask & (1u << fb_)) && !r.converged[1]) st |= kStHaloDivmax0 << fb_;
      if (st) atomicOr(&status[e], st);
    }
    // work list of k_halo_knots_fast (layout at kPendingHead): the deepest-running knots --
    // the highest k -- at the front, the rest from the back of the buffer downwards
    if (any) {
      atomicAdd(&npend[e], 1);
      // (which of the pair is still open rides in the item: the consumer then need not read
      //  the levels table -- a dependent round trip at the front of every listed knot)
      const int item = (int)((bz * n_epoch + e) * NK + ik) | (need_a ? kItemOpenA : 0) |
                       (need_b ? kItemOpenB : 0);
      const int cap = 3 * n_epoch * NK;
      if (4 * ik >= 3 * NK) pending[kPendingHead + atomicAdd(&pending[0], 1)] = item;
      else pending[kPendingHead + cap - 1 - atomicAdd(&pending[2], 1)] = item;
    }
  }
}

// ---------------------------------------------------------------------------
// The coarse samples of the listed knots: one slot of the context's sample buffer per listed
// knot (list position -> slot), filled by k_halo_knots_samples -- a launch of its own over ALL
// listed knots, so that the 2^LC + 1 NFW transforms of a knot are chip time (a thread each)
// and not 2 x 20 us on the critical path of the block that then sums the knot's levels.
// Layout of a slot (doubles): F0[kDeepF], F1[kDeepF].
// A sample array holds the even-numbered samples first and the odd ones from kDeepOdd on
// (deep_pos): every phase of k_halo_knots_fast then reads LDS at unit stride -- the self-check
// predicts every odd sample from the even ones (stride-2 doubles: a two-way bank conflict by
// construction in position order), and a level's new nodes are the odd samples of the level
// below.  kDeepOdd = 16 mod 32: the two halves of a run of consecutive samples fall into
// different banks.
// ---------------------------------------------------------------------------
constexpr int kDeepOdd = 1040;
constexpr int kDeepF = kDeepOdd + (1 << (kDeepCoarse - 1));      // 2064
constexpr int kDeepSlot = 2 * kDeepF;                            // (33 024 bytes: 516 lines of 64)
constexpr int kDeepPsum = 24;                                    // per (slot, chunk): levels 11..1 x 2, padded
__host__ __device__ inline int deep_pos(int q) { return (q & 1) ? kDeepOdd + (q >> 1) : (q >> 1); }
// LDS doubles in front of a block's sample arrays (the epoch's splines, the reductions'
// scratch), rounded to 16 bytes: the arrays are copied as double2
__host__ __device__ inline int deep_f_off(int NM) { return (NM + 8 * (NM - 1) + kDeepScratch + 1) & ~1; }

// grid: any (blocks stride over the work items), block 256.  The node table is level-major, and
// so is the work: chunk c = 0..7 of a knot is table nodes 256 c + 1 .. 256 c + 256 -- levels
// 1..8 (and the lower end point, node 0, in place of node 1, the upper one, which comes from
// d_endp: the node-table stage evaluated it for every knot), level 9, level 10 (two chunks),
// level 11 (four) -- thread t on node 256 c + 1 + t: the nine fields of a node are read at
// unit stride across the wavefront (in position order they were gathers of runs of 32, 16,
// 8 ... nodes: 50 us per configs[2] launch, most of it waiting), and the stores of a slot are
// contiguous for level 11 -- the odd half of deep_pos order -- and strided by 2, 4 ... for the
// levels below.  Work item = (listed knot, part): part p of `parts` (1, 2, 4, 8) is chunks
// [8 p / parts, 8 (p + 1) / parts).  A chunk's samples belong to one Romberg level (chunk 0:
// wavefronts 2-3 level 8, wavefront 1 level 7, wavefront 0 by lane ranges), so the per-level
// sums scipy's rows need are taken here, from the registers: row (slot * 8 + chunk) of psum,
// entry 2 (LC - level) + f, zero for the levels a chunk does not hold.
constexpr int kDeepChunks = 8;
template <int LC>
__global__ __launch_bounds__(256) void k_halo_knots_samples(
    chomp_config cfg, TabLayout L, const SiCiTab* __restrict__ sici_g, int g0, int g1, int g2,
    unsigned mask, int n_epoch, const int* __restrict__ pending, const double* __restrict__ nodes,
    const double* __restrict__ endp, double* __restrict__ samples, double* __restrict__ psum,
    int parts, int slot_lo, int slot_hi) {
  static_assert(LC == 11, "the chunk -> level map below is the level-11 table's");
  constexpr int NC = 1 << LC;
  __shared__ SiCiTab S;
  __shared__ double xs[4][2];
  __shared__ double w0[2][64];
  const int NK = L.NK, tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int count_front = pending[0], count = count_front + pending[2];
  const int hi = count < slot_hi ? count : slot_hi;
  const int n_work = (hi - slot_lo) * parts;
  if ((int)blockIdx.x >= n_work) return;
  copy_doubles(reinterpret_cast<double*>(&S), reinterpret_cast<const double*>(sici_g),
               (int)(sizeof(SiCiTab) / sizeof(double)));
  __syncthreads();
  const bool exclusion = (mask & kMaskExclusion) != 0;
  const int per = kDeepChunks / parts;
  for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
    const int li = slot_lo + w / parts, part = w % parts;
    const int item = kItemIndex & (li < count_front
                                       ? pending[kPendingHead + li]
                                       : pending[kPendingHead + 3 * n_epoch * NK - 1 - (li - count_front)]);
    const int ik = item % NK, e = (item / NK) % n_epoch, zg = item / (NK * n_epoch);
    const int group = zg == 0 ? g0 : (zg == 1 ? g1 : g2);
    if (group < 0 || group > 2) continue;          // (never listed)
    const double* nd = nodes + ((size_t)e * 3 + group) * kNodeStride;
    const KnotK kk = KnotK::uniform(linspace_at(log(cfg.k_min), log(cfg.k_max), NK, ik));
    double* slot = samples + (size_t)(li - slot_lo) * kDeepSlot;
    for (int c = part * per; c < (part + 1) * per; ++c) {
      // (chunk 0, thread 0: node 0 -- the lower end point -- instead of node 1)
      const int idx = (c == 0 && tid == 0) ? 0 : 256 * c + 1 + tid;
      double o[2];
      node_pair(S, kk, exclusion, nd[idx], nd[kNodeCount + idx], nd[2 * kNodeCount + idx],
                nd[3 * kNodeCount + idx], nd[4 * kNodeCount + idx], nd[5 * kNodeCount + idx],
                nd[6 * kNodeCount + idx], nd[7 * kNodeCount + idx], nd[8 * kNodeCount + idx], o);
      int q = 0;                                   // level-major -> position
      if (idx >= 2) {
        const int m = idx - 1;
        const int lv = 32 - __builtin_clz((unsigned)m);
        q = (2 * (m - (1 << (lv - 1))) + 1) << (LC - lv);
      }
      const int at = deep_pos(q);
      slot[at] = o[0];
      slot[kDeepF + at] = o[1];
      if (c == 0 && tid == 0) {                    // the upper end point (node 1 of the table)
        const double* ep = endp + ((size_t)e * 3 + group) * 2 * NK + 2 * ik;
        slot[deep_pos(NC)] = ep[0];
        slot[kDeepF + deep_pos(NC)] = ep[1];
        o[0] = 0.0;                                // (an end point: in no level's sum)
        o[1] = 0.0;
      }
      const double x0 = wave_sum(o[0]), x1 = wave_sum(o[1]);
      __syncthreads();                             // (xs, w0: the previous chunk's were read)
      if (ln == 0) { xs[wv][0] = x0; xs[wv][1] = x1; }
      if (c == 0 && wv == 0) { w0[0][ln] = o[0]; w0[1][ln] = o[1]; }
      __syncthreads();
      if (tid < 24) {
        const int f = tid & 1, l = LC - (tid >> 1);
        double v = 0.0;
        if (c >= 1) {
          const int lc = c == 1 ? 9 : (c <= 3 ? 10 : 11);
          if (l == lc) v = (xs[0][f] + xs[1][f]) + (xs[2][f] + xs[3][f]);
        } else if (l == 8) {
          v = xs[2][f] + xs[3][f];
        } else if (l == 7) {
          v = xs[1][f];
        } else if (l >= 1 && l <= 6) {             // threads 2^(l - 1) .. 2^l - 1 of wavefront 0
          for (int t = 1 << (l - 1); t < (1 << l); ++t) v += w0[f][t];
        }
        psum[((size_t)(li - slot_lo) * kDeepChunks + c) * kDeepPsum + tid] = v;
      }
    }
  }
}

// Dynamic LDS of k_halo_knots_fast (bytes).
// (the deep levels' weights and the Romberg rows' for the context's divmax, not for the largest
//  one allowed: 7 KB of 59 at the default 20 -- what lets a CU hold three blocks)
__host__ __device__ inline int deep_w_levels(int divmax, int LC) {
  const int n = divmax - LC;
  return n < 0 ? 0 : (n > kDeepWLevels ? kDeepWLevels : n);
}
template <int LC>
inline size_t deep_fast_lds(int NM, int divmax) {
  const size_t deep = (size_t)(deep_f_off(NM) + 2 * kDeepF +
                               deep_w_levels(divmax, LC) * kDeepWStride + (divmax + 1) * 32) *
                          sizeof(double);
  return deep;                     // (> finalize_lds_doubles(NK) for any NK <= 512 at LC >= 11)
}
// ... of k_halo_knots_literal.
inline size_t deep_literal_lds(int NM, int NK) {
  size_t d = (size_t)(NM + 8 * (NM - 1) + kDeepScratch);
  if (d < (size_t)finalize_lds_doubles(NK)) d = (size_t)finalize_lds_doubles(NK);
  return d * sizeof(double);
}

// Development stamps (tools/dev_knot_stamps.py builds with -DCHOMP_STAMPS; absent from the
// product build): s_memtime at the phase boundaries of the first knot each block draws.
#ifdef CHOMP_STAMPS
constexpr int kStampBlocks = 2048, kStampSlots = 24;
__device__ long long g_ks[kStampBlocks * kStampSlots];
#define KSTAMP(k)                                                                     \
  do {                                                                                \
    if (first_item && threadIdx.x == 0 && blockIdx.x < kStampBlocks && !from_eval)    \
      g_ks[blockIdx.x * kStampSlots + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#define KSTAMP_VALUE(k, v)                                                            \
  do {                                                                                \
    if (first_item && threadIdx.x == 0 && blockIdx.x < kStampBlocks && !from_eval)    \
      g_ks[blockIdx.x * kStampSlots + (k)] = (long long)(v);                          \
  } while (0)
#define KSTAMP_BLOCK(k, v)                                                            \
  do {                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < kStampBlocks && !from_eval)                  \
      g_ks[blockIdx.x * kStampSlots + (k)] = (long long)(v);                          \
  } while (0)
#else
#define KSTAMP_BLOCK(k, v) do { } while (0)
#define KSTAMP(k) do { } while (0)
#define KSTAMP_VALUE(k, v) do { } while (0)
#endif

// One more arrival at epoch e (its token, or one of its listed knots done) by a whole block:
// whoever brings npend[e] to zero finalises the epoch.  fences: knots were written by other
// blocks of this or the previous launch's kernels since the counter was armed.
// wrote: what THIS block has written that the finalising block will read -- kArriveNothing
// (the token), kArrivePlain (plain stores: a fence, i.e. a write-back of the XCD's L2, in front
// of the count), kArriveThrough (thread 0 stored its results with agent scope, knot_store():
// written through, and only their completion is waited for -- the fence was 2 us of every
// listed knot's chain).
enum { kArriveNothing = 0, kArrivePlain = 1, kArriveThrough = 2 };
__device__ __forceinline__ void knot_store(double* q, double v) {
  __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void deep_arrive(const chomp_config& cfg, const TabLayout& L,
                                            Epoch* __restrict__ epochs_rw, double* __restrict__ tab,
                                            int e, unsigned fam_mask, unsigned* __restrict__ status,
                                            int* __restrict__ npend, bool fences, int* last_sh,
                                            double* sm, int wrote = kArrivePlain) {
  __syncthreads();                 // (the block's results are written)
  if (threadIdx.x == 0) {
    if (fences && wrote == kArrivePlain) __threadfence();   // ... and visible before the count moves
    if (fences && wrote != kArriveNothing) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *last_sh = atomicSub(&npend[e], 1) == 1 ? 1 : 0;
  }
  __syncthreads();
  if (*last_sh) {                  // block-uniform
    // (the knots of the other blocks: agent-scope loads, see halo_finalize_block; everything
    //  else it reads is the previous launch's)
    halo_finalize_block(cfg, L, epochs_rw, tab, e, fam_mask, status, sm, fences);
  }
}

// grid >= n_epoch (blocks draw knots from the list k_halo_knots left), block NT.  An epoch's
// set-up ends with halo_finalize_block once all its knots are final: npend[e] counts its
// listed knots plus one token, which block e takes first -- an epoch with nothing listed
// (every P_mm epoch at the default precision) is finalised right there, otherwise by the
// block that completes its last knot, here or in k_halo_knots_literal.  (The lists are cleared
// by the next set-up's node-table stage, halo_epoch_begin: blocks of this launch may still be
// polling their heads.)
// deepw: deep_weights_host(LC, divmax).  A knot the scheme cannot do -- more break points than
// max_rough, more node-by-node intervals than max_fine, a self-check estimate above tol,
// coarse samples missing from the node table, or all_literal (chomp_set_tuning
// CHOMP_TUNE_DEEP_LITERAL: the checker) -- is NOT evaluated here: it is handed on, still
// counted in npend[e], to the second list, which k_halo_knots_literal works off behind this
// launch.  (The literal evaluation inlined here cost every knot of the fast path its
// registers: 256 VGPRs + 652 bytes of scratch per lane.)
// stats (optional): [0] knots done by the fast path, [1] by the literal one; why literal:
// [2] too many break points, [3] too many node-by-node intervals, [4] self-check; [5] largest
// self-check estimate seen (float bits).
// NT: kDeepThreads, or kDeepThreadsFew when there are at most about two knots per CU to do (one
// epoch): the launch then lasts as long as one knot, and a knot's phases are spread wider.
// SELF: the set-up has no HOD integrands (P_mm, linear): nothing is ever listed at the default
// precision, the launch is the epochs' finalisation -- and a listed knot (tightened
// halo_precision) can only be of the smooth group 0, which this instance evaluates literally
// right here, so that the headline chain does not carry an always-empty hand-over launch.
// EVAL: whether the instance can evaluate the integrand at a node that is not on the coarse
// grid -- the nodes of a break-point interval whose value cannot be read off a neighbouring
// branch's continuation, the margin above a singular satellite onset (alpha != 1).  The
// integrand (NFW transform, mass function, HOD moments, fully inlined) is what takes this
// kernel to 256 VGPRs and into scratch; with the default kind of HOD (alpha = 1) a knot of a
// thousand needs it.  The host launches the EVAL instance when some epoch's HOD has
// alpha != 1; otherwise the lean one, which hands a knot that turns out to need an
// evaluation (stats[6]: two break points closer than a stencil, ~7 % of configs[2]'s listed
// knots) on to a second launch of the EVAL instance behind it (from_eval: that launch draws
// the list positions the lean one left at pending_eval_base(); the knots' samples are still
// in their slots).
template <int NT>
__device__ __forceinline__ void deep_literal_loop(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs, double* __restrict__ tab,
    const chomp_halo_par* __restrict__ profile, const HodDev* __restrict__ hod,
    const SiCiTab* __restrict__ sici_g, int g0, int g1, int g2, unsigned mask, int n_epoch,
    int* __restrict__ pending, int* __restrict__ npend, Epoch* __restrict__ epochs_rw,
    unsigned fam_mask, unsigned* __restrict__ status, int* __restrict__ stats);

template <int LC, int NT, bool SELF, bool EVAL>
__device__ __forceinline__ void deep_fast_body(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs, double* __restrict__ tab,
    const SiCiTab* __restrict__ sici_g, int g0, int g1, int g2, unsigned mask, int n_epoch,
    int* __restrict__ pending, int* __restrict__ npend, Epoch* __restrict__ epochs_rw,
    unsigned fam_mask, unsigned* __restrict__ status, const double* __restrict__ deepw,
    int all_literal, double tol, int max_rough, int max_fine, int* __restrict__ stats,
    const double* __restrict__ samples, const double* __restrict__ psum, int parts,
    int round, int slot_lo, int slot_hi, int from_eval, const DeepPlan* __restrict__ plans) {
  static_assert(LC == kNodeTabLevel, "the coarse samples are the node table's grid");
  constexpr int NC = 1 << LC;
  constexpr int NWV = NT / 64;
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  // (the Si/Ci tables only where the instance evaluates the NFW transform)
  constexpr bool kNeedsSici = SELF || EVAL;
  __shared__ std::conditional_t<kNeedsSici, SiCiTab, double> S_store;
  SiCiTab& S = *reinterpret_cast<SiCiTab*>(&S_store);
  __shared__ int item_sh, last_sh, bail_sh;
  __shared__ DeepPlan PL;                          // the (epoch, group)'s break-point plan
  // whether a node-by-node interval's nodes may be read off the one-sided continuations of the
  // neighbouring smooth segments (the plan's geometry and this knot's eighth differences)
  __shared__ unsigned char fine_poly[kDeepMaxFine];
  // per smooth segment [lo, hi]: the sum of the samples lo .. hi - 7 (see the deep rounds)
  __shared__ double seg_sum[2][kDeepMaxRough + 1], seg_slot[NWV][2][kDeepMaxRough + 1];
  __shared__ double rows_park[2][RombergRows2::kPark];
  const int NK = L.NK;
  const int tid = threadIdx.x;
  const int wv = tid >> 6, ln = tid & 63;
  KSTAMP_BLOCK(18, __builtin_amdgcn_s_memtime());
  const int count_front = pending[0], count = count_front + pending[2];
  // (with an empty list no block of this launch writes a knot: every value the finalisation
  //  reads comes from the previous launch, and no fence is needed)
  const bool fences = count != 0;
  if (round == 0 && !from_eval && (int)blockIdx.x < n_epoch) {
    if (count == 0) {
      // (nothing listed anywhere: every epoch's counter stands at its token, and block e
      //  finalises epoch e without the atomic's round trip -- the headline chain's last launch
      //  is nothing but this)
      halo_finalize_block(cfg, L, epochs_rw, tab, (int)blockIdx.x, fam_mask, status, sm, false);
    } else {
      deep_arrive(cfg, L, epochs_rw, tab, (int)blockIdx.x, fam_mask, status, npend, fences, &last_sh, sm,
                  kArriveNothing);
    }
  }
  // this round's share of the list: the knots whose samples are in the buffer's slots now
  const int item_hi = count < slot_hi ? count : slot_hi;
  if (item_hi <= slot_lo) return;  // nothing listed (for this round): no traffic on the queue head
  int* lit_items = pending + pending_literal_base((size_t)n_epoch, NK);
  int* ev_items = pending + pending_eval_base((size_t)n_epoch, NK) + slot_lo;
  const int ev_count = from_eval ? pending[kPendingEvCount + round] : 0;
  if (from_eval && ev_count == 0) return;
  // what does not depend on the knot, once per block: Si/Ci tables; the deep levels' weights
  // and the Romberg rows' weights where the finalisation of an epoch cannot reach them
  double* const w_all = sm + deep_f_off(L.NM) + 2 * kDeepF;
  const bool w_safe = finalize_lds_doubles(NK) <= deep_f_off(L.NM) + 2 * kDeepF;
  auto stage_weights = [&]() {
    const int nlev = deep_w_levels(cfg.divmax, LC);
    copy_doubles(w_all, deepw, nlev * kDeepWStride);
    romberg_weights_to_lds(w_all + nlev * kDeepWStride, cfg.divmax);   // (the rows' weights)
  };
  if constexpr (kNeedsSici)
    copy_doubles(reinterpret_cast<double*>(&S), reinterpret_cast<const double*>(sici_g),
                 (int)(sizeof(SiCiTab) / sizeof(double)));
  stage_weights();
#ifdef CHOMP_STAMPS
  bool first_item = true;
  int n_items = 0;
#endif
  // (drawing the next knot while the current one is worked on -- the atomic's round trip off the
  //  chain -- was measured and is worse: the blocks that start first, on the deepest knots, then
  //  also hold the first of the knots left over, 100 against 88 us per configs[2] launch; and a
  //  block's FIRST draw issued in front of its staging, to overlap the two: C3 0.317 -> 0.325 ms)
  // (a block's FIRST knot is its own number in the list -- no draw: the atomic's round trip,
  //  ~2.5 us, off the front of every block's chain, and the list's order, deepest knots first,
  //  is the dispatch order of the blocks anyway; the knots beyond the grid are drawn)
  bool own_first = !from_eval;
  for (;;) {
#ifdef CHOMP_STAMPS
    first_item = (n_items++ == 0);
#endif
    __syncthreads();               // (previous item done with E, sm)
    if (tid == 0) {
      if (from_eval) {
        const int d = atomicAdd(&pending[kPendingEvDraw + round], 1);
        item_sh = d < ev_count ? ev_items[d] : item_hi;
      } else if (own_first) {
        item_sh = slot_lo + (int)blockIdx.x;
      } else {
        item_sh = slot_lo + (int)gridDim.x + atomicAdd(&pending[kPendingDraw + round], 1);
      }
    }
    own_first = false;
    __syncthreads();
    if (item_sh >= item_hi) {      // block-uniform
      KSTAMP_BLOCK(19, __builtin_amdgcn_s_memtime());
#ifdef CHOMP_STAMPS
      KSTAMP_BLOCK(20, n_items - 1);
#endif
      return;
    }
    const int item_raw = item_sh < count_front
                             ? pending[kPendingHead + item_sh]
                             : pending[kPendingHead + 3 * n_epoch * NK - 1 - (item_sh - count_front)];
    const int item = item_raw & kItemIndex;
    const int ik = item % NK, e = (item / NK) % n_epoch, zg = item / (NK * n_epoch);
    const int group = zg == 0 ? g0 : (zg == 1 ? g1 : g2);
    double* t = tab + (size_t)e * L.stride;
    double* levs = t + L.off_levels;
    const int fa = group_fa(group < 0 ? 0 : group), fb = group_fb(group < 0 ? 0 : group);
    // (which of the pair is open: from the item's flags -- the lister's levels[] == pending marks)
    const bool pa = group >= 0 && group <= 2 && group != 2 && (mask & (1u << fa)) &&
                    (item_raw & kItemOpenA) != 0;
    const bool pb = group >= 0 && group <= 2 && (mask & (1u << fb)) && (item_raw & kItemOpenB) != 0;
    if (!pa && !pb) {              // (never listed; keep the count right)
      deep_arrive(cfg, L, epochs_rw, tab, e, fam_mask, status, npend, fences, &last_sh, sm,
                  kArriveNothing);
      if (!w_safe) { __syncthreads(); stage_weights(); }
      continue;
    }
    if constexpr (SELF) {
      // (group 0 only; the block-wide literal Romberg on the smooth h_m / pp_mm pair)
      double* nu_knots = sm;
      double* lnm_pp = nu_knots + L.NM;
      copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
                   kEpochDoubles);
      copy_doubles(nu_knots, t + L.off_nu, L.NM);
      copy_doubles(lnm_pp, t + L.off_lnm_pp, 4 * (L.NM - 1));
      __syncthreads();
      HaloCtx c{&E, &S, nu_knots, lnm_pp, L.NM,
                linspace_at(log(cfg.k_min), log(cfg.k_max), NK, ik), (mask & kMaskExclusion) != 0};
      IntegrandMM f{c};
      const RombergOut<2> r = romberg_group<NWV, 2>(f, group_lower(E, 0), log(E.nu_max),
                                                   cfg.global_precision, cfg.halo_precision,
                                                   cfg.divmax, sm + L.NM + 8 * (L.NM - 1));
      if (tid == 0) {
        if (pa) { t[L.off_knot[fa] + ik] = r.value[0]; levs[fa * NK + ik] = (double)r.level[0]; }
        if (pb) { t[L.off_knot[fb] + ik] = r.value[1]; levs[fb * NK + ik] = (double)r.level[1]; }
        unsigned st = 0u;
        if (pa && !r.converged[0]) st |= kStHaloDivmax0 << fa;
        if (pb && !r.converged[1]) st |= kStHaloDivmax0 << fb;
        if (st) atomicOr(&status[e], st);
        if (stats) atomicAdd(&stats[1], 1);
      }
      deep_arrive(cfg, L, epochs_rw, tab, e, fam_mask, status, npend, fences, &last_sh, sm);
      if (!w_safe) { __syncthreads(); stage_weights(); }
      continue;
    }
    // hand the knot on where this scheme does not apply at all
    if (all_literal || cfg.divmax <= LC || !(mask & kMaskDeepNodes)) {
      if (tid == 0) lit_items[atomicAdd(&pending[4], 1)] = item;
      continue;
    }
    KSTAMP(0);
    // (the sampling launch's level sums, its chunks in order: on their way while the samples are copied)
    double lsum_pre = 0.0;
    if (tid < 2 * (LC + 1) && tid % (LC + 1) >= 1) {
      const int f = tid / (LC + 1), l = tid % (LC + 1);
      const double* ps = psum + (size_t)(item_sh - slot_lo) * kDeepChunks * kDeepPsum + 2 * (LC - l) + f;
      for (int pt = 0; pt < kDeepChunks; ++pt) lsum_pre += ps[pt * kDeepPsum];
    }
    // ---- the (epoch, group)'s break-point plan (k_halo_knots' extra block built it)
    copy_doubles(reinterpret_cast<double*>(&PL),
                 reinterpret_cast<const double*>(plans + (size_t)e * 3 + group), kDeepPlanDoubles);
    double* nu_knots = sm;
    double* lnm_pp = nu_knots + L.NM;
    if constexpr (EVAL) {            // (the integrand's tables: only where nodes are evaluated)
      copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
                   kEpochDoubles);
      copy_doubles(nu_knots, t + L.off_nu, L.NM);
      copy_doubles(lnm_pp, t + L.off_lnm_pp, 4 * (L.NM - 1));
    }
    if (tid == 0) bail_sh = 0;
    // ---- the knot's coarse samples: k_halo_knots_samples left them in slot (list position)
    // of the sample buffer, already in deep_pos order, with the sums of levels LC - 7 .. LC
    // beside them: a contiguous copy into LDS
    double* red = sm + L.NM + 8 * (L.NM - 1);
    double* F0 = sm + deep_f_off(L.NM);
    double* F1 = F0 + kDeepF;                              // (both in deep_pos order)
    double* W = F1 + kDeepF;                               // [divmax - LC][kDeepWStride]
    const double* Ctab = W + deep_w_levels(cfg.divmax, LC) * kDeepWStride;   // [divmax + 1][32]
    {
      const double2* src = reinterpret_cast<const double2*>(samples + (size_t)(item_sh - slot_lo) * kDeepSlot);
      double2* dst = reinterpret_cast<double2*>(F0);
      for (int i = tid; i < kDeepF; i += NT) dst[i] = src[i];              // F0 and F1
    }
    __syncthreads();
    KSTAMP(1);
    KSTAMP(2);
    if (PL.flag != 0) {              // block-uniform: too many break points / intervals
      if (tid == 0) {
        lit_items[atomicAdd(&pending[4], 1)] = item;
        if (stats) atomicAdd(&stats[PL.flag == 1 ? 2 : 3], 1);
      }
      continue;
    }
    const double a = PL.a, b = PL.b;
    HaloCtx c{&E, &S, nu_knots, lnm_pp, L.NM,
              linspace_at(log(cfg.k_min), log(cfg.k_max), NK, ik), (mask & kMaskExclusion) != 0};
    bool literal = false, to_eval = false;
    int flip = 0;
    RombergRows2 R;
    R.ctab = Ctab;
    {
      // ---- levels 0..LC: their sums (the sampling launch's parts in order; the lowest three
      // from the samples), then scipy's rows -- all LC extrapolations at once (they do not
      // depend on each other: only the stopping test walks through them in order)
      {
        double* lsum = red + 2 * NWV;                          // [2][LC + 1] (behind group_sum's slots)
        if (tid < 2 * (LC + 1)) {
          const int f = tid / (LC + 1), l = tid % (LC + 1);
          const double v = lsum_pre;
          lsum[f * (LC + 1) + l] = v;
        }
        __syncthreads();
        R.start_levels<LC>(b - a, cfg.global_precision, cfg.halo_precision,
                           0.5 * (F0[0] + F0[deep_pos(NC)]), 0.5 * (F1[0] + F1[deep_pos(NC)]),
                           lsum, lsum + LC + 1, pa, pb);
      }
      KSTAMP(3);
      const int nf_all = PL.n_fine, ns = PL.n_seg;
      if (!R.all_done()) {
        // ---- a break-point interval between two smooth segments: on either side of the
        // break the integrand is the smooth continuation of its neighbour's branch (each
        // `if` of halo.py:1038-1041, 1084-1086, hod.py:189-230 switches between analytic
        // expressions), so a node inside it need not be EVALUATED: its branch is decided by
        // the plan's xstar, its value read off the degree-7 polynomial through the last /
        // first 8 samples of that neighbour, continued by less than one coarse spacing.
        // The eighth difference at the segment's end bounds that continuation's error
        // (0.2 of it at mid-interval); a singular satellite onset (alpha != 1) is not
        // continued across: those intervals have no smooth neighbour on that side.
        if (tid < nf_all) {
          const int i = PL.fine[tid];
          bool ok = PL.geom[tid] != 0;
          if (ok) {
            // |Delta^8 F| at the two ends against the size of the samples there
            const double w8[9] = {1.0, -8.0, 28.0, -56.0, 70.0, -56.0, 28.0, -8.0, 1.0};
            double dl0 = 0.0, dl1 = 0.0, dr0 = 0.0, dr1 = 0.0, sc0 = 0.0, sc1 = 0.0;
#pragma unroll
            for (int m = 0; m < 9; ++m) {
              const int pl = deep_pos(i - 8 + m), pr = deep_pos(i + 1 + m);
              dl0 = fma(w8[m], F0[pl], dl0);
              dl1 = fma(w8[m], F1[pl], dl1);
              dr0 = fma(w8[m], F0[pr], dr0);
              dr1 = fma(w8[m], F1[pr], dr1);
              sc0 += fabs(F0[pl]) + fabs(F0[pr]);
              sc1 += fabs(F1[pl]) + fabs(F1[pr]);
            }
            // (budget: the nodes of a break-point interval are 1 / NC of a level's; with up to
            //  four such intervals an error of tol NC / 4 of the samples' size per node keeps
            //  the level sum within tol -- 5e-7 at the defaults; sc sums 18 samples)
            const double lim = tol * (double)NC / (4.0 * 18.0);
            ok = fabs(dl0) <= lim * sc0 && fabs(dr0) <= lim * sc0 &&
                 fabs(dl1) <= lim * sc1 && fabs(dr1) <= lim * sc1;
          }
          fine_poly[tid] = ok ? 1 : 0;
        }
        // ---- per smooth segment the sum of its samples lo .. hi - 7: all a deep level
        // needs of the segment's interior (below); every segment in one pass, one exchange
        {
#pragma unroll
          for (int sgi = 0; sgi <= kDeepMaxRough; ++sgi) {
            double a0 = 0.0, a1 = 0.0;
            if (sgi < ns) {                                  // block-uniform
              const int hi7 = PL.seg_hi[sgi] - 7;
              for (int q = PL.seg_lo[sgi] + tid; q <= hi7; q += NT) {
                const int at = deep_pos(q);
                a0 += F0[at];
                a1 += F1[at];
              }
              a0 = wave_sum(a0);
              a1 = wave_sum(a1);
              if (ln == 0) { seg_slot[wv][0][sgi] = a0; seg_slot[wv][1][sgi] = a1; }
            }
          }
        }
        __syncthreads();
        if (tid < 2 * (kDeepMaxRough + 1)) {
          const int f = tid / (kDeepMaxRough + 1), sgi = tid % (kDeepMaxRough + 1);
          double v = 0.0;
          if (sgi < ns) {
#pragma unroll
            for (int w2 = 0; w2 < NWV; ++w2) v += seg_slot[w2][f][sgi];
          }
          seg_sum[f][sgi] = v;
        }
        // (visible to the rounds below: the self-check's exchanges carry barriers)
        KSTAMP(4);
        // ---- self-check: the same machinery one level up.  Every odd sample is predicted
        // from the even ones (stencils of twice the spacing, shifted at segment ends
        // exactly as below) and compared with its true value; at the spacing actually
        // used the interpolation error is 2^8 times smaller.
        double e0 = 0.0, e1 = 0.0, m0 = 0.0, m1 = 0.0;
        for (int ep = tid; ep < NC / 2; ep += NT) {
          // the segment that holds both intervals 2 ep and 2 ep + 1 (none: node by node)
          int sg = -1;
          for (int x = 0; x < ns; ++x)
            if (2 * ep >= PL.seg_lo[x] && 2 * ep + 1 < PL.seg_hi[x]) sg = x;
          if (sg < 0) continue;
          const int lo_e = (PL.seg_lo[sg] + 1) >> 1, hi_e = PL.seg_hi[sg] >> 1;   // even nodes / 2
          if (hi_e - lo_e + 1 < kDeepStencil) continue;
          int st = ep - 3;
          st = st < lo_e ? lo_e : (st > hi_e - 7 ? hi_e - 7 : st);
          const double* w = W + (ep - st) * kDeepStencil;
          double p0 = 0.0, p1 = 0.0;
#pragma unroll
          for (int m = 0; m < kDeepStencil; ++m) {
            p0 = fma(w[m], F0[st + m], p0);                // (even sample 2 (st + m))
            p1 = fma(w[m], F1[st + m], p1);
          }
          e0 += fabs(p0 - F0[kDeepOdd + ep]);              // (odd sample 2 ep + 1)
          e1 += fabs(p1 - F1[kDeepOdd + ep]);
        }
        for (int q = tid; q <= NC / 2; q += NT) { m0 += F0[q]; m1 += F1[q]; }
        for (int q = tid; q < NC / 2; q += NT) { m0 += F0[kDeepOdd + q]; m1 += F1[kDeepOdd + q]; }
        e0 = group_sum<NWV>(e0, red, flip);
        e1 = group_sum<NWV>(e1, red, flip);
        m0 = group_sum<NWV>(m0, red, flip);
        m1 = group_sum<NWV>(m1, red, flip);
        const bool bad0 = !R.done[0] && !(e0 * (1.0 / 256.0) <= tol * fabs(m0));
        const bool bad1 = !R.done[1] && !(e1 * (1.0 / 256.0) <= tol * fabs(m1));
        if (bad0 || bad1) literal = true;
        bool need_eval = false;
        if constexpr (!EVAL) {   // an interval that has to be evaluated node by node
          for (int x = 0; x < nf_all; ++x) need_eval = need_eval || fine_poly[x] == 0;
        }
        if (stats && tid == 0) {
          if (!literal && need_eval) atomicAdd(&stats[6], 1);
          if (literal) atomicAdd(&stats[4], 1);
          const float r0 = R.done[0] ? 0.0f : (float)(e0 * (1.0 / 256.0) / fabs(m0));
          const float r1 = R.done[1] ? 0.0f : (float)(e1 * (1.0 / 256.0) / fabs(m1));
          atomicMax(&stats[5], __float_as_int(fmaxf(r0, r1)));   // (positive floats order as ints)
        }
        if (need_eval && !literal) to_eval = true;
      }
      if (literal) {               // block-uniform: on to k_halo_knots_literal, still counted
        if (tid == 0) lit_items[atomicAdd(&pending[4], 1)] = item;
        continue;
      }
      if (to_eval) {               // ... to the evaluating instance behind this launch
        if (tid == 0) ev_items[atomicAdd(&pending[kPendingEvCount + round], 1)] = item_sh;
        continue;
      }
      KSTAMP(5);
      KSTAMP_VALUE(22, nf_all);
      {
        // ---- deeper levels, kDeepRound at a time (their sums are independent; what a level
        // costs here is latency -- a handful of node-by-node evaluations and two reductions --
        // so a pass over three levels takes little longer than one; a knot that stops at the
        // first of them has summed two levels for nothing): weighted sums of the samples + the
        // break-point intervals, then the rows one by one
        const int nf = nf_all;
        for (int lv0 = LC + 1; lv0 <= cfg.divmax && !R.all_done(); lv0 += kDeepRound) {
          const int ng = cfg.divmax - lv0 + 1 < kDeepRound ? cfg.divmax - lv0 + 1 : kDeepRound;
          // (read back behind the round's exchange; two buffers in turn: a wavefront may still
          //  be reading the last round's while wavefront 0 writes this one's)
          double* const park = rows_park[((lv0 - LC - 1) / kDeepRound) & 1];
          R.park(park);
          double s0[kDeepRound], s1[kDeepRound];
#pragma unroll
          for (int g = 0; g < kDeepRound; ++g) { s0[g] = 0.0; s1[g] = 0.0; }
          const double* Wl = W + (size_t)(lv0 - LC - 1) * kDeepWStride;
          // The smooth segments.  Interval i of a segment [lo, hi] takes the stencil starting
          // at st = clamp(i - 3, lo, hi - 7), offset o = i - st: o = 0, 1, 2 for the first three
          // intervals, 4, 5, 6 for the last three, 3 for all those between -- whose total is
          // sum_m W[3][m] D_m with D_m = sum of the samples lo + m .. hi - 7 + m: range sums that
          // do not depend on the level (D_0 is seg_sum; D_(m+1) = D_m - F[lo + m] + F[hi - 6 + m]).
          // One task per (segment, level of the round, {three left, interior, three right}):
          // <= 9 x 3 x 7 threads with 8 multiply-adds each, where a pass over all 2048 intervals
          // cost 40 LDS reads per interval and level round (6.7 us of a ~11 us round).
          {
            if (tid < ns * ng * 7) {
              const int part = tid % 7, g = (tid / 7) % ng, sgi = tid / (7 * ng);
              const int lo = PL.seg_lo[sgi], hi = PL.seg_hi[sgi];
              const double* w = Wl + g * kDeepWStride + part * kDeepStencil;
              double v0 = 0.0, v1 = 0.0;
              if (part != 3) {
                const int st = part < 3 ? lo : hi - 7;
#pragma unroll
                for (int m = 0; m < kDeepStencil; ++m) {
                  const int at = deep_pos(st + m);
                  v0 = fma(w[m], F0[at], v0);
                  v1 = fma(w[m], F1[at], v1);
                }
              } else {
                double d0 = seg_sum[0][sgi], d1 = seg_sum[1][sgi];
#pragma unroll
                for (int m = 0; m < kDeepStencil; ++m) {
                  v0 = fma(w[m], d0, v0);
                  v1 = fma(w[m], d1, v1);
                  if (m < kDeepStencil - 1) {
                    const int ah = deep_pos(hi - 6 + m), al = deep_pos(lo + m);
                    d0 += F0[ah] - F0[al];
                    d1 += F1[ah] - F1[al];
                  }
                }
              }
#pragma unroll
              for (int gg = 0; gg < kDeepRound; ++gg)
                if (gg == g) { s0[gg] += v0; s1[gg] += v1; }
            }
          }
          if (lv0 == LC + 1) KSTAMP(10);
          // the break-point intervals: level lv0 + g has n0 << g nodes in each
          const int n0 = 1 << (lv0 - 1 - LC);
          const int per0 = nf * n0;                            // nodes of the round's first level
          const int total = per0 * ((1 << ng) - 1);
          for (int idx = tid; idx < total; idx += NT) {
            int g = 0, r = idx;
            while (r >= (per0 << g)) { r -= per0 << g; ++g; }
            const int n = n0 << g;
            const int xi = r / n, rr = r % n, iv = PL.fine[xi];
            const long j = (long)iv * n + rr;
            const double h = ldexp(b - a, 1 - (lv0 + g));
            const double x = (a + 0.5 * h) + h * (double)j;
            double o[2];
            if (fine_poly[xi]) {
              // (the node's branch: the plan's tstar on the interval's grid of 2^(divmax - LC)
              //  parts, where this node is the odd multiple 2 rr + 1 of 2^(divmax - level))
              const bool left = ((2 * rr + 1) << (cfg.divmax - (lv0 + g))) < PL.tstar[xi];
              const int s0i = left ? iv - 7 : iv + 1;
              const double t = (left ? 7.0 : -1.0) + ((double)rr + 0.5) / (double)n;
              lagrange8_pair(t, F0, F1, s0i, [](int q) { return deep_pos(q); }, &o[0], &o[1]);
            } else {
              if constexpr (EVAL) {
                int st;
                halo_eval_coded(group, c, x, o, &st);
              } else {             // (this instance carries no integrand: see EVAL)
                o[0] = o[1] = 0.0;
                bail_sh = 1;
              }
            }
#pragma unroll
            for (int gg = 0; gg < kDeepRound; ++gg)
              if (gg == g) { s0[gg] += o[0]; s1[gg] += o[1]; }
          }
          if (lv0 == LC + 1) KSTAMP(11);
          // one exchange for the 2 ng sums
          {
            double* slot = red + (flip ? 2 * kDeepRound * NWV : 0);
#pragma unroll
            for (int g = 0; g < kDeepRound; ++g) {
              const double x0 = wave_sum(s0[g]), x1 = wave_sum(s1[g]);
              if ((tid & 63) == 0) {
                slot[(2 * g) * NWV + (tid >> 6)] = x0;
                slot[(2 * g + 1) * NWV + (tid >> 6)] = x1;
              }
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < kDeepRound; ++g) {
              double t0 = 0.0, t1 = 0.0;
#pragma unroll
              for (int w2 = 0; w2 < NWV; ++w2) {
                t0 += slot[(2 * g) * NWV + w2];
                t1 += slot[(2 * g + 1) * NWV + w2];
              }
              s0[g] = t0; s1[g] = t1;
            }
            flip ^= 1;
          }
          if (lv0 == LC + 1) KSTAMP(12);
          if constexpr (!EVAL) {
            // (read behind the exchange's barrier: block-uniform; cannot happen while the
            //  pre-check above sends every knot with such an interval on, kept as the guard)
            if (bail_sh) { to_eval = true; break; }
          }
          R.unpark(park);
#pragma unroll
          for (int g = 0; g < kDeepRound; ++g)
            if (g < ng && !R.all_done()) R.advance(lv0 + g, s0[g], s1[g]);
          KSTAMP(5 + (lv0 - LC - 1) / kDeepRound + 1);
        }
      }
      if (to_eval) {               // (!EVAL only) still counted in npend[e]
        if (tid == 0) {
          ev_items[atomicAdd(&pending[kPendingEvCount + round], 1)] = item_sh;
          if (stats) atomicAdd(&stats[6], 1);
        }
        continue;
      }
    }
    KSTAMP(16);
    KSTAMP_VALUE(23, R.level[0] > R.level[1] ? R.level[0] : R.level[1]);
    if (tid == 0) {
      if (pa) { knot_store(&t[L.off_knot[fa] + ik], R.value[0]); knot_store(&levs[fa * NK + ik], (double)R.level[0]); }
      if (pb) { knot_store(&t[L.off_knot[fb] + ik], R.value[1]); knot_store(&levs[fb * NK + ik], (double)R.level[1]); }
      unsigned st = 0u;              // divmax exhausted (halo.py:1065-1071 and alike)
      if (pa && !R.done[0]) st |= kStHaloDivmax0 << fa;
      if (pb && !R.done[1]) st |= kStHaloDivmax0 << fb;
      if (st) atomicOr(&status[e], st);
      if (stats) atomicAdd(&stats[0], 1);
    }
    deep_arrive(cfg, L, epochs_rw, tab, e, fam_mask, status, npend, fences, &last_sh, sm,
                kArriveThrough);
    KSTAMP(17);
    if (!w_safe) { __syncthreads(); stage_weights(); }
  }   // next item
}

// LIT: the instance also works off the list of knots handed on to the literal evaluation
// (deep_literal_loop) once its own list is empty -- the launch behind the lean instance does
// both, so that the chain carries one near-empty launch behind the fast sums, not two.
template <int LC, int NT, bool SELF, bool EVAL, bool LIT = false>
// (register caps, by measurement: the lean instance three blocks of 256 to a CU; the instance
//  that evaluates nodes, as the MAIN pass of a set-up with alpha != 1, capped at 256 registers
//  -- it spills ~180 bytes in its node loop and is still 13 % faster than uncapped at one block
//  per CU, 0.697 against 0.799 ms on a configs[2]-sized batch with alpha = 0.9; the instances
//  that only see the rare hand-overs -- SELF, LIT -- uncapped and free of scratch)
__global__ __launch_bounds__(NT, (SELF || LIT) ? 1 : (EVAL ? 512 / NT : (NT == 256 ? 3 : 1))) void k_halo_knots_fast(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs, double* __restrict__ tab,
    const SiCiTab* __restrict__ sici_g, int g0, int g1, int g2, unsigned mask, int n_epoch,
    int* __restrict__ pending, int* __restrict__ npend, Epoch* __restrict__ epochs_rw,
    unsigned fam_mask, unsigned* __restrict__ status, const double* __restrict__ deepw,
    int all_literal, double tol, int max_rough, int max_fine, int* __restrict__ stats,
    const double* __restrict__ samples, const double* __restrict__ psum, int parts,
    int round, int slot_lo, int slot_hi, int from_eval, const DeepPlan* __restrict__ plans,
    const chomp_halo_par* __restrict__ profile, const HodDev* __restrict__ hod) {
  deep_fast_body<LC, NT, SELF, EVAL>(cfg, L, epochs, tab, sici_g, g0, g1, g2, mask, n_epoch, pending,
                                     npend, epochs_rw, fam_mask, status, deepw, all_literal, tol,
                                     max_rough, max_fine, stats, samples, psum, parts, round,
                                     slot_lo, slot_hi, from_eval, plans);
  if constexpr (LIT) {
    __syncthreads();
    deep_literal_loop<NT>(cfg, L, epochs, tab, profile, hod, sici_g, g0, g1, g2, mask, n_epoch,
                          pending, npend, epochs_rw, fam_mask, status, stats);
  }
}

// The knots k_halo_knots_fast handed on (pending[4] of them, behind pending_literal_base):
// every node of scipy's Romberg evaluated (deep_literal) -- the checker of the fast path and
// its fallback.  grid: any (blocks draw from the list), block NT.  Launched behind
// k_halo_knots_fast whenever knots can be listed at all; with an empty list (the rule) every
// block returns after one read.
template <int NT>
__device__ __forceinline__ void deep_literal_loop(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs, double* __restrict__ tab,
    const chomp_halo_par* __restrict__ profile, const HodDev* __restrict__ hod,
    const SiCiTab* __restrict__ sici_g, int g0, int g1, int g2, unsigned mask, int n_epoch,
    int* __restrict__ pending, int* __restrict__ npend, Epoch* __restrict__ epochs_rw,
    unsigned fam_mask, unsigned* __restrict__ status, int* __restrict__ stats) {
  constexpr int NWV = NT / 64;
  extern __shared__ __align__(16) double sm[];
  __shared__ Epoch E;
  __shared__ SiCiTab S;
  __shared__ int item_sh, last_sh;
  const int count = pending[4];
  if (count == 0) return;
  const int NK = L.NK;
  const int tid = threadIdx.x;
  const int* items = pending + pending_literal_base((size_t)n_epoch, NK);
  for (;;) {
    __syncthreads();
    if (tid == 0) item_sh = atomicAdd(&pending[5], 1);
    __syncthreads();
    if (item_sh >= count) return;  // block-uniform
    const int item = items[item_sh];
    const int ik = item % NK, e = (item / NK) % n_epoch, zg = item / (NK * n_epoch);
    const int group = zg == 0 ? g0 : (zg == 1 ? g1 : g2);
    double* t = tab + (size_t)e * L.stride;
    double* levs = t + L.off_levels;
    const int fa = group_fa(group), fb = group_fb(group);
    const bool pa = group != 2 && (mask & (1u << fa)) && levs[fa * NK + ik] == kPendingLevel;
    const bool pb = (mask & (1u << fb)) && levs[fb * NK + ik] == kPendingLevel;
    HaloLds H;
    H.stage(L, E, S, epochs, e, t, profile, hod, sici_g, sm);
    HaloCtx c{&E, &S, H.nu_knots, H.lnm_pp, L.NM,
              linspace_at(log(cfg.k_min), log(cfg.k_max), NK, ik), (mask & kMaskExclusion) != 0};
    double val[2];
    int lev[2];
    bool conv[2];
    deep_literal<NWV>(cfg, c, E, group, pa, H.rest, val, lev, conv);
    if (tid == 0) {
      if (pa) { t[L.off_knot[fa] + ik] = val[0]; levs[fa * NK + ik] = (double)lev[0]; }
      if (pb) { t[L.off_knot[fb] + ik] = val[1]; levs[fb * NK + ik] = (double)lev[1]; }
      unsigned st = 0u;              // divmax exhausted (halo.py:1065-1071 and alike)
      if (pa && !conv[0]) st |= kStHaloDivmax0 << fa;
      if (pb && !conv[1]) st |= kStHaloDivmax0 << fb;
      if (st) atomicOr(&status[e], st);
      if (stats) atomicAdd(&stats[1], 1);
    }
    deep_arrive(cfg, L, epochs_rw, tab, e, fam_mask, status, npend, true, &last_sh, sm);
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_halo_knots_literal(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs, double* __restrict__ tab,
    const chomp_halo_par* __restrict__ profile, const HodDev* __restrict__ hod,
    const SiCiTab* __restrict__ sici_g, int g0, int g1, int g2, unsigned mask, int n_epoch,
    int* __restrict__ pending, int* __restrict__ npend, Epoch* __restrict__ epochs_rw,
    unsigned fam_mask, unsigned* __restrict__ status, int* __restrict__ stats) {
  deep_literal_loop<NT>(cfg, L, epochs, tab, profile, hod, sici_g, g0, g1, g2, mask, n_epoch,
                        pending, npend, epochs_rw, fam_mask, status, stats);
}

}  // namespace chomp
