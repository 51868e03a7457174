// chomp_mass_kernels.h -- Stage K, first half (gfx950): everything up to the mass
// function of an epoch (= one (cosmology, z) pair).
//
//   k_sigma_nodes   cosmology-only tables, once per distinct cosmology of the batch: the
//                   sigma(R) node table, the sigma_8 integral (cosmology.py:118-119), the
//                   coarse ln S(R) table that aims the mass-limit search, and the
//                   closed-form part of every epoch record (cosmology.py:39-119)
//   k_epoch_probe   comoving distance + the mass-limit search of
//                   MassFunction._set_mass_limits (mass_function.py:160-203)
//   k_nu_mass       MassFunction._initialize_splines nu_m loop (mass_function.py:205-210);
//                   the last block of an epoch to finish goes on with the splines, m_star and
//                   the f / bias normalisations (mass_function.py:212-241, Tinker 532-564)
//                   and, when the halo model follows in the same call (chomp_stage_k), with
//                   the node tables of the halo integrals (chomp_halo_kernels.h)
//
// Also here: the layout of the per-epoch table block (TabLayout) and of the node tables,
// shared by the other kernel headers.
//
// Execution shape: one integral (or a pair sharing nodes) per group of wavefronts;
// tables are staged in LDS or read through L2; reductions are wavefront butterflies.
// No MFMA: there is no dense contraction anywhere on this path.
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
  // pinned host words (device-accessible) the block that finalises an epoch's halo set-up
  // mirrors the epoch's status word into -- (set-up sequence number << 32) | word, one 8-byte
  // store: the host knows the word is this set-up's by the number, without an event -- or
  // nullptr (chomp_status_post)
  unsigned long long* h_status;
  unsigned h_seq;
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
  L.h_status = nullptr;
  L.h_seq = 0u;
  return L;
}

// Families: index into off_knot / off_kpp.
enum { F_HM = 0, F_PPMM = 1, F_HG = 2, F_PPGM = 3, F_PPGG = 4 };

// Node tables of the halo integrals: every knot k of an epoch integrates over the
// SAME ln(nu) nodes, so everything that does not depend on k (nu f(nu), b(nu), M(nu),
// concentration, r_s, HOD moments) is tabulated once per (epoch, integration range)
// on the level-kNodeLevel Romberg grid, stored level by level so a level's nodes are
// contiguous.  Deeper levels fall back to direct evaluation.
// The table has room for one level more (kNodeTabLevel): with integrands that can run beyond
// the node table (the HOD ones) its nodes are filled in too, and k_halo_knots_fast reads the
// 2^11 + 1 coarse samples of a listed knot off them instead of evaluating each from scratch.
constexpr int kNodeLevel = 10;
constexpr int kNodeTabLevel = 11;
constexpr int kNodeBase = (1 << kNodeLevel) + 1;        // nodes k_halo_knots reads
constexpr int kNodeCount = (1 << kNodeTabLevel) + 1;    // nodes of the table (the field stride)
constexpr int kNodeFields = 9;   // wA, wB, ln_rs, con, ln_cp, inv_mass_k, state (bit 0: flag),
                                 // r_s, 1 / ((1 + c) r_s): k r_s and its reciprocal are then two
                                 // multiplications per (knot, node) -- no exp, no division
constexpr int kNodeStride = kNodeFields * kNodeCount + 8;   // doubles per (epoch, group);
                                                            // tail: integration limits a, b
__host__ __device__ inline int node_index(int lev, long j) {
  return lev == 0 ? (int)j : 1 + (1 << (lev - 1)) + (int)j;
}
// A pointer into global memory of which every lane of the wavefront holds the same value, told to
// the compiler: loads through it take the scalar base + 32-bit vector offset form instead of a
// 64-bit address computed per lane (four vector instructions per table node).
typedef const double __attribute__((address_space(1)))* UniformDoubles;
__device__ __forceinline__ UniformDoubles uniform_ptr(const double* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (UniformDoubles)(((unsigned long long)hi << 32) | lo);
}

constexpr int kKnotScratch = romberg_scratch<4, 2>();   // LDS doubles of a knot block
constexpr int kSearchJ = 2048;        // candidates per walking direction
constexpr int kEpochDoubles = (int)(sizeof(Epoch) / sizeof(double));
static_assert(sizeof(Epoch) % sizeof(double) == 0, "Epoch must be 8-byte granular");

__device__ __forceinline__ bool same_cosmology(const Epoch& a, const Epoch& b) {
  return a.om0 == b.om0 && a.ob0 == b.ob0 && a.ol0 == b.ol0 && a.or0 == b.or0 &&
         a.tcmb == b.tcmb && a.h == b.h && a.ns == b.ns;
}

// Per-epoch status word (chomp_get_status): the device-side copies of the CHOMP_ST_* bits.
constexpr unsigned kStMassMinSaturated = CHOMP_ST_MASS_MIN_SATURATED;
constexpr unsigned kStMassMaxSaturated = CHOMP_ST_MASS_MAX_SATURATED;
constexpr unsigned kStSearchExhausted = CHOMP_ST_MASS_SEARCH_EXHAUSTED;
constexpr unsigned kStSigmaDivmax = CHOMP_ST_SIGMA_DIVMAX;
constexpr unsigned kStHaloDivmax0 = CHOMP_ST_HALO_DIVMAX_H_M;       // << family index
constexpr unsigned kStNonfinite = CHOMP_ST_NONFINITE;
constexpr unsigned kStHaloBits = (31u * CHOMP_ST_HALO_DIVMAX_H_M) | CHOMP_ST_NONFINITE;

// Cooperative copy of POD blocks as doubles.
// Development stamps (-DCHOMP_STAMPS, absent from the product build): s_memtime at the phase
// boundaries of every block of k_mass_nodes (tools/dev_mass_stamps.py; block (x, 0, z) < (64, 1, 8))
// or, with -DCHOMP_STAMPS=2, of k_halo_knots<1> (block (x, y, 0) < (64, 14, 1)).
#ifdef CHOMP_STAMPS
constexpr int kMStampSlots = 16;
static __device__ long long g_ms[64 * 16 * kMStampSlots];
#endif
#if defined(CHOMP_STAMPS) && CHOMP_STAMPS == 3   /* k_nu_table<., 1>, block (x, y) < (64, 50) */
#define NUSTAMP(k, v)                                                                          \
  do {                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 64 && blockIdx.y < 50)                                \
      g_ms[(blockIdx.x * 50 + blockIdx.y) * 4 + (k)] = (long long)(v);                         \
  } while (0)
#else
#define NUSTAMP(k, v) do { } while (0)
#endif
#if defined(CHOMP_STAMPS) && CHOMP_STAMPS == 4   /* k_epoch_probe<., 0>, block (x, y) < (64, 8) */
#define PSTAMP(k)                                                                              \
  do {                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 64 && blockIdx.y < 8)                                 \
      g_ms[(blockIdx.x * 8 + blockIdx.y) * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
#if defined(CHOMP_STAMPS) && CHOMP_STAMPS == 1
#define MSTAMP(k)                                                                              \
  do {                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 64 && blockIdx.y == 0 && blockIdx.z < 8)              \
      g_ms[(blockIdx.x * 8 + blockIdx.z) * kMStampSlots + (k)] =                               \
          (long long)__builtin_amdgcn_s_memtime();                                             \
  } while (0)
#else
#define MSTAMP(k) do { } while (0)
#endif
#if defined(CHOMP_STAMPS) && CHOMP_STAMPS == 2
#define KNSTAMP(k, v)                                                                          \
  do {                                                                                         \
    if ((threadIdx.x & 63) == 0 && blockIdx.y < 14 && blockIdx.x < 64 && blockIdx.z == 0)      \
      g_ms[((blockIdx.x * 14 + blockIdx.y) * 4 + (threadIdx.x >> 6)) * 4 + (k)] = (long long)(v); \
  } while (0)
#else
#define KNSTAMP(k, v) do { } while (0)
#endif

__device__ __forceinline__ void copy_doubles(double* dst, const double* src, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// sigma(R) node table.  For 14.0662/k_max < R < 0.1/k_min (0.14 < R < 100 Mpc/h at
// the default limits) SingleEpoch.sigma_r integrates over the fixed range
// [ln k_min, ln k_max] (cosmology.py:611-632), so every such integral of an epoch
// visits the same ln k nodes: k_j and the R-independent factor
// (k_j/H0)^(3+n) T(k_j)^2 of Delta^2 are tabulated once per epoch on the
// level-kSigmaLevel Romberg grid (level-major order) by k_sigma_nodes (the factor is
// stored divided by k_j^6, see SigmaTabIntegrand).
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
constexpr int kSigmaNodeBlocks = (kSigmaCount + 255) / 256;
constexpr int kSigmaOffPart = kSigmaOffLnS + kSGrid;
// doubles per cosmology: k[], d2[], I8 = int dlnk d2 W(8k)^2 (sigma_8 normalisation; the slot
// after it is the arrival counter of the node blocks), ln S[], and the node blocks'
// per-level partial sums of the sigma_8 integrand
constexpr int kSigmaOffG = kSigmaOffPart + kSigmaNodeBlocks * (kSigmaLevel + 1);
// Outside that range of R sigma_r's limits follow R (cosmology.py:611-632) and no node table
// can be shared; the R-independent factor of the integrand is then INTERPOLATED instead of
// recomputed (exp, log, three divisions per node): (k/H0)^(3+n) T(k)^2 / k^6 and k itself on
// a uniform grid of kGTabN intervals in ln k over [ln(k_min / 100), ln(100 k_max)] -- the
// widest range sigma_r ever integrates over -- with kGTabPad extra points at either end for
// the 6-point Lagrange stencil.  At 8192 intervals (d ln k = 2.5e-3) the interpolation error
// is ~1e-13 relative; only the no-wiggle transfer function is smooth enough for this (the
// BAO instances keep the direct evaluation).
constexpr int kGTabN = 8192;
constexpr int kGTabPad = 3;
static_assert(kGTabN == kGTabIntervals, "chomp_math.h: epoch_k_range");
constexpr int kGTabCount = kGTabN + 1 + 2 * kGTabPad;
constexpr int kGTabBlocks = (kGTabCount + 255) / 256;
// doubles per cosmology: ..., then g / k^6 [kGTabCount] and k [kGTabCount]
constexpr int kSigmaStride = kSigmaOffG + 2 * kGTabCount;

// The same integrand where sigma_r's limits follow R: Delta^2 W^2 with the R-independent
// factor and k interpolated from the cosmology's uniform ln k table (kGTab*; 6-point
// Lagrange, k through a short series of exp over the fraction of a step).
struct SigmaInterpIntegrand {
  UniformDoubles g;        // g / k^6 on the grid (index 0 = first pad point)
  double xlo, dx, inv_dx;
  double scale, amp2_nine_over_r6, amp2;
  // k R < 1 over the whole range (wave-uniform): the window in the reference's own form
  // 3 (sin x / x^3 - cos x / x^2), whose rounding error at x << 1 is what ends a saturated
  // mass-limit walk (search_status); s - x c is exactly 0 there and would never end it
  bool tiny_r;
  SinCosLead lead = sincos_lead();
  __device__ __forceinline__ double operator()(double ln_k) const {
    const double u = (ln_k - xlo) * inv_dx;
    int i = (int)u;
    i = i < 0 ? 0 : (i > kGTabN - 1 ? kGTabN - 1 : i);
    const double t = u - (double)i;
    UniformDoubles q = g + i + kGTabPad - 2;           // stencil nodes -2 .. 3 around interval i
    const double a = t + 2.0, b = t + 1.0, d = t - 1.0, e = t - 2.0, f = t - 3.0;
    const double ab = a * b, ef = e * f, cd = t * d;
    const double gv = q[0] * (b * cd * ef) * (-1.0 / 120.0) + q[1] * (a * cd * ef) * (1.0 / 24.0) +
                      q[2] * (ab * d * ef) * (-1.0 / 12.0) + q[3] * (ab * t * ef) * (1.0 / 12.0) +
                      q[4] * (ab * cd * f) * (-1.0 / 24.0) + q[5] * (ab * cd * e) * (1.0 / 120.0);
    const double y = t * dx;                           // <= 2.6e-3: four terms reach 1e-16
    const double ey = fma(y, fma(y, fma(y, fma(y, 1.0 / 24.0, 1.0 / 6.0), 0.5), 1.0), 1.0);
    const double k = q[2 + kGTabCount] * ey;
    const double kR = scale * k;
    if (tiny_r) {
      double s, c;
      fast_sincos_pm(kR, &s, &c);
      const double kR2 = kR * kR, k3 = k * k * k;
      const double W = 3.0 * (s / (kR2 * kR) - c / kR2);
      return amp2 * (gv * (k3 * k3)) * (W * W);
    }
    const double w = tophat_numer_pm(kR, lead);
    return amp2_nine_over_r6 * gv * (w * w);
  }
};

// Everything here depends on the cosmology only, not on z, so it is built once per
// distinct cosmology of the batch ("slot"; the z-axis of a (k, z) grid is one slot).
// grid (kSigmaNodeBlocks + kSGrid, n_slots + ceil(n_epoch / 256)), block 256;
// first[s] = an epoch that has cosmology s.  For y < n_slots: the node-table blocks, then
// kSGrid blocks for the coarse ln S(R) table (direct evaluation).  The sigma_8 integral
// (cosmology.py:118-119) lives on the same nodes as the table: every node block also sums
// the sigma_8 integrand of its nodes per Romberg level, and the last one to finish replays
// scipy's rows and stopping test from the level sums -- the integral costs one pass over
// the table instead of a chain of levels.  The rows y >= n_slots fill in the
// closed-form part of every epoch record (SingleEpoch.__init__ minus its two integrals),
// one epoch per thread, while the cosmology-only integrals run.
// BAO: the context's transfer function (chomp_set_transfer), fixed at compile time.
// sigma_8's integral I8 = int dlnk (k/H0)^(3+n) T^2 W(8k)^2 from the per-level partial sums the
// node-table blocks of a cosmology left (n + kSigmaOffPart: nb rows of kSigmaLevel + 1): scipy's
// rows and stopping test on the level sums; beyond the table (or off its range) the integral
// directly.  Whole block (>= 64 threads; barriers inside); red: romberg_scratch<4, 1>() doubles
// when blockDim.x == 256 (the direct fallback is a four-wavefront Romberg), else one wavefront's.
template <bool BAO>
__device__ __forceinline__ void sigma8_from_parts(const chomp_config& cfg, const Epoch& E,
                                                  double* __restrict__ n, int nb, double* red) {
  const double a = log(cfg.k_min), b = log(cfg.k_max);
  const bool tab8 = 0.1 / 8.0 > cfg.k_min && 14.0662 / 8.0 < cfg.k_max && cfg.divmax >= 1;
  __shared__ double stage[kSigmaNodeBlocks * (kSigmaLevel + 1)];
  __shared__ double S[kSigmaLevel + 1];
  __shared__ double i8;
  __shared__ int converged;
  for (int q = threadIdx.x; q < nb * (kSigmaLevel + 1); q += blockDim.x)   // all loads in flight
    stage[q] = __hip_atomic_load(n + kSigmaOffPart + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x <= kSigmaLevel) {
    double t = 0.0;
    for (int q = 0; q < nb; ++q) t += stage[q * (kSigmaLevel + 1) + threadIdx.x];
    S[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    // scipy.integrate.romberg's rows and stopping test on the level sums, as the replay of
    // chomp_romberg.h does them: lane m holds the trapezoid estimate T_m, row i is
    // sum_m C[i][m] T_m (one multiply and one butterfly per row)
    const int lane = threadIdx.x;
    const double range = b - a;
    double ordsum = 0.5 * S[0], nn = 1.0;
    double result = range * ordsum, prev = result;
    double Tl = lane == 0 ? result : 0.0;
    int conv = 0;
    const int top = cfg.divmax < kSigmaLevel ? cfg.divmax : kSigmaLevel;
    // (the rows' weights first, all reads in flight: one dependent read of the constant table
    //  per row costs this tail ~1 us each)
    double crow[kSigmaLevel + 1];
#pragma unroll
    for (int i = 1; i <= kSigmaLevel; ++i) crow[i] = CHOMP_ROMBERG_C[i][lane & 31];
#pragma unroll
    for (int i = 1; i <= kSigmaLevel; ++i) {
      if (tab8 && i <= top && !conv) {
        nn *= 2.0;
        ordsum += S[i];
        const double Ti = ldexp(range * ordsum, -i);     // (/ nn, nn = 2^i: the same bits)
        if (lane == i) Tl = Ti;
        result = wave_sum32(crow[i] * Tl);
        const double err = fabs(result - prev);
        prev = result;
        if (err < cfg.global_precision || err < cfg.cosmo_precision * fabs(result)) conv = 1;
      }
    }
    // (divmax below the table's level: scipy returns the last row, unconverged)
    if (tab8 && !conv && cfg.divmax <= kSigmaLevel) conv = 1;
    if (lane == 0) {
      converged = conv;
      i8 = result;
    }
  }
  __syncthreads();
  if (!converged) {                // beyond the table (or off its range): direct evaluation
    double lo, hi;
    sigma_limits(E, 8.0, &lo, &hi);
    SigmaIntegrandT<BAO> f{&E, 8.0};              // sigma_norm = 1: amp * integral
    double s2;
    if (blockDim.x == 256) {
      s2 = romberg1<4>(f, lo, hi, cfg.global_precision, cfg.cosmo_precision, cfg.divmax, red);
    } else {                       // (block-uniform) the first wavefront alone
      s2 = 0.0;
      if (threadIdx.x < 64)
        s2 = romberg1<1>(f, lo, hi, cfg.global_precision, cfg.cosmo_precision, cfg.divmax, nullptr);
    }
    if (threadIdx.x == 0) i8 = s2;
  }
  if (threadIdx.x == 0) n[kSigmaOffI8] = i8;
}

// NPT: table nodes per thread.  1: most blocks, shortest launch (a (k, z) grid has ONE cosmology
// and the launch is a link of its latency chain).  4: a batch of many cosmologies (a design, an
// MCMC population) -- a quarter of the blocks, each paying the serial head (the transfer-function
// constants, by one thread) once for four times the nodes; the 48 ln S integrals -- two thirds
// of a cosmology's work when every node is evaluated directly -- are left to k_sigma_lns, behind
// this launch.  18 blocks per cosmology instead of 114 (k_sigma_nodes was 56 % of a
// 1024-cosmology step).  Same node values; level sums in a different order (sigma_8 to ~1e-16).
template <int NPT>
__host__ __device__ constexpr int sigma_node_blocks() { return (kSigmaCount + 256 * NPT - 1) / (256 * NPT); }
template <int NPT>
__host__ __device__ constexpr int sigma_lns_blocks() { return NPT == 1 ? kSGrid : 0; }
template <int NPT>
__host__ __device__ constexpr int sigma_gtab_blocks() { return (kGTabCount + 256 * NPT - 1) / (256 * NPT); }

template <bool BAO, int NPT>
__global__ __launch_bounds__(256) void k_sigma_nodes(chomp_config cfg,
                                                     const chomp_cosmo* __restrict__ cosmo,
                                                     const double* __restrict__ zin,
                                                     const int* __restrict__ first,
                                                     const int* __restrict__ slots, int n_slots,
                                                     int n_epoch, Epoch* __restrict__ epochs,
                                                     double* __restrict__ snodes,
                                                     unsigned* __restrict__ status) {
  static_assert(NPT == 1 || NPT == 4, "nodes per thread");
  static_assert(kSGrid % 4 == 0, "four ln S points to a block");
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
    epoch_background(B, cfg.cosmo_precision, cfg.k_min, cfg.k_max, BAO ? 1 : 0);
    B.cosmo_slot = slots[e];
    epochs[e] = B;
    status[e] = 0u;
    return;
  }
  const int slot = blockIdx.y, e = first[slot];
  constexpr int nb = sigma_node_blocks<NPT>();    // node-table blocks
  constexpr int nl = sigma_lns_blocks<NPT>();     // ln S blocks
  if (threadIdx.x == 0) {
    const chomp_cosmo c = cosmo[e];
    E.om0 = c.omega_m0; E.ob0 = c.omega_b0; E.ol0 = c.omega_l0; E.or0 = c.omega_r0;
    E.tcmb = c.cmb_temp; E.h = c.h; E.sigma8 = c.sigma_8; E.ns = c.n_scalar;
    E.z = zin[e];
    epoch_shape_only(E, cfg.k_min, cfg.k_max, BAO ? 1 : 0);    // (amp = 1: bare integrals)
  }
  __syncthreads();
  double* n = snodes + (size_t)slot * kSigmaStride;
  if ((int)blockIdx.x >= nb + nl) {               // ---- interpolation table of the direct path
    const double xlo = log(cfg.k_min / 100.0), xhi = log(cfg.k_max * 100.0);
#pragma unroll
    for (int r = 0; r < NPT; ++r) {
      const int i = (((int)blockIdx.x - nb - nl) * NPT + r) * 256 + (int)threadIdx.x;
      if (i < kGTabCount) {
        const double x = xlo + (xhi - xlo) * ((double)(i - kGTabPad) / (double)kGTabN);
        const double k = exp(x);
        const double T = transfer_t<BAO>(E, k);
        const double k3 = k * k * k;
        n[kSigmaOffG + i] = exp((3.0 + E.ns) * (x - E.ln_H0)) * T * T / (k3 * k3);
        n[kSigmaOffG + kGTabCount + i] = k;
      }
    }
    return;
  }
  if ((int)blockIdx.x >= nb) {
    // ln S point (the ln S points only aim the search: 1e-5 and at most 2^10 panels are plenty
    // -- only R > 100 Mpc/h would go on to 2^11, where the cap costs ~1e-4 of an estimate that
    // the probes certify anyway)
    const int i = (int)blockIdx.x - nb;
    const double R = exp(make_sgrid(cfg.k_min, cfg.k_max).ln_r(i));
    double lo, hi;
    sigma_limits(E, R, &lo, &hi);
    SigmaIntegrandT<BAO> f{&E, R};              // sigma_norm = 1: amp * integral
    const double s2 = romberg1<4>(f, lo, hi, cfg.global_precision, 1e-5,
                                  cfg.divmax < 10 ? cfg.divmax : 10, red);
    if (threadIdx.x == 0) n[kSigmaOffLnS + i] = log(s2);
    return;
  }
  const double a = log(cfg.k_min), b = log(cfg.k_max);
  // sigma_8 on the table's range? (cosmology.py:611-632 keeps [k_min, k_max] for R = 8
  // with any sensible limits; otherwise the finisher integrates directly)
  const bool tab8 = 0.1 / 8.0 > cfg.k_min && 14.0662 / 8.0 < cfg.k_max && cfg.divmax >= 1;
  // per-level sums of this block's nodes (level-major order: a run of 256 nodes holds <= 2
  // levels, the first run levels 0..8)
  __shared__ double wsum[4][kSigmaLevel + 1];
  __shared__ int last_block;
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) <= kSigmaLevel) wsum[wv][threadIdx.x & 63] = 0.0;
#pragma unroll
  for (int r = 0; r < NPT; ++r) {
    const int first_idx = ((int)blockIdx.x * NPT + r) * 256;
    const int idx = first_idx + (int)threadIdx.x;
    const bool live = idx < kSigmaCount;
    int lev = 0;
    double g = 0.0;
    if (live) {
      double x;
      if (idx < 2) {
        x = idx == 0 ? a : b;
      } else {
        const int m = idx - 1;
        lev = 32 - __builtin_clz((unsigned)m);
        const long j = m - (1 << (lev - 1));
        const double h = ldexp(b - a, 1 - lev);
        x = (a + 0.5 * h) + h * (double)j;
      }
      const double k = exp(x);
      const double T = transfer_t<BAO>(E, k);
      n[idx] = k;
      // Delta^2 shape over k^6: W(kR)^2 = 9 (sin y - y cos y)^2 / (k R)^6 then needs no division
      const double k3 = k * k * k;
      const double d2k6 = exp((3.0 + E.ns) * (x - E.ln_H0)) * T * T / (k3 * k3);
      n[kSigmaCount + idx] = d2k6;
      const double t = tophat_numer_pm(8.0 * k);
      g = d2k6 * (9.0 / 262144.0) * (t * t);        // 9 / 8^6
    }
    if (first_idx < kSigmaCount) {                  // (block-uniform)
      const int last_idx = first_idx + 255;
      const int lev_lo = first_idx < 2 ? 0 : 32 - __builtin_clz((unsigned)(first_idx - 1));
      const int lev_hi = last_idx < 2 ? 0 : 32 - __builtin_clz((unsigned)(last_idx - 1));
      for (int l = lev_lo; l <= lev_hi && l <= kSigmaLevel; ++l) {
        const double v = wave_sum(live && lev == l ? g : 0.0);
        if ((threadIdx.x & 63) == 0) wsum[wv][l] += v;
      }
    }
  }
  double* part = n + kSigmaOffPart + (size_t)blockIdx.x * (kSigmaLevel + 1);
  __syncthreads();
  if (threadIdx.x <= kSigmaLevel) {
    // (agent-scope stores: the only thing another block of this launch reads -- the last one,
    //  with agent-scope loads -- written through; each writer waits for its own store, the
    //  barrier below for all of them: no fence, i.e. no write-back of the XCD's whole L2, in
    //  front of the arrival count.  The tables are read by the next launch.)
    __hip_atomic_store(part + threadIdx.x,
                       ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) +
                           wsum[3][threadIdx.x],
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if constexpr (NPT != 1) {
    return;                        // (k_sigma_lns, behind this launch, finishes sigma_8)
  } else {
    int* arrivals = reinterpret_cast<int*>(n + kSigmaOffI8 + 1);
    __syncthreads();
    if (threadIdx.x == 0) last_block = atomicAdd(arrivals, 1) == nb - 1 ? 1 : 0;
    __syncthreads();
    if (!last_block) return;
    // ---- last node block of this cosmology: sigma_8 from the level sums (agent-scope loads)
    if (threadIdx.x == 0) *arrivals = 0;
    sigma8_from_parts<BAO>(cfg, E, n, nb, red);
  }
}

// The interpolated integrand with the g table in LDS and k from exp (k_sigma_lns).
struct SigmaInterpLds {
  static constexpr bool kLaneMajor = true;   // (the six taps below: a gather from LDS at an index ~ ln k)
  static constexpr bool kGeometric = true;   // (k = exp(ln k): from the lane's previous node by a multiplication)
  const double* g;         // g / k^6 on the grid, in LDS (index 0 = first pad point)
  double xlo, dx, inv_dx, scale, nine_over_r6;
  bool tiny_r;
  SinCosLead lead = sincos_lead();
  __device__ __forceinline__ double operator()(double ln_k) const { return with_exp(ln_k, exp(ln_k)); }
  __device__ __forceinline__ double with_exp(double ln_k, double k) const {
    const double u = (ln_k - xlo) * inv_dx;
    int i = (int)u;
    i = i < 0 ? 0 : (i > kGTabN - 1 ? kGTabN - 1 : i);
    const double t = u - (double)i;
    const double* q = g + i + kGTabPad - 2;
    const double a = t + 2.0, b = t + 1.0, d = t - 1.0, e = t - 2.0, f = t - 3.0;
    const double ab = a * b, ef = e * f, cd = t * d;
    const double gv = q[0] * (b * cd * ef) * (-1.0 / 120.0) + q[1] * (a * cd * ef) * (1.0 / 24.0) +
                      q[2] * (ab * d * ef) * (-1.0 / 12.0) + q[3] * (ab * t * ef) * (1.0 / 12.0) +
                      q[4] * (ab * cd * f) * (-1.0 / 24.0) + q[5] * (ab * cd * e) * (1.0 / 120.0);
    const double kR = scale * k;
    if (tiny_r) {
      double s, c;
      fast_sincos_pm(kR, &s, &c);
      const double kR2 = kR * kR, k3 = k * k * k;
      const double W = 3.0 * (s / (kR2 * kR) - c / kR2);
      return (gv * (k3 * k3)) * (W * W);
    }
    const double w = tophat_numer_pm(kR, lead);
    return nine_over_r6 * gv * (w * w);
  }
};

// k_sigma_lns: grid n_slots, block 512, dynamic LDS kGTabCount doubles.  sigma_8's integral and
// the coarse ln S(R) table
// of every cosmology of a large batch (behind k_sigma_nodes<.., 4>, which leaves it out): 48
// Romberg integrals, one per wavefront, six rounds of eight, on the integrand interpolated from
// the cosmology's g table -- a third of the instructions of the direct evaluation -- which the
// block first stages into LDS: read from HBM they cost a thousand cosmologies 6 GB of scattered
// 64-byte fetches (measured: 5 ms), staged 67 MB of streaming reads.
constexpr int kLnsThreads = 512;
template <bool BAO>
__global__ __launch_bounds__(kLnsThreads) void k_sigma_lns(chomp_config cfg,
                                                           const chomp_cosmo* __restrict__ cosmo,
                                                           const double* __restrict__ zin,
                                                           const int* __restrict__ first,
                                                           double* __restrict__ snodes) {
  extern __shared__ __align__(16) double gl[];
  __shared__ Epoch E;
  const int slot = blockIdx.x, e = first[slot];
  double* n = snodes + (size_t)slot * kSigmaStride;
  if (threadIdx.x == 0) {
    const chomp_cosmo c = cosmo[e];
    E.om0 = c.omega_m0; E.ob0 = c.omega_b0; E.ol0 = c.omega_l0; E.or0 = c.omega_r0;
    E.tcmb = c.cmb_temp; E.h = c.h; E.sigma8 = c.sigma_8; E.ns = c.n_scalar;
    E.z = zin[e];
    epoch_shape_only(E, cfg.k_min, cfg.k_max, BAO ? 1 : 0);    // (amp = 1: bare integrals)
  }
  if constexpr (!BAO) copy_doubles(gl, n + kSigmaOffG, kGTabCount);
  __syncthreads();
  // sigma_8 from the node blocks' level sums (k_sigma_nodes<.., 4> leaves that to this kernel:
  // a fence and an arrival count per block cost a thousand cosmologies more than the tables)
  sigma8_from_parts<BAO>(cfg, E, n, sigma_node_blocks<4>(), nullptr);
  __syncthreads();
  const SGrid G = make_sgrid(cfg.k_min, cfg.k_max);
  const double xlo = log(cfg.k_min / 100.0), xhi = log(cfg.k_max * 100.0);
  const double dx = (xhi - xlo) / (double)kGTabN;
  const int dmax = cfg.divmax < 10 ? cfg.divmax : 10;
  for (int i = (int)(threadIdx.x >> 6); i < kSGrid; i += kLnsThreads / 64) {
    const double R = exp(G.ln_r(i));
    double lo, hi;
    sigma_limits(E, R, &lo, &hi);
    double s2;
    if constexpr (BAO) {
      SigmaIntegrandT<BAO> f{&E, R};              // (the wiggles are not smooth enough to interpolate)
      s2 = romberg1<1>(f, lo, hi, cfg.global_precision, 1e-5, dmax, nullptr);
    } else {
      const double r3 = R * R * R;
      SigmaInterpLds f{gl, xlo, dx, 1.0 / dx, R, 9.0 / (r3 * r3), 100.0 * E.k_max * R < 1.0};
      if (dmax >= 6) {             // levels 0..6 in one pass (seven dependent round trips less)
        Scalar1<SigmaInterpLds> w{f};
        static_assert(detail::lane_major<Scalar1<SigmaInterpLds>>::value, "lane-major level loop");
        const double fb[1] = {f(hi)};
        s2 = romberg_wave6<1>(w, lo, hi, fb, cfg.global_precision, 1e-5, dmax).value[0];
      } else {
        s2 = romberg1<1>(f, lo, hi, cfg.global_precision, 1e-5, dmax, nullptr);
      }
    }
    if ((threadIdx.x & 63) == 0) n[kSigmaOffLnS + i] = log(s2);
  }
}

// Delta^2(k) W(kR)^2 / (amp sigma_norm^2) from the table (levels <= kSigmaLevel); beyond it
// the nodes no longer come from the level-major table but the R-independent factor is still
// interpolated (SigmaInterpIntegrand; the wiggle transfer function: evaluated directly) --
// the integrals of the largest masses walk on to level 14, whose 8192 new nodes would
// otherwise cost four times a table node each.
template <bool BAO>
struct SigmaTabIntegrand {
  const Epoch* e;
  UniformDoubles node;      // this cosmology's table: k_j ...
  UniformDoubles node_g;    // ... then (k_j/H0)^(3+n) T^2 / k_j^6
  double scale, inv_amp;
  double nine_over_r6;     // 9 / R^6
  SigmaInterpIntegrand deep;   // (same normalisation: amp2 = 1)
  __device__ __forceinline__ void operator()(double ln_k, double (&out)[1], int lev,
                                             long j) const {
    if (lev <= kSigmaLevel) {
      // (& 0x3fff: no-op on a node index, < kSigmaCount; it shows the code generator that the
      //  byte offset fits the 32-bit vector offset of a load with a scalar base)
      const unsigned idx = (unsigned)node_index(lev, j) & 0x3fffu;
      const double kR = scale * node[idx];
      const double t = tophat_numer_pm(kR, deep.lead);
      out[0] = node_g[idx] * nine_over_r6 * (t * t);
    } else if (BAO) {
      SigmaIntegrandT<BAO> f{e, scale};
      out[0] = f(ln_k) * inv_amp;
    } else {
      out[0] = deep(ln_k);
    }
  }
};

// sigma^2(R) = int dlnk Delta^2 W^2 (cosmology.py:602-642) with the whole group on
// one Romberg integral; `rtol` is cosmo_precision for reference-exact values.
// UNROLL > 1 overlaps the table loads of several nodes (worth it where few blocks share a
// CU, as in k_epoch_init; with many resident blocks the extra registers cost more).
template <int NW, int UNROLL = 1, bool BAO = false>
__device__ __forceinline__ double sigma2_block(const Epoch& E, const double* snode, double R,
                                               const chomp_config& cfg, double rtol,
                                               double* red, bool* converged = nullptr,
                                               const RombergLoose* loose = nullptr,
                                               int* level_out = nullptr) {
  const double need_min = 1.0 / R / 10.0, need_max = 1.0 / R * 14.0662;
  const double amp2 = E.amp * E.sigma_norm * E.sigma_norm;
  if (need_min > E.k_min && need_max < E.k_max) {          // fixed range: table path
    // (sigma_limits leaves [k_min, k_max] as it is here: its logarithms are the record's)
    const double lo = E.ln_k_min, hi = E.ln_k_max;
    const double r3 = R * R * R;
    const double nine_r6 = 9.0 / (r3 * r3);
    SigmaTabIntegrand<BAO> f{&E, uniform_ptr(snode), uniform_ptr(snode + kSigmaCount), R, 1.0 / amp2,
                             nine_r6,
                             SigmaInterpIntegrand{uniform_ptr(snode + kSigmaOffG), E.gtab_xlo, E.gtab_dx,
                                                  E.gtab_inv_dx, R, nine_r6, 1.0, false}};
    RombergLoose ls{0.0, 0.0, 0.0, 0.0, 0.0};          // (the integral is sigma^2 / amp2 here)
    if (loose) ls = RombergLoose{loose->rtol, loose->lo1 / amp2, loose->hi1 / amp2,
                                 loose->lo2 / amp2, loose->hi2 / amp2};
    RombergOut<1> r;
    bool fused = false;
    if constexpr (NW == 1) fused = cfg.divmax >= 6;
    if (fused) {
      // one wavefront: levels 0..6 in ONE pass (lane p on node p of the level-6 grid, the upper
      // end point by every lane) and their rows replayed -- walked one by one they are seven
      // dependent round trips of ~2 us each with a node or less per lane, half of a typical
      // integral of the nu table
      double fb[1];
      f(hi, fb, 0, 1);
      r = romberg_wave6<1>(f, lo, hi, fb, cfg.global_precision, rtol, cfg.divmax, nullptr,
                           loose ? &ls : nullptr);
    } else {
      r = romberg_group<NW, 1, SigmaTabIntegrand<BAO>, UNROLL>(
          f, lo, hi, cfg.global_precision, rtol, cfg.divmax, red, nullptr, loose ? &ls : nullptr);
    }
    if (converged) *converged = r.converged[0];
    if (level_out) *level_out = r.level[0];
    return amp2 * r.value[0];
  }
  double lo, hi;
  sigma_limits(E, R, &lo, &hi);
  if constexpr (!BAO) {
    const double r3 = R * R * R;
    SigmaInterpIntegrand f{uniform_ptr(snode + kSigmaOffG), E.gtab_xlo, E.gtab_dx, E.gtab_inv_dx, R,
                           amp2 * 9.0 / (r3 * r3), amp2, 100.0 * E.k_max * R < 1.0};
    Scalar1<SigmaInterpIntegrand> w{f};
    RombergOut<1> r;
    bool fused = false;
    if constexpr (NW == 1) fused = cfg.divmax >= 6;
    if (fused) {
      const double fb[1] = {f(hi)};
      r = romberg_wave6<1>(w, lo, hi, fb, cfg.global_precision, rtol, cfg.divmax, nullptr, loose);
    } else {
      r = romberg_group<NW, 1>(w, lo, hi, cfg.global_precision, rtol, cfg.divmax, red, nullptr, loose);
    }
    if (converged) *converged = r.converged[0];
    if (level_out) *level_out = 100 + r.level[0];     // (100 +: the interpolated integrand)
    return r.value[0];
  } else {
    SigmaIntegrandT<BAO> f{&E, R};
    Scalar1<SigmaIntegrandT<BAO>> w{f};
    const RombergOut<1> r = romberg_group<NW, 1>(w, lo, hi, cfg.global_precision, rtol, cfg.divmax, red,
                                                 nullptr, loose);
    if (converged) *converged = r.converged[0];
    return r.value[0];
  }
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
template <int NW, int UNROLL = 1, bool BAO = false>
__device__ __forceinline__ double nu_of_mass_block(const Epoch& E, const double* snode,
                                                   double mass, const chomp_config& cfg,
                                                   double rtol, double* red,
                                                   bool* converged = nullptr,
                                                   const RombergLoose* loose = nullptr,
                                                   int* level_out = nullptr) {
  const double s2 = sigma2_block<NW, UNROLL, BAO>(E, snode, scale_of_mass(E, mass), cfg, rtol, red,
                                                  converged, loose, level_out);
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

// nu(M) at a probe.  The probe only has to decide on which side of the band edges nu lies,
// so the Romberg may stop at a looser tolerance as soon as its result is clear of both edges
// by kAmbiguous (in ln nu); a result within that of an edge walks on to the reference's
// tolerance (RombergLoose: the same rows, no second integral), so every comparison that
// decides the stopping step is either clear of the edge or exact.
// nu(M) at a probe.  The probe only has to decide on which side of the band edges nu lies,
// so the Romberg may stop at a looser tolerance as soon as its result is clear of both edges
// by kAmbiguous (in ln nu); a result within that of an edge walks on to the reference's
// tolerance (RombergLoose: the same rows, no second integral), so every comparison that
// decides the stopping step is either clear of the edge or exact.
template <int NW, bool BAO>
__device__ __noinline__ double nu_probe(const Epoch& E, const double* snode, double m,
                                           const chomp_config& cfg, double thr_lo,
                                           double thr_hi, double* red) {
  // (a cosmo_precision looser than the probe tolerance is used as it is: the reference's
  // decision rests on exactly that integral)
  const double kAmbiguous = 2e-5;    // (4e-6 measured: no probe of configs[1] falls in either window)
  const double rtol_probe = 1e-6;
  if (!(cfg.cosmo_precision < rtol_probe))
    return nu_of_mass_block<NW, 1, BAO>(E, snode, m, cfg, cfg.cosmo_precision, red);
  // nu = delta_c^2 / sigma^2: the windows in sigma^2
  const double d2 = E.delta_c * E.delta_c;
  const double up = 1.0 + kAmbiguous, dn = 1.0 - kAmbiguous;
  const RombergLoose loose{rtol_probe, d2 / thr_hi * dn, d2 / thr_hi * up, d2 / thr_lo * dn,
                           d2 / thr_lo * up};
  // (two nodes in flight: their table loads overlap; more costs registers, not time)
  return nu_of_mass_block<NW, 2, BAO>(E, snode, m, cfg, cfg.cosmo_precision, red, nullptr, &loose);
}

// The bracketing secant search on exact integrals (whole block).  Returns the mass the
// reference's walk stops at; *n_eval counts the sigma integrals.  Seeds (from probes that
// did not certify the estimate): seed_dir != 0 fixes the walking direction; candidate
// seed_jl is known to FAIL the threshold test (nu_l: its nu, exact or estimated -- it only
// steers the secant); seed_jh >= 0 is known to PASS it with exact nu_h.
template <int NW, bool BAO>
__device__ double search_side_exact(const Epoch& E, const double* snode, int side,
                                    const chomp_config& cfg, const double* cand, double* red,
                                    int* n_eval, bool* exhausted, int seed_dir = 0,
                                    int seed_jl = 0, double nu_l = 0.0, int seed_jh = -1,
                                    double nu_h = 0.0) {
  const SideThresholds T = side_thresholds(side, cand);
  double mass = T.down[0];
  int dir = seed_dir;
  double nu0 = nu_l;
  if (seed_dir == 0) {
    nu0 = nu_probe<NW, BAO>(E, snode, T.down[0], cfg, T.thr_lo, T.thr_hi, red);
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
      const double nu = nu_probe<NW, BAO>(E, snode, tab[jp], cfg, T.thr_lo, T.thr_hi, red);
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
        if (jh < 0 && jp == kSearchJ - 1) { jh = jp; *exhausted = true; break; }   // table exhausted
      }
    }
    if (jh < 0) jh = jl;
    mass = tab[jh];
  }
  return mass;
}

// Status bits of a finished walk (chomp_get_status).  mass_min: once k R < 0.2 over sigma_r's
// whole k range (clamped at 100 k_max, cosmology.py:627-632) the top-hat window is 1 to
// 4e-3, nu(M) has converged to a constant -- above the band, or the walk would have ended
// earlier -- and only the rounding error of 3 (sin x / x^3 - cos x / x^2) at small x, whose
// variance biases sigma^2 upwards like eps^2 / x^4, ends the reference's walk: the step it
// stops at is a property of the libm in use (DESIGN.md "Known limit of parity").  At
// k R = 0.2 a 5 % step still moves nu by 3e-5 against a rounding error of 1e-7; a factor two
// below that the two are equal.  mass_max: the walk ended with sigma_r's range clamped at
// k_min / 100 (cosmology.py:617-622).
__device__ __forceinline__ unsigned search_status(const Epoch& E, int side, double mass,
                                                  bool exhausted) {
  const double R = scale_of_mass(E, mass);
  unsigned st = 0u;
  if (side == 0 && 100.0 * E.k_max * R < 0.2) st |= kStMassMinSaturated;
  if (side == 1 && 0.1 / R <= E.k_min / 100.0) st |= kStMassMaxSaturated;
  if (exhausted) st |= kStSearchExhausted;
  return st;
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
  const double ln_margin = 0.00995033085; // ln(1 + margin)
  SidePlan P{false, 0, 0, 0.0, false};
  // ln nu = ln_nu_c - ln S(ln R); the candidates are M_0 * 1.05^(-+j), so ln R moves by
  // ln(1.05) / 3 per step (to the estimate's accuracy).  The four logarithms this needs are
  // one call with a different argument in each of four lanes.
  const int lane = threadIdx.x & 63;
  const double arg = lane == 0 ? E.delta_c * E.delta_c / (E.amp * E.sigma_norm * E.sigma_norm)
                   : lane == 1 ? 3.0 * T.down[0] / (4.0 * kPi * E.rho_bar)
                   : lane == 2 ? T.thr_hi : T.thr_lo;
  const double lg = log(arg);
  const double ln_nu_c = readlane_d(lg, 0), x0 = readlane_d(lg, 1) * (1.0 / 3.0);
  const double ln_hi = readlane_d(lg, 2), ln_lo = readlane_d(lg, 3);
  const double step = 0.016263388 /* ln(1.05) / 3 */;
  const double ln_nu0 = ln_nu_c - ln_s_estimate(G, lns, x0);
  P.nu_start = exp(ln_nu0);
  if (!(ln_nu0 == ln_nu0)) return P;
  if (ln_nu0 > ln_hi + ln_margin) P.dir = -1;
  else if (ln_nu0 < ln_lo - ln_margin) P.dir = +1;
  else if (ln_nu0 >= ln_lo + ln_margin && ln_nu0 <= ln_hi - ln_margin) {
    P.ok = true;                          // inside the band with room to spare
    return P;
  } else {                                // on an edge: probe the start and its neighbours
    P.ok = true;
    P.at_edge = true;
    P.dir = P.nu_start > 0.5 * (T.thr_lo + T.thr_hi) ? -1 : +1;
    P.j = 2;
    return P;
  }
  (void)margin;
  const double ln_thr = P.dir < 0 ? ln_hi : ln_lo;
  // nu is monotone along the table, so the first passing index is the minimum over the
  // passing ones; a candidate outside the ln S table counts as passing, and is rejected
  // below if it turns out to be the first.  Two rounds: every (kSearchJ / 256)-th candidate,
  // then the candidates between the last failing and the first passing one of those.
  auto passes = [&](int j) {
    const double ln_nu = ln_nu_c - ln_s_estimate(G, lns, x0 + (double)(P.dir * j) * step);
    return !(ln_nu == ln_nu) || (P.dir < 0 ? !(ln_thr < ln_nu) : !(ln_thr > ln_nu));
  };
  constexpr int kStride = kSearchJ / 256;
  __syncthreads();
  if (threadIdx.x == 0) *sh = kSearchJ;
  __syncthreads();
  for (int t = (int)threadIdx.x; t < 256; t += (int)blockDim.x) {   // (one round for a block of 256)
    const int j = kStride * (1 + t) - 1;                       // 7, 15, ..., 2047
    if (j < kSearchJ && passes(j)) atomicMin(sh, j);
  }
  __syncthreads();
  const int jc = *sh;                     // first passing coarse candidate (or kSearchJ)
  __syncthreads();
  if (jc < kSearchJ && (int)threadIdx.x < kStride - 1) {
    const int j = jc - (kStride - 1) + (int)threadIdx.x;       // jc - 7 .. jc - 1
    if (j >= 1 && passes(j)) atomicMin(sh, j);
  }
  __syncthreads();
  const int j = *sh;
  if (j >= kSearchJ) return P;                   // nothing passes: exact search
  const double lsj = ln_s_estimate(G, lns, x0 + (double)(P.dir * j) * step);
  if (!(lsj == lsj)) return P;                   // the walk leaves the ln S table
  P.j = j;
  P.ok = true;
  return P;
}

// k_epoch_probe (chomp_probe_kernel.h) is compiled in its own translation unit, chomp_probe.hip
// -- WITH LLVM's machine LICM, which every other kernel of the library is faster without
// (chomp_amd/_lib.py: HIPCC_FLAGS): 38.5 against 41.8 us per C2 launch.  This is its launcher.
void launch_epoch_probe(bool bao, unsigned n_epoch, hipStream_t stream, const chomp_config& cfg,
                        Epoch* epochs, double* search, const double* cand, const double* snodes,
                        double* probe, int* count, unsigned* status);
constexpr int kProbes = 4;
constexpr int kProbeStride = 24;   // doubles per epoch: nu[2][kProbes], chi, pad[3], plan[2][4]

// ---------------------------------------------------------------------------
// k_nu_table: one sigma(R) Romberg of the nu table per block of 64 NW threads,
// nu_i = nu_m(exp(ln_mass_i)) (mass_function.py:205-210).  NW = 1 (one wavefront per integral:
// most integrals in flight) for a batch of epochs; NW = 4 when the whole launch is a few dozen
// integrals (one epoch) and lasts as long as one of them.
// epochs_fastest != 0: grid (n_epoch, NM), the LARGEST mass first -- its integral is the longest
// of an epoch's fifty (level 12 where the others stop at 10-11), and a launch of single-wavefront
// integrals ends with whatever was dispatched last (a (k, z) grid: one cosmology, one sigma(R)
// node table for everybody).  epochs_fastest == 0: grid (NM, n_epoch), an epoch's fifty masses
// side by side -- a batch of many cosmologies, where the blocks in flight should share as few of
// the 131 KB node tables as possible (1024 cosmologies, epochs fastest: 600 us against 420).
// ---------------------------------------------------------------------------
template <bool BAO, int NW>
__global__ __launch_bounds__(64 * NW) void k_nu_table(chomp_config cfg, TabLayout L,
                                                      const Epoch* __restrict__ epochs,
                                                      const double* __restrict__ search,
                                                      const double* __restrict__ snodes,
                                                      double* __restrict__ tab,
                                                      unsigned* __restrict__ status,
                                                      int epochs_fastest) {
  __shared__ Epoch E;
  __shared__ double red[romberg_scratch<NW, 1>()];
  int e = epochs_fastest ? (int)blockIdx.x : (int)blockIdx.y;
  int i = (int)blockIdx.x;
  if (!epochs_fastest) {
    // A cosmology per epoch: an epoch's NM integrals read the same 131 KB node table, and
    // workgroups go to the eight XCDs -- eight L2s -- round-robin by linear index: (mass, epoch)
    // order puts every table into all eight.  Linear block b runs on XCD b mod 8; within an XCD
    // the blocks walk through (epoch, mass) with the epochs = that XCD mod 8 -- one L2 per table
    // (1024 cosmologies: 1 GB of table fills per launch down to 134 MB).
    const int NM = (int)gridDim.x, n = (int)gridDim.y;
    const int b = (int)blockIdx.x + NM * (int)blockIdx.y;
    if (b < 8 * NM * (n / 8)) {
      const int slot = b >> 3;
      e = (slot / NM) * 8 + (b & 7);
      i = slot % NM;
    }
  }
  if (epochs_fastest) {
    // Largest mass first -- but the first two and the last two rows take four masses from the
    // MIDDLE of the table.  Blocks go to the SIMDs round-robin (1024 of them: a SIMD gets linear
    // blocks b, b + 1024, b + 2048, ...), a 64-epoch launch is 3200 wavefronts, so the SIMDs of
    // the first two rows get a fourth one from the last two; the SIMDs run at their VALU issue
    // rate from the first microsecond to the last (tools/dev_nu_stamps3.py), so the launch lasts
    // as long as the most loaded of them -- and the largest masses (Romberg level 12) are the
    // longest integrals, the middle ones (level 10-11 on the node table) the shortest.
    const int NM = (int)gridDim.y, y = (int)blockIdx.y, m0 = NM / 2 - 1;
    if (NM < 8) i = NM - 1 - y;
    else if (y < 2) i = m0 + y;
    else if (y >= NM - 2) i = m0 + 2 + (y - (NM - 2));
    else { i = NM - 1 - (y - 2); if (i <= m0 + 3) i -= 4; }
  }
  NUSTAMP(0, __builtin_amdgcn_s_memrealtime());
  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),
               kEpochDoubles);
  __syncthreads();
  NUSTAMP(1, __builtin_amdgcn_s_memrealtime());
  const double* snode = snodes + (size_t)E.cosmo_slot * kSigmaStride;
  const double ln_lo = search[(e * 2 + 0) * 2], ln_hi = search[(e * 2 + 1) * 2];
  const double lnm = linspace_at(ln_lo, ln_hi, L.NM, i);
  bool conv = true;
#if defined(CHOMP_STAMPS) && CHOMP_STAMPS == 3
  int lev_dbg = 0;
  const double nu = nu_of_mass_block<NW, 1, BAO>(E, snode, exp(lnm), cfg, cfg.cosmo_precision,
                                                 red, &conv, nullptr, &lev_dbg);
  NUSTAMP(1, lev_dbg);
#else
  const double nu = nu_of_mass_block<NW, 1, BAO>(E, snode, exp(lnm), cfg, cfg.cosmo_precision,
                                                 red, &conv);
#endif
  NUSTAMP(2, __builtin_amdgcn_s_memrealtime());
  NUSTAMP(3, (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32));   // HW_ID, XCC_ID
  if (threadIdx.x == 0) {
    double* t = tab + (size_t)e * L.stride;
    t[L.off_ln_mass + i] = lnm;
    t[L.off_nu + i] = nu;
    if (!conv) atomicOr(&status[e], kStSigmaDivmax);     // scipy: AccuracyWarning, last row kept
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
// mass_setup_block: MassFunction._initialize_splines / _normalize for one epoch
// (mass_function.py:212-241, Tinker 532-564) by a whole block of 256 threads, once the
// epoch's nu table is complete.  E: the epoch record in LDS (completed here; written back
// with the spline coefficients if `publish`); M: the LDS carve-up below, which afterwards holds the ln M grid, the nu knots and
// the pp coefficients of nu(ln M) and ln M(nu) for whoever goes on in the same block.
// ---------------------------------------------------------------------------
struct MassLds {
  double *x_lnm, *y_nu, *c_nu, *c_lnm, *work, *gl, *red;
  __device__ __forceinline__ void carve(double* sm, int NM) {
    x_lnm = sm;                      // [NM]
    y_nu = x_lnm + NM;               // [NM]
    c_nu = y_nu + NM;                // [4(NM-1)]
    c_lnm = c_nu + 4 * (NM - 1);     // [4(NM-1)]
    work = c_lnm + 4 * (NM - 1);     // [18 NM]
    gl = work + 18 * NM;             // [32]
    red = gl + 32;                   // [romberg_scratch<4, 2>()]
  }
};
__host__ __device__ inline int mass_lds_doubles(int NM) {
  return 2 * NM + 8 * (NM - 1) + 18 * NM + 32 + romberg_scratch<4, 2>();
}

__device__ __forceinline__ void mass_setup_block(
    const chomp_config& cfg, const TabLayout& L, Epoch& E, Epoch* __restrict__ epochs, int e,
    bool publish, double ln_mass_min, double ln_mass_max, int n_search, double* __restrict__ t,
    const chomp_halo_par& hp, int mf_kind, const TinkerTab* __restrict__ tinker,
    const double* __restrict__ gl16, const MassLds& M) {
  const int NM = L.NM;
  copy_doubles(M.x_lnm, t + L.off_ln_mass, NM);
  copy_doubles(M.y_nu, t + L.off_nu, NM);
  copy_doubles(M.gl, gl16, 32);
  __syncthreads();
  MSTAMP(1);
  {   // nu(ln M) on threads 0..127, ln M(nu) on threads 128..255, in lockstep
    const int sys = threadIdx.x >> 7, tid = threadIdx.x & 127;
    spline_build_pcr(sys == 0 ? M.x_lnm : M.y_nu, sys == 0 ? M.y_nu : M.x_lnm, NM,
                     sys == 0 ? M.c_nu : M.c_lnm, M.work + sys * 9 * NM, tid, 128, true);
  }
  MSTAMP(2);
  if (threadIdx.x < 64) {
    // (one wavefront: the five Tinker parameters -- a look-up in the 9-row table from global
    //  memory and a power of 1 + z each, mass_function.py:547-564 -- in five lanes at once, m_star
    //  in a sixth: done one after the other by thread 0 they were 6.7 us of every block's chain
    //  on a Tinker set-up (tools/dev_mass_stamps.py).  Same calls on the same arguments.)
    const int lane = (int)threadIdx.x;
    const double mf_delta_v = (hp.delta_v == -1.0) ? E.delta_v : hp.delta_v;
    double mine = 0.0;
    if (mf_kind == CHOMP_MF_TINKER && lane < 5) {
      const double ld = log(mf_delta_v);
      const double opz = 1.0 + E.z;
      mine = spline_eval(tinker->x, tinker->c[lane], 9, ld);
      const double ex = lane == 1 ? 0.20 : (lane == 2 ? -0.01 : (lane == 3 ? -0.08 : 0.27));
      if (lane > 0) mine *= pow(opz, ex);
    }
    if (lane == 5) mine = exp(spline_eval(M.y_nu, M.c_lnm, NM, 1.0));   // :223
    const double t_alpha = readlane_d(mine, 0), t_beta = readlane_d(mine, 1);
    const double t_gamma = readlane_d(mine, 2), t_phi = readlane_d(mine, 3);
    const double t_eta = readlane_d(mine, 4), m_star = readlane_d(mine, 5);
    if (lane == 0) {
      E.ln_mass_min = ln_mass_min;
      E.ln_mass_max = ln_mass_max;
      E.n_search = n_search;
      E.nu_min = 1.001 * M.y_nu[0];                       // mass_function.py:212-213
      E.nu_max = 0.999 * M.y_nu[NM - 1];
      E.m_star = m_star;
      E.stq = hp.stq;
      E.st_a = hp.st_little_a;
      E.mf_delta_v = mf_delta_v;
      E.mf_kind = mf_kind;
      E.f_norm = 1.0;
      E.bias_norm = 1.0;
      E.ln_st_a = log(hp.st_little_a);
      E.ln_t_beta = 0.0;
      if (mf_kind == CHOMP_MF_TINKER) {
        E.t_alpha = t_alpha;
        E.t_beta = t_beta;
        E.t_gamma = t_gamma;
        E.t_phi = t_phi;
        E.t_eta = t_eta;
        E.ln_t_beta = log(E.t_beta);
        tinker_bias_constants(E);
      }
    }
  }
  __syncthreads();
  MSTAMP(3);
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
  if (literal) {
    if (mf_kind == CHOMP_MF_ST) {
      FnuLin f{&E};
      const double norm = romberg1<4>(f, E.nu_min, E.nu_max, cfg.global_precision,
                                      cfg.mass_precision, cfg.divmax, M.red);
      __syncthreads();
      if (threadIdx.x == 0) E.f_norm = 1.0 / norm;
      __syncthreads();
    }
    FnuBiasLin f{&E};
    const double norm = romberg1<4>(f, E.nu_min, E.nu_max, cfg.global_precision,
                                    cfg.mass_precision, cfg.divmax, M.red);
    __syncthreads();
    if (threadIdx.x == 0) E.bias_norm = 1.0 / norm;
    __syncthreads();
  } else {
    // both integrals in one pass over the 128 nodes (f_norm = bias_norm = 1 in E meanwhile):
    // int f and int f b; the bias integrand of the reference carries the NORMALISED f, i.e.
    // f_norm times the raw integral
    constexpr int NT = 256;
    const double wd = (b - a) / 8.0;
    double p1 = 0.0, p2 = 0.0;
    for (int idx = threadIdx.x; idx < 16 * 8; idx += NT) {
      const int pn = idx >> 4, q = idx & 15;
      const double mid = a + wd * ((double)pn + 0.5);
      const double nu = exp(mid + 0.5 * wd * M.gl[q]);
      const double fn = f_nu(E, nu) * nu;
      p1 += M.gl[16 + q] * fn;
      p2 += M.gl[16 + q] * (fn * bias_nu(E, nu));
    }
    const double i1 = 0.5 * wd * group_sum<4>(p1, M.red, flip);
    const double i2 = 0.5 * wd * group_sum<4>(p2, M.red, flip);
    __syncthreads();
    if (threadIdx.x == 0) {
      const double fnorm = mf_kind == CHOMP_MF_ST ? 1.0 / i1 : 1.0;   // Tinker: f is normalised
      E.f_norm = fnorm;
      E.bias_norm = 1.0 / (fnorm * i2);
    }
    __syncthreads();
  }
  MSTAMP(4);
  if (!publish) return;
  copy_doubles(reinterpret_cast<double*>(&epochs[e]), reinterpret_cast<const double*>(&E),
               kEpochDoubles);
  copy_doubles(t + L.off_nu_pp, M.c_nu, 4 * (NM - 1));
  copy_doubles(t + L.off_lnm_pp, M.c_lnm, 4 * (NM - 1));
}

}  // namespace chomp
