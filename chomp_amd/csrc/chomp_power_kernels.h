// chomp_power_kernels.h -- Stage E (gfx950): evaluation of the spectra from the tables.
//
//   k_power         generic per-sample kernel: Halo.linear_power/power_mm/power_gm/power_gg
//                   (halo.py:266-439), HaloFit.power_* (halo.py:1325-1413)
//   k_power_prep + k_power_stream   row-major streaming of a large (k, z) grid
//   k_power_grid    row-walking streaming kernel (epochs of different cosmologies, small grids)
//   k_power_grid_lanes  per-lane pass for unsorted / ragged / out-of-range k groups
//   k_power_extrap  constants of Halo(extrapolate=True) above k_max
//   k_sigma_r, k_y_nfw, k_eval   point lookups of the mirror classes
#pragma once

#include "chomp_halo_kernels.h"

namespace chomp {

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
// BAO (here and below): the context's transfer function, a compile-time choice in device
// code (chomp_set_transfer; transfer_t in chomp_math.h).
template <bool BAO>
__device__ __forceinline__ double power_tail(const Epoch& E, const double* x, int w, double kv,
                                             double k_max) {
  if (w == CHOMP_P_MM) return linear_power_t<BAO>(E, kv) * x[0];
  const double* vs = w == CHOMP_P_GM ? x + 1 : x + 3;          // value at k_max, log-slope
  return pow(kv / k_max, vs[1]) * vs[0];
}

template <bool BAO>
__device__ __forceinline__ double halofit_mm(const Epoch& E, double k) {
  // halo.py:1339-1360
  const double lk = log(k);
  const double dk = delta_k_ln_t<BAO>(E, lk, k);
  const double y = k / E.hf_k_s;
  const double d2q = dk * (pow(1.0 + dk, E.hf_beta_n) / (1.0 + E.hf_alpha_n * dk) *
                           exp(-(y / 4.0 + y * y / 8.0)));
  const double d2h = (E.hf_a_n * pow(y, 3.0 * E.hf_f1) /
                      (1.0 + E.hf_b_n * pow(y, E.hf_f2) +
                       pow(E.hf_c_n * E.hf_f3 * y, 3.0 - E.hf_gamma_n))) /
                     (1.0 + E.hf_mu_n / y + E.hf_nu_n / (y * y));
  return 2.0 * kPi * kPi / (k * k * k) * (d2q + d2h);
}

// halofit_mm for a caller that holds ln k as well: every power becomes one exp of a
// linear form in ln k, the linear spectrum comes from power_shape (halo.py:1339-1360).
template <bool BAO>
__device__ __forceinline__ double halofit_mm_ln(const Epoch& E, double amp2, double lk, double k) {
  const double k3 = k * k * k;
  const double dk = amp2 * power_shape_t<BAO>(E, lk, k) * k3 * (1.0 / (2.0 * kPi * kPi));
  const double ln_y = lk - log(E.hf_k_s);               // (log of a per-epoch constant)
  const double y = exp(ln_y);
  const double d2q = dk * exp(E.hf_beta_n * fast_log(1.0 + dk) - (y * 0.25 + y * y * 0.125)) /
                     (1.0 + E.hf_alpha_n * dk);
  const double t_b = exp(E.hf_f2 * ln_y);
  const double t_c = exp((3.0 - E.hf_gamma_n) * (log(E.hf_c_n * E.hf_f3) + ln_y));
  const double inv_y = 1.0 / y;
  const double d2h = E.hf_a_n * exp(3.0 * E.hf_f1 * ln_y) /
                     ((1.0 + E.hf_b_n * t_b + t_c) *
                      (1.0 + E.hf_mu_n * inv_y + E.hf_nu_n * inv_y * inv_y));
  return 2.0 * kPi * kPi / k3 * (d2q + d2h);
}

// P(k) of one epoch from tables staged in LDS: shared by k_power and by the
// projection integrands (correlation.py:270-275, 387-392 call halo.power_*).
struct PowerEval {
  const Epoch* E;
  const double *ca, *cb, *cp;     // pp coefficients: h_a, h_b, 1-halo term
  int NK, w;
  bool halofit, extrap;
  const double* tail;             // misc[3..7] of the epoch (k_power_extrap)
  double x0, dx, inv_dx, amp2, k_min, k_max, c_lo;

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
    inv_dx = 1.0 / dx;
    amp2 = 0.0;                     // set by finish(): *Els may still be being staged
    c_lo = 0.0;
  }
  __device__ __forceinline__ bool needs_tables() const {
    return w != CHOMP_P_LIN && !(halofit && w == CHOMP_P_MM);
  }
  // after the barrier: k < k_min constant (halo.py:314-317)
  template <bool BAO>
  __device__ __forceinline__ void finish_t() {
    amp2 = E->amp * E->sigma_norm * E->sigma_norm;
    if (w != CHOMP_P_LIN && !halofit) {
      const double ha = pp_poly(ca, 0, 0.0), hb = pp_poly(cb, 0, 0.0), p0 = pp_poly(cp, 0, 0.0);
      c_lo = ha * hb + p0 / linear_power_t<BAO>(*E, k_min);
    }
  }
  // (the projection kernels only run on no-wiggle contexts)
  __device__ __forceinline__ void finish() { finish_t<false>(); }
  __device__ __forceinline__ double operator()(double kv) const { return eval_t<false>(kv); }
  template <bool BAO>
  __device__ __forceinline__ double eval_t(double kv) const {
    if (w == CHOMP_P_LIN) return linear_power_t<BAO>(*E, kv);
    if (halofit) {
      const double pmm = halofit_mm<BAO>(*E, kv);
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
    if (kv < k_min) return linear_power_t<BAO>(*E, kv) * c_lo;
    if (extrap ? kv < k_max : kv <= k_max) {
      const double lk = log(kv);
      const double ha = spline_eval_uniform(x0, dx, ca, NK, lk);
      const double hb = spline_eval_uniform(x0, dx, cb, NK, lk);
      const double pp = spline_eval_uniform(x0, dx, cp, NK, lk);
      const double plin = 2.0 * kPi * kPi * delta_k_ln_t<BAO>(*E, lk, kv) / (kv * kv * kv);
      return plin * ha * hb + pp;
    }
    if (extrap) return power_tail<BAO>(*E, tail, w, kv, k_max);
    return 0.0;                                           // k > k_max (or NaN)
  }
  // The same spectrum for callers that integrate over ln k (w(theta), xi(r)) and so hold
  // both ln k and k: no logarithm is retaken, the three splines share one interval
  // lookup, and the linear spectrum uses the two-division arrangement of Stage E
  // (power_shape).  Agrees with operator() to rounding.
  // HF: compile-time copy of `halofit`, so that each instance carries one formula only.
  // BAO: the context's transfer function (compile time, as in eval_t).
  template <bool HF, bool BAO = false>
  __device__ __forceinline__ double at_ln(double lk, double kv) const {
    const bool in = kv >= k_min && kv <= k_max;
    if (HF) {
      if (w != CHOMP_P_MM && !in) return 0.0;             // halo.py:649-672 range rule
      const double pmm = halofit_mm_ln<BAO>(*E, amp2, lk, kv);
      if (w == CHOMP_P_MM) return pmm;
      int i = (int)floor((lk - x0) * inv_dx);
      i = i < 0 ? 0 : (i > NK - 2 ? NK - 2 : i);
      const double d = lk - (x0 + dx * (double)i);
      return pmm * pp_poly(ca, i, d) * pp_poly(cb, i, d) + pp_poly(cp, i, d);
    }
    if (w == CHOMP_P_LIN || !in || (extrap && !(kv < k_max))) {
      // the rarely taken branches (eval_t's, without the HaloFit formula this instance never uses)
      if (w == CHOMP_P_LIN) return linear_power_t<BAO>(*E, kv);
      if (kv < k_min) return linear_power_t<BAO>(*E, kv) * c_lo;
      if (extrap && kv >= k_max) return power_tail<BAO>(*E, tail, w, kv, k_max);
      return 0.0;                                         // k > k_max (or NaN)
    }
    int i = (int)floor((lk - x0) * inv_dx);
    i = i < 0 ? 0 : (i > NK - 2 ? NK - 2 : i);
    const double d = lk - (x0 + dx * (double)i);
    const double ha = pp_poly(ca, i, d), hb = pp_poly(cb, i, d), pp = pp_poly(cp, i, d);
    return amp2 * power_shape_t<BAO>(*E, lk, kv) * ha * hb + pp;
  }
};

// Halo(extrapolate=True): the constants of the continuation above k_max, from the knot
// tables of spectrum w (halo.py:300-312: misc[3]; :341-352 / :405-416: value at k_max and
// mean log-slope over knots -7..-1 into misc[4,5] (gm) / misc[6,7] (gg)).  grid n, block 64.
template <bool BAO>
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
    lv[i] = log(linear_power_t<BAO>(E, exp(x)) * ka[j] * kb[j] + kp[j]);
  }
  __syncthreads();
  if (i == 0) {
    const double plin = linear_power_t<BAO>(E, cfg.k_max);
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

template <bool BAO>
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
  P.template finish_t<BAO>();
  double* o = out + (size_t)blockIdx.y * nk;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk;
       i += (size_t)gridDim.x * blockDim.x)
    o[i] = P.template eval_t<BAO>(k[i]);
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
template <bool BAO>
__device__ __forceinline__ double power_lane(const chomp_config& cfg, const TabLayout& L,
                                             const Epoch& E, const double* t, int fa, int fb,
                                             int fp, int w, bool extrap, double kv) {
  if (w == CHOMP_P_LIN) return linear_power_t<BAO>(E, kv);
  const double x0 = log(cfg.k_min);
  const double dx = (log(cfg.k_max) - x0) / (double)(L.NK - 1);
  if (kv < cfg.k_min) {
    const double c_lo = t[L.off_kpp[fa]] * t[L.off_kpp[fb]] +
                        t[L.off_kpp[fp]] / linear_power_t<BAO>(E, cfg.k_min);
    return linear_power_t<BAO>(E, kv) * c_lo;
  }
  if (extrap ? kv < cfg.k_max : kv <= cfg.k_max) {
    const double lk = log(kv);
    const double ha = spline_eval_uniform(x0, dx, t + L.off_kpp[fa], L.NK, lk);
    const double hb = spline_eval_uniform(x0, dx, t + L.off_kpp[fb], L.NK, lk);
    const double pp = spline_eval_uniform(x0, dx, t + L.off_kpp[fp], L.NK, lk);
    return 2.0 * kPi * kPi * delta_k_ln_t<BAO>(E, lk, kv) / (kv * kv * kv) * ha * hb + pp;
  }
  if (extrap) return power_tail<BAO>(E, t + L.off_misc + 3, w, kv, cfg.k_max);
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

// One wavefront's 128 k (two per lane) for the epochs [q_lo, q_hi) on the per-lane path: any
// k, any knot interval, any order.  In-range k still re-use the Eisenstein-Hu shape across
// epochs of one cosmology; k outside [k_min, k_max] take the full formula (halo.py:314-320).
template <bool BAO>
__device__ __forceinline__ void power_lanes_range(const chomp_config& cfg, const TabLayout& L,
                                                  const Epoch* __restrict__ epochs,
                                                  const double* __restrict__ tab, int w,
                                                  bool extrap, int epoch0, int q_lo, int q_hi,
                                                  const KLanes& s, size_t nk,
                                                  double* __restrict__ out) {
  const PowerFam F = power_families(w);
  const int NK = L.NK;
  const double x0 = log(cfg.k_min);
  const double dx = (log(cfg.k_max) - x0) / (double)(NK - 1);
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
      sh0 = power_shape_t<BAO>(E, s.lk0, s.k0);
      sh1 = power_shape_t<BAO>(E, s.lk1, s.k1);
    }
    if (s.have0) {
      double r;
      if (s.in0 && w != CHOMP_P_LIN) {
        const double ha = pp_poly(t + L.off_kpp[F.fa], s.idx0, e0);
        const double hb = pp_poly(t + L.off_kpp[F.fb], s.idx0, e0);
        const double pp = pp_poly(t + L.off_kpp[F.fp], s.idx0, e0);
        r = fma(A * sh0, ha * hb, pp);
      } else {
        r = power_lane<BAO>(cfg, L, E, t, F.fa, F.fb, F.fp, w, extrap, s.k0);
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
        r = power_lane<BAO>(cfg, L, E, t, F.fa, F.fb, F.fp, w, extrap, s.k1);
      }
      o[1] = r;
    }
  }
}

// The row-walking streaming pass (epochs of different cosmologies, or small grids).
// grid (ceil(nk / 512), ceil(n_epoch / epochs_per_y)), block 256.  A wavefront whose k
// do not qualify for the fast path only enters itself in the slow list.
template <bool BAO>
__global__ __launch_bounds__(256) void k_power_grid(chomp_config cfg, TabLayout L,
                                                    const Epoch* __restrict__ epochs,
                                                    const double* __restrict__ tab, int w,
                                                    int epoch0, int n_epoch, int epochs_per_y,
                                                    int rot,
                                                    const double* __restrict__ k, size_t nk,
                                                    double* __restrict__ out,
                                                    int* __restrict__ slow, int parity,
                                                    int inline_lanes, bool extrap) {
  if (!inline_lanes) slow_list_begin(slow, parity);
  const int k_group = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
  const PowerFam F = power_families(w);
  const bool same_ab = F.fa == F.fb;
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
  const int q_lo = blockIdx.y * epochs_per_y;
  int q_hi = q_lo + epochs_per_y;
  if (q_hi > n_epoch) q_hi = n_epoch;
  if (!fast) {
    if (inline_lanes) {            // small grids: this block's epochs on the per-lane path, here
      if ((size_t)k_group * 128 < nk)
        power_lanes_range<BAO>(cfg, L, epochs, tab, w, extrap, epoch0, q_lo, q_hi, s, nk, out);
    } else if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && (size_t)k_group * 128 < nk) {
      slow_list_append(slow, parity, k_group);
    }
    return;
  }
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
      shape0 = power_shape_t<BAO>(E, s.lk0, k0);
      shape1 = power_shape_t<BAO>(E, s.lk1, k1);
    }
    double ha0 = fma(fma(fma(c.a3, d0, c.a2), d0, c.a1), d0, c.a0);
    double pp0 = fma(fma(fma(c.p3, d0, c.p2), d0, c.p1), d0, c.p0);
    double ha1 = fma(fma(fma(c.a3, d1, c.a2), d1, c.a1), d1, c.a0);
    double pp1 = fma(fma(fma(c.p3, d1, c.p2), d1, c.p1), d1, c.p0);
    double hb0 = ha0, hb1 = ha1;   // P_mm and P_gg multiply a 2-halo factor by itself
    if (!same_ab) {                // wave-uniform
      hb0 = fma(fma(fma(c.b3, d0, c.b2), d0, c.b1), d0, c.b0);
      hb1 = fma(fma(fma(c.b3, d1, c.b2), d1, c.b1), d1, c.b0);
    }
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
template <bool BAO>
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
  const double sh0 = power_shape_t<BAO>(E, s.lk0, s.k0), sh1 = power_shape_t<BAO>(E, s.lk1, s.k1);
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
  const bool same_ab = F.fa == F.fb;
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
    double pp0 = fma(fma(fma(c.p3, d0, c.p2), d0, c.p1), d0, c.p0);
    double ha1 = fma(fma(fma(c.a3, d1, c.a2), d1, c.a1), d1, c.a0);
    double pp1 = fma(fma(fma(c.p3, d1, c.p2), d1, c.p1), d1, c.p0);
    double hb0 = ha0, hb1 = ha1;   // P_mm and P_gg multiply a 2-halo factor by itself
    if (!same_ab) {                // wave-uniform
      hb0 = fma(fma(fma(c.b3, d0, c.b2), d0, c.b1), d0, c.b0);
      hb1 = fma(fma(fma(c.b3, d1, c.b2), d1, c.b1), d1, c.b0);
    }
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
// The per-lane pass of the streaming shape: 1-D grid; every wavefront walks work items
// (listed k group, chunk of epochs).  The chunk length adapts to the length of the list: few
// listed groups -> short chunks over many wavefronts (the per-epoch coefficient loads of this
// path are a dependent chain), a fully listed grid -> one item per group.
template <bool BAO>
__global__ __launch_bounds__(256) void k_power_grid_lanes(
    chomp_config cfg, TabLayout L, const Epoch* __restrict__ epochs,
    const double* __restrict__ tab, int w, bool extrap, int epoch0, int n_epoch,
    const double* __restrict__ k, size_t nk, double* __restrict__ out,
    const int* __restrict__ slow, int parity) {
  const int count = slow[parity];
  if (count == 0) return;
  const int NK = L.NK;
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
    power_lanes_range<BAO>(cfg, L, epochs, tab, w, extrap, epoch0, q_lo, q_hi, s, nk, out);
  }
}

// sigma_r at arbitrary scales (SingleEpoch.sigma_r): grid n, block 256.
template <bool BAO>
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
  const double s2 = sigma2_block<4, 1, BAO>(E, snodes + (size_t)E.cosmo_slot * kSigmaStride, scale[blockIdx.x], cfg,
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
      case CHOMP_EV_DELTA_K:
        r = E.with_bao ? delta_k_ln_t<true>(E, log(v), v) : delta_k_ln_t<false>(E, log(v), v);
        break;
      default: r = 0.0;
    }
    out[i] = r;
  }
}

}  // namespace chomp
