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
//   * lane m holds the trapezoid estimate T_m and the extrapolated value of row i is
//     the weighted sum R[i][i] = sum_m C[i][m] T_m (weights = the Richardson
//     recurrence unrolled once, exactly, in tools/gen_special_tables.py): one
//     multiply and one butterfly per row instead of an i-step dependent chain;
//   * every lane of the group ends up with the same sums, so the stopping test
//     is wave-uniform and needs no broadcast;
//   * for multi-wavefront groups the first round evaluates the whole level-L0 grid
//     (one node per thread) instead of walking levels 0..L0 with mostly idle lanes.
//
// Nodes follow SciPy's formula lox + h*j with h = (b-a)/2^(i-1), lox = a + h/2.
#pragma once

#include <type_traits>

#include <hip/hip_runtime.h>

namespace chomp {

// 64-lane all-reduce of a double on the DPP path: xor-1 / xor-2 inside quads, two
// row rotations inside the 16-lane rows, then the four row totals through v_readlane
// (the generic __shfl_xor butterfly lowers to six dependent ds_bpermute round trips
// per 32-bit half, ~10x the latency).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_move<0x124>(v);      // row_ror:4
  v += dpp_move<0x128>(v);      // row_ror:8  -> every lane holds its row's total
  return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

// The same sum where only lanes 0..31 (0..15) can hold anything but zero -- the Romberg rows: lane
// m < 32 holds C[i][m] T_m -- without the rows of lanes that hold the zeros: the bits of wave_sum
// of the same vector ((r0 + r16) + (0 + 0)), six (nine) vector instructions fewer per row.
// What lanes 32..63 (16..63) hold does not matter.
__device__ __forceinline__ double wave_sum32(double v) {
  asm("" : "+v"(v));            // (the caller's product, rounded: not contracted into the first add)
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x124>(v);
  v += dpp_move<0x128>(v);
  return readlane_d(v, 0) + readlane_d(v, 16);
}
__device__ __forceinline__ double wave_sum16(double v) {
  asm("" : "+v"(v));
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x124>(v);
  v += dpp_move<0x128>(v);
  return readlane_d(v, 0);
}

__device__ __forceinline__ double dpp_or_shfl_xor(double v, int offset) {
  return __shfl_xor(v, offset, 64);
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

constexpr int kMaxDivmax = 30;   // chomp_ctx_create enforces divmax <= 30

template <int NF>
struct RombergOut {
  double value[NF];
  int level[NF];
  bool converged[NF];   // false: divmax exhausted (SciPy would warn and return value)
};

// LDS doubles a group needs: 2*NW for the sum exchange + NF*(7 NW + 2) for the
// per-wavefront stride sums of the fused first round (NW > 1 only).
template <int NW, int NF>
constexpr int romberg_scratch() {
  return NW == 1 ? 1 : 2 * NW + NF * (7 * NW + 2);
}

namespace detail {
// An integrand may declare `static constexpr bool kLaneMajor = true`: see romberg_wave6's level loop.
template <class F, class = void>
struct lane_major : std::false_type {};
template <class F>
struct lane_major<F, std::void_t<decltype(F::kLaneMajor)>> : std::bool_constant<F::kLaneMajor> {};
// ... and `static constexpr bool kGeometric = true` with `void geometric(x, exp(x), out)`.
template <class F, class = void>
struct geometric : std::false_type {};
template <class F>
struct geometric<F, std::void_t<decltype(F::kGeometric)>> : std::bool_constant<F::kGeometric> {};
// Integrands may take the node's (level, index-within-level) besides x, so that
// table-driven integrands can look their node up; plain ones take (x, out).
template <class F, int NF>
__device__ __forceinline__ auto call_f(const F& f, double x, double (&o)[NF], int lev, long j,
                                       int) -> decltype(f(x, o, lev, j), void()) {
  f(x, o, lev, j);
}
template <class F, int NF>
__device__ __forceinline__ void call_f(const F& f, double x, double (&o)[NF], int, long, long) {
  f(x, o);
}
// An integrand may offer `bool fast(x, out, level, j) const`: the common case as straight-line
// code (no branch that depends on the node), false when the node needs the general operator().
// The unrolled level loop then issues UNROLL of them back to back -- their table reads overlap,
// which a branch per node prevents -- and repairs the rare refusals afterwards.
template <class F, int NF>
__device__ __forceinline__ auto fast_f(const F& f, double x, double (&o)[NF], int lev, long j,
                                       int) -> decltype(f.fast(x, o, lev, j)) {
  return f.fast(x, o, lev, j);
}
template <class F, int NF>
__device__ __forceinline__ bool fast_f(const F& f, double x, double (&o)[NF], int lev, long j,
                                       long) {
  call_f<F, NF>(f, x, o, lev, j, 0);
  return true;
}
}  // namespace detail

// F: void operator()(double x, double (&out)[NF]) const
//    or void operator()(double x, double (&out)[NF], int level, long j) const
// where the node is the j-th new point of `level` (level 0: j = 0 -> a, 1 -> b).
//
// NW > 1: the first round evaluates the whole level-L0 grid (2^L0 + 1 = 32 NW + 1
// points, one per thread) at once; the Romberg rows 0..L0 and their stopping tests
// are then replayed from the per-level sums, so an integral that the reference
// stops at level <= L0 still returns exactly that level's value.
// UNROLL: node evaluations issued together in the level loop (their table loads overlap);
// the partial sums are still added in node order, so the result does not depend on it.
// dump (optional, global or LDS memory, kRombergDump * NF doubles): the state the
// integral stopped in -- per integrand the 32 trapezoid estimates T_m, then the running
// node sum and the last row's value -- so that another kernel can carry it on to deeper
// levels (k_halo_deep_level / k_halo_deep_advance).
constexpr int kRombergDump = 34;
// Optional second, looser stopping rule for an integral whose value only has to DECIDE a
// comparison: a row also ends the integral when err < rtol |result| with this (looser) rtol
// and the result lies outside both windows (lo1, hi1), (lo2, hi2) around the values it is
// compared with.  A result inside a window walks on to the regular tolerance: the same rows,
// so the value is then exactly what the regular rule alone would have returned.
struct RombergLoose {
  double rtol, lo1, hi1, lo2, hi2;
  // Not before level kMinLevel.  Two consecutive rows of a coarse grid can agree by accident:
  // sigma^2(R) of a 107 M_sun/h halo at z = 0.96 (tools/soak.py 12 160, case 96) has rows 3 and 4
  // -- 9 and 17 nodes over sixteen e-folds of k -- 3.9e-7 apart and both 0.9 % off, on the other
  // side of the band edge: the probe "passed", the walk stopped a step early, no flag.  The
  // reference's own rule (1.48e-8) walks on there; this shortcut must not be what decides.
  static constexpr int kMinLevel = 8;
  __device__ __forceinline__ bool decides(double err, double cur, int level) const {
    return level >= kMinLevel && err < rtol * fabs(cur) && !(cur > lo1 && cur < hi1) &&
           !(cur > lo2 && cur < hi2);
  }
};
// An integral carried on by another kernel from the state romberg_group dumped when it ran out
// of its divmax at `level` (dump: kRombergDump doubles of one integrand): the same rows and
// stopping test from the sums of the further levels.  Every lane of every wavefront holds the
// same state.
struct RombergResume {
  double ordsum, Tl, prev, value, range, n, tol, rtol;
  int level;
  bool done;
  __device__ __forceinline__ void load(const double* dump, int level_, double range_, double tol_,
                                       double rtol_) {
    const int lane = threadIdx.x & 63;
    Tl = lane < 32 ? dump[lane] : 0.0;
    ordsum = dump[32];
    prev = dump[33];
    value = prev;
    range = range_; tol = tol_; rtol = rtol_;
    level = level_;
    n = (double)(1L << level_);
    done = false;
  }
  // c_il: CHOMP_ROMBERG_C[i][lane & 31], read by the caller BEFORE the level's nodes (a
  // dependent read here would add its latency to every level)
  __device__ __forceinline__ void advance(int i, double S, double c_il) {
    const int lane = threadIdx.x & 63;
    n *= 2.0;
    ordsum += S;
    const double Ti = ldexp(range * ordsum, -i);          // (/ n, n = 2^i: the same bits, one instruction)
    if (lane == i) Tl = Ti;
    const double cur = wave_sum32(c_il * Tl);
    const double err = fabs(cur - prev);
    prev = cur;
    value = cur;
    level = i;
    if (err < tol || err < rtol * fabs(cur)) done = true;
  }
};

template <int NW, int NF, class F, int UNROLL = 1>
__device__ __forceinline__ RombergOut<NF> romberg_group(const F& f, double a, double b,
                                                        double tol, double rtol,
                                                        int divmax, double* red,
                                                        double* dump = nullptr,
                                                        const RombergLoose* loose = nullptr) {
  constexpr int NT = 64 * NW;
  const int lane = threadIdx.x & 63;
  const int gt = (NW == 1) ? lane : (int)threadIdx.x;
  const int cl = lane & 31;                      // column of the weight table
  int flip = 0;
  const double intrange = b - a;

  // Lane m (< 32) holds the trapezoid estimate T_m; the Romberg value of row i is
  // R[i][i] = sum_m C[i][m] T_m (CHOMP_ROMBERG_C: the Richardson recurrence unrolled
  // on coefficient vectors), one multiply and one wavefront butterfly per row instead
  // of an i-step dependent chain.
  double ordsum[NF], Tl[NF], prev[NF];
  bool done[NF];
  RombergOut<NF> out;
  bool all_done = false;

  auto advance = [&](int q, int i, double S, double n, double c_il) {
    ordsum[q] += S;
    const double Ti = ldexp(intrange * ordsum[q], -i);   // R[i][0] (/ n, n = 2^i: the same bits)
    if (lane == i) Tl[q] = Ti;
    const double cur = wave_sum32(c_il * Tl[q]);
    const double err = fabs(cur - prev[q]);
    prev[q] = cur;
    out.value[q] = cur;
    out.level[q] = i;
    if (err < tol || err < rtol * fabs(cur)) done[q] = true;
    if (loose != nullptr && loose->decides(err, cur, i)) done[q] = true;
  };

  int i0;   // first level handled by the generic loop
  if constexpr (NW == 1) {
    // T_0: the two end points (lanes 0 and 1)
    double v[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) v[q] = 0.0;
    if (gt < 2) detail::call_f<F, NF>(f, gt == 0 ? a : b, v, 0, (long)gt, 0);
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      ordsum[q] = 0.5 * group_sum<NW>(v[q], red, flip);
      out.value[q] = intrange * ordsum[q];
      out.level[q] = 0;
      prev[q] = out.value[q];
      Tl[q] = (lane == 0) ? out.value[q] : 0.0;
      done[q] = false;
    }
    i0 = 1;
  } else {
    constexpr int L0max = (NW == 2) ? 6 : (NW == 4) ? 7 : (NW == 8) ? 8 : 9;
    static_assert(NW == 2 || NW == 4 || NW == 8 || NW == 16, "NW must be 1,2,4,8,16");
    const int L0 = divmax < L0max ? divmax : L0max;
    const int N0 = 1 << L0;                              // intervals of the fused grid
    double* cw = red + 2 * NW;                           // [NF][7 NW + 2]
    constexpr int FS = 7 * NW + 2;
    // weight rows 1..L0, fetched before the node evaluations hide their latency
    double crow[L0max + 1];
#pragma unroll
    for (int i = 1; i <= L0max; ++i) crow[i] = CHOMP_ROMBERG_C[i][cl];
    double v[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) v[q] = 0.0;
    if (gt <= N0) {
      int lev;
      long j;
      double x;
      if (gt == 0) { lev = 0; j = 0; x = a; }
      else if (gt == N0) { lev = 0; j = 1; x = b; }
      else {
        const int tz = __builtin_ctz((unsigned)gt);
        lev = L0 - tz;
        j = (long)(((gt >> tz) - 1) >> 1);
        const double h = ldexp(intrange, 1 - lev);
        x = (a + 0.5 * h) + h * (double)j;
      }
      detail::call_f<F, NF>(f, x, v, lev, j, 0);
    }
    // Per-level sums of the fused grid.  Thread p holds node p; nodes of level l are
    // the multiples of s = N0 >> l that are not multiples of 2s, so with
    // C_s = sum of the interior nodes at multiples of s, S_l = C_s - C_2s.  One xor
    // butterfly per wavefront yields its C_32 .. C_1 in lane 0 on the way (after the
    // steps with offsets >= s, lane 0 holds the lanes that are multiples of s); the
    // strides >= 64 only involve lane 0 of each wavefront.
    const int wv = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      double* c = cw + q * FS;
      if (gt == 0) c[7 * NW] = v[q];                      // f(a)
      if (gt == N0) c[7 * NW + 1] = v[q];                 // f(b)
      double x = (gt > 0 && gt < N0) ? v[q] : 0.0;        // interior nodes only
      if (lane == 0) c[wv * 7 + 6] = x;
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        x += __shfl_xor(x, 32 >> st, 64);
        if (lane == 0) c[wv * 7 + st] = x;                // C_(32 >> st) of this wavefront
      }
    }
    __syncthreads();
    double Sl[NF];     // lane l: sum of level l (lane 0: the end points)
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const double* c = cw + q * FS;
      double Cs = 0.0, C2s = 0.0;                         // for this lane's level
      if (lane >= 1 && lane <= L0) {
        const int s = N0 >> lane;                         // stride of level `lane`
        auto C_of = [&](int stride) {
          double t = 0.0;
          if (stride >= N0) return t;                     // no interior multiples
          if (stride < 64) {
            const int st = 5 - (31 - __builtin_clz((unsigned)stride));   // 32>>st == stride
            for (int w = 0; w < NW; ++w) t += c[w * 7 + st];
          } else {
            const int ws = stride >> 6;
            for (int w = ws; w * 64 < N0; w += ws) t += c[w * 7 + 6];
          }
          return t;
        };
        Cs = C_of(s);
        C2s = C_of(2 * s);
      }
      Sl[q] = (lane == 0) ? 0.5 * (c[7 * NW] + c[7 * NW + 1]) : (Cs - C2s);
    }
    __syncthreads();          // cw may be reused by a later call
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      ordsum[q] = __shfl(Sl[q], 0, 64);
      out.value[q] = intrange * ordsum[q];
      out.level[q] = 0;
      prev[q] = out.value[q];
      Tl[q] = (lane == 0) ? out.value[q] : 0.0;
      done[q] = false;
    }
    double n = 1.0;
#pragma unroll
    for (int i = 1; i <= L0max; ++i) {
      if (i <= L0 && !all_done) {
        n *= 2.0;
        all_done = true;
#pragma unroll
        for (int q = 0; q < NF; ++q) {
          const double S = __shfl(Sl[q], i, 64);
          if (!done[q]) advance(q, i, S, n, crow[i]);
          all_done = all_done && done[q];
        }
      }
    }
    i0 = L0 + 1;
  }

  double n = (double)(1L << (i0 - 1));
  for (int i = i0; i <= divmax && !all_done; ++i) {
    const double c_il = CHOMP_ROMBERG_C[i][cl];          // latency hidden by the nodes
    n *= 2.0;
    const long numtosum = 1L << (i - 1);
    const double h = ldexp(intrange, 1 - i);              // (intrange / numtosum)
    const double lox = a + 0.5 * h;
    double part[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) part[q] = 0.0;
    long j = gt;
    if (UNROLL > 1) {
      for (; j + (UNROLL - 1) * (long)NT < numtosum; j += UNROLL * (long)NT) {
        double v[UNROLL][NF];
        bool ok = true;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
          ok = detail::fast_f<F, NF>(f, lox + h * (double)(j + u * (long)NT), v[u], i,
                                     j + u * (long)NT, 0) && ok;
        if (!ok) {                                 // (rare: some node of the batch refused)
          for (int u = 0; u < UNROLL; ++u)
            detail::call_f<F, NF>(f, lox + h * (double)(j + u * (long)NT), v[u], i,
                                  j + u * (long)NT, 0);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
          for (int q = 0; q < NF; ++q) part[q] += v[u][q];
      }
    }
    for (; j < numtosum; j += NT) {
      double v[NF];
      detail::call_f<F, NF>(f, lox + h * (double)j, v, i, j, 0);
#pragma unroll
      for (int q = 0; q < NF; ++q) part[q] += v[q];
    }
    all_done = true;
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const double S = group_sum<NW>(part[q], red, flip);
      if (!done[q]) advance(q, i, S, n, c_il);
      all_done = all_done && done[q];
    }
  }
#pragma unroll
  for (int q = 0; q < NF; ++q) out.converged[q] = done[q];
  if (dump != nullptr && (NW == 1 || threadIdx.x < 64)) {
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      if (lane < 32) dump[q * kRombergDump + lane] = Tl[q];
      if (lane == 0) {
        dump[q * kRombergDump + 32] = ordsum[q];
        dump[q * kRombergDump + 33] = prev[q];
      }
    }
  }
  return out;
}

// One wavefront per integral (or NF integrals sharing their nodes) with a first round that
// fills it exactly: lane p evaluates node p of the level-6 grid (p = 0: the lower end point;
// 64 intervals, 65 points) and the values at the UPPER end point are handed in (fb: the
// caller has them from a table, e.g. one end-point evaluation per knot done elsewhere).
// Rows 0..6 and their stopping tests are replayed from the per-level sums exactly as in
// romberg_group's multi-wavefront first round; deeper levels walk on as romberg_group<1>.
// A knot that scipy stops at level 6 costs one evaluation per lane, at level 7 two.
// Needs divmax >= 6.  dump (optional, kRombergDump * NF doubles, LDS or global): the state the
// integral stopped in, as romberg_group leaves it, for whoever carries it on (RombergResume).
// loose (optional): the probes' early stopping rule (RombergLoose), as in romberg_group.
template <int NF, class F>
__device__ __forceinline__ RombergOut<NF> romberg_wave6(const F& f, double a, double b,
                                                        const double (&fb)[NF], double tol,
                                                        double rtol, int divmax,
                                                        double* dump = nullptr,
                                                        const RombergLoose* loose = nullptr) {
  constexpr int L0max = 6;
  const int lane = threadIdx.x & 63;
  const int cl = lane & 31;
  const double intrange = b - a;
  double ordsum[NF], Tl[NF], prev[NF];
  bool done[NF];
  RombergOut<NF> out;
  bool all_done = false;
  auto advance = [&](int q, int i, double S, double n, double c_il) {
    ordsum[q] += S;
    const double Ti = ldexp(intrange * ordsum[q], -i);   // R[i][0] (/ n, n = 2^i: the same bits)
    if (lane == i) Tl[q] = Ti;
    const double cur = wave_sum32(c_il * Tl[q]);
    const double err = fabs(cur - prev[q]);
    prev[q] = cur;
    out.value[q] = cur;
    out.level[q] = i;
    if (err < tol || err < rtol * fabs(cur)) done[q] = true;
    if (loose != nullptr && loose->decides(err, cur, i)) done[q] = true;
  };
  constexpr int L0 = L0max, N0 = 1 << L0max;             // (the caller guarantees divmax >= 6)
  double crow[L0max + 1];
#pragma unroll
  for (int i = 1; i <= L0max; ++i) crow[i] = CHOMP_ROMBERG_C[i][cl];
  double v[NF];
#pragma unroll
  for (int q = 0; q < NF; ++q) v[q] = 0.0;
  if (lane < N0) {
    int lev = 0;
    long j = 0;
    double x = a;
    if (lane > 0) {
      const int tz = __builtin_ctz((unsigned)lane);
      lev = L0 - tz;
      j = (long)(((lane >> tz) - 1) >> 1);
      const double h = ldexp(intrange, 1 - lev);
      x = (a + 0.5 * h) + h * (double)j;
    }
    detail::call_f<F, NF>(f, x, v, lev, j, 0);
  }
  // The sums of levels 1..6: level l is the lanes whose lowest set bit is 2^(6 - l) (lane 32;
  // 16 and 48; ...; the odd lanes).  Inside a row of 16 lanes by DPP (three adds), the four rows
  // through v_readlane -- the xor butterfly this replaces took six dependent ds_bpermute round
  // trips per integrand (the sums then came out as differences of nested partial sums).
  double n = 1.0;
#pragma unroll
  for (int q = 0; q < NF; ++q) {
    double Ls[L0max + 1];
    {
      const double x0 = lane > 0 ? v[q] : 0.0;                   // (lane 0: the lower end point)
      const double x8 = x0 + dpp_move<0x128>(x0);                // lanes {i, i + 8} of a row
      const double x4 = x8 + dpp_move<0x124>(x8);                // = mod 4
      const double x2 = x4 + dpp_move<0x4E>(x4);                 // = mod 2
      Ls[6] = (readlane_d(x2, 1) + readlane_d(x2, 17)) + (readlane_d(x2, 33) + readlane_d(x2, 49));
      Ls[5] = (readlane_d(x4, 2) + readlane_d(x4, 18)) + (readlane_d(x4, 34) + readlane_d(x4, 50));
      Ls[4] = (readlane_d(x8, 4) + readlane_d(x8, 20)) + (readlane_d(x8, 36) + readlane_d(x8, 52));
      Ls[3] = (readlane_d(x0, 8) + readlane_d(x0, 24)) + (readlane_d(x0, 40) + readlane_d(x0, 56));
      Ls[2] = readlane_d(x0, 16) + readlane_d(x0, 48);
      Ls[1] = readlane_d(x0, 32);
    }
    ordsum[q] = 0.5 * (readlane_d(v[q], 0) + fb[q]);
    out.value[q] = intrange * ordsum[q];
    out.level[q] = 0;
    prev[q] = out.value[q];
    Tl[q] = (lane == 0) ? out.value[q] : 0.0;
    done[q] = false;
    // rows 1..L0 of this integrand: lane l forms T_l from the running sum of the level sums
    // (Ls[i], above), the L0
    // extrapolations R[i][i] are taken back to back -- they do not depend on each other; row
    // by row each waited for the butterfly of the one before, six round trips per integrand in
    // front of every integral's first deeper level -- and only the stopping test walks through
    // them in order.  Same operations on the same operands as the row-by-row replay.
    double nq = 1.0;
    {
      double os = ordsum[q], mine = ordsum[q];
      double osum[L0max + 1];
      osum[0] = os;
#pragma unroll
      for (int i = 1; i <= L0max; ++i) {
        os += Ls[i];
        osum[i] = os;
        if (i <= lane) mine = os;
      }
      const double T = lane <= L0max ? ldexp(intrange * mine, -lane) : 0.0;   // (/ 2^lane)
      double cur[L0max + 1];
#pragma unroll
      for (int i = 1; i <= L0max; ++i) cur[i] = wave_sum16(lane <= i ? crow[i] * T : 0.0);
      int stop = 0;
#pragma unroll
      for (int i = 1; i <= L0max; ++i) {
        if (!done[q]) {
          nq *= 2.0;
          const double err = fabs(cur[i] - prev[q]);
          prev[q] = cur[i];
          out.value[q] = cur[i];
          out.level[q] = i;
          stop = i;
          if (err < tol || err < rtol * fabs(cur[i])) done[q] = true;
          // (the probes' loose rule decides nothing at these levels: RombergLoose::kMinLevel)
          static_assert(L0max < RombergLoose::kMinLevel, "rows of the first pass: exact rule only");
        }
      }
      // (the state the row-by-row replay leaves: the sums and T_l up to the row reached)
      double upto = osum[0];
#pragma unroll
      for (int i = 1; i <= L0max; ++i)
        if (i == stop) upto = osum[i];
      ordsum[q] = upto;
      Tl[q] = lane <= stop ? T : 0.0;
    }
    n = nq > n ? nq : n;
  }
  all_done = true;
#pragma unroll
  for (int q = 0; q < NF; ++q) all_done = all_done && done[q];
  // (an integrand that stopped early keeps its own n; the walk below continues from L0 for
  //  the others, whose n is 2^L0)
  n = (double)(1L << L0);
  for (int i = L0 + 1; i <= divmax && !all_done; ++i) {
    const double c_il = CHOMP_ROMBERG_C[i][cl];
    n *= 2.0;
    const long numtosum = 1L << (i - 1);
    const double h = ldexp(intrange, 1 - i);              // (intrange / numtosum)
    const double lox = a + 0.5 * h;
    double part[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) part[q] = 0.0;
    if constexpr (detail::lane_major<F>::value) {
      // (an integrand that GATHERS from an LDS table at an index proportional to x: a lane
      //  takes numtosum / 64 consecutive nodes, so that the lanes of a wavefront are a 64th of
      //  the range apart at every level -- the same bank pattern as the first pass -- instead of
      //  one node spacing apart, which at the deep levels is a few table entries: 4-8 lanes to
      //  a bank)
      const long per = numtosum >> 6;                      // (i > L0 = 6: numtosum >= 64)
      if constexpr (detail::geometric<F>::value) {
        // (... and whose abscissa is a logarithm it needs the exponential of: a lane's nodes
        //  are a constant factor exp(h) apart, so one exponential per lane and level and a
        //  multiplication per node -- <= 16 of them in a row here, 2e-15 of drift)
        const double ratio = exp(h);
        double ex = exp(lox + h * (double)((long)lane * per));
        for (long p = 0; p < per; ++p) {
          const long j = (long)lane * per + p;
          double w[NF];
          f.geometric(lox + h * (double)j, ex, w);
          ex *= ratio;
#pragma unroll
          for (int q = 0; q < NF; ++q) part[q] += w[q];
        }
      } else {
        for (long p = 0; p < per; ++p) {
          const long j = (long)lane * per + p;
          double w[NF];
          detail::call_f<F, NF>(f, lox + h * (double)j, w, i, j, 0);
#pragma unroll
          for (int q = 0; q < NF; ++q) part[q] += w[q];
        }
      }
    } else {
      for (long j = lane; j < numtosum; j += 64) {
        double w[NF];
        detail::call_f<F, NF>(f, lox + h * (double)j, w, i, j, 0);
#pragma unroll
        for (int q = 0; q < NF; ++q) part[q] += w[q];
      }
    }
    all_done = true;
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const double S = wave_sum(part[q]);
      if (!done[q]) advance(q, i, S, n, c_il);
      all_done = all_done && done[q];
    }
  }
#pragma unroll
  for (int q = 0; q < NF; ++q) out.converged[q] = done[q];
  if (dump != nullptr) {
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      if (lane < 32) dump[q * kRombergDump + lane] = Tl[q];
      if (lane == 0) {
        dump[q * kRombergDump + 32] = ordsum[q];
        dump[q * kRombergDump + 33] = prev[q];
      }
    }
  }
  return out;
}

// Single-integrand convenience wrapper: F is double operator()(double).
template <class F>
struct Scalar1 {
  static constexpr bool kLaneMajor = detail::lane_major<F>::value;
  static constexpr bool kGeometric = detail::geometric<F>::value;
  const F& f;
  __device__ __forceinline__ void geometric(double x, double ex, double (&out)[1]) const {
    if constexpr (kGeometric) out[0] = f.with_exp(x, ex); else out[0] = f(x);
  }
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
