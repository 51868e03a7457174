// chomp_probe_kernel.h -- k_epoch_probe, the certifying probes of the mass-limit search
// (chomp_probe.hip instantiates the four-wavefront phases, chomp_capi.hip the single-wavefront
// probing phase of a large batch: see launch_epoch_probe in chomp_mass_kernels.h).
#pragma once

#include "chomp_mass_kernels.h"

namespace chomp {

// grid (n_epoch, 2 * kProbes), block 64 * kInitNW.  blockIdx.y = kProbes * side + p
// certifies candidate j - 2 + p of side 0 (mass_min) / 1 (mass_max); role kProbes also
// does the comoving distance (or only that, with fixed mass limits; a ninth block per epoch
// for it was measured: 576 blocks for 512 resident ones, 34.1 against 32.8 us).  The last block of
// an epoch to finish combines the results (count[e], reset by it): a side whose probes
// show "fails at c - 1, passes at c" is settled; any other (the estimate off by more than
// the probes cover, a walk that leaves the ln S table: rare) falls back to the bracketing
// secant search on exact integrals, seeded with what the probes established.
// epochs[e] holds the closed-form part of the record (k_sigma_nodes) on entry and the
// complete record on exit.
// PHASE 0: as described (one launch: a (k, z) grid's set-up is a latency chain).  A large
// batch is throughput-bound, and there the fence in front of every block's arrival count -- an
// L2 write-back on this chip -- costs more than the probes: PHASE 1 (the same grid) only
// probes, PHASE 2 (grid n_epoch) certifies behind the kernel boundary.  Same numbers.
// NW: wavefronts per block.  kInitNW for a (k, z) grid (the launch lasts as long as one probe:
// four wavefronts on each) and for the certifying phase; 1 for the probing phase of a large
// batch -- its 8 n_epoch integrals fill the chip as single wavefronts, with no barrier or LDS
// hand-over per Romberg level (1024 epochs: 241 -> 72 us; the four-wavefront kernel ran at 41 % of
// its VALU issue rate).
template <bool BAO, int PHASE, int NW = kInitNW>
__global__ __launch_bounds__(64 * NW, NW == 1 ? 4 : 1) void k_epoch_probe(
    chomp_config cfg, Epoch* __restrict__ epochs, double* __restrict__ search,
    const double* __restrict__ cand, const double* __restrict__ snodes,
    double* __restrict__ probe, int* __restrict__ count, unsigned* __restrict__ status) {
  __shared__ Epoch E;
  __shared__ double red[romberg_scratch<NW, 1>()];
  __shared__ double lns[kSGrid];   // the cosmology's coarse ln S(R) table
  __shared__ int last, sh_j;
  __shared__ double seeds[2][6];   // per side: dir, jl, nu_l, jh, nu_h, n_eval (uncertified)
  __shared__ int open_side[2];
  // (roles in reverse dispatch order: the mass_max-side probes -- Romberg levels 12-13 -- first)
  const int e = blockIdx.x, role = PHASE == 2 ? 0 : 2 * kProbes - 1 - (int)blockIdx.y;
  const bool chi_role = PHASE != 2 && role == kProbes;
  const int side = role / kProbes, p = role % kProbes;
  const bool fixed = cfg.mass_min > 0.0 && cfg.mass_max > 0.0;     // mass_function.py:163-170
  if (fixed && !chi_role) return;
  PSTAMP(0);
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
  PSTAMP(1);
  if (chi_role) {                  // comoving distance, cosmology.py:106-110
    EIntegrand f{E.om0, E.ol0, E.or0, E.H0};
    const double chi = romberg1<NW>(f, 0.0, E.z, cfg.global_precision,
                                         cfg.cosmo_precision, cfg.divmax, red);
    __syncthreads();
    if (threadIdx.x == 0) {
      E.chi = chi;
      __hip_atomic_store(pr + 2 * kProbes, chi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
  if constexpr (PHASE != 2) {
    // ---- this block's probe: candidate j - 2 + p of its side
    PSTAMP(2);
    const SidePlan plan = plan_side(E, lns, side, cand, &sh_j);
    PSTAMP(3);
    const SideThresholds T = side_thresholds(side, cand);
    double nu_mine = NAN;
    if (plan.ok && plan.dir != 0) {
      const int c = plan.j - 2 + p;
      // (away from an edge candidate 0 fails by the margin of the estimate)
      if ((c > 0 || (c == 0 && plan.at_edge)) && c < kSearchJ) {
        const double* tab = plan.dir < 0 ? T.down : T.up;
        nu_mine = nu_probe<NW, BAO>(E, snode, tab[c], cfg, T.thr_lo, T.thr_hi, red);
      }
    }
    if (threadIdx.x == 0) {
      // (agent-scope stores: written through to where the last block's agent-scope loads read
      //  them -- PHASE 0 then needs no fence, i.e. no write-back of this XCD's whole L2, in
      //  front of its arrival count, only the stores' completion)
      auto put = [](double* q, double v) {
        __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      put(pr + role, nu_mine);
      if (p == 0) {
        double* pl = pr + 2 * kProbes + 4 + 4 * side;
        put(pl, plan.ok ? (plan.at_edge ? 2.0 : 1.0) : 0.0);
        put(pl + 1, (double)plan.dir);
        put(pl + 2, (double)plan.j);
        put(pl + 3, plan.nu_start);
      }
    }
  }
  PSTAMP(4);
  if constexpr (PHASE != 1) {
  if constexpr (PHASE == 0) {
    if (threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the record's stores (above) have landed
      last = atomicAdd(&count[e], 1) == 2 * kProbes - 1 ? 1 : 0;
    }
    __syncthreads();
    PSTAMP(5);
    if (!last) return;
    // (no fence on this side: everything the last block reads of the others -- the probe record
    //  -- is read with agent-scope loads, issued behind the arrival count's return)
  }
  // ---- last block of the epoch: certify both sides (thread 0: scalar logic on 8 numbers).
  // The epoch's probe record -- written by the other seven blocks -- is fetched by 24 lanes at
  // once, one agent-scope load each, and the logic reads the copy in LDS: read where they are
  // used, the ~20 loads were a chain of dependent round trips, 9 of the launch's 38 us
  // (tools/dev_probe_stamps4.py: arrival at 25 us, end at 34.5).
  __shared__ double rec[kProbeStride];
  if (threadIdx.x < kProbeStride)
    rec[threadIdx.x] = __hip_atomic_load(pr + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x < 2) {           // (the two sides side by side: lane 0 mass_min, lane 1 mass_max)
    auto peek = [&](const double* q) { return rec[q - pr]; };
    {
      const int sd = (int)threadIdx.x;
      const double* pl = pr + 2 * kProbes + 4 + 4 * sd;
      const double mode = peek(pl);
      const bool ok = mode != 0.0, at_edge = mode == 2.0;
      const int dir = (int)peek(pl + 1);
      const int j = (int)peek(pl + 2);
      double nu_start = peek(pl + 3);
      const SideThresholds S = side_thresholds(sd, cand);
      double mass = S.down[0];
      int n_eval = 0;
      bool certified = ok;
      int seed_dir = 0, seed_jl = 0, seed_jh = -1;
      double nu_l = 0.0, nu_h = 0.0;
      bool walk = ok && dir != 0;
      if (at_edge) {               // the exact nu of the starting mass decides the direction
        nu_start = peek(pr + kProbes * sd);
        const int dir_exact = S.thr_hi < nu_start ? -1 : (S.thr_lo > nu_start ? +1 : 0);
        n_eval = 1;
        if (!(nu_start == nu_start)) { certified = false; walk = false; n_eval = 0; }
        else if (dir_exact == 0) walk = false;                   // stays at the start: done
        else if (dir_exact != dir) {                             // guessed the other way
          certified = false; walk = false;
          seed_dir = dir_exact; seed_jl = 0; nu_l = nu_start;
        }
      }
      if (walk) {
        const double* tab = dir < 0 ? S.down : S.up;
        const double thr = dir < 0 ? S.thr_hi : S.thr_lo;
        // status of candidates j - 2 .. j + 1: 0 fails, 1 passes, -1 unknown
        // (scalars, not arrays: dynamic indexing would put them in scratch)
        int st0, st1, st2, st3;
        double nu0, nu1, nu2, nu3;
        auto classify = [&](int q, int* st, double* nu) {
          const int c = j - 2 + q;
          *nu = peek(pr + kProbes * sd + q);
          if (c <= 0) *st = 0;     // (at an edge: candidate 0 fails exactly, see above)
          else if (!(*nu == *nu)) *st = -1;
          else { *st = (dir < 0 ? !(thr < *nu) : !(thr > *nu)) ? 1 : 0; ++n_eval; }
        };
        classify(0, &st0, &nu0); classify(1, &st1, &nu1);
        classify(2, &st2, &nu2); classify(3, &st3, &nu3);
        certified = false;
        int first_pass = -1, before = -1;        // status of the candidate before it
        double nu_first = 0.0;
        if (st3 == 1) { first_pass = 3; nu_first = nu3; before = st2; }
        if (st2 == 1) { first_pass = 2; nu_first = nu2; before = st1; }
        if (st1 == 1) { first_pass = 1; nu_first = nu1; before = st0; }
        if (st0 == 1) { first_pass = 0; nu_first = nu0; before = -1; }
        if (first_pass > 0 && before == 0) {                     // fails at c - 1, passes at c
          certified = true;
          mass = tab[j - 2 + first_pass];
        } else if (first_pass == 0 && j - 2 == 1) {              // passes at 1, 0 fails
          certified = true;
          mass = tab[1];
        }
        if (!certified) {          // the estimate was off by more than the probes cover:
          seed_dir = dir;          // the exact search starts from what they established
          seed_jl = 0; nu_l = nu_start;
          if (st0 == 0 && j - 2 > 0) { seed_jl = j - 2; nu_l = nu0; }
          if (st1 == 0 && j - 1 > 0) { seed_jl = j - 1; nu_l = nu1; }
          if (st2 == 0 && j > 0) { seed_jl = j; nu_l = nu2; }
          if (st3 == 0 && j + 1 > 0) { seed_jl = j + 1; nu_l = nu3; }
          if (first_pass >= 0 && j - 2 + first_pass > seed_jl) {
            seed_jh = j - 2 + first_pass; nu_h = nu_first;
          }
        }
      }
      seeds[sd][0] = (double)seed_dir; seeds[sd][1] = (double)seed_jl; seeds[sd][2] = nu_l;
      seeds[sd][3] = (double)seed_jh; seeds[sd][4] = nu_h; seeds[sd][5] = (double)n_eval;
      open_side[sd] = certified ? 0 : 1;
      if (certified) {
        search[(e * 2 + sd) * 2 + 0] = log(mass);
        search[(e * 2 + sd) * 2 + 1] = (double)n_eval;
        const unsigned st = search_status(E, sd, mass, false);
        if (st) atomicOr(&status[e], st);
      }
    }
    if (threadIdx.x == 0) {
      E.chi = peek(pr + 2 * kProbes);
      if (PHASE == 0) count[e] = 0;
    }
  }
  __syncthreads();
  for (int sd = 0; sd < 2; ++sd) {
    if (!open_side[sd]) continue;                // block-uniform
    int n_eval = (int)seeds[sd][5];
    bool exhausted = false;
    const double mass = search_side_exact<NW, BAO>(
        E, snode, sd, cfg, cand, red, &n_eval, &exhausted, (int)seeds[sd][0], (int)seeds[sd][1],
        seeds[sd][2], (int)seeds[sd][3], seeds[sd][4]);
    if (threadIdx.x == 0) {
      search[(e * 2 + sd) * 2 + 0] = log(mass);
      search[(e * 2 + sd) * 2 + 1] = (double)n_eval;
      const unsigned st = search_status(E, sd, mass, exhausted);
      if (st) atomicOr(&status[e], st);
    }
  }
  __syncthreads();
  copy_doubles(reinterpret_cast<double*>(&epochs[e]), reinterpret_cast<const double*>(&E),
               kEpochDoubles);
  PSTAMP(6);
  }  // (PHASE != 1)
}

}  // namespace chomp
