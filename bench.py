#!/usr/bin/env python3
"""Benchmark of the halo-model P(k, z) hot path on 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline (BASELINE.json configs[1]): a "step" is one full pass of the hot path --
Stage K (every table of the 64 redshifts: sigma_8 normalisation, mass-limit search,
nu table, splines, normalisations, n_bar, the 50-knot 2-halo / 1-halo integrals)
followed by Stage E (P_mm on the 4096 x 64 (k, z) grid), with k resident in HBM
and, for N > 1, the all-gather that re-assembles the grid on every rank.  Rank 0
prints ONE JSON line (contract in the task description) carrying

  roofline          the streaming Stage-E kernel on the enlarged grid of SURVEY 8(d)
                    (HIP events on the kernel's own stream), the whole call beside it,
                    and the kernel the timed C2 step actually runs
  roofline_stage_k  Stage K of the timed step against the vector-fp64 peak (FLOP per
                    step from the committed SQ counters of the same kernels)
  cpu_baseline      the NumPy oracle = a port of the reference's algorithm, timed on
                    this box's host cores on a bounded sample of the same workload
  other_configs     configs[2..4] (c3, c4, c5) timed in the same process, GPU side
  dropin            the reference-shaped loop over one Halo object (ms per z, samples/s)

N > 1 (torch.distributed.run, one rank per GPU, RCCL): the headline is the STRONG split
of north_star / SURVEY 8(e) -- configs[1]'s 64 redshift rows dealt over the ranks, one
all-gather per step; the same run then times the WEAK split (64 rows per GPU) and reports
it under `weak_scaling` (`--scaling weak` swaps the two).  Rank 0 times the CPU baseline
before it touches the GPU at every N.

Every leg reports ms_per_step_median / _p95 / _max beside its mean (HIP events behind
every step, a pass of its own), and `dropin` is the reference-shaped call pattern: one
Halo, set_redshift(z) + power_mm(k) per redshift, host arrays in and out.

--workload c3 | c4 | c5 runs one of the other configs as the headline instead.
"""
import argparse
import json
import os
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NK, NZ, Z_MAX = 4096, 64, 1.5
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6   # vector fp64: half the 157.3 TFLOP/s fp32 vector peak
N_THETA, N_ELL = 1024, 2048
# SQ counter summaries of Stage K (tools/prof.sh sq -> tools/collect_profiles.py), newest first
STAGE_K_COUNTER_FILES = ("round4_stage_k_counters.json", "round3_stage_k_counters.json",
                         "round2_stage_k_counters.json")
PROFILE_ROUND = "round4"


# ---------------------------------------------------------------------------
# CPU baselines (the oracle: tests' checker, here only timed)
# ---------------------------------------------------------------------------
def _oracle_row(args):
    """One redshift row through the oracle (module-level: used by the process pool)."""
    which, mass_function, z = args
    from oracle import chomp_oracle as o
    fam = {"power_mm": ("mm",), "power_gm": ("gm",)}[which]
    k = numpy.logspace(-3, 2, NK)
    e = o.epoch(None, float(z))
    m = o.mass_table(e, kind=mass_function)
    t = o.halo_table(e, m, families=fam)
    return float(o.halo_power(t, fam[0], k).sum())


def cpu_baseline(which, mass_function, sample_z, pool_z):
    """Time the oracle (port of the reference's CPU algorithm) on `sample_z` redshifts x
    the 4096-point k grid with one host thread (what a user of the reference gets: it is
    single-threaded Python), then on `pool_z` with one process per core of this box's
    CPU share.  Called before anything touches the GPU (the pool forks)."""
    import multiprocessing
    t0 = time.perf_counter()
    for z in sample_z:
        _oracle_row((which, mass_function, z))
    dt = time.perf_counter() - t0
    cores = min(16, len(os.sched_getaffinity(0)))
    with multiprocessing.get_context("fork").Pool(cores) as pool:
        pool.map(_oracle_row, [(which, mass_function, pool_z[0])] * cores)     # warm the workers
        t0 = time.perf_counter()
        pool.map(_oracle_row, [(which, mass_function, z) for z in pool_z], chunksize=1)
        dtp = time.perf_counter() - t0
    return {"value": len(sample_z) * NK / dt, "unit": "samples/s", "cores": 1,
            "kind": "port",
            "sample": "%d of the %d redshifts x %d k, oracle/chomp_oracle.py (NumPy/SciPy "
                      "restatement of the reference, adaptive Romberg), %.1f s"
                      % (len(sample_z), NZ, NK, dt),
            "pool": {"value": len(pool_z) * NK / dtp, "unit": "samples/s", "cores": cores,
                     "sample": "%d redshifts x %d k over %d forked processes, %.1f s"
                               % (len(pool_z), NK, cores, dtp)},
            "host_cores_available": os.cpu_count()}


def projection_baseline(ggl):
    """The oracle on a bounded sample of the projection workload (one thread): the set-up
    (MultiEpoch, windows, 50-knot kernel, halo tables at z_bar) and a few theta / l."""
    from oracle import chomp_oracle as o
    d2r = numpy.pi / 180.0
    t0 = time.perf_counter()
    me = o.multi_epoch(0.0, 5.0)
    wa = o.window_table("galaxy", o.dndz_maglim(0.0, 2.0, 2.0, 0.3, 2.0), me)
    wb = (o.window_table("convergence", o.dndz_gaussian(0.0, 2.0, 1.0, 0.2), me) if ggl else
          o.window_table("galaxy", o.dndz_maglim(0.0, 2.0, 2.0, 0.3, 2.0), me))
    kt = o.kernel_table(1e-6 * d2r, 100.0 * d2r, wa, wb, me, bessel_order=2 if ggl else 0)
    e = o.epoch(None, float(kt.z_bar))
    fam = ("gm",) if ggl else ("gg",)
    t = o.halo_table(e, o.mass_table(e), o.zheng(), families=fam)
    if ggl:
        o.halofit_table(t)
    D_z = float(o.me_growth(me, kt.z_bar))
    t_setup = time.perf_counter() - t0
    power = ((lambda k: o.halofit_power(t, "gm", k)) if ggl else
             (lambda k: o.halo_power(t, "gg", k)))
    theta = numpy.logspace(-3, 0, N_THETA)[::128] * d2r
    ell = numpy.logspace(1, 4, N_ELL)[::128]
    t0 = time.perf_counter()
    o.wtheta(kt, power, theta, t.k_min, t.k_max, D_z)
    o.cell(kt, power, ell, D_z)
    t_eval = time.perf_counter() - t0
    n = theta.size + ell.size
    full = t_setup + t_eval * (N_THETA + N_ELL) / n      # what the whole workload would take
    return {"value": (N_THETA + N_ELL) / full, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": "oracle set-up (%.1f s) + %d of %d theta and %d of %d l (%.1f s), the "
                      "evaluation time scaled to the full grid" % (
                          t_setup, theta.size, N_THETA, ell.size, N_ELL, t_eval),
            "host_cores_available": os.cpu_count()}


# ---------------------------------------------------------------------------
# GPU legs
# ---------------------------------------------------------------------------
class Dist(object):
    """What a leg needs to know about the job: ranks, fences, the max over ranks."""

    def __init__(self, args, world, rank, local, dev):
        self.args, self.world, self.rank, self.local, self.dev = args, world, rank, local, dev

    def fence(self):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize(self.dev)
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, seconds):
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64,
                         device="cpu" if self.args.rehearse else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


def rehearsal_gathers():
    """--rehearse only (every rank on ONE GPU, gloo): device tensors cannot go through gloo, so
    the two gathers of chomp_amd.grid are replaced by versions staged through host memory.  Same
    shapes, same re-ordering; a functional check of the N > 1 path, never a measurement."""
    import torch
    import torch.distributed as dist
    from chomp_amd import grid

    def gather_rows_async(local, n_all, world, out=None, **_):
        rpr = grid.rows_per_rank(n_all, world)
        assert local.shape[0] == rpr
        host = torch.empty((world * rpr, local.shape[1]), dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.cpu().contiguous())
        return grid.PendingRows(host.to(local.device), None, n_all, world)

    def gather_samples(local, n_all, world, **_):
        per = local.shape[0]
        host = torch.empty(world * per, dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.cpu().contiguous())
        return host.to(local.device).view(world, per).t().reshape(-1)[:n_all]

    grid.gather_rows_async = gather_rows_async
    grid.gather_samples = gather_samples


def step_distribution(step, n, stream, dev):
    """Per-step durations of n further steps (HIP events on the stream the steps are queued on,
    one behind every step; run AFTER a leg's timed region so that the headline's K steps carry
    no event packets): median / p95 / max beside the leg's mean, and where the slowest step
    was -- a transient is reported, not averaged away."""
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize(dev)
    ev[0].record(stream)
    for i in range(n):
        step()
        ev[i + 1].record(stream)
    torch.cuda.synchronize(dev)
    d = numpy.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n)])
    return {"ms_per_step_median": float(numpy.median(d)),
            "ms_per_step_p95": float(numpy.percentile(d, 95)),
            "ms_per_step_max": float(d.max()), "ms_per_step_min": float(d.min()),
            "slowest_step_index": int(d.argmax()), "steps_in_distribution": int(n),
            "clock": "HIP events behind every step, a pass of its own after the timed region"}


def dropin_leg(D, passes=3):
    """The reference-shaped call pattern (halo.py:255-264, 277-320): ONE Halo
    object, `set_redshift(z)` + `power_mm(k)` per redshift with host arrays in and out -- what
    an existing script does, unchanged.  Every z is a full single-epoch Stage K (a latency
    chain: each launch lasts as long as its slowest unit) + Stage E + the result's copy to the
    host; the headline batches the 64 redshifts into one set-up instead."""
    import warnings
    from chomp_amd import halo
    k = numpy.logspace(-3, 2, NK)
    z = numpy.linspace(0.0, Z_MAX, NZ)
    h = halo.Halo(float(z[-1]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for zz in z[:8]:
            h.set_redshift(float(zz))
            out = h.power_mm(k)
        per = []
        for _ in range(passes):
            t0 = time.perf_counter()
            for zz in z:
                h.set_redshift(float(zz))
                out = h.power_mm(k)
            per.append(time.perf_counter() - t0)
    assert out.shape == (NK,) and bool(numpy.isfinite(out).all())
    best = min(per)
    return {"workload": "one Halo: %d x (set_redshift(z) + power_mm(k[%d])), host arrays in and out"
                        % (NZ, NK),
            "ms_per_z": 1e3 * best / NZ, "ms_per_pass": 1e3 * best,
            "samples_per_s": NZ * NK / best, "passes": passes,
            "ms_per_pass_all": [1e3 * t for t in per], "clock": "host wall clock, best pass"}


def grid_leg(D, which, mf, nz, steps, warmup, stream):
    """`steps` timed steps of the (k, z) grid workload with nz redshift rows in all, dealt
    over the ranks (interleaved).  Steps are software-pipelined for N > 1: the all-gather of
    step i (RCCL's own stream) overlaps Stage K of step i + 1; every step's gather has
    completed and been re-ordered inside the timed region.  Returns (seconds, hg, k, out)."""
    import torch
    from chomp_amd import grid
    z = numpy.linspace(0.0, Z_MAX, nz)
    hg = grid.HaloGrid(z, mass_function=mf, device=D.local, stream=stream.cuda_stream,
                       rank=D.rank, world=D.world)
    k = torch.logspace(-3, 2, NK, dtype=torch.float64, device=D.dev)

    def run(n_steps):
        out, pending = None, None
        for _ in range(n_steps):
            hg.setup(which)                             # Stage K, this rank's redshifts
            nxt = hg.power_all_async(which, k)          # Stage E + all-gather launch
            if pending is not None:
                out = pending.wait()
            pending = nxt
        if pending is not None:
            out = pending.wait()
        return out

    out = run(warmup)
    D.fence()
    t0 = time.perf_counter()
    out = run(steps)
    D.fence()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    assert out.shape == (nz, NK) and bool(torch.isfinite(out).all())
    hg.bench_step = lambda: run(1)          # (step_distribution: one whole step, gather included)
    return elapsed, hg, k, out


BATCH_SIZES = (64, 256, 1024)
COSMO_FIELDS = ("omega_m0", "omega_b0", "omega_l0", "omega_r0", "cmb_temp", "h", "sigma_8",
                "n_scalar", "w0", "wa")


def batch_cosmologies(n, draws, seed):
    """`draws` parameter sets for n epochs, each a float64 [n, 10] block in chomp_cosmo's field
    order: WMAP7 with Omega_m and sigma_8 jittered by +-2 % per epoch, flat (Omega_L follows)."""
    from chomp_amd import defaults
    base = numpy.array([float(defaults.default_cosmo_dict[f]) for f in COSMO_FIELDS])
    rng = numpy.random.default_rng(seed)
    out = []
    for _ in range(draws):
        a = numpy.tile(base, (n, 1))
        om = (base[0] + base[3]) * (1.0 + 0.02 * rng.uniform(-1.0, 1.0, n))
        a[:, 0] = om - base[3]
        a[:, 2] = 1.0 - om
        a[:, 6] = base[6] * (1.0 + 0.02 * rng.uniform(-1.0, 1.0, n))
        out.append(a)
    return out


def batch_leg(D, n, distinct, steps, warmup, stream, which="power_mm", mf="st"):
    """SURVEY 8(f) rank 1, the batch-over-parameters axis (simulation_design.py:116-155, the MCMC
    loop of examples/example_script.py:141-143): n epochs (z = linspace(0, 1.5, n)) x 4096 k per
    step.  distinct=False: one cosmology for all (configs[1] made longer; the cosmology-only work
    is shared and no parameter changes between steps).  distinct=True: every epoch its own
    cosmology AND a fresh draw every step (a pool of 4 pre-packed draws taken in turn, so that
    every step's block differs from what the device holds: no upload is skipped, the host packs
    nothing inside the step).  Returns the result dict of one batch_scaling entry."""
    import torch
    from chomp_amd import grid
    z = numpy.linspace(0.0, Z_MAX, n)
    hg = grid.HaloGrid(z, mass_function=mf, device=D.local, stream=stream.cuda_stream)
    pool = None
    if distinct:
        pool = [hg.ctx.pack_cosmo(a, n) for a in batch_cosmologies(n, 4, 1000 + n)]
    k = torch.logspace(-3, 2, NK, dtype=torch.float64, device=D.dev)
    out = torch.empty((n, NK), dtype=torch.float64, device=D.dev)
    turn = [0]

    def setup():
        if pool is not None:
            hg.set_parameters(cosmo=pool[turn[0] % len(pool)])
            turn[0] += 1
        hg.setup(which)

    def step():
        setup()
        hg.power(which, k, out=out)

    for _ in range(warmup):
        step()
    D.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    D.fence()
    elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(out).all())
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(3, steps // 2)
    ev0.record(stream)
    for _ in range(reps):
        setup()
    ev1.record(stream)
    torch.cuda.synchronize(D.dev)
    t_k = ev0.elapsed_time(ev1) / reps * 1e-3
    status = hg.status()
    res = {"n_epoch": n, "distinct_cosmologies": bool(distinct),
           "parameters_change_every_step": bool(distinct),
           "ms_per_step": 1e3 * elapsed / steps, "samples_per_s": n * NK * steps / elapsed,
           "us_per_epoch": 1e6 * elapsed / steps / n, "stage_k_ms": t_k * 1e3, "steps": steps,
           "epochs_flagged": int(numpy.count_nonzero(status))}
    kind = "distinct" if distinct else "one"
    rk = stage_k_roofline(t_k, "batch_%d_%s" % (n, kind))
    if rk is not None:
        res["stage_k_frac"] = rk["frac"]
        res["stage_k_tflops"] = rk["achieved"]
        res["flop_per_step"] = rk["flop_per_step"]
        res["flop_source"] = "SQ counters of this batch (%s)" % rk["note"].split("counters: ")[-1]
        if "frac_at_round3_flop_count" in rk:
            res["stage_k_frac_at_round3_flop_count"] = rk["frac_at_round3_flop_count"]
    else:
        # no counter pass of this size: FLOP per epoch of the measured 1024-epoch batch of the
        # same kind (the per-epoch work does not depend on the batch size; the cosmology-only
        # part is counted per epoch there too)
        ref = stage_k_roofline(t_k, "batch_1024_%s" % kind)
        if ref is not None:
            flop = ref["flop_per_step"] * n / 1024.0
            res["stage_k_tflops"] = flop / t_k / 1e12
            res["stage_k_frac"] = flop / t_k / 1e12 / FP64_VALU_PEAK_TFLOPS
            res["flop_per_step"] = flop
            res["flop_source"] = "scaled by n / 1024 from the counters of the 1024-epoch batch"
        else:
            res["stage_k_frac"] = None
    return res


def projection_leg(D, ggl, steps, warmup):
    """configs[3] (c4: gal-gal clustering, J0, P_gg) and configs[4] (c5: galaxy-galaxy
    lensing, J2, HaloFit P_gm): a step = the whole projection path -- MultiEpoch chi(z),
    both windows, the 50-knot Bessel kernel and z_bar, the halo tables at z_bar, then
    w(theta) at 1024 theta and C_l at 2048 l -- with theta / l sharded over the ranks
    (set-up replicated, as SURVEY 8(e) prescribes) and one all-gather each."""
    import contextlib
    import torch
    from chomp_amd import cosmology, correlation, halo, kernel, grid
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    with contextlib.redirect_stdout(sys.stderr):     # (the reference's z_max warning)
        lens_a = kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0)
        lens_b = kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0)
    wa = kernel.WindowFunctionGalaxy(lens_a, cm)
    if ggl:
        wb = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2), cm)
        kern = kernel.GalaxyGalaxyLensingKernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
        h = halo.HaloFit(0.0)
        spec = "power_gm"
    else:
        wb = kernel.WindowFunctionGalaxy(lens_b, cm)
        kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
        h = halo.Halo(0.0)
        spec = "power_gg"
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")               # (divmax warnings of the HOD integrands)
        corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec=spec)
    theta = torch.logspace(-3, 0, N_THETA, dtype=torch.float64, device=D.dev) * d2r
    ell = torch.logspace(1, 4, N_ELL, dtype=torch.float64, device=D.dev)
    my_theta = grid.shard_samples(theta, D.rank, D.world)
    my_ell = grid.shard_samples(ell, D.rank, D.world)

    def step():
        # forget every table: the step rebuilds the projection and the halo model
        kern._done.clear()
        h._epoch_sig = None
        h._nbar_valid = False
        h._reset_flags(all_tables=True)
        if ggl:
            h._initialized_sigma_spline = False
        ctx, code = corr._prepare(defer_status=True)   # (device in, device out: no sync)
        # (one call for both observables: C_l runs beside w(theta) on the context's side stream)
        w, c = ctx.wtheta_cell(code, 0, corr._k_lim[0], corr._k_lim[1], corr.D_z, my_theta, my_ell)
        if D.world > 1:                      # one all-gather per output array
            w = grid.gather_samples(w, N_THETA, D.world)
            c = grid.gather_samples(c, N_ELL, D.world)
        return w, c

    import warnings as _w
    with _w.catch_warnings():
        _w.simplefilter("ignore")
        for _ in range(warmup):
            w, c = step()
        D.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            w, c = step()
        D.fence()
        elapsed = D.max_over_ranks(time.perf_counter() - t0)
        assert w.numel() == N_THETA and c.numel() == N_ELL
        assert bool(torch.isfinite(w).all()) and bool((c > 0).all())
        dist = None
        if D.world == 1:
            dist = step_distribution(step, steps, torch.cuda.current_stream(D.dev), D.dev)
        # ---- the same step replayed from ONE HIP graph (the side stream's forks and joins are
        # captured with the rest): an eager step is ~25 launches on two streams and close to
        # host-bound; a replay is one host call, every kernel of the step still runs
        graph = None
        if D.world == 1 and not D.args.no_graph:
            try:
                cur = torch.cuda.current_stream(D.dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cur):
                    wg, cg = step()
                for _ in range(max(2, warmup)):
                    g.replay()
                D.fence()
                t0 = time.perf_counter()
                for _ in range(steps):
                    g.replay()
                D.fence()
                eg = time.perf_counter() - t0
                same = bool(torch.equal(wg, w)) and bool(torch.equal(cg, c))
                graph = dict({"ms_per_step": 1e3 * eg / steps, "steps": steps,
                              "equals_eager_step_bit_for_bit": same},
                             **{k: v for k, v in step_distribution(g.replay, steps, cur, D.dev).items()
                                if k.startswith("ms_per_step_") or k == "slowest_step_index"})
            except Exception as exc:   # noqa: BLE001  (reported, never fatal for the eager numbers)
                if os.environ.get("CHOMP_BENCH_DEBUG"):
                    raise
                graph = {"error": str(exc)[:300]}
    return elapsed, dist, graph


def stage_k_roofline(stage_k_seconds, workload):
    """Stage K against the vector-fp64 peak: FLOP per step from the SQ counters committed
    under profiles/ (rocprofv3 --pmc SQ_INSTS_VALU_*_F64 of the same kernels on the same
    workload; 64 lanes x (2 FMA + ADD + MUL + TRANS) per wavefront instruction)."""
    cnt = None
    for name in STAGE_K_COUNTER_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                cnt = json.load(fh)[workload]
            counters_file = name
            break
        except (OSError, ValueError, KeyError):
            continue
    if cnt is None:
        return None
    flop = float(cnt["fp64_flop_per_step"])
    out = {"bound": "fp64_valu", "achieved": flop / stage_k_seconds / 1e12,
           "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": flop / stage_k_seconds / 1e12 / FP64_VALU_PEAK_TFLOPS,
           "flop_per_step": flop, "stage_k_ms": stage_k_seconds * 1e3,
           "valu_insts_per_step": cnt.get("valu_insts_per_step"),
           "kernels": cnt.get("kernels"),
           "note": "fp64 operations the kernels of THIS round execute for the step (SQ counters); "
                   "chain of %d dependent launches; counters: profiles/%s"
                   % (len(cnt.get("kernels", [])), counters_file)}
    # The same work priced at what the previous round's kernels executed for it: a round that
    # removes redundant fp64 operations (divisions by powers of two, per-wavefront copies of a
    # prologue) lowers `flop_per_step` with the time, and `frac` alone then hides the speed-up.
    try:
        with open(os.path.join(ROOT, "profiles", "round3_stage_k_counters.json")) as fh:
            f3 = float(json.load(fh)[workload]["fp64_flop_per_step"])
        out["flop_per_step_round3_kernels"] = f3
        out["frac_at_round3_flop_count"] = f3 / stage_k_seconds / 1e12 / FP64_VALU_PEAK_TFLOPS
    except (OSError, ValueError, KeyError):
        pass
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU)
    with torch.distributed.run as CHILDREN of this process -- which never initialises the GPU
    and is never replaced by another program -- relay rank 0's JSON line and return the
    launcher's exit code."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
        "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # --standalone: the launcher picks the rendezvous port itself (a port found here by binding
    # and closing a socket could be taken by someone else before the ranks get to it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr",
           "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(n),
           os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for text in proc.stdout:
        if text.startswith("{") and '"metric"' in text:
            line = text.rstrip("\n")
        else:
            sys.stderr.write(text)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited cleanly but rank 0 printed no result\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5", "batch"])
    ap.add_argument("--batch-n", type=int, default=1024,
                    help="--workload batch: epochs per step")
    ap.add_argument("--batch-distinct", type=int, default=1,
                    help="--workload batch: 1 = a cosmology per epoch, redrawn every step; "
                         "0 = one cosmology")
    ap.add_argument("--no-batch-scaling", action="store_true")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1, c2 / c3: which split is the headline (the other one is timed "
                         "too and reported beside it).  Default: strong -- configs[1]'s 64 "
                         "redshift rows dealt over the ranks, as north_star shards it")
    ap.add_argument("--roofline-nk", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true",
                    help="development runs: the timed steps only (no stage split, roofline, "
                         "other configs)")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="c4 / c5: do not also time the step replayed from a HIP graph")
    ap.add_argument("--rehearse", action="store_true",
                    help="test-only: run the N > 1 path on ONE GPU (every rank on device 0, "
                         "gloo all-gather through host memory); the numbers mean nothing")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has not
        # imported torch.cuda or touched HIP in any way and never will.
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N` or "
                 "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)"
                 % (args.gpus, world))
    proj = args.workload in ("c4", "c5")
    which = "power_mm" if args.workload == "c2" else "power_gm"
    mf = "st" if args.workload == "c2" else "tinker"
    baseline = None
    if rank == 0 and not args.no_cpu_baseline and args.workload != "batch":
        # rank 0 only, at every N; before this process initialises the GPU or joins the process
        # group: the pool forks (the other ranks wait for it in init_process_group)
        if proj:
            baseline = projection_baseline(args.workload == "c5")
        else:
            z1 = numpy.linspace(0.0, Z_MAX, NZ)
            if args.workload == "c2":
                baseline = cpu_baseline(which, mf, z1, numpy.concatenate([z1, z1]))
            else:
                baseline = cpu_baseline(which, mf, z1[[0, 21, 42, 63]], z1[::4])

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    if args.rehearse:
        local = 0
        os.environ["CHOMP_DEVICE"] = "0"      # the mirror classes pick their device from here
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        backend = dist.get_backend()
    D = Dist(args, world, rank, local, dev)
    # A stream of our own: on the legacy NULL stream every torch fill / memset on this
    # image serialises against the whole device (+0.1 ms per step, measured).
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    # rccl_world_size: ranks that talk RCCL -- None when no nccl group exists (one rank without
    # a process group, or the gloo rehearsal: neither says anything about RCCL)
    job = {"rccl_world_size": dist.get_world_size() if backend == "nccl" else None,
           "world_size": world, "backend": backend or "none"}
    if args.rehearse and world > 1:
        rehearsal_gathers()

    def finish(res):
        if rank == 0:
            print(json.dumps(res))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()

    # ---- the batch-over-parameters axis as the headline (profiling runs: tools/prof.sh sq)
    if args.workload == "batch":
        assert world == 1, "--workload batch is a one-GPU measurement"
        b = batch_leg(D, args.batch_n, bool(args.batch_distinct), args.steps, args.warmup, stream)
        return finish({"metric": "halo-model P(k,z) samples/sec (batch over parameters)",
                       "value": b["samples_per_s"], "unit": "samples/s", "n_gpus": 1,
                       "steps": args.steps, "warmup": args.warmup, "ms_per_step": b["ms_per_step"],
                       "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                       "dtype": "f64", "data": "synthetic",
                       "config": {"workload": "power_mm, %d epochs x %d k, %s" % (
                           args.batch_n, NK, "a cosmology per epoch, redrawn every step"
                           if args.batch_distinct else "one cosmology")},
                       "job": job, "batch": b})

    # ---- projection workloads as the headline
    if proj:
        ggl = args.workload == "c5"
        elapsed, pdist, pgraph = projection_leg(D, ggl, args.steps, args.warmup)
        res = {"metric": "Limber w(theta) + C_l samples/sec (projection and halo set-up included)",
               "value": (N_THETA + N_ELL) * args.steps / elapsed, "unit": "samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic" + (" (REHEARSAL on one GPU: not a measurement)"
                                      if args.rehearse else ""),
               "config": {"workload": "configs[%d]: %s, %d theta (logspace -3..0 deg) + %d l "
                                      "(logspace 1..4), WMAP7, halo at z_bar" % (
                                          4 if ggl else 3,
                                          "galaxy-galaxy lensing J2 kernel + HaloFit power_gm" if ggl
                                          else "gal-gal clustering J0 kernel + power_gg",
                                          N_THETA, N_ELL),
                          "n_theta": N_THETA, "n_ell": N_ELL,
                          "sharding": "theta and l interleaved over %d rank(s), set-up "
                                      "replicated, one all-gather each" % world},
               "job": job, "roofline": None}
        if pdist is not None:
            res.update(pdist)
        res["launch"] = "eager: one host call per kernel, two streams"
        if pgraph is not None:
            res["hip_graph_replay"] = pgraph
            if pgraph.get("equals_eager_step_bit_for_bit") and pgraph["ms_per_step"] < res["ms_per_step"]:
                # (the faster way to run the SAME step is the headline; the eager one stays beside it)
                res["eager"] = {k: res[k] for k in list(res) if k.startswith("ms_per_step")}
                res["eager"]["value"] = res["value"]
                res["ms_per_step"] = pgraph["ms_per_step"]
                res["value"] = (N_THETA + N_ELL) / (1e-3 * pgraph["ms_per_step"])
                for k in list(res):
                    if k.startswith("ms_per_step_") and k in pgraph:
                        res[k] = pgraph[k]
                res["launch"] = "HIP graph replay of the whole step (one host call)"
        if baseline is not None:
            res["cpu_baseline"] = baseline
        return finish(res)

    # ---- the (k, z) grid workloads
    # weak: every GPU carries configs[1]'s 64 redshift rows (global grid 4096 k x 64 N z; at
    # N = 1 exactly configs[1]); strong: configs[1]'s 64 rows dealt over the ranks.
    nz_of = {"weak": NZ * world, "strong": NZ}
    # (one rank: both splits are configs[1] itself; the label stays "weak" -- per-GPU work fixed
    #  -- so that a scaling run's N = 1 line equals the plain bench line)
    head = args.scaling or ("weak" if world == 1 else "strong")
    elapsed, hg, k, out = grid_leg(D, which, mf, nz_of[head], args.steps, args.warmup, stream)
    head_dist = step_distribution(hg.bench_step, args.steps, stream, dev)
    nz = nz_of[head]
    ms_per_step = 1e3 * elapsed / args.steps
    value = nz * NK * args.steps / elapsed
    n_local = len(hg.idx)
    rows = None
    if world > 1:
        t = torch.zeros(world, dtype=torch.int64, device="cpu" if args.rehearse else dev)
        t[rank] = n_local
        dist.all_reduce(t)
        rows = [int(x) for x in t.tolist()]
    job["rows_per_rank"] = rows if rows is not None else [n_local]
    other_split = None
    if world > 1 and not args.no_roofline:
        alt = "strong" if head == "weak" else "weak"
        e2, _, _, _ = grid_leg(D, which, mf, nz_of[alt], args.steps, args.warmup, stream)
        other_split = {"scaling": alt, "nz": nz_of[alt], "ms_per_step": 1e3 * e2 / args.steps,
                       "value": nz_of[alt] * NK * args.steps / e2, "unit": "samples/s",
                       "rows_per_rank": [len(range(r, nz_of[alt], world)) for r in range(world)]}

    if args.no_roofline:
        return finish(dict({"metric": "halo-model P(k,z) samples/sec (development run)",
                            "value": value, "unit": "samples/s", "n_gpus": world,
                            "steps": args.steps, "warmup": args.warmup,
                            "ms_per_step": ms_per_step, "scaling": head, "job": job},
                           **head_dist))

    # ---- stage split and rooflines (rank 0's shard; HIP events on the kernel stream)
    def timed(fn, reps):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize(dev)
        ev0.record(stream)
        for _ in range(reps):
            fn()
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        return ev0.elapsed_time(ev1) / reps * 1e-3

    t_setup = timed(lambda: hg.setup(which), max(3, args.steps // 2))
    buf = torch.empty((n_local, NK), dtype=torch.float64, device=dev)
    t_e_c2 = timed(lambda: hg.power(which, k, out=buf), 50)
    nk_big = args.roofline_nk
    k_big = torch.logspace(-3, 2, nk_big, dtype=torch.float64, device=dev)
    buf_big = torch.empty((n_local, nk_big), dtype=torch.float64, device=dev)
    # Per-kernel durations of the call: HIP events recorded by the library around its launches,
    # on the stream they run on; the last call of each back-to-back train of 40.  NO private
    # warm-up: the trains start right behind the compute-bound Stage K phase, where the first
    # ~100 launches run at a varying, lower rate while the device's clocks settle, and
    # `avg_launch_us` (hence roofline.frac) is the mean over ALL trains -- what a rocprofv3
    # --stats average of the same command sees.  The settled rate (last 5 trains) is reported
    # beside it as frac_steady.
    hg.ctx.set_timing(True)
    per_kernel = []
    t_calls = []
    for _ in range(12):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(40):
            hg.power(which, k_big, out=buf_big)
        ev1.record(stream)
        per_kernel.append(hg.ctx.get_timing())
        ev1.synchronize()
        t_calls.append(ev0.elapsed_time(ev1) / 40 * 1e-3)
    t_prep, t_stream, t_lanes = (float(x) * 1e-6 for x in numpy.mean(per_kernel, axis=0))
    t_stream_steady = float(numpy.mean([p[1] for p in per_kernel[-5:]])) * 1e-6
    t_e_big = float(numpy.mean(t_calls))
    # the same grid registered once (chomp_power_plan: the k-only table is kept, a grid
    # without k groups for the per-lane pass skips that launch): what a caller who evaluates
    # many (cosmology-preserving) set-ups on one k grid sees per call
    hg.ctx.power_plan(k_big)
    for _ in range(50):
        hg.power(which, k_big, out=buf_big)
    per_kernel = []
    for _ in range(5):
        for _ in range(40):
            hg.power(which, k_big, out=buf_big)
        per_kernel.append(hg.ctx.get_timing())
    hg.ctx.set_timing(False)
    tp_prep, tp_stream, tp_lanes = (float(x) * 1e-6 for x in numpy.mean(per_kernel, axis=0))
    t_e_planned = timed(lambda: hg.power(which, k_big, out=buf_big), 100)
    # Algorithmic bytes of one Stage-E launch: k is read once (8 B per k) and one
    # P value is written per (k, z) sample (8 B).  SURVEY 8(d) prices the per-z
    # explicit-k call at 16 B/sample (k re-read for every z); the grid launch shares
    # the k read across its z rows, so the honest figure for THIS launch shape is
    # 8 nk + 8 nk nz.  Both are reported; `achieved` uses the launch's own bytes.
    bytes_big = 8.0 * nk_big + 8.0 * n_local * nk_big
    bytes_c2 = 8.0 * NK + 8.0 * n_local * NK
    traffic = traffic_call = None
    try:      # HBM bytes per launch from rocprofv3 PMC passes (profiles/)
        with open(os.path.join(ROOT, "profiles", "stage_e_pmc.json")) as fh:
            pmc = json.load(fh)
        if pmc.get("nk") == nk_big and pmc.get("nz") == n_local:
            traffic_call = pmc["hbm_bytes_per_launch"]
            name = [n for n in pmc["WRITE_SIZE_KiB_raw_by_kernel"] if n.startswith("k_power_stream")][0]
            traffic = 1024.0 * (2.0 * pmc["FETCH_SIZE_KiB_raw_by_kernel"][name] +
                                pmc["WRITE_SIZE_KiB_raw_by_kernel"][name])
    except (OSError, ValueError, KeyError, IndexError):
        pass
    roof = {"bound": "hbm", "kernel": "k_power_stream (the dominant kernel of a Stage E call on a "
                      "large grid: it writes every output sample; HIP events around its launch on "
                      "the context's stream)",
            "workload": "%d k x %d z (enlarged grid, SURVEY 8(d))" % (nk_big, n_local),
            "achieved": bytes_big / t_stream / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_big / t_stream / 1e9 / HBM_PEAK_GBS,
            "frac_steady": bytes_big / t_stream_steady / 1e9 / HBM_PEAK_GBS,
            "steady_launch_us": t_stream_steady * 1e6,
            "averaging": "avg_launch_us / frac: mean over 12 trains of 40 calls starting right "
                         "behind Stage K (no private warm-up; comparable with a rocprofv3 --stats "
                         "average); steady: the last 5 trains",
            "traffic": traffic,
            "traffic_source": "profiles/stage_e_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                              "passes of this command, FETCH doubled per MI355X_MICROARCH.md); "
                              "not re-measured by this run",
            "bytes_per_launch": bytes_big, "avg_launch_us": t_stream * 1e6,
            # the whole chomp_power call as a caller sees it (HIP events around the trains)
            "whole_call": {"kernels_us": {"k_power_prep": t_prep * 1e6,
                                          "k_power_stream": t_stream * 1e6,
                                          "k_power_grid_lanes": t_lanes * 1e6},
                           "avg_call_us": t_e_big * 1e6, "traffic": traffic_call,
                           "achieved": bytes_big / t_e_big / 1e9,
                           "frac": bytes_big / t_e_big / 1e9 / HBM_PEAK_GBS},
            "whole_call_registered_grid": {
                "kernels_us": {"k_power_prep": tp_prep * 1e6, "k_power_stream": tp_stream * 1e6,
                               "k_power_grid_lanes": tp_lanes * 1e6},
                "avg_call_us": t_e_planned * 1e6,
                "achieved": bytes_big / t_e_planned / 1e9,
                "frac": bytes_big / t_e_planned / 1e9 / HBM_PEAK_GBS,
                "note": "chomp_power_plan: the k-only table of a registered k grid is kept "
                        "across calls"},
            "achieved_at_16B_per_sample": 16.0 * n_local * nk_big / t_e_big / 1e9,
            # what the TIMED step runs: a 4096 x 64 grid is 2 MB -- one small launch of the
            # row-walking kernel, bound by its launch, not by HBM
            "timed_step_kernel": {"kernel": "k_power_grid", "achieved": bytes_c2 / t_e_c2 / 1e9,
                                  "frac": bytes_c2 / t_e_c2 / 1e9 / HBM_PEAK_GBS,
                                  "avg_launch_us": t_e_c2 * 1e6, "bytes_per_launch": bytes_c2},
            "samples_per_s_stage_e_only": n_local * nk_big / t_e_big}
    del buf_big, k_big

    res = {
        "metric": "halo-model P(k,z) samples/sec (Stage K set-up + Stage E grid)",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": head, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" + (" (REHEARSAL on one GPU: not a measurement)"
                                               if args.rehearse else ""),
        "config": {"workload": "configs[%d]: %s, WMAP7, %s mass function, %d k "
                               "(logspace -3..2) x %d z (linspace 0..%.1f)"
                               % (1 if args.workload == "c2" else 2, which,
                                  "Sheth-Tormen" if mf == "st" else "Tinker10 + Zheng HOD",
                                  NK, nz, Z_MAX),
                   "nk": NK, "nz": nz, "sharding": "z interleaved over %d rank(s), one all-gather per step"
                               "%s" % (world, ", overlapped with the next step's Stage K"
                                       if world > 1 else "")},
        "job": job,
        "stage_split_rank0": {"stage_k_ms": t_setup * 1e3, "stage_e_ms": t_e_c2 * 1e3,
                              "n_local_z": n_local},
        "roofline": roof,
    }
    res.update(head_dist)
    if n_local == NZ:
        rk = stage_k_roofline(t_setup, args.workload)
        if rk is not None:
            res["roofline_stage_k"] = rk
    if other_split is not None:
        res["strong_scaling" if other_split["scaling"] == "strong" else "weak_scaling"] = other_split
    if baseline is not None:
        res["cpu_baseline"] = baseline

    # ---- the other configs, GPU side, same process (N = 1 only)
    if world == 1 and not args.no_other_configs and args.workload == "c2":
        other = {}
        del hg
        n_o = 30                                     # (timed steps of each of these legs)
        e3, hg3, k3, _ = grid_leg(D, "power_gm", "tinker", NZ, n_o, 3, stream)
        d3 = step_distribution(hg3.bench_step, n_o, stream, dev)
        ts3 = timed(lambda: hg3.setup("power_gm"), 5)
        f3 = stage_k_roofline(ts3, "c3")
        other["c3"] = {"workload": "configs[2]: power_gm, Tinker10 + Zheng07, 4096 k x 64 z",
                       "ms_per_step": 1e3 * e3 / n_o, "value": NZ * NK * n_o / e3,
                       "unit": "samples/s", "steps": n_o,
                       "dominant_kernel": "k_halo_knots_fast (knots beyond the node tables: "
                                          "level sums from 2049 coarse samples)",
                       "stage_k_ms": ts3 * 1e3,
                       "frac": f3["frac"] if f3 else None,
                       "frac_of": "vector-fp64 peak, Stage K (FLOP per step: profiles/%s)"
                                  % STAGE_K_COUNTER_FILES[0],
                       "deep_knots_fast_literal_cumulative": list(hg3.ctx.deep_stats())}
        other["c3"].update(d3)
        del hg3
        for name, ggl in (("c4", False), ("c5", True)):
            ep, pd, pg = projection_leg(D, ggl, n_o, 3)
            other[name] = {"workload": "configs[%d]: %s, 1024 theta + 2048 l" % (
                               4 if ggl else 3, "GGL J2 kernel + HaloFit power_gm" if ggl
                               else "clustering J0 kernel + power_gg"),
                           "ms_per_step": 1e3 * ep / n_o, "value": (N_THETA + N_ELL) * n_o / ep,
                           "unit": "samples/s", "steps": n_o,
                           "dominant_kernel": "none above a fifth of the step: k_halo_knots_fast "
                                              "(knots beyond the node tables), k_cell + k_cell_deep, "
                                              "k_wtheta_nodes / _moments / _fast (level sums from "
                                              "prefix moments)",
                           "frac": None,
                           "profile": "profiles/%s_kernel_stats_%s.csv" % (PROFILE_ROUND, name)}
            if pd is not None:
                other[name].update(pd)
            other[name]["launch"] = "eager: one host call per kernel, two streams"
            if pg is not None:
                other[name]["hip_graph_replay"] = pg
                if pg.get("equals_eager_step_bit_for_bit") and pg["ms_per_step"] < other[name]["ms_per_step"]:
                    other[name]["eager"] = {k: other[name][k] for k in list(other[name])
                                            if k.startswith("ms_per_step") or k == "value"}
                    other[name]["ms_per_step"] = pg["ms_per_step"]
                    other[name]["value"] = (N_THETA + N_ELL) / (1e-3 * pg["ms_per_step"])
                    for k in list(other[name]):
                        if k.startswith("ms_per_step_") and k in pg:
                            other[name][k] = pg[k]
                    other[name]["launch"] = "HIP graph replay of the whole step (one host call)"
        res["other_configs"] = other
        # ---- the reference-shaped call pattern (one Halo, set_redshift + power_mm per z, host
        # arrays): what a drop-in script sees, beside the batched headline
        res["dropin"] = dropin_leg(D)
        res["dropin"]["vs_headline"] = res["dropin"]["samples_per_s"] / value
    # ---- the batch-over-parameters axis: where Stage K fills the chip
    if world == 1 and not args.no_batch_scaling and args.workload == "c2":
        try:
            del hg
        except NameError:
            pass
        res["batch_scaling"] = [batch_leg(D, n, distinct, 10 if n < 1024 else 6, 2, stream)
                                for distinct in (False, True) for n in BATCH_SIZES]
    finish(res)


if __name__ == "__main__":
    main()
