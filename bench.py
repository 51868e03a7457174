#!/usr/bin/env python3
"""Benchmark of the halo-model P(k, z) hot path on 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W

Default workload (the headline): a "step" is one full pass of the hot path over
BASELINE.json's configs[1]:
Stage K (every table of the 64 redshifts: sigma_8 normalisation, mass-limit search,
nu table, splines, normalisations, n_bar, the 50-knot 2-halo / 1-halo integrals)
followed by Stage E (P_mm on the 4096 x 64 (k, z) grid), with k resident in HBM
and, for N > 1, the all-gather that re-assembles the grid on every rank.  Rank 0
prints ONE JSON line (contract in the task description) carrying `roofline`
(Stage-E kernel, HIP events on the kernel's own stream) and `cpu_baseline` (the
NumPy oracle = a port of the reference's algorithm, timed on this box's host
cores on a bounded sample of the same workload).

--workload c3 is configs[2] (same grid, Tinker10 + Zheng07 P_gm); c4 / c5 are the
projection configs (w(theta) at 1024 theta + C_l at 2048 l, see projection_bench).
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


N_THETA, N_ELL = 1024, 2048


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


def projection_bench(args, world, rank, local):
    """configs[3] (c4: gal-gal clustering, J0, P_gg) and configs[4] (c5: galaxy-galaxy
    lensing, J2, HaloFit P_gm): a step = the whole projection path -- MultiEpoch chi(z),
    both windows, the 50-knot Bessel kernel and z_bar, the halo tables at z_bar, then
    w(theta) at 1024 theta and C_l at 2048 l -- with theta / l sharded over the ranks
    (set-up replicated, as SURVEY 8(e) prescribes) and one all-gather each."""
    ggl = args.workload == "c5"
    baseline = None
    if world == 1 and not args.no_cpu_baseline:
        baseline = projection_baseline(ggl)
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    if args.rehearse:
        local = 0
        os.environ["CHOMP_DEVICE"] = "0"      # the mirror classes pick their device from here
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.rehearse else "nccl",
                                **({} if args.rehearse else {"device_id": dev}))
    from chomp_amd import cosmology, correlation, halo, kernel
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    import contextlib
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
    corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec=spec)
    theta = torch.logspace(-3, 0, N_THETA, dtype=torch.float64, device=dev) * d2r
    ell = torch.logspace(1, 4, N_ELL, dtype=torch.float64, device=dev)
    from chomp_amd import grid
    my_theta = grid.shard_samples(theta, rank, world)
    my_ell = grid.shard_samples(ell, rank, world)

    def step():
        # forget every table: the step rebuilds the projection and the halo model
        kern._done.clear()
        h._epoch_sig = None
        h._nbar_valid = False
        h._reset_flags(all_tables=True)
        if ggl:
            h._initialized_sigma_spline = False
        ctx, code = corr._prepare()
        w = ctx.wtheta(code, 0, corr._k_lim[0], corr._k_lim[1], corr.D_z, my_theta)
        c = ctx.cell(code, 0, corr.D_z, my_ell)
        if world > 1:                      # one all-gather per output array
            w = grid.gather_samples(w, N_THETA, world, via_host=args.rehearse)
            c = grid.gather_samples(c, N_ELL, world, via_host=args.rehearse)
        return w, c

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        w, c = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w, c = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert w.numel() == N_THETA and c.numel() == N_ELL
    assert bool(torch.isfinite(w).all()) and bool((c > 0).all())
    if rank == 0:
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
               "roofline": None}
        if baseline is not None:
            res["cpu_baseline"] = baseline
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--roofline-nk", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true",
                    help="development runs: the timed steps only (no stage split, roofline)")
    ap.add_argument("--rehearse", action="store_true",
                    help="test-only: run the N > 1 path on ONE GPU (every rank on device 0, "
                         "gloo all-gather through host memory); the numbers mean nothing")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node N"
    if args.workload in ("c4", "c5"):
        return projection_bench(args, world, rank, local)
    which = "power_mm" if args.workload == "c2" else "power_gm"
    mf = "st" if args.workload == "c2" else "tinker"
    baseline = None
    if world == 1 and not args.no_cpu_baseline:
        # before the GPU is initialised: the pool forks
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
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from chomp_amd import grid
    # weak scaling: every GPU carries configs[1]'s 64 redshift rows, so the global
    # grid is 4096 k x (64 N) z; at N = 1 this is exactly configs[1].
    nz = NZ * world
    z = numpy.linspace(0.0, Z_MAX, nz)
    # A stream of our own: on the legacy NULL stream every torch fill / memset on this
    # image serialises against the whole device (+0.1 ms per step, measured).
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    hg = grid.HaloGrid(z, mass_function=mf, device=local, stream=stream.cuda_stream,
                       rank=rank, world=world)
    k = torch.logspace(-3, 2, NK, dtype=torch.float64, device=dev)

    def run(n_steps):
        """n_steps steps, software-pipelined for N > 1: the all-gather of step i (RCCL's
        own stream) overlaps Stage K of step i + 1; every step's gather has completed and
        been re-ordered before this returns."""
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

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    out = run(args.warmup)
    fence()
    t0 = time.perf_counter()
    out = run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cpu" if args.rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape == (nz, NK) and bool(torch.isfinite(out).all())
    ms_per_step = 1e3 * elapsed / args.steps
    value = nz * NK * args.steps / elapsed

    if args.no_roofline:
        if rank == 0:
            print(json.dumps({"metric": "halo-model P(k,z) samples/sec (development run)",
                              "value": value, "unit": "samples/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": ms_per_step}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    # ---- stage split and roofline (rank 0's shard; HIP events on the kernel stream)
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

    n_local = len(hg.idx)
    t_setup = timed(lambda: hg.setup(which), max(3, args.steps // 2))
    buf = torch.empty((n_local, NK), dtype=torch.float64, device=dev)
    t_e_c2 = timed(lambda: hg.power(which, k, out=buf), 50)
    nk_big = args.roofline_nk
    k_big = torch.logspace(-3, 2, nk_big, dtype=torch.float64, device=dev)
    buf_big = torch.empty((n_local, nk_big), dtype=torch.float64, device=dev)
    # (the first ~100 launches after the compute-bound Stage K phase run at a varying,
    #  lower rate while the device's clocks settle -- rocprof trace in DESIGN.md section 6 --
    #  so the streaming stage gets a warm-up train of its own)
    for _ in range(200):
        hg.power(which, k_big, out=buf_big)
    t_e_big = timed(lambda: hg.power(which, k_big, out=buf_big), 100)
    # per-kernel durations of the same call (HIP events recorded by the library around its
    # three launches, on the stream they run on): the last call of 5 back-to-back trains
    hg.ctx.set_timing(True)
    per_kernel = []
    for _ in range(5):
        for _ in range(40):
            hg.power(which, k_big, out=buf_big)
        per_kernel.append(hg.ctx.get_timing())
    hg.ctx.set_timing(False)
    t_prep, t_stream, t_lanes = (float(x) * 1e-6 for x in numpy.mean(per_kernel, axis=0))
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
    roof = {"bound": "hbm", "kernel": "k_power_stream (the dominant kernel of a Stage E call: it writes "
                      "every output sample; HIP events around its launch on the context's stream)",
            "workload": "%d k x %d z (enlarged grid, SURVEY 8(d))" % (nk_big, n_local),
            "achieved": bytes_big / t_stream / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_big / t_stream / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
            "bytes_per_launch": bytes_big, "avg_launch_us": t_stream * 1e6,
            # the whole chomp_power call = k_power_prep + k_power_stream + k_power_grid_lanes
            # (HIP events around 100 back-to-back calls): what a caller of Stage E sees
            "whole_call": {"kernels_us": {"k_power_prep": t_prep * 1e6,
                                          "k_power_stream": t_stream * 1e6,
                                          "k_power_grid_lanes": t_lanes * 1e6},
                           "avg_call_us": t_e_big * 1e6, "traffic": traffic_call,
                           "achieved": bytes_big / t_e_big / 1e9,
                           "frac": bytes_big / t_e_big / 1e9 / HBM_PEAK_GBS},
            "achieved_at_16B_per_sample": 16.0 * n_local * nk_big / t_e_big / 1e9,
            "c2_grid": {"achieved": bytes_c2 / t_e_c2 / 1e9, "avg_launch_us": t_e_c2 * 1e6,
                        "bytes_per_launch": bytes_c2},
            "samples_per_s_stage_e_only": n_local * nk_big / t_e_big}
    del buf_big, k_big

    if rank == 0:
        res = {
            "metric": "halo-model P(k,z) samples/sec (Stage K set-up + Stage E grid)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
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
            "stage_split_rank0": {"stage_k_ms": t_setup * 1e3, "stage_e_ms": t_e_c2 * 1e3,
                                  "n_local_z": n_local},
            "roofline": roof,
        }
        if baseline is not None:
            res["cpu_baseline"] = baseline
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
