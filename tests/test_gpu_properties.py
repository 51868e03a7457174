"""GPU tests at BASELINE.json's full sizes through size-independent properties, plus
edge cases of the boundary (empty / unsorted / out-of-range / NaN inputs, non-default
precision dictionaries, fixed mass limits) checked against the oracle."""
import copy

import numpy
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-4
PROJ_RTOL = 1e-9     # w(theta), C_l against the reference's vectors (measured 4e-11 / 7e-12)


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_c2_full_grid_properties(torch_mod):
    """configs[1] at full size (4096 k x 64 z): the grid launch must equal the
    reference-shaped per-z calls, be invariant under permutation / chunking of k
    (fast scalar-cache path vs per-lane path), and reduce to P_lin h_m^2 + pp_mm at
    the knots."""
    torch = torch_mod
    from chomp_amd import grid, halo, _lib
    z = numpy.linspace(0.0, 1.5, 64)
    hg = grid.HaloGrid(z)
    k = torch.logspace(-3, 2, 4096, dtype=torch.float64, device="cuda")
    full = hg.power("power_mm", k).cpu().numpy()
    assert full.shape == (64, 4096) and numpy.all(numpy.isfinite(full)) and numpy.all(full > 0)
    # (1) per-z calls through the mirror class (one context per Halo)
    for i in (0, 17, 63):
        h = halo.Halo(float(z[i]))
        assert rel_err(h.power_mm(k.cpu().numpy()), full[i]) < 1e-12
    # (2) permutation invariance: a shuffled k takes the per-lane path
    perm = torch.randperm(4096, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    shuf = hg.power("power_mm", k[perm].contiguous()).cpu().numpy()
    assert numpy.array_equal(shuf, full[:, perm.cpu().numpy()]) or \
        rel_err(shuf, full[:, perm.cpu().numpy()]) < 1e-13
    # (3) chunking invariance (odd chunk -> unaligned / non-vector path)
    a = hg.power("power_mm", k[:1001].contiguous()).cpu().numpy()
    b = hg.power("power_mm", k[1001:].contiguous()).cpu().numpy()
    assert rel_err(numpy.concatenate([a, b], axis=1), full) < 1e-13
    # (4) dense grid: 2^18 k exercises the scalar-cache fast path; sub-sampling it
    #     must reproduce the coarse grid
    kd = torch.logspace(-3, 2, (1 << 18) + 1, dtype=torch.float64, device="cuda")
    dense = hg.power("power_mm", kd)
    kk = kd[::64].contiguous()
    coarse = hg.power("power_mm", kk)
    assert rel_err(dense[:, ::64].cpu().numpy(), coarse.cpu().numpy()) < 1e-12
    # (5) at the knots P = P_lin h_m^2 + pp_mm
    lnk = numpy.linspace(numpy.log(1e-3), numpy.log(1e2), 50)[1:-1]
    kn = numpy.exp(lnk)
    pk = hg.power("power_mm", kn)
    lin = hg.power("linear_power", kn)
    for i in (0, 40):
        hm, pp = hg.ctx.table("h_m", i)[1:-1], hg.ctx.table("pp_mm", i)[1:-1]
        assert rel_err(pk[i], lin[i] * hm * hm + pp) < 1e-11


def test_stage_e_launch_shapes_agree(torch_mod):
    """Stage E has three launch shapes -- the row-major streaming pair (k_power_prep +
    k_power_stream, large grids of one cosmology), the row-walking kernel and the generic
    per-sample kernel -- plus the per-lane pass for k groups that are unsorted, ragged or
    out of range.  All must return the same numbers on the same input."""
    torch = torch_mod
    from chomp_amd import grid, _lib
    gen = torch.Generator("cuda").manual_seed(5)

    def run(hg, which, k, stream):
        hg.ctx.set_tuning(_lib.TUNE_E_STREAM_MIN, 0 if stream else 1 << 62)
        out = hg.power(which, k).clone()
        torch.cuda.synchronize()
        return out

    hg = grid.HaloGrid(numpy.linspace(0.0, 1.5, 6), mass_function="tinker")
    hg.setup("power_gg")
    cases = {
        "sorted": torch.logspace(-3, 2, 1 << 16, dtype=torch.float64, device="cuda"),
        "ragged, beyond both ends": torch.logspace(-4, 3, 70002, dtype=torch.float64, device="cuda"),
        "tiny": torch.logspace(-3, 2, 6, dtype=torch.float64, device="cuda"),
    }
    sh = torch.logspace(-3.5, 2.5, 1 << 15, dtype=torch.float64, device="cuda")
    cases["shuffled"] = sh[torch.randperm(sh.numel(), device="cuda", generator=gen)].contiguous()
    bad = cases["sorted"].clone()
    bad[::1000] = float("nan"); bad[1::1000] = 0.0; bad[2::1000] = -1.0; bad[3::1000] = float("inf")
    cases["nan / zero / negative / inf"] = bad
    for name, k in cases.items():
        for which in ("power_mm", "power_gm", "power_gg"):
            a = run(hg, which, k, True)
            b = run(hg, which, k, False)
            assert a.shape == (6, k.numel())
            assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), (name, which)
            # generic kernel: an 8-byte-aligned (not 16) slice cannot take either fast shape
            c = run(hg, which, k[1:], True)
            fin = torch.isfinite(a[:, 1:])
            assert torch.equal(fin, torch.isfinite(c)), (name, which)
            err = ((a[:, 1:] - c).abs() / c.abs().clamp_min(1e-300))[fin].max().item()
            assert err < 1e-12, (name, which, err)
    # odd number of rows (one row per block) and a sub-range of the epochs
    k = cases["sorted"]
    hg.ctx.set_tuning(_lib.TUNE_E_STREAM_MIN, 0)
    part = torch.empty((3, k.numel()), dtype=torch.float64, device="cuda")
    hg.ctx.power(3, k, 2, 3, out=part)               # P_gg, epochs 2..4
    full = run(hg, "power_gg", k, False)
    assert torch.equal(part, full[2:5])


def test_mass_limit_search_matches_the_walk():
    """The aim-and-certify search of k_epoch_init must stop on exactly the step the
    reference's 5 % walk stops at (mass_function.py:160-203), for cosmologies and redshifts
    well away from the fixtures: the ln M limits are compared bit for bit with the
    oracle's literal walk.

    The one regime where that cannot hold is REPORTED, not skipped: a walk that runs on until
    k R < 0.2 over the whole (clamped, cosmology.py:627-632) k range of sigma_r (M <~ 0.1
    M_sun/h; z >~ 0.8 with a low sigma_8 and Omega_m; the reference's search diverges near
    z ~ 1.9).  There nu(M) has converged to a constant ABOVE the band in exact arithmetic --
    the walk would never end -- and what ends it is the rounding error of the top-hat window
    3 (sin x / x^3 - cos x / x^2) at x << 1, whose variance biases sigma^2 upwards like
    eps^2 / x^4: the step the reference stops at is a property of the libm in use, not of the
    model (observed: 5 steps apart at M ~ 3e-8).  Such an epoch carries
    CHOMP_ST_MASS_MIN_SATURATED in its status word -- exactly when the oracle's walk ends in
    that regime too -- and every epoch without the flag is bit-equal."""
    from chomp_amd import grid, _lib
    from oracle import chomp_oracle as o
    rng = numpy.random.default_rng(20)
    base = o.default_cosmo_dict
    cds, zs = [], []
    for i in range(24):
        om = rng.uniform(0.2, 0.4)
        cds.append(dict(base, omega_m0=om - base["omega_r0"], omega_l0=1.0 - om,
                        omega_b0=rng.uniform(0.035, 0.055), h=rng.uniform(0.6, 0.8),
                        sigma_8=rng.uniform(0.65, 0.95), n_scalar=rng.uniform(0.9, 1.02)))
        zs.append(float(rng.choice([0.0, 0.05, 0.3, 0.77, 1.1, 1.45, 1.7]) if i < 14
                        else rng.uniform(0.0, 1.7)))
    hg = grid.HaloGrid(numpy.array(zs), cosmo_dict=cds)
    hg.setup("power_mm")
    status = hg.status()
    bad, n_checked, n_flagged = [], 0, 0
    for i in range(24):
        e = o.epoch(cds[i], zs[i])
        lo, hi, _ = o.mass_limits(e)
        sc = hg.ctx.scalars(i)
        assert numpy.isfinite(sc["ln_mass_min"]) and sc["ln_mass_max"] == hi, (i, zs[i])
        # did the oracle's walk end with k R < 0.2 over sigma_r's whole (clamped) k range?
        R = (3.0 * numpy.exp(lo) / (4.0 * numpy.pi * o.rho_bar(e))) ** (1.0 / 3.0)
        saturated = 100.0 * e.limits["k_max"] * R < 0.2
        flagged = bool(status[i] & _lib.ST_MASS_MIN_SATURATED)
        assert flagged == saturated, (i, zs[i], int(status[i]), float(lo), float(sc["ln_mass_min"]))
        assert not status[i] & (_lib.ST_MASS_MAX_SATURATED | _lib.ST_MASS_SEARCH_EXHAUSTED)
        if flagged:
            n_flagged += 1
            assert lo < numpy.log(1.0) and sc["ln_mass_min"] < numpy.log(1.0)
        else:
            n_checked += 1
            if sc["ln_mass_min"] != lo:
                bad.append((i, zs[i], float(sc["ln_mass_min"]), float(lo)))
    assert not bad, bad
    assert n_checked >= 18 and n_flagged >= 1, (n_checked, n_flagged)


def test_status_word_reports_what_scipy_warned_about():
    """chomp_get_status: an exhausted divmax (scipy's AccuracyWarning in the reference,
    halo.py:1065-1071 and alike) and a saturated mass-limit search reach the caller -- per
    epoch from the batch API, as Python warnings from the drop-in classes -- and a clean
    epoch reports 0."""
    import warnings
    from chomp_amd import grid, halo, hod, _lib
    from oracle import chomp_oracle as o
    hg = grid.HaloGrid(numpy.array([0.0, 1.0]))
    hg.setup("power_mm")
    assert list(hg.status()) == [0, 0]
    # P_gm at the default precision: the discontinuous HOD integrands exhaust divmax = 20
    hg.setup("power_gm")
    st = hg.status()
    for i, z in enumerate((0.0, 1.0)):
        e = o.epoch(None, z)
        m = o.mass_table(e)
        o.DIVMAX_EXCEEDED[0] = 0
        o.halo_table(e, m, o.zheng(), families=("gm",))
        oracle_warned = o.DIVMAX_EXCEEDED[0] > 0
        lev = hg.ctx.table("levels", i).reshape(5, -1)
        at_top = {name: bool((lev[f] == 20).any()) for f, name in
                  enumerate(("h_m", "pp_mm", "h_g", "pp_gm", "pp_gg")) if name != "pp_gg"}
        flagged = {name: bool(st[i] & bit) for name, bit in _lib.ST_HALO_DIVMAX.items()}
        assert any(flagged.values()) == oracle_warned, (z, flagged, oracle_warned)
        for name, f in flagged.items():
            if f:
                assert at_top[name], (z, name)       # a flag needs a knot that ran to divmax
        assert not st[i] & (_lib.ST_SATURATED | _lib.ST_NONFINITE | _lib.ST_SIGMA_DIVMAX)
    # the drop-in class warns like the reference did
    h = halo.Halo(0.0, hod.HODZheng())
    with pytest.warns(_lib.ChompAccuracyWarning, match="divmax"):
        h.power_gm(numpy.array([0.1, 1.0]))
    # a saturated search: low sigma_8 and Omega_m at z ~ 1.4 (one of the three cases a
    # randomised soak of 40 cosmologies found 5 % away from the oracle)
    cd = dict(o.default_cosmo_dict, omega_m0=0.22563, omega_b0=0.04499, omega_l0=0.77429,
              h=0.7203, sigma_8=0.70574, n_scalar=0.93183)
    hs = grid.HaloGrid(numpy.array([1.3923165344405541]), cosmo_dict=cd)
    hs.setup("power_mm")
    assert hs.status()[0] & _lib.ST_MASS_MIN_SATURATED
    assert any("saturated" in m for m in _lib.describe_status(int(hs.status()[0])))
    with pytest.warns(_lib.ChompParityWarning, match="saturated"):
        hs.status(warn=True)


def test_posted_status_is_the_status_and_warns_when_looked_at():
    """chomp_status_post / chomp_status_wait: the words copied behind a set-up equal a
    synchronous chomp_get_status, later work does not change them, and the drop-in classes'
    device-resident path (Correlation with torch inputs: no synchronisation anywhere) still
    delivers the divmax warning -- when the status is looked at."""
    import torch
    import warnings
    from chomp_amd import correlation, cosmology, grid, halo, kernel, _lib
    hg = grid.HaloGrid(numpy.array([0.0, 1.0]))
    with pytest.raises(_lib.ChompError, match='status_post'):
        hg.ctx.status_wait(0, 1)
    hg.setup("power_gm")
    hg.ctx.status_post()
    now = hg.status()
    assert now.any()
    hg.setup("power_mm")                     # clears the device's words ...
    assert list(hg.status()) == [0, 0]
    assert list(hg.ctx.status_wait(0, 2)) == list(now)     # ... not the posted copy
    # the words of a post are the finalising blocks' own writes into one of two pinned regions:
    # set-ups that come round to the posted region again collect it first, nothing is lost
    hg.setup("power_gm")
    hg.ctx.status_post()
    for _ in range(3):
        hg.setup("power_mm")
    assert list(hg.ctx.status_wait(0, 2)) == list(now)
    assert list(hg.ctx.status_wait(1, 1)) == [now[1]]
    hg.ctx.status_post()                     # ... and the next post is the last set-up's
    assert list(hg.ctx.status_wait(0, 2)) == [0, 0]
    # ... nor when the batch grows and the pinned words are reallocated under an unread post
    hg.setup("power_gm")
    hg.ctx.status_post()
    wide = grid.HaloGrid(numpy.linspace(0.0, 1.0, 9))
    hg.ctx.epochs_set(wide._c_cosmo, wide._z)
    assert list(hg.ctx.status_wait(0, 2)) == list(now)
    hg.ctx.epochs_set(hg._c_cosmo, hg._z)
    # a post behind a set-up that did not end in a halo set-up (mass function only) is a copy
    hg.ctx.epochs_set(hg._c_cosmo, hg._z)
    hg.ctx.status_post()
    assert list(hg.ctx.status_wait(0, 2)) == list(hg.ctx.status(0, 2))
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
    h = halo.Halo(0.0)
    corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec="power_gg")
    theta = torch.logspace(-3, 0, 16, dtype=torch.float64, device="cuda") * d2r
    with warnings.catch_warnings():
        warnings.simplefilter("error")       # nothing may warn (or synchronise) here
        ctx, code = corr._prepare(defer_status=True)
        w = ctx.wtheta(code, 0, corr._k_lim[0], corr._k_lim[1], corr.D_z, theta)
    assert h._status_pending
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        word = h.status                      # (whether pp_gg at z_bar exhausts divmax is not the point)
    assert not h._status_pending and word == int(ctx.status(0, 1)[0])
    assert bool(torch.isfinite(w).all())
    # P_gm at z = 0 does exhaust divmax (test_status_word_...): built without a word to the
    # caller, reported when looked at
    h2 = halo.Halo(0.0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        h2._sync(_lib.FAM_GM, defer_status=True)
    assert h2._status_pending
    with pytest.warns(_lib.ChompAccuracyWarning, match="divmax"):
        word = h2.status
    assert word & _lib.ST_HALO_DIVMAX["pp_gm"] and not h2._status_pending
    # ... and a rebuild does not lose an unread word: it is delivered first
    h2._reset_flags(all_tables=True)
    h2._sync(_lib.FAM_GM, defer_status=True)
    assert h2._status_pending
    h2._reset_flags(all_tables=True)
    with pytest.warns(_lib.ChompAccuracyWarning, match="divmax"):
        h2._sync(_lib.FAM_GM, defer_status=True)


def test_abi_error_codes_and_call_order():
    """The C ABI reports misuse instead of computing garbage: stages called before their
    prerequisites (CHOMP_ERR_STATE), bad arguments (CHOMP_ERR_ARG -> ValueError), features
    outside the scope (CHOMP_ERR_SCOPE), each with a message from chomp_last_error."""
    from chomp_amd import _lib, cosmology, defaults
    ctx = cosmology._context()
    k = numpy.logspace(-2, 1, 8)
    with pytest.raises(_lib.ChompError, match="epochs_set"):
        ctx.power(_lib.P_MM, k, 0, 1)
    ctx.epochs_set(defaults.default_cosmo_dict, [0.2])
    with pytest.raises(_lib.ChompError, match="mass_setup"):
        ctx.halo_setup(defaults.default_halo_dict, __import__("chomp_amd").hod.HODZheng(), _lib.FAM_MM)
    ctx.mass_setup(defaults.default_halo_dict, _lib.MF_ST)
    with pytest.raises(_lib.ChompError, match="not built"):
        ctx.power(_lib.P_MM, k, 0, 1)                       # knot tables missing
    assert ctx.power(_lib.P_LIN, k, 0, 1).shape == (1, 8)   # needs no tables
    from chomp_amd import hod
    ctx.halo_setup(defaults.default_halo_dict, hod.HODZheng(), _lib.FAM_MM)
    assert numpy.all(ctx.power(_lib.P_MM, k, 0, 1) > 0)
    with pytest.raises(_lib.ChompError, match="not built"):
        ctx.power(_lib.P_GG, k, 0, 1)
    with pytest.raises(ValueError):
        ctx.power(7, k, 0, 1)                               # unknown spectrum
    with pytest.raises(ValueError):
        ctx.power(_lib.P_MM, k, 0, 2)                       # epoch range
    with pytest.raises(_lib.ChompError, match="halofit_setup"):
        ctx.power(_lib.P_MM | _lib.P_HALOFIT, k, 0, 1)
    with pytest.raises(_lib.ChompError, match="kernel_setup"):
        ctx.wtheta(_lib.P_MM, 0, 1e-3, 1e2, 1.0, numpy.array([1e-3]))
    with pytest.raises(ValueError):
        ctx.xi3d(_lib.P_MM, 0, 1.0, 0.5, numpy.array([1.0]))    # k_max < k_min
    with pytest.raises(_lib.ChompScopeError):
        ctx.epochs_set(dict(defaults.default_cosmo_dict, w0=-0.9), [0.2])
    with pytest.raises(_lib.ChompScopeError):
        ctx.halo_setup(dict(defaults.default_halo_dict, alpha=-1.5), hod.HODZheng(), _lib.FAM_MM)
    # the context stays usable after every one of them
    ctx.epochs_set(defaults.default_cosmo_dict, [0.2])
    ctx.mass_setup(defaults.default_halo_dict, _lib.MF_ST)
    ctx.halo_setup(defaults.default_halo_dict, hod.HODZheng(), _lib.FAM_MM)
    assert numpy.all(numpy.isfinite(ctx.power(_lib.P_MM, k, 0, 1)))


def test_range_branches_and_bad_inputs():
    """halo.py:314-320 semantics for every kind of k the reference accepts."""
    from chomp_amd import halo
    h = halo.Halo(0.3)
    k = numpy.array([0.0, -1.0, 1e-17, 1e-5, 9.99e-4, 1e-3, 1.0, 100.0, 100.00000001,
                     1e3, numpy.inf, numpy.nan])
    out = h.power_mm(k)
    lin = h.linear_power(k)
    assert out.shape == k.shape
    assert numpy.all(out[8:] == 0.0)                 # k > k_max, inf, nan -> 0
    assert numpy.all(out[:8] > 0)
    # below k_min: P_lin(k) * constant (the value at k_min over P_lin(k_min))
    ratio = out[3:5] / lin[3:5]
    assert abs(ratio[0] / ratio[1] - 1) < 1e-13
    assert abs(ratio[0] / (out[5] / lin[5]) - 1) < 1e-12
    assert lin[0] == 1e-16 and lin[1] == 1e-16 and lin[2] == 1e-16   # k <= 1e-16
    assert h.power_mm(numpy.array([])).shape == (0,)
    assert h.power_gm(numpy.empty((0, 3))).shape == (0, 3)


@pytest.mark.parametrize("over", [
    dict(halo_npoints=30, mass_npoints=40),
    dict(divmax=6),
    dict(halo_precision=1.48e-4, cosmo_precision=1.48e-6, mass_precision=1.48e-6),
])
def test_non_default_precision_vs_oracle(over):
    """Contexts snapshot defaults.default_precision; other point counts / Romberg
    depths / tolerances must follow the reference algorithm just the same."""
    from chomp_amd import defaults, halo
    from oracle import chomp_oracle as o
    saved = copy.deepcopy(defaults.default_precision)
    try:
        defaults.default_precision.update(over)
        prec = dict(o.default_precision, **over)
        k = numpy.logspace(-3, 2, 64)
        h = halo.Halo(0.4)
        got_mm, got_gm = h.power_mm(k), h.power_gm(k)
        e = o.epoch(None, 0.4, prec=prec)
        t = o.halo_table(e, o.mass_table(e), o.zheng(prec=prec), families=("mm", "gm"))
        assert rel_err(got_mm, o.halo_power(t, "mm", k)) < RTOL
        assert rel_err(got_gm, o.halo_power(t, "gm", k)) < RTOL
    finally:
        defaults.default_precision.clear()
        defaults.default_precision.update(saved)


def test_tight_halo_precision_lists_smooth_knots_too():
    """With halo_precision tightened to 1.48e-9 the smooth h_m / pp_mm integrals do not converge
    within the node tables either.  A set-up without HOD integrands (P_mm) then evaluates its
    listed knots literally inside k_halo_knots_fast<.., true> (no hand-over launch on that
    chain); in a P_gm set-up the same knots take the fast level sums.  Values and Romberg
    levels against the oracle."""
    from chomp_amd import defaults, halo
    from oracle import chomp_oracle as o
    saved = copy.deepcopy(defaults.default_precision)
    try:
        over = dict(halo_precision=1.48e-9)
        defaults.default_precision.update(over)
        prec = dict(o.default_precision, **over)
        k = numpy.logspace(-3, 2, 48)
        e = o.epoch(None, 0.3, prec=prec)
        t = o.halo_table(e, o.mass_table(e), o.zheng(prec=prec), families=("mm", "gm"))
        h = halo.Halo(0.3)
        got_mm = h.power_mm(k)
        ctx = h._sync(0)
        f, l = ctx.deep_stats()
        assert f == 0 and l >= 1, (f, l)               # listed, and done in the same launch
        lev = ctx.table("levels", 0).reshape(5, -1)
        assert lev[:2].max() > 10
        assert numpy.array_equal(lev[0], t.levels["_h_m_integrand"])
        assert numpy.array_equal(lev[1], t.levels["_pp_mm_integrand"])
        assert rel_err(ctx.table("h_m", 0), t.h_m) < 1e-9 and rel_err(ctx.table("pp_mm", 0), t.pp_mm) < 1e-9
        assert rel_err(got_mm, o.halo_power(t, "mm", k)) < 1e-8
        h2 = halo.Halo(0.3)
        got_gm = h2.power_gm(k)
        ctx2 = h2._sync(0)
        f2, l2 = ctx2.deep_stats()
        assert f2 > 10 and l2 == 0, (f2, l2)
        lev2 = ctx2.table("levels", 0).reshape(5, -1)
        assert numpy.array_equal(lev2[0], t.levels["_h_m_integrand"])
        assert rel_err(ctx2.table("h_m", 0), t.h_m) < 1e-9
        assert rel_err(got_gm, o.halo_power(t, "gm", k)) < RTOL
    finally:
        defaults.default_precision.clear()
        defaults.default_precision.update(saved)


def test_fixed_mass_limits_vs_oracle():
    """defaults.default_limits mass_min/mass_max > 0 skip the search
    (mass_function.py:163-170)."""
    from chomp_amd import defaults, halo
    from oracle import chomp_oracle as o
    saved = dict(defaults.default_limits)
    try:
        defaults.default_limits.update(mass_min=1e10, mass_max=3e15)
        lim = dict(o.default_limits, mass_min=1e10, mass_max=3e15)
        k = numpy.logspace(-3, 2, 48)
        h = halo.Halo(0.2)
        got = h.power_mm(k)
        e = o.epoch(None, 0.2, limits=lim)
        t = o.halo_table(e, o.mass_table(e), families=("mm",))
        assert rel_err(got, o.halo_power(t, "mm", k)) < RTOL
        assert h.mass.ln_mass_min == numpy.log(1e10)
    finally:
        defaults.default_limits.clear()
        defaults.default_limits.update(saved)


def test_batch_of_cosmologies_vs_oracle():
    """The epoch axis also carries different cosmologies / HODs (the SimulationDesign
    axis, simulation_design.py:116-155): each row must equal its own oracle run."""
    from chomp_amd import grid
    from oracle import chomp_oracle as o
    base = o.default_cosmo_dict
    cds = [dict(base), dict(base, sigma_8=0.75, n_scalar=0.98),
           dict(base, omega_m0=0.31 - base["omega_r0"], omega_l0=0.69, h=0.68)]
    hods = [dict(o.default_hod_dict), dict(o.default_hod_dict, log_M_min=12.5, log_M_0=12.5),
            dict(o.default_hod_dict, sigma=0.3)]
    z = numpy.array([0.1, 0.6, 1.1])
    k = numpy.logspace(-3, 2, 40)
    hg = grid.HaloGrid(z, cosmo_dict=cds, hod_dict=hods, mass_function="tinker")
    got = hg.power("power_gm", k)
    for i in range(3):
        e = o.epoch(cds[i], float(z[i]))
        t = o.halo_table(e, o.mass_table(e, kind="tinker"), o.zheng(hods[i]), families=("gm",))
        assert rel_err(got[i], o.halo_power(t, "gm", k)) < RTOL, i


def _full_size_projection(ggl):
    from chomp_amd import cosmology, kernel
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    if ggl:
        wb = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2), cm)
        return kernel.GalaxyGalaxyLensingKernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
    wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    return kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)


def test_c4_full_size_vs_reference():
    """G6b / configs[3] at full size: all 1024 theta and 2048 l, power_gg and power_mm, against
    the reference's own vectors (1e-9: the same Romberg rows on the same integrand values).
    Evaluating in chunks must give identical values (every theta / l is an independent
    integral: the sharding unit)."""
    from chomp_amd import correlation, halo
    g = load_golden("g6b_limber_galgal_full")
    kern = _full_size_projection(False)
    for ps in ("power_gg", "power_mm"):
        h = halo.Halo(0.0)
        corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec=ps)
        w = corr.correlation(g["theta"])
        assert w.shape == (1024,)
        assert rel_err(w, g["w_" + ps]) < PROJ_RTOL, ps
        cf = correlation.CorrelationFourier(10, 1e4, kern, input_halo=h, powSpec=ps)
        cl = cf.correlation(g["ell"])
        assert cl.shape == (2048,) and numpy.all(cl > 0)
        assert rel_err(cl, g["cl_" + ps]) < PROJ_RTOL, ps
        if ps == "power_gg":
            halves = numpy.concatenate([corr.correlation(g["theta"][:500]),
                                        corr.correlation(g["theta"][500:])])
            assert numpy.array_equal(halves, w)
            pick = numpy.arange(3, 2048, 97)
            assert numpy.array_equal(cf.correlation(g["ell"][pick]), cl[pick])


def test_c5_full_size_vs_reference():
    """G7b / configs[4] at full size: w_GGL at 1024 theta and C_l at 2048 l, J2 kernel +
    HaloFit power_gm, in the fixture's call order (HaloFit built and first evaluated at z = 0)."""
    from chomp_amd import correlation, halo
    g = load_golden("g7b_ggl_halofit_full")
    kern = _full_size_projection(True)
    hf = halo.HaloFit(0.0)
    hf.power_mm(numpy.logspace(-3, 2, 8))
    corr = correlation.Correlation(0.001, 1.0, kern, input_halo=hf, power_spec="power_gm")
    assert abs(corr.D_z / float(g["D_z"]) - 1) < 1e-9
    assert rel_err(corr.correlation(g["theta"]), g["w_ggl"]) < PROJ_RTOL
    cf = correlation.CorrelationFourier(10, 1e4, kern, input_halo=hf, powSpec="power_gm")
    assert rel_err(cf.correlation(g["ell"]), g["cl_ggl"]) < PROJ_RTOL


def test_stage_e_timing_facility():
    """chomp_set_timing / chomp_get_timing: per-kernel HIP-event durations of a streaming
    Stage E call; a state error when the last call took another launch shape."""
    import torch
    from chomp_amd import grid, _lib
    hg = grid.HaloGrid(numpy.linspace(0.0, 1.0, 8))
    k = torch.logspace(-3, 2, 1 << 20, dtype=torch.float64, device="cuda")
    hg.ctx.set_timing(True)
    ref = hg.power("power_mm", k)
    us = hg.ctx.get_timing()
    assert us.shape == (3,) and numpy.all(us > 0) and us[1] > us[2]
    hg.ctx.set_timing(False)
    assert torch.equal(ref, hg.power("power_mm", k))          # timing does not change results
    with pytest.raises(_lib.ChompError):
        hg.ctx.get_timing()
    hg.ctx.set_timing(True)
    hg.power("power_mm", k[:4096].contiguous())               # small grid: row-walking shape
    with pytest.raises(_lib.ChompError):
        hg.ctx.get_timing()


def test_deep_knot_paths_agree():
    """The HOD knots that run beyond the node tables (Romberg levels 11..20, up to 2^20 nodes
    in the reference) are done by k_halo_knots_fast: level sums from 2049 coarse samples
    (degree-7 Lagrange weights) plus node-by-node evaluation of the few coarse intervals that
    hold a break point of the integrand.  Its checker evaluates every node literally
    (CHOMP_TUNE_DEEP_LITERAL): the knot values must agree to 1e-9 -- four orders inside the
    Romberg tolerance that decides the rows -- the stopping levels at every knot, and the
    spectra to 1e-9; Sheth-Tormen and Tinker, three Zheng HODs, redshifts of configs[2]."""
    from chomp_amd import grid, _lib
    z = numpy.array([0.0, 0.1, 0.55, 1.2, 1.5])
    k = numpy.logspace(-3, 2, 200)
    hods = [dict(log_M_min=12.14, sigma=0.15, log_M_0=12.14, log_M_1p=13.43, alpha=1.0),
            dict(log_M_min=11.9, sigma=0.35, log_M_0=12.3, log_M_1p=13.1, alpha=0.8),
            dict(log_M_min=12.5, sigma=0.0, log_M_0=12.0, log_M_1p=13.6, alpha=1.2)]
    n_fast = 0
    for mf in ("st", "tinker"):
        for hd in hods:
            fast = grid.HaloGrid(z, mass_function=mf, hod_dict=hd)
            lit = grid.HaloGrid(z, mass_function=mf, hod_dict=hd)
            lit.ctx = fast.ctx.__class__(fast.ctx.config, device=fast.ctx.device)
            lit.ctx.set_tuning(_lib.TUNE_DEEP_LITERAL, 1)
            for which in ("power_gm", "power_gg"):
                pf, pl = fast.power(which, k), lit.power(which, k)
                assert numpy.max(numpy.abs(pf / pl - 1)) < 1e-9, (mf, hd, which)
                for i in range(z.size):
                    for name in (("h_m", "h_g", "pp_gm") if which == "power_gm" else ("h_g", "pp_gg")):
                        a, b = fast.ctx.table(name, i), lit.ctx.table(name, i)
                        assert numpy.max(numpy.abs(a / b - 1)) < 1e-9, (mf, hd, which, i, name)
                    la = fast.ctx.table("levels", i).reshape(5, -1)
                    lb = lit.ctx.table("levels", i).reshape(5, -1)
                    rows = (0, 2, 3) if which == "power_gm" else (2, 4)
                    assert numpy.array_equal(la[list(rows)], lb[list(rows)]), (mf, hd, which, i)
            f, l = fast.ctx.deep_stats()
            assert f > 0 and l == 0, (f, l)       # no knot fell back to the literal route
            n_fast += f
            f, l = lit.ctx.deep_stats()
            assert f == 0 and l > 0
    assert n_fast > 500


def test_deep_knots_leave_the_fast_route_by_themselves():
    """The three exits of k_halo_knots_fast -- a self-check estimate above the threshold, more
    break points than allowed, more node-by-node intervals than allowed -- hand a knot to
    k_halo_knots_literal behind the launch.  With the default thresholds no knot of any tested
    model takes them, so the thresholds are moved (CHOMP_TUNE_DEEP_TOL / _MAX_BREAKS /
    _MAX_FINE) until SOME BUT NOT ALL knots of a 5-epoch Tinker batch leave: knots of one
    launch then finish on both routes, in both kernels, epochs are finalised by either -- and
    tables, stopping levels and spectra must equal the all-literal run."""
    from chomp_amd import grid, _lib
    z = numpy.array([0.0, 0.3, 0.7, 1.1, 1.5])
    k = numpy.logspace(-3, 2, 120)
    hd = dict(log_M_min=12.0, sigma=0.25, log_M_0=12.2, log_M_1p=13.2, alpha=0.9)

    def run(tune):
        hg = grid.HaloGrid(z, mass_function="tinker", hod_dict=hd)
        hg.ctx = hg.ctx.__class__(hg.ctx.config, device=hg.ctx.device)      # (a context of its own)
        for what, v in tune:
            hg.ctx.set_tuning(what, v)
        out, tabs, status = {}, {}, []
        for w, names in (("power_gm", ("h_m", "h_g", "pp_gm")), ("power_gg", ("h_g", "pp_gg"))):
            out[w] = hg.power(w, k)                    # (a set-up builds the requested families only)
            for i in range(z.size):
                for n in names + ("levels",):
                    tabs[(w, n, i)] = hg.ctx.table(n, i)
            status.append(hg.status())
        f, l = hg.ctx.deep_stats()
        return out, tabs, f, l, dict(hg.ctx.deep_detail), numpy.concatenate(status)

    ref = run([(_lib.TUNE_DEEP_LITERAL, 1)])
    assert ref[2] == 0 and ref[3] > 50
    base = run([])
    assert base[3] == 0 and base[2] == ref[3]
    worst = base[4]["worst_estimate"]
    assert 0.0 < worst < 1e-9

    def same_as_literal(r, why):
        for w in ref[0]:
            assert numpy.max(numpy.abs(r[0][w] / ref[0][w] - 1)) < 1e-9, (why, w)
        for key, t in ref[1].items():
            if key[1] == "levels":             # (rows of the families this set-up built)
                rows = [0, 2, 3] if key[0] == "power_gm" else [2, 4]
                assert numpy.array_equal(r[1][key].reshape(5, -1)[rows], t.reshape(5, -1)[rows]), (why, key)
            else:
                assert numpy.max(numpy.abs(r[1][key] / t - 1)) < 1e-9, (why, key)
        assert numpy.array_equal(r[5], ref[5]), why                  # status words
        assert r[2] + r[3] == ref[3], why                            # every listed knot done once

    # (1) the self-check: a threshold inside the range of the knots' estimates
    mixed = None
    for div in (3.0, 10.0, 30.0, 100.0, 1000.0, 1e4):
        r = run([(_lib.TUNE_DEEP_TOL, max(1, int(worst / div / 1e-15)))])
        if r[2] > 0 and r[3] > 0:
            mixed = r
            break
    assert mixed is not None, "no threshold splits the knots"
    assert mixed[4]["self_check"] == mixed[3] and mixed[4]["too_many_breaks"] == 0
    same_as_literal(mixed, "self-check")
    # ... and at 0 every knot that reaches the check leaves
    r = run([(_lib.TUNE_DEEP_TOL, 0)])
    assert r[3] > 0 and r[4]["self_check"] == r[3]
    same_as_literal(r, "tol 0")
    # (2) break points: pp_gm has two (<N> = 1, the satellite onset at M_0), pp_gg one
    r = run([(_lib.TUNE_DEEP_MAX_BREAKS, 1)])
    assert r[2] > 0 and r[3] > 0 and r[4]["too_many_breaks"] == r[3], (r[2], r[3], r[4])
    same_as_literal(r, "max breaks 1")
    r = run([(_lib.TUNE_DEEP_MAX_BREAKS, 0)])
    assert r[3] > 0 and r[4]["too_many_breaks"] == r[3]
    same_as_literal(r, "max breaks 0")
    # (3) node-by-node intervals: alpha != 1 makes the satellite onset singular, 17 intervals
    # above it are evaluated node by node
    r = run([(_lib.TUNE_DEEP_MAX_FINE, 4)])
    assert r[3] > 0 and r[4]["too_many_fine"] == r[3], (r[2], r[3], r[4])
    same_as_literal(r, "max fine 4")


def test_step_replays_from_a_hip_graph():
    """A whole step (Stage K + Stage E) captured into a HIP graph and replayed: same bits as
    the eager calls, also when replays and eager calls alternate (the Stage E work-list
    counters must not depend on host-side state that a replay does not see)."""
    import torch
    from chomp_amd import grid
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        z = numpy.linspace(0.0, 1.2, 16)
        k = torch.logspace(-3.2, 2.2, 4096, dtype=torch.float64, device="cuda")   # some k out of range
        out = torch.zeros((16, 4096), dtype=torch.float64, device="cuda")
        hg = grid.HaloGrid(z, stream=s.cuda_stream)

        def step():
            hg.setup("power_mm")
            hg.power("power_mm", k, out=out)
        step()
        step()
        torch.cuda.synchronize()
        ref = out.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
        torch.cuda.synchronize()
        for i in range(7):
            out.zero_()
            g.replay()
            if i % 3 == 2:
                step()                     # an eager call in between
            torch.cuda.synchronize()
            assert torch.equal(out, ref), i


def test_graph_capture_refuses_what_a_replay_cannot_honour():
    """Under stream capture the library must not synchronise or reallocate: a parameter block
    that differs from what the device holds, or a work buffer that would have to grow, is an
    error (CHOMP_ERR_STATE) -- not a broken capture -- and buffers a captured graph references
    stay alive when a later eager call outgrows them (include/chomp_mi355x.h, "HIP graphs")."""
    import torch
    from chomp_amd import grid, _lib
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        z = numpy.linspace(0.0, 1.0, 8)
        k = torch.logspace(-3, 2, 2048, dtype=torch.float64, device="cuda")
        out = torch.zeros((8, 2048), dtype=torch.float64, device="cuda")
        hg = grid.HaloGrid(z, stream=s.cuda_stream)
        hg.setup("power_mm")
        hg.power("power_mm", k, out=out)
        torch.cuda.synchronize()
        ref = out.clone()
        # (a) changed parameters while capturing
        hg2 = grid.HaloGrid(z + 0.1, stream=s.cuda_stream)
        hg2.ctx = hg.ctx                      # same context, other redshifts
        g = torch.cuda.CUDAGraph()
        with pytest.raises(_lib.ChompError, match="captur"):
            with torch.cuda.graph(g, stream=s):
                hg2.setup("power_mm")
        torch.cuda.synchronize()
        # (b) the same parameters capture fine ...
        hg.setup("power_mm")
        hg.power("power_mm", k, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            hg.setup("power_mm")
            hg.power("power_mm", k, out=out)
        torch.cuda.synchronize()
        # ... (c) and a later, larger eager call must not free what the graph references
        big = grid.HaloGrid(numpy.linspace(0.0, 1.0, 24), stream=s.cuda_stream)
        big.ctx = hg.ctx
        big.setup("power_mm")
        kb = torch.logspace(-3, 2, 1 << 15, dtype=torch.float64, device="cuda")
        big.power("power_mm", kb)
        torch.cuda.synchronize()
        out.zero_()
        g.replay()                            # stale buffers, but alive: no fault
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        # eager again with the original batch: the original numbers
        hg.setup("power_mm")
        hg.power("power_mm", k, out=out)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def test_power_plan_reuses_the_k_table(torch_mod):
    """chomp_power_plan: a registered k grid skips the k-only step (and, when it holds no k
    group for the per-lane pass, that launch too).  Same bits as unregistered calls, across
    spectra and set-ups of the same cosmology; another cosmology or grid ends it silently."""
    torch = torch_mod
    from chomp_amd import grid, _lib
    hg = grid.HaloGrid(numpy.linspace(0.0, 1.5, 8), mass_function="tinker")
    hg.setup("power_gg")
    hg.ctx.set_tuning(_lib.TUNE_E_STREAM_MIN, 0)              # streaming shape on a small grid
    clean = torch.logspace(-3, 2, 1 << 14, dtype=torch.float64, device="cuda")
    ragged = torch.logspace(-4, 3, 1 << 14, dtype=torch.float64, device="cuda")   # out-of-range ends
    for k in (clean, ragged):
        ref = {w: hg.power(w, k).clone() for w in ("power_mm", "power_gm", "power_gg")}
        hg.ctx.power_plan(k)
        for _ in range(2):
            for w, r in ref.items():
                assert torch.equal(torch.nan_to_num(hg.power(w, k), nan=-7.0),
                                   torch.nan_to_num(r, nan=-7.0)), w
        hg.setup("power_gg")                                  # same cosmology: still registered
        assert torch.equal(hg.power("power_gm", k), ref["power_gm"])
    # another grid in between, then the registered one again (now unregistered: recomputed)
    other = hg.power("power_mm", clean)
    assert torch.equal(hg.power("power_mm", ragged), ref["power_mm"])
    assert torch.isfinite(other).all()
    # another cosmology under the same pointer: the registration must not be used
    hg.ctx.power_plan(clean)
    cd = dict(__import__("chomp_amd").defaults.default_cosmo_dict, sigma_8=0.75)
    hg2 = grid.HaloGrid(numpy.linspace(0.0, 1.5, 8), cosmo_dict=cd, mass_function="tinker")
    hg2.ctx = hg.ctx
    hg2.setup("power_mm")
    got = hg2.power("power_mm", clean)
    fresh = grid.HaloGrid(numpy.linspace(0.0, 1.5, 8), cosmo_dict=cd, mass_function="tinker")
    assert torch.equal(got, fresh.power("power_mm", clean))


def test_roctx_ranges_switch_on_and_off():
    """CHOMP_TUNE_ROCTX: the stage ranges (roctxRangePush / Pop through a dlopen'ed marker
    library) can be switched on and off around a step without changing a result."""
    from chomp_amd import grid, _lib
    hg = grid.HaloGrid(numpy.array([0.0, 0.7]))
    k = numpy.logspace(-2, 1, 16)
    ref = hg.power("power_mm", k)
    hg.ctx.set_tuning(_lib.TUNE_ROCTX, 1)
    hg.setup("power_mm")
    assert numpy.array_equal(hg.power("power_mm", k), ref)
    hg.ctx.set_tuning(_lib.TUNE_ROCTX, 0)
    hg.setup("power_mm")
    assert numpy.array_equal(hg.power("power_mm", k), ref)


def test_launch_shapes_by_batch_size_agree():
    """The set-up kernels pick their block shapes by the size of the batch -- four wavefronts per
    sigma(R) integral and a block per knot pair for one epoch, one wavefront each for a batch,
    single-wavefront knot blocks once the knots outnumber the chip's slots (48 epochs x 50 knots
    x 2 groups) -- and the tables of an epoch do not depend on which batch it came in (to the
    order of the additions)."""
    from chomp_amd import grid
    k = numpy.logspace(-2.5, 1.5, 37)
    z = numpy.linspace(0.05, 1.2, 48)
    big = grid.HaloGrid(z)
    p_big = big.power("power_gm", k)
    for i in (0, 17, 47):
        one = grid.HaloGrid(z[i:i + 1])
        p_one = one.power("power_gm", k)[0]
        assert numpy.max(numpy.abs(p_one / p_big[i] - 1)) < 1e-12, i
        for name in ("nu", "pp_gm", "h_g"):
            a, b = one.ctx.table(name, 0), big.ctx.table(name, i)
            assert numpy.max(numpy.abs(a - b)) <= 1e-12 * numpy.max(numpy.abs(b)), (i, name)
        la = one.ctx.table("levels", 0).reshape(5, -1)[[0, 2, 3]]      # h_m, h_g, pp_gm: the built ones
        lb = big.ctx.table("levels", i).reshape(5, -1)[[0, 2, 3]]
        assert numpy.array_equal(la, lb), i
    mid = grid.HaloGrid(z[:8])                   # 8 x 50 x 2 = 800 knots: four-knot blocks
    p_mid = mid.power("power_gm", k)
    assert numpy.max(numpy.abs(p_mid / p_big[:8] - 1)) < 1e-12


def test_batch_shapes_with_other_point_counts():
    """The four-knots-to-a-block kernel with its cooperative tail and the permuted dispatch order
    of the nu table, at point counts other than 50 (30 knots: the last block of an epoch holds
    fewer than four; 40 masses): a 16-epoch batch against its epochs one by one, which take the
    block-per-knot / four-wavefronts-per-integral shapes."""
    from chomp_amd import defaults, grid
    saved = copy.deepcopy(defaults.default_precision)
    try:
        defaults.default_precision.update(dict(halo_npoints=30, mass_npoints=40))
        k = numpy.logspace(-2.5, 1.5, 29)
        z = numpy.linspace(0.0, 1.4, 16)
        big = grid.HaloGrid(z)
        p_big = big.power("power_mm", k)
        assert not big.status().any()
        for i in (0, 7, 15):
            one = grid.HaloGrid(z[i:i + 1])
            p_one = one.power("power_mm", k)[0]
            assert numpy.array_equal(one.ctx.table("ln_mass", 0), big.ctx.table("ln_mass", i))
            for name in ("nu", "h_m", "pp_mm"):
                a, b = one.ctx.table(name, 0), big.ctx.table(name, i)
                assert a.shape == b.shape and numpy.max(numpy.abs(a - b)) <= 1e-12 * numpy.max(numpy.abs(b)), (i, name)
            la = one.ctx.table("levels", 0).reshape(5, -1)[:2]
            lb = big.ctx.table("levels", i).reshape(5, -1)[:2]
            assert numpy.array_equal(la, lb), i
            assert numpy.max(numpy.abs(p_one / p_big[i] - 1)) < 1e-12, i
    finally:
        defaults.default_precision.clear()
        defaults.default_precision.update(saved)


def test_many_cosmologies_in_one_batch_equal_single_epochs():
    """A design or MCMC batch is throughput-, not latency-bound, and takes other launch shapes:
    from 16 distinct cosmologies on the cosmology-only tables are built four nodes per thread
    without arrival counts, sigma_8 and the aiming table by k_sigma_lns; from 128 epochs on the
    probes of the mass-limit search and their certification are two launches.  Every epoch of
    such a batch must come out as it does on its own -- mass limits bit for bit, tables to the
    order of the additions."""
    from chomp_amd import grid
    rng = numpy.random.default_rng(11)
    base = dict(grid.defaults.default_cosmo_dict)
    n = 140
    cds, zs = [], []
    for i in range(n):
        om = rng.uniform(0.24, 0.34)
        cds.append(dict(base, omega_m0=om - base["omega_r0"], omega_l0=1.0 - om,
                        sigma_8=rng.uniform(0.75, 0.88), h=rng.uniform(0.65, 0.75),
                        n_scalar=rng.uniform(0.94, 1.0)))
        zs.append(float(rng.uniform(0.0, 1.2)))
    k = numpy.logspace(-3, 2, 50)
    big = grid.HaloGrid(numpy.array(zs), cosmo_dict=cds)
    p_big = big.power("power_mm", k)
    assert not big.status().any()
    for i in (0, 7, 13, 19, 77, 139):
        one = grid.HaloGrid(numpy.array([zs[i]]), cosmo_dict=cds[i])
        p_one = one.power("power_mm", k)[0]
        a, b = one.ctx.scalars(0), big.ctx.scalars(i)
        assert a["ln_mass_min"] == b["ln_mass_min"] and a["ln_mass_max"] == b["ln_mass_max"], i
        assert abs(a["sigma_norm"] / b["sigma_norm"] - 1) < 1e-13, i
        assert numpy.array_equal(one.ctx.table("ln_mass", 0), big.ctx.table("ln_mass", i))
        assert numpy.max(numpy.abs(one.ctx.table("nu", 0) / big.ctx.table("nu", i) - 1)) < 1e-12, i
        assert numpy.max(numpy.abs(p_one / p_big[i] - 1)) < 1e-12, i


def test_large_batch_shapes_vs_oracle():
    """The launch shapes only a large batch takes -- from 16 distinct cosmologies on
    k_sigma_nodes<., 4> + k_sigma_lns (cosmology-only tables without arrival counts, sigma_8
    and the aiming table behind the kernel boundary; k_nu_table's XCD-aware block order, whose
    last 147 mod 8 epochs take the plain one), from 128 epochs on k_epoch_probe<., 1, 1>
    + <., 2> (single-wavefront probes, then their certification as a launch of its own),
    single-wavefront knot blocks -- pinned DIRECTLY to the oracle, not to the device's own
    single-epoch answer: nine epochs of a 147-epoch design (the consumer: simulation_design.py:116-155) against
    oracle.chomp_oracle -- the ln M limits of the mass function bit for bit wherever the
    status word does not flag a saturated search, sigma_norm, the nu table and P_mm(k) to 1e-7."""
    from chomp_amd import grid, _lib
    from oracle import chomp_oracle as o
    rng = numpy.random.default_rng(404)
    base = dict(o.default_cosmo_dict)
    n, n_cosmo = 147, 24
    pool = []
    for _ in range(n_cosmo):
        om = rng.uniform(0.25, 0.33)
        pool.append(dict(base, omega_m0=om - base["omega_r0"], omega_l0=1.0 - om,
                         omega_b0=rng.uniform(0.04, 0.05), h=rng.uniform(0.66, 0.74),
                         sigma_8=rng.uniform(0.77, 0.86), n_scalar=rng.uniform(0.94, 0.99)))
    cds = [pool[i % n_cosmo] for i in range(n)]          # every cosmology at six redshifts
    zs = [float(z) for z in rng.uniform(0.0, 1.3, n)]
    k = numpy.logspace(-3, 2, 48)
    big = grid.HaloGrid(numpy.array(zs), cosmo_dict=cds)
    p_big = big.power("power_mm", k)
    status = big.status()
    n_exact = 0
    for i in (0, 5, 23, 24, 71, 100, 127, 143, 146):
        e = o.epoch(cds[i], zs[i])
        lo, hi, _ = o.mass_limits(e)
        sc = big.ctx.scalars(i)
        flagged = bool(status[i] & (_lib.ST_SATURATED | _lib.ST_MASS_SEARCH_EXHAUSTED))
        if not flagged:
            assert sc["ln_mass_min"] == lo and sc["ln_mass_max"] == hi, (i, zs[i])
            n_exact += 1
        m = o.mass_table(e)
        assert abs(sc["sigma_norm"] / e.sigma_norm - 1) < 1e-9, i
        if not flagged:
            assert numpy.max(numpy.abs(big.ctx.table("nu", i) / m.nu_arr - 1)) < 1e-7, i
            t = o.halo_table(e, m, families=("mm",))
            ref = o.halo_power(t, "mm", k)
            ok = ref > 0
            assert numpy.max(numpy.abs(p_big[i][ok] / ref[ok] - 1)) < 1e-7, (i, zs[i])
            assert numpy.array_equal(p_big[i][~ok], ref[~ok])
    assert n_exact >= 6


def test_probe_shortcut_does_not_decide_on_coarse_rows():
    """A case the randomised soak found (tools/soak.py 12 160, case 96): sigma^2(R) of the
    candidate one 5 % step above the reference's mass_min has Romberg rows 3 and 4 -- 9 and 17
    nodes -- that agree to 3.9e-7 and are both 0.9 % off, on the passing side of the band edge.
    The probes' loose stopping rule (a device-only shortcut: 1e-6, clear of the edges) took
    that for a converged integral and the search stopped a step early, unflagged, P_mm 6.6e-4
    off.  The shortcut now decides nothing below Romberg level 8: the limits are the
    reference's walk's, bit for bit, alone and as epoch 96 of 160 (single-wavefront probes)."""
    from chomp_amd import grid
    from oracle import chomp_oracle as o
    cd = dict(o.default_cosmo_dict, omega_m0=0.282801919784524, omega_b0=0.04606049457071416,
              omega_l0=0.717113386337925, h=0.615423691856908, sigma_8=0.7320155736315984,
              n_scalar=0.9301320614204301)
    z = 0.9637352201784997
    e = o.epoch(cd, z)
    lo, hi, _ = o.mass_limits(e)
    ref = o.halo_power(o.halo_table(e, o.mass_table(e), families=("mm",)), "mm", numpy.logspace(-3, 2, 24))
    for n, idx in ((1, 0), (160, 96)):
        zs = numpy.linspace(0.0, 1.2, n)
        zs[idx] = z
        hg = grid.HaloGrid(zs, cosmo_dict=[cd] * n)
        p = hg.power("power_mm", numpy.logspace(-3, 2, 24))
        sc = hg.ctx.scalars(idx)
        assert hg.status()[idx] == 0
        assert sc["ln_mass_min"] == lo and sc["ln_mass_max"] == hi, (n, sc["ln_mass_min"] - lo)
        assert numpy.max(numpy.abs(p[idx] / ref - 1)) < 1e-7


def test_soak_seed_flag_iff_mismatch():
    """One seed of tools/soak.py as a test: random cosmologies, HODs and redshifts, device
    P_mm (and P_gm for the first few) against the oracle.  Every epoch the status word does
    not flag agrees to 1e-7 -- and a saturated mass-limit search, the one documented limit of
    parity (the reference's own answer is decided by rounding noise there), is FLAGGED: flag <=>
    the oracle's walk ends in that regime too."""
    from chomp_amd import grid, _lib
    from oracle import chomp_oracle as o
    rng = numpy.random.default_rng(7)
    n, n_gm = 16, 2
    k = numpy.logspace(-3, 2, 40)
    cos, zs, hods = [], [], []
    for i in range(n):
        c = dict(o.default_cosmo_dict)
        c["omega_m0"] = rng.uniform(0.2, 0.4) - c["omega_r0"]
        c["omega_l0"] = 1.0 - c["omega_m0"] - c["omega_r0"]
        c["omega_b0"] = rng.uniform(0.035, 0.055)
        c["h"] = rng.uniform(0.6, 0.8)
        c["sigma_8"] = rng.uniform(0.7, 0.9)
        c["n_scalar"] = rng.uniform(0.92, 1.0)
        cos.append(c)
        zs.append(float(rng.uniform(0.0, 1.5)))
        h = dict(o.default_hod_dict)
        h["log_M_min"] = rng.uniform(11.8, 12.6)
        h["log_M_0"] = h["log_M_min"]
        h["sigma"] = rng.uniform(0.1, 0.4)
        h["log_M_1p"] = h["log_M_min"] + rng.uniform(1.0, 1.5)
        hods.append(h)
    g = grid.HaloGrid(numpy.array(zs), cosmo_dict=cos, hod_dict=hods)
    pm = g.power("power_mm", k)
    status = g.status()
    pg = g.power("power_gm", k)
    n_flagged = 0
    for i in range(n):
        e = o.epoch(cos[i], zs[i])
        lo, _, _ = o.mass_limits(e)
        R = (3.0 * numpy.exp(lo) / (4.0 * numpy.pi * o.rho_bar(e))) ** (1.0 / 3.0)
        saturated = 100.0 * e.limits["k_max"] * R < 0.2
        flagged = bool(status[i] & (_lib.ST_SATURATED | _lib.ST_MASS_SEARCH_EXHAUSTED))
        assert flagged == bool(saturated), (i, zs[i], int(status[i]))
        if flagged:
            n_flagged += 1
            continue
        fam = ("mm", "gm") if i < n_gm else ("mm",)
        tb = o.halo_table(e, o.mass_table(e), o.zheng(hods[i]), families=fam)
        ref = o.halo_power(tb, "mm", k)
        ok = ref > 0
        assert numpy.max(numpy.abs(pm[i][ok] / ref[ok] - 1)) < 1e-7, (i, zs[i])
        if i < n_gm:
            refg = o.halo_power(tb, "gm", k)
            assert numpy.max(numpy.abs(pg[i][ok] / refg[ok] - 1)) < 1e-7, (i, zs[i])
    assert n_flagged <= 3


@pytest.mark.parametrize("alpha", [1.0, 0.9])
def test_deep_knots_in_rounds_of_the_sample_buffer(alpha):
    """The sample buffer of the listed knots holds a slot for every knot that can be listed only
    up to a budget; beyond it (hundreds of epochs) the sampling and the summing launches work
    the list off in rounds of `slots` knots -- per-round draw counters, the epochs' tokens in
    round 0, the hand-over launch behind every round, the literal list behind the last.
    CHOMP_TUNE_DEEP_SLOTS shrinks the buffer until a 6-epoch batch takes several rounds: every
    number must come out as in one round, bit for bit -- with the lean instance and its
    hand-over launch (alpha = 1), with the evaluating instance as the main pass (alpha != 1),
    and with knots leaving for the literal evaluation in the middle of it."""
    from chomp_amd import grid, _lib
    z = numpy.linspace(0.0, 1.4, 6)
    k = numpy.logspace(-3, 2, 90)
    hd = dict(log_M_min=12.1, sigma=0.2, log_M_0=12.2, log_M_1p=13.3, alpha=alpha)

    def run(tune):
        hg = grid.HaloGrid(z, mass_function="tinker", hod_dict=hd)
        hg.ctx = hg.ctx.__class__(hg.ctx.config, device=hg.ctx.device)      # (a context of its own)
        for what, v in tune:
            hg.ctx.set_tuning(what, v)
        out = {w: hg.power(w, k) for w in ("power_gm", "power_gg")}
        tabs = {(n, i): hg.ctx.table(n, i) for i in range(z.size) for n in ("h_g", "pp_gg", "levels")}
        f, l = hg.ctx.deep_stats()
        return out, tabs, f, l, hg.status()

    one = run([])
    assert one[2] > 60 and one[3] == 0
    for slots in (17, 64):
        many = run([(_lib.TUNE_DEEP_SLOTS, slots)])
        assert many[2] == one[2] and many[3] == 0, slots
        for w in one[0]:
            assert numpy.array_equal(many[0][w], one[0][w]), (slots, w)
        for key in one[1]:
            assert numpy.array_equal(many[1][key], one[1][key]), (slots, key)
        assert numpy.array_equal(many[4], one[4])
    # knots that leave for the literal evaluation, in rounds: same numbers as in one round
    lit1 = run([(_lib.TUNE_DEEP_MAX_BREAKS, 1)])
    litn = run([(_lib.TUNE_DEEP_MAX_BREAKS, 1), (_lib.TUNE_DEEP_SLOTS, 23)])
    assert lit1[3] > 0 and (litn[2], litn[3]) == (lit1[2], lit1[3])
    for w in one[0]:
        assert numpy.array_equal(litn[0][w], lit1[0][w]), w
        assert numpy.max(numpy.abs(lit1[0][w] / one[0][w] - 1)) < 1e-9, w


def test_deep_knots_with_other_point_counts():
    """The fast deep-level sums at table sizes other than 50 x 50 (an odd mass_npoints moves the
    sample arrays' place in a block's LDS, which are copied 16 bytes at a time; 30 knots): every
    listed knot against the literal evaluation of every Romberg node."""
    from chomp_amd import defaults, grid, _lib
    saved = copy.deepcopy(defaults.default_precision)
    try:
        defaults.default_precision.update(dict(halo_npoints=30, mass_npoints=41))
        z = numpy.array([0.0, 0.8])
        k = numpy.logspace(-3, 2, 60)

        def run(tune):
            hg = grid.HaloGrid(z, mass_function="tinker")
            for what, v in tune:
                hg.ctx.set_tuning(what, v)
            return hg.power("power_gm", k), [hg.ctx.table("levels", i) for i in range(2)], hg.ctx.deep_stats()

        fast, lit = run([]), run([(_lib.TUNE_DEEP_LITERAL, 1)])
        assert fast[2][0] > 10 and fast[2][1] == 0 and lit[2][0] == 0
        assert numpy.max(numpy.abs(fast[0] / lit[0] - 1)) < 1e-9
        for a, b in zip(fast[1], lit[1]):
            assert numpy.array_equal(a, b)
    finally:
        defaults.default_precision.clear()
        defaults.default_precision.update(saved)
