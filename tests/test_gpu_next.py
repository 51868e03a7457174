"""GPU parity of the SURVEY 8(f) "next" rows built so far: Halo(extrapolate=True) and
HaloExclusion (rank 2), Correlation3d and the ASCII writers (rank 3), against vectors
produced by running the reference (tests/golden/g8, g9)."""
import os

import numpy
import pytest
from scipy.interpolate import InterpolatedUnivariateSpline

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def test_extrapolation_above_k_max():
    """halo.py:300-312, 341-367, 405-431."""
    from chomp_amd import halo
    g = load_golden("g8_extrapolate_exclusion")
    for i, z in enumerate(g["z"]):
        h = halo.Halo(float(z), extrapolate=True)
        assert h.get_extrapolation() is True
        assert rel_err(h.power_mm(g["k"]), g["ext_mm_%d" % i]) < RTOL
        assert rel_err(h.power_gm(g["k"]), g["ext_gm_%d" % i]) < RTOL
        assert rel_err(h.power_gg(g["k"]), g["ext_gg_%d" % i]) < RTOL
    # switching it off restores the zero branch; on again (set_extrapolation, :113-121)
    h.set_extrapolation(False)
    assert h.power_gm(numpy.array([250.0]))[0] == 0.0
    h.set_extrapolation(True)
    assert abs(h.power_gm(numpy.array([250.0]))[0] / g["ext_gm_1"][-1] - 1) < RTOL
    # a large device-resident grid reaching beyond k_max (streaming + per-lane passes)
    import torch
    kd = torch.logspace(-4, 3, 1 << 16, dtype=torch.float64, device="cuda")
    ctx = h._sync(7)
    big = ctx.power(h._power_code(1), kd, 0, 1)[0].cpu().numpy()
    ref = numpy.exp(numpy.interp(numpy.log(g["k"][:36]), numpy.log(kd.cpu().numpy()), numpy.log(big)))
    assert rel_err(ref, g["ext_mm_1"][:36]) < 2e-3          # (interpolation of the dense grid)
    small = h.power_mm(kd.cpu().numpy()[::997])
    assert numpy.array_equal(small, big[::997])


def test_halo_exclusion():
    """halo.py:1201-1233."""
    from chomp_amd import halo
    g = load_golden("g8_extrapolate_exclusion")
    hx = halo.HaloExclusion(0.0)
    assert rel_err(hx.power_mm(g["kx"]), g["excl_mm"]) < RTOL
    assert rel_err(hx.power_gm(g["kx"]), g["excl_gm"]) < RTOL
    assert rel_err(hx.power_gg(g["kx"]), g["excl_gg"]) < RTOL
    k_knots = numpy.exp(g["excl_ln_k"])[:-1]
    assert rel_err(hx._h_m(k_knots), g["excl_h_m"][:-1]) < RTOL
    assert rel_err(hx._h_g(k_knots), g["excl_h_g"][:-1]) < RTOL
    # the plain Halo differs visibly (the window suppresses the 2-halo term at high k)
    h = halo.Halo(0.0)
    assert rel_err(h.power_mm(g["kx"]), g["excl_mm"]) > 1e-2


def test_correlation3d():
    """correlation.py:408-510, all 50 separations of each case."""
    from chomp_amd import correlation
    g = load_golden("g9_correlation3d")
    for tag, kw in (("mm", dict(powSpec="power_mm")), ("gg", dict(powSpec="power_gg")),
                    ("mm_wide", dict(powSpec="power_mm", k_min=1e-4, k_max=1e3))):
        c3 = correlation.Correlation3d(0.1, 50.0, redshift=0.0, **kw)
        assert c3.halo.get_extrapolation() == bool(g["extrap_" + tag])
        c3.compute_correlation()
        assert numpy.allclose(c3.r_array, g["r_array"], rtol=1e-14)
        scale = numpy.max(numpy.abs(g["xi_" + tag]))
        # xi changes sign: relative to the local magnitude where it is not crossing zero
        big = numpy.abs(g["xi_" + tag]) > 1e-3 * scale
        assert rel_err(c3.xi_array[big], g["xi_" + tag][big]) < RTOL, tag
        assert numpy.max(numpy.abs(c3.xi_array - g["xi_" + tag])) < RTOL * scale, tag
        got = c3.correlation(g["r_test"])
        assert numpy.max(numpy.abs(got - g["corr_" + tag])) < RTOL * scale, tag
        assert got[g["r_test"] <= 0.1].tolist() == [0.0] * int(numpy.sum(g["r_test"] <= 0.1))


def test_correlation3d_with_halofit():
    """Correlation3d over a HaloFit spectrum (correlation.py:408-510 with halo.py:1236-1412 as
    input_halo): raw_correlation / compute_correlation / correlation against the oracle."""
    from chomp_amd import correlation, halo
    from oracle import chomp_oracle as o
    e = o.epoch(None, 0.0)
    t = o.halo_table(e, o.mass_table(e), o.zheng(), families=("mm", "gm"))
    o.halofit_table(t)
    r = numpy.logspace(-1, numpy.log10(50.0), 7)
    for ps, fam in (("power_mm", "mm"), ("power_gm", "gm")):
        c3 = correlation.Correlation3d(0.1, 50.0, redshift=0.0, input_halo=halo.HaloFit(0.0),
                                       powSpec=ps)
        got = c3.raw_correlation(r)
        ref = o.xi3d_raw(lambda k: o.halofit_power(t, fam, k), r, t.k_min, t.k_max)
        scale = numpy.max(numpy.abs(ref))
        assert numpy.max(numpy.abs(got - ref)) < RTOL * scale, ps
        c3.compute_correlation()
        assert numpy.all(numpy.isfinite(c3.xi_array)) and c3.xi_array.size == 50
        assert numpy.shape(c3.correlation(numpy.array([1.0, 10.0]))) == (2,)


def test_spline_eval_matches_fitpack():
    from chomp_amd import cosmology
    ctx = cosmology._context()
    rng = numpy.random.default_rng(3)
    xk = numpy.sort(rng.uniform(0.0, 10.0, 50))
    yk = numpy.sin(xk) * numpy.exp(0.1 * xk)
    x = numpy.concatenate([rng.uniform(xk[0], xk[-1], 500), xk, [xk[0] - 0.2, xk[-1] + 0.3]])
    ref = InterpolatedUnivariateSpline(xk, yk)(x)
    assert numpy.max(numpy.abs(ctx.spline_eval(xk, yk, x) - ref)) < 1e-12
    sp = InterpolatedUnivariateSpline(xk, yk)
    inside = x[(x > xk[0]) & (x < xk[-1])]
    dref = numpy.array([sp.derivatives(v)[1] for v in inside])
    assert numpy.max(numpy.abs(ctx.spline_eval(xk, yk, inside, deriv=1) - dref)) < 1e-11
    with pytest.raises(ValueError):
        ctx.spline_eval(xk[::-1].copy(), yk, x)


def test_ascii_writers(tmp_path):
    """halo.py:587-647, kernel.py:342-355, 765-781, correlation.py:277-289: same columns
    and number formats as the reference's files."""
    from chomp_amd import cosmology, correlation, halo, kernel
    h = halo.Halo(0.0)
    fn = str(tmp_path / "halo.txt")
    h.write(fn)
    rows = numpy.loadtxt(fn)
    assert rows.shape == (50, 5)
    k = rows[:, 0]
    assert numpy.allclose(k, numpy.exp(h._ln_k_array), rtol=0, atol=1e-10)
    assert numpy.allclose(rows[:-1, 2], h.power_mm(numpy.exp(h._ln_k_array))[:-1], rtol=1e-6, atol=1e-10)
    assert rows[-1, 2] == 0.0            # exp(log(k_max)) > k_max: the reference's last row
    open(fn).readline().startswith("#ttype1 = k [Mpc/h]")
    h.write_power_components(str(tmp_path / "comp.txt"))
    comp = numpy.loadtxt(str(tmp_path / "comp.txt"))
    assert comp.shape == (50, 6) and numpy.all(comp[:-1, 1] > 0)
    h.write_halo(str(tmp_path / "prof.txt"))
    prof = numpy.loadtxt(str(tmp_path / "prof.txt"))
    assert prof.shape == (50, 5) and numpy.all(prof[:, 2] > 0)
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
    kern.write(str(tmp_path / "kern.txt"))
    kt = numpy.loadtxt(str(tmp_path / "kern.txt"))
    assert kt.shape == (50, 2) and numpy.allclose(kt[:, 1], kern._kernel_array, rtol=1e-9)
    wa.write(str(tmp_path / "win.txt"))
    wt = numpy.loadtxt(str(tmp_path / "win.txt"))
    assert wt.shape[1] == 2 and wt.shape[0] == wa._chi_array.size
    corr = correlation.Correlation(0.01, 1.0, kern, input_halo=h, power_spec="power_mm")
    corr.compute_correlation()
    corr.write(str(tmp_path / "w.txt"))
    wt = numpy.loadtxt(str(tmp_path / "w.txt"))
    assert numpy.allclose(wt[:, 0], corr.theta_array / d2r, rtol=1e-9)
    assert numpy.allclose(wt[:, 1], corr.wtheta_array, rtol=1e-9)


def test_simulation_design_batched_equals_loop():
    """simulation_design.py:116-155: the design points of a
    SimulationDesign are one batch of epochs; the batch must give what the reference-shaped
    point-by-point loop gives, and what the oracle gives for a point."""
    from chomp_amd import halo, simulation_design as sd
    from oracle import chomp_oracle as o
    numpy.random.seed(12)
    k = numpy.logspace(-3, 2, 24)
    params = {"omega_m0": [0.27, 0.22, 0.35], "sigma_8": [0.8, 0.7, 0.9],
              "log_M_min": [12.1, 11.8, 12.6]}
    des = sd.SimulationDesignFlatUniverse(halo.Halo(0.3), "power_gm", params, n_design=6,
                                          independent_var=k)
    batched = des.run_design()
    assert des._batched() and batched.shape == (24, 6)
    points = des.points.copy()
    loop = sd.SimulationDesignFlatUniverse(halo.Halo(0.3), "power_gm", params, n_design=6,
                                           independent_var=k)
    loop._init_design_points()
    loop.points = points
    looped = loop.run_design(batched=False)
    assert rel_err(batched.values, looped.values) < 1e-12
    # one point against the oracle
    i = 4
    pt = points.iloc[i]
    cd = dict(o.default_cosmo_dict, omega_m0=pt["omega_m0"], sigma_8=pt["sigma_8"])
    cd["omega_l0"] = 1.0 - cd["omega_m0"] - cd["omega_r0"]
    hd = dict(o.default_hod_dict, log_M_min=pt["log_M_min"])
    e = o.epoch(cd, 0.3)
    t = o.halo_table(e, o.mass_table(e), o.zheng(hd), families=("gm",))
    assert rel_err(batched.values[:, i], o.halo_power(t, "gm", k)) < RTOL
    des.write("/dev/null")


def test_simulation_design_reports_status_of_its_points():
    """The batch path of SimulationDesign must not swallow what the point-by-point loop says
    through Halo._sync: a design point whose mass-limit search saturates (the reference's own
    answer is decided by rounding noise there; P(k) may differ by percents) raises a
    ChompParityWarning naming the point, its word is kept in design_status, and with_status=True
    hands the words back beside the frame (an int64 Series, never a row of the float frame)."""
    import warnings
    from chomp_amd import halo, simulation_design as sd, _lib
    numpy.random.seed(5)
    k = numpy.logspace(-3, 2, 16)
    params = {"omega_m0": [0.27, 0.21, 0.33], "omega_b0": [0.045, 0.04, 0.05],
              "h": [0.7, 0.65, 0.75], "sigma_8": [0.8, 0.69, 0.9], "n_scalar": [0.96, 0.92, 1.0]}
    z = 1.3923165344405541
    des = sd.SimulationDesignFlatUniverse(halo.Halo(z), "power_mm", params, n_design=5,
                                          independent_var=k)
    des._init_design_points()
    # point 3 := one of the cases the randomised soak found 5 % from the oracle (flagged there)
    des.points.loc[3, ["omega_m0", "omega_b0", "h", "sigma_8", "n_scalar"]] = [
        0.22563, 0.04499, 0.7203, 0.70574, 0.93183]
    # ... and point 1 := a benign one
    des.points.loc[1, ["omega_m0", "omega_b0", "h", "sigma_8", "n_scalar"]] = [
        0.30, 0.045, 0.7, 0.85, 0.96]
    with pytest.warns(_lib.ChompParityWarning, match="design point 3: .*saturated"):
        frame, words = des.run_design(with_status=True)
    assert des._batched() and frame.shape == (16, 5)    # the reference's frame and nothing else
    assert frame.values.dtype == numpy.float64 and words.dtype == numpy.int64
    st = des.design_status
    assert words is st and list(st.index) == list(des.points.index)
    assert int(st[3]) & _lib.ST_MASS_MIN_SATURATED and int(st[1]) == 0
    assert des.design_values is frame
    # the loop path says the same about the same point
    loop = sd.SimulationDesignFlatUniverse(halo.Halo(z), "power_mm", params, n_design=5,
                                           independent_var=k)
    loop._init_design_points()
    loop.points = des.points.copy()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        looped = loop.run_design(batched=False)
    assert any(issubclass(w.category, _lib.ChompParityWarning) for w in rec)
    assert loop.design_status is None
    ok = [c for c in range(5) if int(st[c]) == 0]
    assert rel_err(des.design_values.values[:, ok], looped.values[:, ok]) < 1e-12


def test_convenience_methods_vs_oracle(tmp_path):
    """Methods of the reference's classes that sit beside the hot path: MassFunction.dndm /
    write (mass_function.py:268-302), MultiEpoch's redshift-argument scalars
    (cosmology.py:977-1110), SingleEpoch.growth_factor_eval / transfer_function."""
    from chomp_amd import cosmology, mass_function
    from oracle import chomp_oracle as o
    e = o.epoch(None, 0.4)
    m = o.mass_table(e)
    mf = mass_function.MassFunction(0.4)
    mass = numpy.logspace(10, 15, 12)
    sp = m.nu_spline
    ref = numpy.array([0.5 * o.rho_bar(e) / (x * x) * o.f_nu(m, o.nu_of_mass(m, x)) *
                       sp.derivatives(numpy.log(x))[1] for x in mass])
    assert rel_err(mf.dndm(mass), ref) < 1e-6
    mf.write(str(tmp_path / "mf.txt"))
    rows = numpy.loadtxt(str(tmp_path / "mf.txt"))
    assert rows.shape == (50, 4) and rel_err(rows[:, 1], m.nu_arr) < 1e-6
    se = cosmology.SingleEpoch(0.4)
    k = numpy.logspace(-3, 2, 20)
    assert rel_err(se.transfer_function(k), o.eh_transfer(e, k)) < 1e-12
    a = numpy.array([0.3, 0.7, 1.0])
    assert rel_err(se.growth_factor_eval(a), o.growth_approx(e, a)) < 1e-14
    se.write(str(tmp_path / "pk.txt"))
    pk = numpy.loadtxt(str(tmp_path / "pk.txt"))
    assert pk.shape[1] == 2 and pk.shape[0] >= 202
    me = cosmology.MultiEpoch(0.0, 2.0)
    z = 0.4
    D = float(me.growth_factor(z))
    assert abs(me.delta_c(z) * D / o.delta_c(e) - 1) < 1e-12
    assert abs(me.delta_v(z) * D / (178.0 / o.omega_m(e) ** 0.55) - 1) < 1e-12
    assert abs(me.rho_bar(z) / o.rho_bar(e) - 1) < 1e-12
    e0 = o.epoch(None, 0.0)
    assert rel_err(me.delta_k(k, z), o.delta_k(e0, k) * D * D) < 1e-12
    assert abs(me.nu_r(8.0, z) / (me.delta_c(z) / (o.sigma_r(e0, 8.0) * D)) ** 2 - 1) < 1e-7
    me.write(str(tmp_path / "me.txt"), str(tmp_path / "me_pk.txt"))
    assert numpy.loadtxt(str(tmp_path / "me.txt")).shape == (50, 8)
    assert numpy.loadtxt(str(tmp_path / "me_pk.txt")).shape == (100, 2)


def test_hod_summary_integrals():
    """halo.py:709-838 against the reference (G10) and, for the satellite fraction the
    reference cannot reach with its own Zheng HOD, against the oracle."""
    from chomp_amd import cosmology, halo, mass_function
    from oracle import chomp_oracle as o
    g = load_golden("g10_hod_stats")
    for i, z in enumerate(g["z"]):
        for tag in ("st", "tinker"):
            kw = {}
            if tag == "tinker":
                cosmo = cosmology.SingleEpoch(float(z))
                kw = dict(cosmo_single_epoch=cosmo,
                          mass_func=mass_function.TinkerMassFunction(float(z), cosmo))
            h = halo.Halo(float(z), **kw)
            ref = g["%s_%d" % (tag, i)]
            assert abs(h.calculate_bias() / ref[0] - 1) < RTOL
            assert abs(h.calculate_m_eff() / ref[1] - 1) < RTOL
            assert abs(h.n_bar / ref[2] - 1) < RTOL
            with pytest.raises(TypeError):
                h.calculate_f_sat()
            e = o.epoch(None, float(z))
            t = o.halo_table(e, o.mass_table(e, kind=tag), families=())
            got = h._sync(0).hod_stats(0, 1)[0]
            assert numpy.allclose(got, o.hod_stats(t), rtol=RTOL)


def test_bao_transfer_function():
    """SURVEY 8(a) row a6, bracketed part: SingleEpoch(with_bao=True) and everything built on
    it, against the reference (G11)."""
    from chomp_amd import cosmology, halo
    g = load_golden("g11_bao")
    z, k = float(g["z"]), g["k"]
    c = cosmology.SingleEpoch(z, with_bao=True)
    assert rel_err(c.transfer_function(k), g["transfer"]) < 1e-9
    assert rel_err(c.linear_power(k), g["linear"]) < RTOL
    assert abs(c._sigma_norm / float(g["sigma_norm"]) - 1) < 1e-7
    assert rel_err(c.sigma_r(g["scale"]), g["sigma_r"]) < 1e-7
    h = halo.Halo(z, cosmo_single_epoch=c)
    assert numpy.array_equal(h.mass._ln_mass_array, g["ln_mass"])
    assert rel_err(h.mass._nu_array, g["nu"]) < 1e-7
    inr = k <= 100.0
    assert rel_err(h.power_mm(k)[inr], g["power_mm"][inr]) < RTOL
    assert rel_err(h.power_gm(k)[inr], g["power_gm"][inr]) < RTOL
    # large device grid through the streaming path, wiggles included
    import torch
    kd = torch.logspace(-3, 2, 1 << 16, dtype=torch.float64, device="cuda")
    ctx = h._sync(3)
    big = ctx.power(1, kd, 0, 1)[0].cpu().numpy()
    sub = h.power_mm(kd.cpu().numpy()[::1021])
    assert numpy.array_equal(sub, big[::1021])
    # the reference's set_cosmology re-runs __init__ without with_bao: so does the mirror
    c.set_cosmology(dict(c.cosmo_dict))
    assert c._with_bao is False
    plain = cosmology.SingleEpoch(z)
    assert rel_err(c.linear_power(k), plain.linear_power(k)) < 1e-12


def test_raw_kernel():
    """kernel.py:678-704: raw_kernel is the chi integral itself; at the table knots it is
    the tabulated value bit for bit, between them it is what the spline approximates."""
    from chomp_amd import kernel
    d = kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0)
    w = kernel.WindowFunctionGalaxy(d)
    wl = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2))
    for kern in (kernel.Kernel(1e-5, 1.0, w, w),
                 kernel.GalaxyGalaxyLensingKernel(1e-5, 1.0, w, wl)):
        knots, tab = kern._ln_ktheta_array, kern._kernel_array
        assert numpy.array_equal(kern.raw_kernel(knots), tab)
        mid = 0.5 * (knots[:-1] + knots[1:])[::7]
        raw = kern.raw_kernel(mid)
        scale = numpy.max(numpy.abs(tab))
        assert numpy.max(numpy.abs(raw - kern.kernel(mid))) < 1e-2 * scale   # spline error of 200 knots
        assert isinstance(kern.raw_kernel(float(mid[0])), float)
    with pytest.raises(Exception):
        kern.kernel_weighted_mean(lambda z: z)


def test_bao_batch_of_many_epochs_equals_the_small_batch():
    """The wiggle transfer function through the launch shapes of a large batch (the C ABI takes
    with_bao for any number of epochs; the drop-in classes only ever ask for one): 130 epochs --
    single-wavefront probes of the mass-limit search (k_epoch_probe<true, 1, 1>), their
    certification behind the kernel boundary, single-wavefront nu-table and knot blocks --
    against the same epochs in batches of four (four wavefronts per integral: other orders of
    summation): the mass limits bit for bit, the nu table and P_mm to 1e-11."""
    import numpy as np
    from chomp_amd import grid, _lib
    z = np.linspace(0.0, 1.3, 130)
    k = np.logspace(-3, 2, 32)

    class BaoGrid(grid.HaloGrid):
        def setup(self, which="power_mm"):
            _, need = grid._WHICH[which]
            self.ctx.epochs_set(self._c_cosmo, self._z, with_bao=True)
            self.ctx.stage_k(self._c_halo, self.kind, self._c_halo, self._c_hod, need)
            self._tables = need

    big = BaoGrid(z)
    p_big = big.power("power_mm", k)
    assert not big.status().any()
    for i0 in (0, 64, 126):
        small = BaoGrid(z[i0:i0 + 4])
        p = small.power("power_mm", k)
        for j in range(4):
            a, b = big.ctx.scalars(i0 + j), small.ctx.scalars(j)
            assert a["ln_mass_min"] == b["ln_mass_min"] and a["ln_mass_max"] == b["ln_mass_max"]
            assert np.max(np.abs(big.ctx.table("nu", i0 + j) / small.ctx.table("nu", j) - 1)) < 1e-11
        assert np.max(np.abs(p_big[i0:i0 + 4] / p - 1)) < 1e-11
    # ... and the wiggles are there: not the no-wiggle spectrum
    plain = grid.HaloGrid(z[:4]).power("power_mm", k)
    assert np.max(np.abs(p_big[:4] / plain - 1)) > 1e-3


def test_gaussian_covariance():
    """SURVEY 8(f) rank 4, Gaussian part: Covariance(corr, corr, nongaussian_cov=False)
    against the reference (G12): the projected-spectrum table over ln K, covariance_G of
    every bin pair, the Poisson term, get_covariance() and the written file."""
    from chomp_amd import correlation, covariance, halo, kernel
    from oracle import chomp_oracle as o
    g = load_golden("g12_covariance_gaussian")
    for tag, ps, kws in (
            ("mag", "power_mm", dict(bins_per_decade=2.0, survey_area_deg2=25.0,
                                     n_a=[1.0e10, 1.0e10], n_b=[1.0e10, 1.0e10], variance=1.0)),
            ("auto", "power_gg", dict(bins_per_decade=3.0, survey_area_deg2=100.0,
                                      n_a=2.0e6, n_b=2.0e6, variance=0.3))):
        wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0))
        wb = wa if tag == "auto" else kernel.WindowFunctionConvergence(
            kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2))
        from chomp_amd import cosmology
        cm = cosmology.MultiEpoch(0.0, 5.0)
        d2r = numpy.pi / 180.0
        kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
        corr = correlation.Correlation(0.01, 1.0, kern, input_halo=halo.Halo(0.0),
                                       power_spec=ps)
        cv = covariance.Covariance(corr, corr, nongaussian_cov=False, power_spec=ps, **kws)
        bins = cv.annular_bins
        assert numpy.allclose([b.center for b in bins], g[tag + "_center"], rtol=1e-15)
        assert numpy.allclose([b.inner for b in bins], g[tag + "_inner"], rtol=1e-15)
        assert list(cv.equal_windows) == list(g[tag + "_equal_windows"])
        assert [bool(x) for x in cv.cosmic_shear] == list(g[tag + "_cosmic_shear"])
        cv._initialize_halo_splines()
        sc = g[tag + "_scalars"]       # D_z, chi_min, chi_max, ln_K_min, ln_K_max, j0, area, z_bar
        assert abs(cv._D_z_a / sc[0] - 1) < 1e-9 and abs(cv._z_bar_G_a - sc[7]) < 1e-12
        assert abs(cv._ln_K_min - sc[3]) < 1e-9 and abs(cv._ln_K_max - sc[4]) < 1e-9
        assert abs(cv._j0_limit / sc[5] - 1) < 1e-14 and abs(cv.area / sc[6] - 1) < 1e-14
        assert numpy.allclose(cv._ln_K_array, g[tag + "_ln_K"], rtol=0, atol=1e-9)
        scale = numpy.max(numpy.abs(g[tag + "_proj"]))
        assert numpy.max(numpy.abs(cv._halo_a_array - g[tag + "_proj"])) < 1e-6 * scale
        big = numpy.abs(g[tag + "_proj"]) > 1e-6 * scale
        assert rel_err(cv._halo_a_array[big], g[tag + "_proj"][big]) < RTOL
        nb = len(bins)
        G = numpy.array([[cv.covariance_G(a.center, b.center, a.delta, b.delta) for b in bins]
                         for a in bins])
        assert rel_err(G, g[tag + "_G"]) < RTOL
        P = numpy.array([cv.covariance_P(b.delta, b.center) for b in bins])
        assert rel_err(P, g[tag + "_P"]) < 1e-12
        full = cv.get_covariance()
        assert full.shape == (nb, nb) and rel_err(full, g[tag + "_cov"]) < RTOL
        assert abs(cv.covariance(bins[0], bins[0]) / g[tag + "_cov"][0, 0] - 1) < RTOL
        assert abs(cv.covariance(bins[0], bins[1]) / g[tag + "_cov"][0, 1] - 1) < RTOL
    # the Romberg levels of the last case against the oracle's (the stopping rule is part
    # of the answer)
    me = o.multi_epoch(0.0, 5.0)
    ow = o.window_table("galaxy", o.dndz_maglim(0.0, 2.0, 2.0, 0.3, 2.0), me)
    kt = o.kernel_table(1e-6 * d2r, 100 * d2r, ow, ow, me)
    e = o.epoch(None, kt.z_bar)
    t = o.halo_table(e, o.mass_table(e), o.zheng(), families=("gg",))
    lev = []
    ocv = o.covariance_table(kt, lambda k: o.halo_power(t, "gg", k), levels=lev)
    agree = numpy.mean(numpy.asarray(lev) == cv._halo_a_levels)
    assert agree == 1.0, (lev, cv._halo_a_levels)
    with pytest.raises(Exception):
        covariance.Covariance(corr, corr)                  # trispectrum terms: scope error


def test_bao_projections():
    """A Halo on SingleEpoch(with_bao=True) through w(theta), C_l, xi(r) and the covariance
    table, against the reference (G13): the wiggle transfer function is carried through the
    projection kernels, not only through P(k)."""
    from chomp_amd import correlation, cosmology, covariance, halo, kernel
    g = load_golden("g13_bao_projections")
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
    assert abs(kern.z_bar - float(g["z_bar"])) < 1e-12
    for ps in ("power_mm", "power_gg"):
        zb = kern.z_bar      # built at z_bar: moving a Halo in z re-creates its cosmology
        h = halo.Halo(zb, cosmo_single_epoch=cosmology.SingleEpoch(zb, with_bao=True))
        corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec=ps)
        assert h.cosmo._with_bao
        assert rel_err(corr.correlation(g["theta"]), g["w_" + ps]) < RTOL
        cf = correlation.CorrelationFourier(10, 1e4, kern, input_halo=h, powSpec=ps)
        assert rel_err(cf.correlation(g["ell"]), g["cl_" + ps]) < RTOL
    assert rel_err(h.power_mm(g["k"]), g["p_mm_zbar"]) < RTOL
    # the wiggles are there: the no-wiggle spectrum at the same z differs by percents
    assert numpy.max(numpy.abs(g["p_mm_zbar"] / g["p_mm_zbar_nowiggle"] - 1)) > 0.02
    # ... and a Halo that a Correlation moves to z_bar loses them, as in the reference
    moved = halo.Halo(0.0, cosmo_single_epoch=cosmology.SingleEpoch(0.0, with_bao=True))
    correlation.Correlation(0.001, 1.0, kern, input_halo=moved, power_spec="power_mm")
    assert not moved.cosmo._with_bao
    assert rel_err(moved.power_mm(g["k"]), g["p_mm_moved"]) < RTOL
    assert rel_err(g["p_mm_moved"], g["p_mm_zbar_nowiggle"]) < 1e-12
    # the covariance table integrates the same spectrum: at its peak it is the Limber C_l
    cv = covariance.Covariance(corr, corr, nongaussian_cov=False, power_spec="power_gg")
    cv._initialize_halo_splines()
    assert numpy.all(numpy.isfinite(cv._halo_a_array)) and cv._halo_a_array.max() > 0
    h3 = halo.Halo(0.5, cosmo_single_epoch=cosmology.SingleEpoch(0.5, with_bao=True))
    c3 = correlation.Correlation3d(1.0, 150.0, redshift=0.5, input_halo=h3, powSpec="power_mm")
    xi = numpy.array([c3.raw_correlation(x) for x in g["r"]])
    scale = numpy.abs(g["xi_raw"]).max()
    assert numpy.max(numpy.abs(xi - g["xi_raw"])) < 1e-6 * scale
    big = numpy.abs(g["xi_raw"]) > 1e-4 * scale
    assert rel_err(xi[big], g["xi_raw"][big]) < RTOL


def test_boxcar_dndz():
    """The base class dNdz (raw_dndz = 1 between z_min and z_max, kernel.py:26-86) through a
    galaxy and a convergence window, the J0 kernel and w(theta), against the reference (G14)."""
    from chomp_amd import correlation, cosmology, halo, kernel
    g = load_golden("g14_boxcar_dndz")
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    da = kernel.dNdz(0.2, 0.6)
    assert numpy.array_equal(da.dndz(numpy.array([0.1, 0.3, 0.7])) > 0, [False, True, False])
    assert abs(da.norm - 2.5) < 1e-12
    wa = kernel.WindowFunctionGalaxy(da, cm)
    wb = kernel.WindowFunctionConvergence(kernel.dNdz(0.8, 1.2), cm)
    kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
    assert kern.z_bar == float(g["z_bar"])
    ctx = kern._dev()
    info = ctx.kernel_info()
    assert abs(info["norm_a"] / float(g["wa_norm"]) - 1) < 1e-7
    assert abs(info["norm_b"] / float(g["wb_norm"]) - 1) < 1e-7
    assert numpy.allclose(ctx.kernel_table("wa"), g["wa"], rtol=2e-6, atol=1e-16)
    assert numpy.allclose(ctx.kernel_table("wb"), g["wb"], rtol=5e-6, atol=1e-16)
    scale = numpy.max(numpy.abs(g["kernel"]))
    assert numpy.allclose(ctx.kernel_table("kernel"), g["kernel"], rtol=2e-5, atol=2e-6 * scale)
    corr = correlation.Correlation(0.001, 1.0, kern, input_halo=halo.Halo(0.0),
                                   power_spec="power_mm")
    assert rel_err(corr.correlation(g["theta"]), g["w_mm"]) < RTOL

    class Other(kernel.dNdz):              # a user-defined raw_dndz cannot run on the device
        def raw_dndz(self, redshift):
            return redshift
    with pytest.raises(Exception):
        kernel.WindowFunctionGalaxy(Other(0.1, 0.5), cm)._dev()


def test_dndz_interpolation():
    """dNdzInterpolation (kernel.py:181-208): a tabulated p(z) through both windows, the J0
    kernel and w(theta) against the reference (G15); the same table through a smoothing
    spline."""
    from chomp_amd import correlation, cosmology, halo, kernel
    g = load_golden("g15_dndz_interpolation")
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    dist = kernel.dNdzInterpolation(g["z_tab"], g["p_tab"])
    assert dist.z_min == g["z_tab"][0] and dist.z_max == g["z_tab"][-1]
    assert numpy.allclose(dist.dndz(g["z_probe"]), g["dndz_probe"], rtol=1e-7, atol=1e-12)
    wa = kernel.WindowFunctionGalaxy(dist, cm)
    wb = kernel.WindowFunctionConvergence(kernel.dNdzInterpolation(g["z_tab"], g["p_tab"]), cm)
    kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
    assert kern.z_bar == float(g["z_bar"])
    ctx = kern._dev()
    info = ctx.kernel_info()
    assert abs(info["norm_a"] / float(g["wa_norm"]) - 1) < 1e-7
    assert numpy.allclose(ctx.kernel_table("wa"), g["wa"], rtol=2e-6, atol=1e-12 * g["wa"].max())
    assert numpy.allclose(ctx.kernel_table("wb"), g["wb"], rtol=5e-6, atol=1e-12 * g["wb"].max())
    scale = numpy.max(numpy.abs(g["kernel"]))
    assert numpy.allclose(ctx.kernel_table("kernel"), g["kernel"], rtol=2e-5, atol=2e-6 * scale)
    corr = correlation.Correlation(0.001, 1.0, kern, input_halo=halo.Halo(0.0),
                                   power_spec="power_mm")
    assert rel_err(corr.correlation(g["theta"]), g["w_mm"]) < RTOL
    sm = kernel.dNdzInterpolation(g["z_tab"], g["p_tab"], interpolation_order=3, smoothing=1e-4)
    ws = kernel.WindowFunctionGalaxy(sm, cosmology.MultiEpoch(0.0, 5.0))
    tab = ws._dev().kernel_table("wa")
    assert numpy.allclose(tab, g["smooth_wf"], rtol=2e-6, atol=1e-12 * g["smooth_wf"].max())
    sm.normalize()
    assert abs(sm.norm / float(g["smooth_norm"]) - 1) < 1e-7


def test_flat_and_delta_convergence_windows():
    """WindowFunctionFlatConvergence and WindowFunctionConvergenceDelta (kernel.py:487-556)
    against a galaxy window in a J0 kernel and through w(theta), vs the reference (G16)."""
    from chomp_amd import correlation, cosmology, halo, kernel
    g = load_golden("g16_flat_delta_windows")
    d2r = numpy.pi / 180.0
    for tag in ("flat", "delta"):
        cm = cosmology.MultiEpoch(0.0, 5.0)
        wa = kernel.WindowFunctionGalaxy(kernel.dNdzGaussian(0.0, 2.0, 0.5, 0.1), cm)
        wb = (kernel.WindowFunctionFlatConvergence(0.3, 0.9, cm) if tag == "flat"
              else kernel.WindowFunctionConvergenceDelta(1.1, cm))
        kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
        assert kern.z_bar == float(g[tag + "_z_bar"])
        ctx = kern._dev()
        assert numpy.allclose(ctx.kernel_table("wb_chi"), g[tag + "_wb_chi"], rtol=1e-7)
        assert numpy.allclose(ctx.kernel_table("wb"), g[tag + "_wb"], rtol=1e-9, atol=1e-18)
        scale = numpy.max(numpy.abs(g[tag + "_kernel"]))
        assert numpy.allclose(ctx.kernel_table("kernel"), g[tag + "_kernel"], rtol=2e-5,
                              atol=2e-6 * scale)
        corr = correlation.Correlation(0.001, 1.0, kern, input_halo=halo.Halo(0.0),
                                       power_spec="power_mm")
        assert rel_err(corr.correlation(g["theta"]), g[tag + "_w_mm"]) < RTOL
        assert numpy.allclose(wb.window_function(g[tag + "_wb_chi"][5:8]), g[tag + "_wb"][5:8],
                              rtol=1e-9)


def test_halofit_on_the_wiggle_transfer_function():
    """HaloFit(z, cosmo_single_epoch=SingleEpoch(z, with_bao=True)): the sigma table, the fit,
    the three spectra and w(theta) against the reference (G17)."""
    from chomp_amd import correlation, cosmology, halo, kernel
    g = load_golden("g17_halofit_bao")
    z, k = float(g["z"]), g["k"]
    hf = halo.HaloFit(z, cosmo_single_epoch=cosmology.SingleEpoch(z, with_bao=True))
    assert rel_err(hf.power_mm(k), g["mm"]) < RTOL
    inr = k <= 100.0
    assert rel_err(hf.power_gm(k)[inr], g["gm"][inr]) < RTOL
    assert rel_err(hf.power_gg(k)[inr], g["gg"][inr]) < RTOL
    pars = numpy.array([hf._k_s, hf._n_eff, hf._C, hf._a_n, hf._b_n, hf._c_n, hf._gamma_n,
                        hf._alpha_n, hf._beta_n, hf._nu_n])
    assert rel_err(pars, g["pars"]) < 1e-6
    assert numpy.max(numpy.abs(g["mm"] / g["mm_nowiggle"] - 1)) > 0.02      # wiggles are in
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
    zb = kern.z_bar
    assert abs(zb - float(g["z_bar"])) < 1e-12
    hz = halo.HaloFit(zb, cosmo_single_epoch=cosmology.SingleEpoch(zb, with_bao=True))
    corr = correlation.Correlation(0.001, 1.0, kern, input_halo=hz, power_spec="power_mm")
    assert hz.cosmo._with_bao
    assert rel_err(corr.correlation(g["theta"]), g["w_mm"]) < RTOL


def test_dndchi_gaussian():
    """dNdChiGaussian (kernel.py:114-145) as a lens distribution, against the reference (G18)."""
    from chomp_amd import cosmology, kernel
    g = load_golden("g18_dndchi_gaussian")
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    dist = kernel.dNdChiGaussian(600.0, 1800.0, 1200.0, 150.0, cm)
    assert abs(dist.z_min / float(g["z_min"]) - 1) < 1e-8
    assert abs(dist.z_max / float(g["z_max"]) - 1) < 1e-8
    assert numpy.allclose(dist.dndz(g["z_probe"]), g["dndz_probe"], rtol=1e-6, atol=1e-9)
    wa = kernel.WindowFunctionGalaxy(dist, cm)
    wb = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2), cm)
    kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
    assert abs(kern.z_bar / float(g["z_bar"]) - 1) < 1e-9      # (z limits come from a spline inverse)
    ctx = kern._dev()
    assert abs(ctx.kernel_info()["norm_a"] / float(g["wa_norm"]) - 1) < 1e-6
    wa_tab = ctx.kernel_table("wa")
    assert numpy.allclose(wa_tab[:-1], g["wa"][:-1], rtol=5e-6, atol=1e-9 * g["wa"].max())
    # the last knot sits exactly on z_max = redshift(chi_max): whether redshift(chi(z_max)) <= z_max
    # holds there is a matter of the last bit of two spline evaluations (4e-4 of the peak either way)
    assert wa_tab[-1] == 0.0 or abs(wa_tab[-1] / g["wa"][-1] - 1) < 5e-6
    scale = numpy.max(numpy.abs(g["kernel"]))
    assert numpy.allclose(ctx.kernel_table("kernel"), g["kernel"], rtol=2e-4, atol=2e-5 * scale)
