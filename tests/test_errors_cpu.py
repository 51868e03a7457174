"""CPU tests of the host-side mirror: argument handling and error behaviour that the
reference defines (KeyError on missing keys) or that the accelerated scope adds
(ChompScopeError for reference features outside the hot path)."""
import numpy
import pytest


def test_missing_keys_raise_keyerror():
    from chomp_amd import cosmology, halo, hod
    cd = dict(omega_m0=0.3)
    with pytest.raises(KeyError):
        cosmology.SingleEpoch(0.0, cd)
    with pytest.raises(KeyError):
        hod.HODZheng({"log_M_min": 12.0})
    with pytest.raises(KeyError):
        halo.Halo(0.0, halo_dict={"stq": 0.3})


def test_scope_errors():
    from chomp_amd import cosmology, halo, kernel, _lib
    c = cosmology.SingleEpoch(0.0, with_bao=True)     # accelerated
    assert c._with_bao is True
    c.set_cosmology(dict(c.cosmo_dict))               # re-runs __init__: drops it, as the reference
    assert c._with_bao is False
    h = halo.Halo(0.0, extrapolate=True)          # accelerated since SURVEY 8(f) rank 2
    assert h.get_extrapolation() and (h._power_code(_lib.P_GM) & _lib.P_EXTRAPOLATE)
    assert h._power_code(_lib.P_LIN) == _lib.P_LIN
    with pytest.raises(_lib.ChompScopeError):
        halo.Halo(0.0, halo_dict=dict(stq=0.3, st_little_a=0.707, c0=9.0, beta=-0.13,
                                      alpha=-1.5, delta_v=-1.0))
    d = kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2)
    w = kernel.WindowFunctionGalaxy(d)
    with pytest.raises(_lib.ChompScopeError):
        kernel.Kernel(1e-6, 1.0, w, w, force_quad=True)


def test_redshift_distribution_clipping():
    """Constructor-side rules that are pure host logic (kernel.py:101-104, 164-173)."""
    from chomp_amd import kernel
    d = kernel.dNdzGaussian(0.0, 5.0, 1.0, 0.2)
    assert d.z_min == 0.0 and abs(d.z_max - 2.6) < 1e-15
    m = kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0)
    assert abs(m.z_max - 0.3 * numpy.sqrt(-numpy.log(1.48e-8))) < 1e-12
    mi = kernel.dNdzMagLim(0.0, 2.0, 2, 0.3, 2)        # int b: Python-2 1/b == 0
    assert abs(mi.z_max - 0.3) < 1e-15


def test_hod_derived_constants_match_oracle():
    from chomp_amd import hod
    from oracle import chomp_oracle as o
    for hd in (o.default_hod_dict, dict(o.default_hod_dict, sigma=0.71, log_M_min=14.06)):
        a, b = hod.HODZheng(hd), o.zheng(hd)
        assert abs(a.first_moment_zero / b.first_moment_zero - 1) < 1e-12
        assert a.second_moment_zero == b.second_moment_zero
        assert a._safe_norm == b.safe_norm
        m = numpy.logspace(10, 16, 50)
        assert numpy.allclose(a.first_moment(m), o.zheng_first(b, m), rtol=1e-13, atol=1e-300)
        assert numpy.allclose(a.second_moment(m), o.zheng_second(b, m), rtol=1e-13, atol=1e-300)


def test_theta_bins_match_oracle():
    from oracle import chomp_oracle as o
    from chomp_amd import correlation

    class _K(object):        # only the binning part of Correlation.__init__ is exercised
        pass
    for args in ((0.001, 1.0, 5.0), (0.01, 10.0, 3.0), (0.5, 0.5, 5.0)):
        c = correlation.Correlation.__new__(correlation.Correlation)
        try:
            correlation.Correlation.__init__(c, args[0], args[1], _K(), bins_per_decade=args[2])
        except AttributeError:
            pass            # stops at the kernel access, after theta_array is built
        assert numpy.allclose(c.theta_array, o.theta_bins(*args), rtol=1e-15)


def test_simulation_design_host_logic():
    """simulation_design.py:17-114: Latin-hypercube sampling and the design frame (no GPU)."""
    from chomp_amd import halo, simulation_design as sd
    numpy.random.seed(3)
    P = sd.random_lhs(16, 3)
    assert P.shape == (16, 3) and numpy.all((P >= 0) & (P < 1))
    for j in range(3):                      # one point per stratum in every dimension
        assert sorted(numpy.floor(P[:, j] * 16).astype(int).tolist()) == list(range(16))
    params = {"omega_m0": [0.27, 0.2, 0.4], "log_M_min": [12.1, 11.5, 12.8], "c0": [9, 7, 11]}
    des = sd.SimulationDesign(halo.Halo(0.0), "power_gm", params, n_design=5,
                              independent_var=numpy.logspace(-2, 1, 4))
    assert des._param_types == ["cosmo_dict", "hod_dict", "halo_dict"]
    assert des._vary_cosmology and des._vary_hod and des._vary_halo
    assert not des._batched()               # halo parameters vary: point-by-point loop
    des._init_design_points()
    assert des.points.shape == (5, 3)
    for key, (_, lo, hi) in params.items():
        assert numpy.all((des.points[key] >= lo) & (des.points[key] <= hi))
    des2 = sd.SimulationDesign(halo.Halo(0.0), "power_gm", {"sigma_8": [0.8, 0.7, 0.9]},
                               n_design=4, independent_var=numpy.logspace(-2, 1, 4))
    assert des2._batched()


def test_correlation3d_host_logic():
    """correlation.py:414-456: r grid, extrapolation switch, power-spectrum fallback."""
    from chomp_amd import correlation, halo
    c3 = correlation.Correlation3d(0.1, 50.0, powSpec="power_mm")
    assert c3.r_array.size == 50 and abs(c3.r_array[0] - 0.1) < 1e-15
    assert not c3.halo.get_extrapolation()
    c3w = correlation.Correlation3d(0.1, 50.0, powSpec="power_mm", k_min=1e-4, k_max=1e3)
    assert c3w.halo.get_extrapolation() and c3w._k_lim == (1e-4, 1e3)
    one = correlation.Correlation3d(2.0, 2.0)
    assert one.r_array.tolist() == [2.0] and one._power_name == "linear_power"
    assert isinstance(halo.HaloExclusion(0.2), halo.Halo)


def test_covariance_host_logic():
    """covariance.py:46-130: bins, survey scalars and the scope of the accelerated class."""
    from chomp_amd import correlation, covariance, kernel, _lib
    from oracle import chomp_oracle as o
    w = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0))
    ws = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2))
    kern = kernel.Kernel(1e-8, 1.0, w, ws)
    corr = correlation.Correlation.__new__(correlation.Correlation)   # no device here
    corr.log_theta_min = numpy.log10(0.01 * numpy.pi / 180)
    corr.log_theta_max = numpy.log10(1.0 * numpy.pi / 180)
    corr.kernel = kern

    class _H(object):
        power_mm = None
    corr.halo = _H()
    with pytest.raises(_lib.ChompScopeError):
        covariance.Covariance(corr, corr)                      # nongaussian_cov defaults to True
    other = correlation.Correlation.__new__(correlation.Correlation)
    other.__dict__.update(corr.__dict__)
    with pytest.raises(_lib.ChompScopeError):
        covariance.Covariance(corr, other, nongaussian_cov=False)
    cv = covariance.Covariance(corr, corr, bins_per_decade=3.0, survey_area_deg2=100.0,
                               n_a=[2e6, 3e6], n_b=4e6, variance=0.3, nongaussian_cov=False)
    inner, outer, center, delta = o.annular_bins(0.01, 1.0, 3.0)
    assert numpy.array_equal([b.center for b in cv.annular_bins], center)
    assert numpy.array_equal([b.delta for b in cv.annular_bins], delta)
    assert (cv.n_a1, cv.n_a2, cv.n_b1, cv.n_b2) == (2e6, 3e6, 4e6, 4e6)
    assert cv.cosmic_shear == [False, False] or list(cv.cosmic_shear) == [0, 0]
    assert cv.proj_power_poisson(0) == 0.0
    assert cv.proj_power_poisson(4) == 0.09 / (2e6 / cv.area)
    p = cv.covariance_P(delta[1], center[1])
    assert abs(p / o.covariance_P(center[1], delta[1], cv.area, 2e6, 3e6, 0.3) - 1) < 1e-14
    with pytest.raises(_lib.ChompScopeError):
        cv.covariance_NG(0.01, 0.01)


def test_halo_sync_retries_the_mass_function_after_a_failed_setup():
    """A set-up that raises (HIP error, scope error, a status warning turned error) must
    not leave the Halo believing its mass function is on the device: the retry runs
    stage_k (mass function + halo model) again, never halo_setup on the old nu tables."""
    from chomp_amd import halo, _lib

    class _Ctx(object):
        def __init__(self):
            self.calls = []
            self.fail = 1

        def epochs_set(self, *a):
            self.calls.append("epochs_set")

        def stage_k(self, *a):
            self.calls.append("stage_k")
            if self.fail:
                self.fail -= 1
                raise _lib.ChompError("injected")

        def halo_setup(self, *a):
            self.calls.append("halo_setup")

        def status_post(self):
            self.calls.append("status_post")

        def warn_status(self, *a, **k):
            return [0]

    h = halo.Halo(0.0)
    h._ctx = ctx = _Ctx()
    with pytest.raises(_lib.ChompError):
        h._sync(_lib.FAM_MM)
    assert h._mass_sig is None and not h._nbar_valid and not h._initialized_h_m
    h._sync(_lib.FAM_MM)
    assert ctx.calls == ["epochs_set", "stage_k", "stage_k", "status_post"]
    assert h._initialized_h_m and h._initialized_pp_mm and h._nbar_valid
    h._sync(_lib.FAM_MM)                     # up to date: nothing more
    assert len(ctx.calls) == 4
    from chomp_amd import defaults
    h.set_hod(dict(defaults.default_hod_dict))   # HOD only: halo model alone
    h._sync(_lib.FAM_GM)
    assert ctx.calls[4:] == ["halo_setup", "status_post"]


def test_loader_refuses_a_library_built_from_other_sources(monkeypatch, tmp_path):
    """chomp_amd/_lib.py: a hash mismatch whose rebuild fails is an error, not a warning
    (a test run must never certify a binary that is not the tree); CHOMP_ALLOW_STALE_LIB=1
    is the explicit opt-in of a box without hipcc."""
    import warnings
    from chomp_amd import _lib
    if not __import__("os").path.exists(_lib.LIB_PATH):
        pytest.skip("no built library in the tree")
    monkeypatch.setattr(_lib, "_lib", None)

    def _no_hipcc(*a, **k):
        raise OSError("hipcc: not found (injected)")
    monkeypatch.setattr(_lib, "build", _no_hipcc)
    monkeypatch.delenv("CHOMP_ALLOW_STALE_LIB", raising=False)
    with pytest.raises(ImportError, match="built from other sources"):
        _lib.lib()
    monkeypatch.setenv("CHOMP_ALLOW_STALE_LIB", "1")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        L = _lib.lib()
    assert hasattr(L, "chomp_power") and any("CHOMP_ALLOW_STALE_LIB" in str(x.message) for x in w)
