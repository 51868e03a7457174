"""CPU tests of the fp64 building blocks used inside the HIP kernels
(chomp_amd/csrc/chomp_math.h), compiled for the host by tests/hostcheck.
They check the device special functions / splines / integrand pieces against
SciPy and the oracle; the kernels themselves are tested with -m gpu."""
import ctypes
import os
import subprocess

import numpy
import pytest
from scipy import special
from scipy.interpolate import InterpolatedUnivariateSpline

from conftest import ROOT, rel_err
from oracle import chomp_oracle as o

HC = os.path.join(ROOT, "tests", "hostcheck")
dp = ctypes.POINTER(ctypes.c_double)


def _p(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def hc():
    so = os.path.join(HC, "libhostcheck.so")
    src = os.path.join(HC, "hostcheck.cpp")
    hdr = os.path.join(ROOT, "chomp_amd", "csrc", "chomp_math.h")
    if (not os.path.exists(so) or
            os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared",
                               "-o", so, src])
    return ctypes.CDLL(so)


def test_sici(hc):
    x = numpy.concatenate([numpy.logspace(-8, numpy.log10(4), 400)[:-1],
                           numpy.linspace(4, 40, 2000), numpy.logspace(1.6, 5, 500),
                           [4.0, 32.0 / 7, 8.0, 16.0, 1e6]])
    si, ci = numpy.empty_like(x), numpy.empty_like(x)
    hc.hc_sici(_p(x), x.size, _p(si), _p(ci))
    rsi, rci = special.sici(x)
    assert numpy.max(numpy.abs(si - rsi)) < 4e-16 * 4
    assert numpy.max(numpy.abs(ci - rci)) < 3e-15     # |Ci| up to 18 at x=1e-8


def test_fast_log(hc):
    rng = numpy.random.default_rng(7)
    x = numpy.concatenate([numpy.logspace(-300, 300, 20001), 1.0 + rng.uniform(-1e-3, 1e-3, 5000),
                           rng.uniform(0.5, 2.0, 20000), [1.0, 2.0, 0.5, numpy.sqrt(0.5), numpy.sqrt(2.0)]])
    out = numpy.empty_like(x)
    hc.hc_fast_log(_p(x), x.size, _p(out))
    ref = numpy.log(x)
    big = numpy.abs(ref) > 1e-3
    assert numpy.max(numpy.abs(out - ref)[big] / numpy.abs(ref)[big]) < 4e-16
    assert numpy.max(numpy.abs(out - ref)[~big]) < 4e-19


def test_fast_sincos(hc):
    rng = numpy.random.default_rng(11)
    x = numpy.concatenate([numpy.logspace(-12, 6, 40001), -numpy.logspace(-8, 5, 5001),
                           rng.uniform(0, 1e4, 50000), numpy.arange(0, 2000) * numpy.pi / 4,
                           [0.0, numpy.pi / 2, numpy.pi, 1e8, 3.3e9]])
    s, c = numpy.empty_like(x), numpy.empty_like(x)
    hc.hc_fast_sincos(_p(x), x.size, _p(s), _p(c))
    assert numpy.max(numpy.abs(s - numpy.sin(x))) < 2.3e-16
    assert numpy.max(numpy.abs(c - numpy.cos(x))) < 2.3e-16
    small = numpy.abs(x) < 1e-3                      # relative accuracy where sin x ~ x
    assert numpy.max(numpy.abs(s[small] - numpy.sin(x[small])) /
                     numpy.maximum(numpy.abs(x[small]), 1e-300)) < 2.3e-16
    big = numpy.array([1e12, 7.7e14, -3e13])          # degrades gracefully, stays in [-1, 1]
    sb, cb = numpy.empty_like(big), numpy.empty_like(big)
    hc.hc_fast_sincos(_p(big), big.size, _p(sb), _p(cb))
    assert numpy.all(numpy.abs(sb) <= 1.0 + 1e-15) and numpy.all(numpy.abs(cb) <= 1.0 + 1e-15)
    assert numpy.max(numpy.abs(sb - numpy.sin(big))) < 1e-15 * 1e15 * 1e-15 + 1e-3


def test_bessel(hc):
    x = numpy.concatenate([numpy.linspace(0, 32, 4001), numpy.linspace(32, 200, 3000),
                           numpy.logspace(2.3, 5, 200)])
    out = numpy.empty_like(x)
    # the path only needs x <= 8th zero (24.4 / 27.4, kernel.py:628-629, 806-807);
    # beyond x ~ 1e3 the phase x - (n/2+1/4) pi itself carries x*eps of error in
    # both implementations
    small = x <= 200.0
    for order, ref in ((0, special.j0(x)), (2, special.jn(2, x))):
        hc.hc_bessel(order, _p(x), x.size, _p(out))
        assert numpy.max(numpy.abs(out - ref)[small]) < 2e-15
        assert numpy.max(numpy.abs(out - ref)[~small] / x[~small]) < 2e-17


def test_notaknot_spline(hc):
    rng = numpy.random.default_rng(7)
    # non-uniform knots (like the nu grid) and uniform knots (like ln k)
    for uniform in (0, 1):
        if uniform:
            x = numpy.linspace(numpy.log(1e-3), numpy.log(1e2), 50)
        else:
            x = numpy.cumsum(rng.uniform(0.05, 2.0, 50))
        y = numpy.sin(x) * numpy.exp(0.1 * x) + 3
        ref = InterpolatedUnivariateSpline(x, y)
        xe = numpy.concatenate([numpy.linspace(x[0] - 1.0, x[-1] + 1.0, 1001), x])
        out = numpy.empty_like(xe)
        hc.hc_spline(_p(x), _p(y), x.size, _p(xe), xe.size, _p(out), uniform)
        assert numpy.max(numpy.abs(out - ref(xe)) / (1 + numpy.abs(ref(xe)))) < 2e-13


def test_parallel_spline_build(hc):
    """The PCR (parallel cyclic reduction) build used inside the kernels gives the
    same not-a-knot spline, on uniform, non-uniform and steeply graded knots."""
    rng = numpy.random.default_rng(11)
    cases = [numpy.linspace(-6.9, 4.6, 50), numpy.cumsum(rng.uniform(0.05, 2.0, 50)),
             numpy.geomspace(0.1, 47.0, 50), numpy.linspace(0.0, 1.0, 100),
             numpy.cumsum(rng.uniform(0.01, 5.0, 9))]
    for x in cases:
        y = numpy.cos(0.7 * x) * numpy.exp(0.05 * x) + 0.1 * x
        ref = InterpolatedUnivariateSpline(x, y)
        xe = numpy.concatenate([numpy.linspace(x[0], x[-1], 777), x])
        out = numpy.empty_like(xe)
        hc.hc_spline_pcr(_p(x), _p(y), x.size, _p(xe), xe.size, _p(out))
        assert numpy.max(numpy.abs(out - ref(xe)) / (1 + numpy.abs(ref(xe)))) < 5e-13


def test_quintic_derivatives(hc):
    """HaloFit's k=5 spline derivatives (halo.py:1289-1292)."""
    x = numpy.linspace(numpy.log(0.1), numpy.log(10.0), 50)
    y = -1.3 * x - 0.21 * x ** 2 + 0.01 * numpy.sin(2 * x) + 0.4
    sp = InterpolatedUnivariateSpline(x, y, k=5)
    d = numpy.empty(2)
    for xq in (-2.0, -0.9957, 0.0, 0.3337, 1.9, x[7], x[0], x[-1]):
        hc.hc_quintic(_p(x), _p(y), x.size, ctypes.c_double(xq), _p(d))
        ref = sp.derivatives(xq)[1:3]
        assert numpy.allclose(d, ref, rtol=1e-9, atol=1e-11), (xq, d, ref)


def _epoch(hc, cd, z, sigma_norm=1.0):
    n = hc.hc_sizeof_epoch()
    buf = (ctypes.c_char * n)()
    c = numpy.array([cd[k] for k in ("omega_m0", "omega_b0", "omega_l0", "omega_r0",
                                     "cmb_temp", "h", "sigma_8", "n_scalar")])
    hc.hc_epoch(_p(c), ctypes.c_double(z), ctypes.c_double(sigma_norm), buf)
    return buf


@pytest.mark.parametrize("z", [0.0, 0.7, 1.5])
def test_background_and_linear_power(hc, z):
    e = o.epoch(None, z)
    buf = _epoch(hc, o.default_cosmo_dict, z, e.sigma_norm)
    sc = numpy.empty(7)
    hc.hc_scalars(buf, _p(sc))
    ref = [e.growth, o.omega_m(e), o.omega_l(e), o.delta_c(e), o.delta_v(e),
           o.rho_bar(e), e.delta_H]
    assert numpy.allclose(sc, ref, rtol=1e-14, atol=0)
    k = numpy.logspace(-5, 4, 500)
    out = numpy.empty_like(k)
    hc.hc_linear_power(buf, _p(k), k.size, _p(out))
    assert rel_err(out, o.linear_power(e, k)) < 2e-13
    # Stage E's two-division arrangement with the short logarithm
    hc.hc_power_shape(buf, _p(k), k.size, _p(out))
    assert rel_err(out, o.linear_power(e, k)) < 2e-13
    for R in (0.05, 8.0, 120.0):
        lo, hi = o.sigma_limits(e, R)
        lnk = numpy.linspace(numpy.log(lo), numpy.log(hi), 999)
        got = numpy.empty_like(lnk)
        hc.hc_sigma_integrand(buf, ctypes.c_double(R), _p(lnk), lnk.size, _p(got))
        ref = o._sigma_integrand(lnk, e, R) / (2 * numpy.pi ** 2)
        assert numpy.max(numpy.abs(got - ref)) < 1e-9 * numpy.max(ref)


def test_bao_transfer_function(hc):
    """cosmology.py:474-538 against the oracle (itself equal to the reference, G11)."""
    z = 0.3
    e = o.epoch(None, z, with_bao=True)
    n = hc.hc_sizeof_epoch()
    buf = (ctypes.c_char * n)()
    cd = o.default_cosmo_dict
    c = numpy.array([cd[k] for k in ("omega_m0", "omega_b0", "omega_l0", "omega_r0",
                                     "cmb_temp", "h", "sigma_8", "n_scalar")])
    hc.hc_epoch_bao(_p(c), ctypes.c_double(z), ctypes.c_double(e.sigma_norm), buf)
    k = numpy.logspace(-5, 4, 700)
    out = numpy.empty_like(k)
    hc.hc_transfer(buf, _p(k), k.size, _p(out))
    assert rel_err(out, o.eh_bao_transfer(e, k)) < 5e-13
    hc.hc_linear_power(buf, _p(k), k.size, _p(out))
    assert rel_err(out, o.linear_power(e, k)) < 1e-12
    hc.hc_power_shape(buf, _p(k), k.size, _p(out))
    assert rel_err(out, o.linear_power(e, k)) < 1e-12
    assert rel_err(o.eh_bao_transfer(e, k), o.eh_transfer(e, k)) > 0.05      # wiggles are there


def test_y_nfw_mass_function_hod(hc):
    z = 0.5
    e = o.epoch(None, z)
    m = o.mass_table(e)
    t = o.halo_table(e, m, families=())
    buf = _epoch(hc, o.default_cosmo_dict, z, e.sigma_norm)
    lnm = numpy.repeat(numpy.linspace(m.ln_mass_min, m.ln_mass_max, 60), 8)
    lnk = numpy.tile(numpy.linspace(numpy.log(1e-3), numpy.log(1e2), 8), 60)
    out = numpy.empty_like(lnm)
    hd = o.default_halo_dict
    hc.hc_y_nfw(buf, ctypes.c_double(hd["c0"]), ctypes.c_double(hd["beta"]),
                ctypes.c_double(hd["delta_v"]), ctypes.c_double(m.m_star),
                _p(lnk), _p(lnm), lnm.size, _p(out))
    ref = numpy.array([o.y_nfw(t, a, numpy.exp(b)) for a, b in zip(lnk, lnm)])
    assert numpy.max(numpy.abs(out - ref)) < 5e-13
    nu = numpy.logspace(numpy.log10(m.nu_min), numpy.log10(m.nu_max), 200)
    f, b = numpy.empty_like(nu), numpy.empty_like(nu)
    par = numpy.array([m.stq, m.st_a, m.f_norm, m.bias_norm])
    hc.hc_mass_function(buf, 0, _p(par), _p(nu), nu.size, _p(f), _p(b))
    assert rel_err(f, o.f_nu(m, nu)) < 1e-13
    assert rel_err(b, o.bias_nu(m, nu)) < 1e-13
    mt = o.mass_table(e, kind="tinker")
    par = numpy.array([mt.delta_v, mt.t_alpha, mt.t_beta, mt.t_gamma, mt.t_phi,
                       mt.t_eta, mt.bias_norm])
    hc.hc_mass_function(buf, 1, _p(par), _p(nu), nu.size, _p(f), _p(b))
    assert rel_err(f, o.f_nu(mt, nu)) < 1e-13
    assert rel_err(b, o.bias_nu(mt, nu)) < 1e-13
    h = o.zheng()
    mass = numpy.logspace(9, 16, 300)
    n1, n2 = numpy.empty_like(mass), numpy.empty_like(mass)
    hod = numpy.array([h.log_M_min, h.sigma, h.log_M_0, h.log_M_1p, h.alpha])
    hc.hc_zheng(buf, _p(hod), _p(mass), mass.size, _p(n1), _p(n2))
    assert numpy.allclose(n1, o.zheng_first(h, mass), rtol=1e-12, atol=1e-300)
    assert numpy.allclose(n2, o.zheng_second(h, mass), rtol=1e-12, atol=1e-300)
