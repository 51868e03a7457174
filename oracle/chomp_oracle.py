"""NumPy/SciPy restatement of the CHOMP hot path (TEST INFRASTRUCTURE ONLY).

Layout: plain functions over small table objects instead of the reference's
class hierarchy, but the *algorithm* is the reference's, quirks included:

  epoch(cosmo, z)                 cosmology.py:39-119   SingleEpoch
  linear_power / delta_k          cosmology.py:449-472, 574-600
  sigma_r / nu_m                  cosmology.py:602-699
  mass_table(...)                 mass_function.py:38-61, 160-255, 436-564
  zheng_* (HOD)                   hod.py:141-230
  halo_table(...)                 halo.py:41-104, 674-707, 839-1086
  halo_power(...)                 halo.py:266-439, 649-672
  halofit_table / halofit_power   halo.py:1236-1412
  multi_epoch(...)                cosmology.py:747-817, 862-953
  dndz_*                          kernel.py:26-179
  window_table(...)               kernel.py:211-387, 410-484
  kernel_table / kernel_eval      kernel.py:559-729, 784-839
  wtheta / cell                   correlation.py:65-123, 242-275, 320-392

Third-party arithmetic used exactly as the reference uses it: SciPy's
``InterpolatedUnivariateSpline`` (FITPACK), ``special.sici/j0/jn/jn_zeros/erf/
erfinv``; ``integrate.romberg`` (removed from SciPy 1.15) is restated in
oracle/romberg.py.

Python-2 semantics the reference's pinned values depend on are explicit here:
``(Omb2)**(3/4) == 1`` (cosmology.py:464) and ``1/b`` floor division for an int
``b`` (kernel.py:167).
"""
import copy
import warnings

import numpy
from scipy import special
from scipy.interpolate import InterpolatedUnivariateSpline

from .romberg import romberg, AccuracyWarning  # noqa: F401

# ---------------------------------------------------------------------------
# Parameter dictionaries: the values of /root/reference/defaults.py:6-92 (data).
# ---------------------------------------------------------------------------
default_cosmo_dict = {
    "omega_m0": 0.278 - 4.15e-5 / 0.7 ** 2, "omega_b0": 0.046,
    "omega_l0": 0.722, "omega_r0": 4.15e-5 / 0.7 ** 2, "cmb_temp": 2.726,
    "h": 0.7, "sigma_8": 0.811, "n_scalar": 0.960, "w0": -1.0, "wa": 0.0}
default_halo_dict = {"stq": 0.3, "st_little_a": 0.707, "c0": 9.0,
                     "beta": -0.13, "alpha": -1, "delta_v": -1.0}
default_hod_dict = {"log_M_min": 12.14, "sigma": 0.15, "log_M_0": 12.14,
                    "log_M_1p": 13.43, "alpha": 1.0}
default_limits = {"k_min": 0.001, "k_max": 100.0, "mass_min": -1,
                  "mass_max": -1}
default_precision = {
    "corr_npoints": 50, "corr_precision": 1.48e-6,
    "cosmo_npoints": 50, "cosmo_precision": 1.48e-8,
    "dNdz_precision": 1.48e-8,
    "halo_npoints": 50, "halo_precision": 1.48e-5, "halo_limit": 100,
    "kernel_npoints": 50, "kernel_precision": 1.48e-6, "kernel_limit": 100,
    "kernel_bessel_limit": 8,
    "mass_npoints": 50, "mass_precision": 1.48e-8,
    "window_npoints": 100, "window_precision": 1.48e-6,
    "global_precision": 1.48e-32, "divmax": 20}


class Table(object):
    """Attribute bag for tabulated state."""
    def __init__(self, **kw):
        self.__dict__.update(kw)


# How often scipy.integrate.romberg would have raised its AccuracyWarning ("divmax exceeded")
# since the counter was last cleared: the tests of the status word compare it with the
# device's CHOMP_ST_*_DIVMAX bits.
DIVMAX_EXCEEDED = [0]


def _rom(f, a, b, rtol, prec, args=()):
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always", AccuracyWarning)
        out = romberg(f, a, b, args=args, vec_func=True,
                      tol=prec["global_precision"], rtol=rtol,
                      divmax=prec["divmax"])
    DIVMAX_EXCEEDED[0] += sum(1 for w in rec if issubclass(w.category, AccuracyWarning))
    return out


# ---------------------------------------------------------------------------
# L1: single-epoch cosmology (cosmology.py:25-728)
# ---------------------------------------------------------------------------
def epoch(cosmo_dict=None, redshift=0.0, limits=None, prec=None, with_bao=False):
    """cosmology.py:39-119.  with_bao: the E&H transfer function with wiggles (:474-538)."""
    cd = default_cosmo_dict if cosmo_dict is None else cosmo_dict
    lim = default_limits if limits is None else limits
    prec = default_precision if prec is None else prec
    if cd["w0"] != -1.0 or cd["wa"] != 0.0:
        raise NotImplementedError("w(z) != -1 is outside the hot-path scope")
    if redshift < 0.0:
        redshift = 0.0
    e = Table(cosmo_dict=cd, z=redshift, prec=prec, limits=lim, with_bao=bool(with_bao))
    e.om0, e.ob0, e.ol0, e.or0 = (cd["omega_m0"], cd["omega_b0"],
                                  cd["omega_l0"], cd["omega_r0"])
    e.tcmb, e.h, e.sigma_8, e.n = (cd["cmb_temp"], cd["h"], cd["sigma_8"],
                                   cd["n_scalar"])
    e.H0 = 100.0 / (2.998 * 10 ** 5)
    p = prec["cosmo_precision"]
    tot = e.om0 + e.ol0 + e.or0
    e.flat = (tot <= 1.0 + p) and (tot >= 1.0 - p)        # :65-71
    e.open = tot <= 1.0 - p                                # :72-75
    e.closed = tot > 1.0 + p                               # :76-79
    e.k_min, e.k_max = lim["k_min"], lim["k_max"]
    e.delta_H = (1.94e-5 * e.om0 ** (-0.785 - 0.05 * numpy.log(e.om0)) *
                 numpy.exp(-0.95 * (e.n - 1) - 0.169 * (e.n - 1) ** 2))  # :83-85
    e.chi = _rom(lambda zz: E(e, zz), 0.0, e.z, prec["cosmo_precision"], prec)
    e.growth_norm = growth_approx(e, 1.0)                  # :112, 326
    e.growth = growth_approx(e, 1.0 / (1.0 + e.z)) / e.growth_norm
    e.sigma_norm = 1.0
    e.sigma_norm = e.sigma_8 * e.growth / sigma_r(e, 8.0)  # :118-119
    return e


def E0(e, z):
    """cosmology.py:165-178 (w = -1 branch)."""
    a = 1.0 / (1.0 + z)
    return e.ol0 + e.om0 / (a * a * a) + e.or0 / (a * a * a * a)


def E(e, z):
    """cosmology.py:153-163: c/H(z) in Mpc/h."""
    return 1.0 / (e.H0 * numpy.sqrt(E0(e, z)))


def growth_approx(e, a):
    """cosmology.py:215-231 -- what growth_factor_eval always returns (:326);
    note ``Omega_m * (4./7.)`` (a product, not a power)."""
    om = e.om0 / a ** 3
    denom = e.ol0 + om
    Omega_m = om / denom
    Omega_L = e.ol0 / denom
    coeff = 5. * Omega_m / (2. / a)
    term1 = Omega_m * (4. / 7.)
    term3 = (1. + 0.5 * Omega_m) * (1. + Omega_L / 70.)
    return coeff / (term1 - Omega_L + term3)


def omega_m(e):
    return e.om0 * (1.0 + e.z) ** 3 / E0(e, e.z)           # :375-382


def omega_l(e):
    return e.ol0 / E0(e, e.z)                               # :384-391


def delta_c(e):
    """cosmology.py:393-407."""
    d = 0.15 * (12.0 * numpy.pi) ** (2.0 / 3.0)
    if e.open:
        d *= omega_m(e) ** 0.0185
    if e.flat and e.om0 < 1.0001:
        d *= omega_m(e) ** 0.0055
    return d


def delta_v(e):
    """cosmology.py:409-423."""
    d = 178.0
    if e.open:
        d /= omega_m(e) ** 0.7
    if e.flat and e.om0 < 1.0001:
        d /= omega_m(e) ** 0.55
    return d / e.growth


def rho_crit(e):
    return 1.879 / (1.989) * 3.086 ** 3 * 1e10 * E0(e, e.z)   # :437-438


def rho_bar(e):
    return rho_crit(e) * omega_m(e)                          # :440-447


def eh_bao_transfer(e, k):
    """cosmology.py:474-538: Eisenstein & Hu (1998) with baryon wiggles, as the reference
    writes it (the second b1, b2 belong to beta_c; T0t's denominator takes q(k))."""
    k = numpy.asarray(k, dtype=float)
    theta = e.tcmb / 2.7
    Ob, Om = e.ob0, e.om0
    Oc, O, h = Om - Ob, Om, e.h
    Obh2, Oh2, ObO = Ob * h ** 2, O * h ** 2, Ob / O
    zeq = 2.5e4 * Oh2 * theta ** (-4)
    keq = 7.46e-2 * Oh2 * theta ** (-2)
    b1 = 0.313 * Oh2 ** (-0.419) * (1. + 0.607 * Oh2 ** 0.674)
    b2 = 0.238 * Oh2 ** 0.223
    zd = 1291. * (Oh2 ** 0.251 / (1. + 0.659 * Oh2 ** 0.828)) * (1. + b1 * Obh2 ** b2)
    R = lambda z: 31.5 * Obh2 * theta ** (-4) * (1000. / z)
    Req, Rd = R(zeq), R(zd)
    s = (2. / (3. * keq)) * numpy.sqrt(6. / Req) * numpy.log(
        (numpy.sqrt(1. + Rd) + numpy.sqrt(Rd + Req)) / (1. + numpy.sqrt(Req)))
    ks = k * h * s
    kSilk = 1.6 * Obh2 ** 0.52 * Oh2 ** 0.73 * (1. + (10.4 * Oh2) ** (-0.95))
    q = k * h / (13.41 * keq)
    G = lambda y: y * (-6. * numpy.sqrt(1. + y) + (2 + 3 * y) * numpy.log(
        (numpy.sqrt(1. + y) + 1.) / (numpy.sqrt(1. + y) - 1.)))
    alpha_b = 2.07 * keq * s * (1. + Rd) ** (-3. / 4.) * G((1. + zeq) / (1. + zd))
    beta_b = 0.5 + ObO + (3. - 2. * ObO) * numpy.sqrt((17.2 * Oh2) ** 2 + 1.)
    C = lambda a: (14.2 / a) + 386. / (1. + 69.9 * q ** 1.08)
    T0t = lambda a, b: numpy.log(numpy.e + 1.8 * b * q) / (
        numpy.log(numpy.e + 1.8 * b * q) + C(a) * q ** 2)
    a1 = (46.9 * Oh2) ** 0.670 * (1. + (32.1 * Oh2) ** (-0.532))
    a2 = (12. * Oh2) ** 0.424 * (1. + (45. * Oh2) ** (-0.582))
    alpha_c = a1 ** (-ObO) * a2 ** (-ObO ** 3)
    b1 = 0.944 * (1. + (458. * Oh2) ** (-0.708)) ** (-1)
    b2 = (0.395 * Oh2) ** (-0.0266)
    beta_c = 1. / (1. + b1 * ((Oc / O) ** b2 - 1))
    f = 1. / (1. + (ks / 5.4) ** 4)
    Tc = f * T0t(1, beta_c) + (1. - f) * T0t(alpha_c, beta_c)
    beta_node = 8.41 * (Oh2 ** 0.435)
    stilde = s / (1. + (beta_node / ks) ** 3) ** (1. / 3.)
    Tb1 = T0t(1., 1.) / (1. + (ks / 5.2) ** 2)
    Tb2 = (alpha_b / (1. + (beta_b / ks) ** 3)) * numpy.exp(-(k * h / kSilk) ** 1.4)
    Tb = numpy.sinc(k * stilde / numpy.pi) * (Tb1 + Tb2)
    return ObO * Tb + (Oc / O) * Tc


def transfer_function(e, k):
    """cosmology.py:556-572."""
    return eh_bao_transfer(e, k) if getattr(e, "with_bao", False) else eh_transfer(e, k)


def eh_transfer(e, k):
    """cosmology.py:449-472, with the reference's deviations from EH98 kept:
    (Omb2)**(3/4) -> **0 under Py2, (1+0.43ks)**4, q = k*theta/Gamma."""
    theta = e.tcmb / 2.7
    Omh2 = e.om0 * e.h ** 2
    Omb2 = e.ob0 * e.h ** 2
    omega_ratio = e.ob0 / e.om0
    s = 44.5 * numpy.log(9.83 / Omh2) / numpy.sqrt(1 + 10.0 * (Omb2) ** 0)
    alpha = (1 - 0.328 * numpy.log(431.0 * Omh2) * omega_ratio +
             0.38 * numpy.log(22.3 * Omh2) * omega_ratio ** 2)
    Gamma_eff = e.om0 * e.h * (alpha + (1 - alpha) / (1 + 0.43 * k * s) ** 4)
    q = k * theta / Gamma_eff
    L0 = numpy.log(2 * numpy.e + 1.8 * q)
    C0 = 14.2 + 731.0 / (1 + 62.5 * q)
    return L0 / (L0 + C0 * q * q)


def delta_k(e, k):
    """cosmology.py:574-587."""
    d = (e.delta_H ** 2 * (k / e.H0) ** (3 + e.n) * transfer_function(e, k) ** 2) / e.h
    return d * (e.growth * e.growth * e.sigma_norm * e.sigma_norm)


def linear_power(e, k):
    """cosmology.py:589-600."""
    k = numpy.asarray(k, dtype=float)
    with numpy.errstate(all="ignore"):
        return numpy.where(k > 1e-16,
                           2.0 * numpy.pi * numpy.pi * delta_k(e, k) / (k * k * k),
                           1e-16)


def _sigma_integrand(ln_k, e, scale):
    """cosmology.py:644-660."""
    k = numpy.exp(ln_k)
    dk = 1.0 * k
    kR = scale * k
    W = 3.0 * (numpy.sin(kR) / kR ** 3 - numpy.cos(kR) / kR ** 2)
    return dk * linear_power(e, k) * W * W * k * k


def sigma_limits(e, scale):
    """cosmology.py:611-632: the k-range adapts to the scale."""
    k_min, k_max = e.k_min, e.k_max
    needed_k_min = 1.0 / scale / 10.0
    needed_k_max = 1.0 / scale * 14.0662
    if needed_k_min <= k_min and needed_k_min > e.k_min / 100.0:
        k_min = needed_k_min
    elif needed_k_min <= k_min and needed_k_min <= e.k_min / 100.0:
        k_min = e.k_min / 100.0
    if needed_k_max >= k_max and needed_k_max < e.k_max * 100.0:
        k_max = needed_k_max
    elif needed_k_max >= k_max and needed_k_max >= e.k_max * 100.0:
        k_max = e.k_max * 100.0
    return k_min, k_max


def sigma_r(e, scale):
    """cosmology.py:602-642."""
    k_min, k_max = sigma_limits(e, scale)
    s2 = _rom(_sigma_integrand, numpy.log(k_min), numpy.log(k_max),
              e.prec["cosmo_precision"], e.prec, args=(e, scale))
    return numpy.sqrt(s2 / (2.0 * numpy.pi * numpy.pi))


def sigma_m(e, mass):
    scale = (3.0 * mass / (4.0 * numpy.pi * rho_bar(e))) ** (1.0 / 3.0)   # :671
    return sigma_r(e, scale)


def nu_m(e, mass):
    sq = delta_c(e) / sigma_m(e, mass)                                    # :698
    return sq * sq


# ---------------------------------------------------------------------------
# L2: mass function (mass_function.py) and HOD (hod.py)
# ---------------------------------------------------------------------------
_TINKER = dict(
    delta=[200, 300, 400, 600, 800, 1200, 1600, 2400, 3200],
    alpha=[0.368, 0.363, 0.385, 0.389, 0.393, 0.365, 0.379, 0.355, 0.327],
    beta=[0.589, 0.585, 0.544, 0.543, 0.564, 0.632, 0.637, 0.673, 0.702],
    gamma=[0.864, 0.922, 0.987, 1.09, 1.20, 1.34, 1.50, 1.68, 1.81],
    phi=[-0.729, -0.789, -0.910, -1.05, -1.20, -1.26, -1.45, -1.50, -1.49],
    eta=[-0.243, -0.261, -0.261, -0.273, -0.278, -0.301, -0.301, -0.319,
         -0.336])   # Tinker et al. 2010 table 4, as in mass_function.py:450-459


def mass_limits(e):
    """mass_function.py:160-203: multiplicative 5 % walk until nu(M) is in band."""
    lim, prec = e.limits, e.prec
    if lim["mass_min"] > 0 and lim["mass_max"] > 0:
        return numpy.log(lim["mass_min"]), numpy.log(lim["mass_max"]), 0
    mass_min, mass_max = 1.0e9, 1.0e16
    n_eval = 0
    while True:
        n_eval += 1
        nu = nu_m(e, mass_min)
        if 0.1 * (1.0 + 0.05) < nu:
            mass_min = mass_min / 1.05
            continue
        elif 0.1 * (1.0 - 0.05) > nu:
            mass_min = mass_min * 1.05
            continue
        n_eval += 1
        nu = nu_m(e, mass_max)
        if 50.0 * (1.0 - 0.05) > nu:
            mass_max = mass_max * 1.05
            continue
        elif 50.0 * (1.0 + 0.05) < nu:
            mass_max = mass_max / 1.05
            continue
        break
    return numpy.log(mass_min), numpy.log(mass_max), n_eval


def mass_table(e, halo_dict=None, kind="st"):
    """mass_function.py:38-61 (+ Tinker 448-492): limits, nu table, splines,
    normalisations.  ``kind`` is 'st' (Sheth-Tormen) or 'tinker'."""
    hd = default_halo_dict if halo_dict is None else halo_dict
    prec = e.prec
    m = Table(kind=kind, e=e, halo_dict=hd)
    m.delta_c = delta_c(e)
    m.delta_v = hd["delta_v"]
    if m.delta_v == -1:
        m.delta_v = delta_v(e)
    m.stq, m.st_a = hd["stq"], hd["st_little_a"]
    m.ln_mass_min, m.ln_mass_max, m.n_search = mass_limits(e)
    m.ln_mass = numpy.linspace(m.ln_mass_min, m.ln_mass_max,
                               prec["mass_npoints"])
    m.nu_arr = numpy.array([nu_m(e, numpy.exp(x)) for x in m.ln_mass])  # :205-210
    m.nu_min = 1.001 * m.nu_arr[0]
    m.nu_max = 0.999 * m.nu_arr[-1]
    m.nu_spline = InterpolatedUnivariateSpline(m.ln_mass, m.nu_arr)
    m.ln_mass_spline = InterpolatedUnivariateSpline(m.nu_arr, m.ln_mass)
    m.m_star = float(numpy.exp(m.ln_mass_spline(1.0)))
    if kind == "tinker":
        lnD = numpy.log(_TINKER["delta"])
        lnd = numpy.log(m.delta_v)
        z = e.z
        m.t_alpha = float(InterpolatedUnivariateSpline(lnD, _TINKER["alpha"])(lnd))
        m.t_beta = float(InterpolatedUnivariateSpline(lnD, _TINKER["beta"])(lnd)) * numpy.power(1 + z, 0.20)
        m.t_phi = float(InterpolatedUnivariateSpline(lnD, _TINKER["phi"])(lnd)) * numpy.power(1 + z, -0.08)
        m.t_eta = float(InterpolatedUnivariateSpline(lnD, _TINKER["eta"])(lnd)) * numpy.power(1 + z, 0.27)
        m.t_gamma = float(InterpolatedUnivariateSpline(lnD, _TINKER["gamma"])(lnd)) * numpy.power(1 + z, -0.01)
    mass_normalize(m)
    return m


def mass_normalize(m):
    """mass_function.py:225-241 (ST) / 532-545 (Tinker: bias only)."""
    prec = m.e.prec
    m.f_norm = 1.0
    if m.kind == "st":
        norm = _rom(lambda x: f_nu(m, x), m.nu_min, m.nu_max,
                    prec["mass_precision"], prec)
        m.f_norm = 1.0 / norm
    m.bias_norm = 1.0
    norm = _rom(lambda x: f_nu(m, x) * bias_nu(m, x), m.nu_min, m.nu_max,
                prec["mass_precision"], prec)
    m.bias_norm = 1.0 / norm


def f_nu(m, nu):
    if m.kind == "st":                                        # :243-255
        nu_prime = nu * m.st_a
        return (m.f_norm * (1.0 + nu_prime ** (-1.0 * m.stq)) *
                numpy.sqrt(nu_prime) * numpy.exp(-0.5 * nu_prime) / nu)
    sqrtnu = numpy.sqrt(nu)                                   # :494-509
    return (m.t_alpha * (1 + numpy.power(m.t_beta * sqrtnu, -2 * m.t_phi)) *
            numpy.power(nu, m.t_eta) * numpy.exp(-m.t_gamma * nu / 2.0) / sqrtnu)


def bias_nu(m, nu):
    if m.kind == "st":                                        # :290-302
        nu_prime = nu * m.st_a
        return m.bias_norm * (
            1.0 + (nu_prime - 1.0) / m.delta_c +
            2.0 * m.stq / (m.delta_c * (1.0 + nu_prime ** m.stq)))
    sqrtnu = numpy.sqrt(nu)                                   # :511-530
    y = numpy.log10(m.delta_v)
    A = 1 + 0.24 * y * numpy.exp(-(4.0 / y) ** 4)
    a = 0.44 * y - 0.88
    B, b = 0.183, 1.5
    C = 0.019 + 0.107 * y + 0.19 * numpy.exp(-(4.0 / y) ** 4)
    c = 2.4
    return m.bias_norm * (1 - A * sqrtnu ** a / (sqrtnu ** a + m.delta_c ** a) +
                          B * sqrtnu ** b + C * sqrtnu ** c)


def mass_of_nu(m, nu):
    return numpy.exp(m.ln_mass_spline(nu))                    # :337-346


def nu_of_mass(m, mass):
    return m.nu_spline(numpy.log(mass))                       # :315-324


def zheng(hod_dict=None, prec=None):
    """hod.py:156-186 (including the ``secon_moment_zero`` typo: the clamp of
    second_moment_zero never applies)."""
    hd = default_hod_dict if hod_dict is None else hod_dict
    prec = default_precision if prec is None else prec
    h = Table(**hd)
    h.first_moment_zero = numpy.power(
        10.0, h.log_M_min + h.sigma * special.erfinv(
            2. * prec['halo_precision'] - 1.0))
    h.second_moment_zero = 10.0 ** h.log_M_0
    h.safe_norm = 10.0 ** (h.log_M_min + 1.0 * h.sigma)
    return h


def zheng_central(h, mass):
    if h.sigma <= 0.0:                                         # :209-212
        return numpy.where(numpy.log10(mass) > h.log_M_min, 1.0, 0.0)
    return 0.5 * (1 + special.erf((numpy.log10(mass) - h.log_M_min) / h.sigma))


def zheng_satellite(h, mass):
    diff = mass - numpy.power(10, h.log_M_0)                   # :226-230
    with numpy.errstate(all="ignore"):
        return numpy.where(diff > 0.0,
                           zheng_central(h, mass) *
                           numpy.power(diff / (10 ** h.log_M_1p), h.alpha),
                           0.0)


def zheng_first(h, mass):
    return zheng_central(h, mass) + zheng_satellite(h, mass)   # :189-191


def zheng_second(h, mass):
    n_sat = zheng_satellite(h, mass)                           # :193-195
    return (2 + n_sat) * n_sat


# ---------------------------------------------------------------------------
# L3: halo model (halo.py)
# ---------------------------------------------------------------------------
def halo_table(e=None, m=None, hod=None, halo_dict=None, families=("mm",), exclusion=False):
    """halo.py:41-104 + the lazy initialisers.  ``families`` selects which of
    the 1-halo/2-halo knot tables to build: 'mm' (h_m, pp_mm), 'gm' (h_m, h_g,
    pp_gm), 'gg' (h_g, pp_gg).  ``exclusion``: HaloExclusion (halo.py:1201-1233), whose
    two 2-halo integrands carry the halo-exclusion mass window."""
    e = epoch() if e is None else e
    hd = default_halo_dict if halo_dict is None else halo_dict
    m = mass_table(e, hd) if m is None else m
    hod = zheng(prec=e.prec) if hod is None else hod
    prec = e.prec
    t = Table(e=e, m=m, hod=hod, halo_dict=hd, exclusion=bool(exclusion))
    t.k_min, t.k_max = e.limits["k_min"], e.limits["k_max"]
    t.ln_k = numpy.linspace(numpy.log(t.k_min), numpy.log(t.k_max),
                            prec["halo_npoints"])             # :52-54
    t.c0 = hd["c0"] / (1.0 + e.z)                             # :71
    t.beta = hd["beta"]
    t.delta_v = hd["delta_v"]
    if t.delta_v == -1:
        t.delta_v = delta_v(e)
    t.rho_bar = rho_bar(e)
    # halo splines (:839-855): ln r_v and ln c against ln M
    masses = numpy.exp(m.ln_mass)
    t.ln_r_v_spline = InterpolatedUnivariateSpline(
        m.ln_mass, numpy.log((3.0 * masses / (4.0 * numpy.pi * t.delta_v *
                                              t.rho_bar)) ** (1.0 / 3.0)))
    t.ln_c_spline = InterpolatedUnivariateSpline(
        m.ln_mass, numpy.log(t.c0 * (masses / m.m_star) ** t.beta))
    _n_bar(t)
    fam = set(families)
    if "mm" in fam or "gm" in fam:
        t.h_m = _knots(t, _h_m_integrand, numpy.log(m.nu_min), None)
        t.h_m_spline = InterpolatedUnivariateSpline(t.ln_k, t.h_m)
    if "mm" in fam:
        t.pp_mm = _knots(t, _pp_mm_integrand, numpy.log(m.nu_min), None) / t.rho_bar
        t.pp_mm_spline = InterpolatedUnivariateSpline(t.ln_k, t.pp_mm)
    if "gm" in fam or "gg" in fam:
        t.h_g = _knots(t, _h_g_integrand, numpy.log(_nu_lo(t, hod.first_moment_zero)),
                       hod.safe_norm) / t.n_bar_over_rho_bar
        t.h_g_spline = InterpolatedUnivariateSpline(t.ln_k, t.h_g)
    if "gm" in fam:
        t.pp_gm = _knots(t, _pp_gm_integrand, numpy.log(_nu_lo(t, hod.first_moment_zero)),
                         hod.safe_norm) / t.n_bar
        t.pp_gm_spline = InterpolatedUnivariateSpline(t.ln_k, t.pp_gm)
    if "gg" in fam:
        t.pp_gg = (_knots(t, _pp_gg_integrand, numpy.log(_nu_lo(t, hod.second_moment_zero)),
                          hod.safe_norm) * t.rho_bar / (t.n_bar * t.n_bar))
        t.pp_gg_spline = InterpolatedUnivariateSpline(t.ln_k, t.pp_gg)
    return t


def _nu_lo(t, moment_zero):
    """halo.py:935-939 / 1002-1006: lower integration limit hint from the HOD."""
    nu_min = t.m.nu_min
    if moment_zero > -1 and moment_zero > numpy.exp(t.m.ln_mass_min):
        nu_min = float(nu_of_mass(t.m, moment_zero))
    return nu_min


def y_nfw(t, ln_k, mass):
    """halo.py:561-585."""
    k = numpy.exp(ln_k)
    lm = numpy.log(mass)
    con = numpy.exp(t.ln_c_spline(lm))
    con_plus = 1.0 + con
    z = k * numpy.exp(t.ln_r_v_spline(lm)) / con
    si_z, ci_z = special.sici(z)
    si_cz, ci_cz = special.sici(con_plus * z)
    rho_km = (numpy.cos(z) * (ci_cz - ci_z) + numpy.sin(z) * (si_cz - si_z) -
              numpy.sin(con * z) / (con_plus * z))
    mass_k = numpy.log(con_plus) - con / con_plus
    return rho_km / mass_k


def _mass_window(t, mass, ln_k):
    """HaloExclusion._mass_window, halo.py:1223-1233: transform of the window that cuts
    halos within two virial radii of each other (1 for the plain Halo)."""
    if not getattr(t, "exclusion", False):
        return 1.0
    k = numpy.exp(ln_k)
    kR = k * 2.0 * numpy.exp(t.ln_r_v_spline(numpy.log(mass)))
    return ((kR * numpy.cos(kR) + kR * kR * kR * special.sici(kR)[1] +
             (2 - kR * kR) * numpy.sin(kR)) / (3.0 * kR))


def _h_m_integrand(ln_nu, t, ln_k, norm):                       # :922-927, 1208-1213
    nu = numpy.exp(ln_nu)
    mass = mass_of_nu(t.m, nu)
    return (norm * nu * _mass_window(t, mass, ln_k) * f_nu(t.m, nu) * bias_nu(t.m, nu) *
            y_nfw(t, ln_k, mass))


def _pp_mm_integrand(ln_nu, t, ln_k, norm):                     # :989-994
    nu = numpy.exp(ln_nu)
    mass = mass_of_nu(t.m, nu)
    y = y_nfw(t, ln_k, mass)
    return norm * nu * f_nu(t.m, nu) * mass * y * y


def _h_g_integrand(ln_nu, t, ln_k, norm):                       # :964-969, 1215-1221
    nu = numpy.exp(ln_nu)
    mass = mass_of_nu(t.m, nu)
    return (norm * nu * _mass_window(t, mass, ln_k) * f_nu(t.m, nu) * bias_nu(t.m, nu) *
            y_nfw(t, ln_k, mass) * zheng_first(t.hod, mass) / mass)


def _pp_gm_integrand(ln_nu, t, ln_k, norm):                     # :1078-1086
    nu = numpy.exp(ln_nu)
    mass = mass_of_nu(t.m, nu)
    y = y_nfw(t, ln_k, mass)
    n_exp = zheng_first(t.hod, mass)
    return numpy.where(n_exp < 1, norm * nu * f_nu(t.m, nu) * n_exp * y,
                       norm * nu * f_nu(t.m, nu) * n_exp * y * y)


def _pp_gg_integrand(ln_nu, t, ln_k, norm):                     # :1032-1041
    nu = numpy.exp(ln_nu)
    mass = mass_of_nu(t.m, nu)
    y = y_nfw(t, ln_k, mass)
    n_pair = zheng_second(t.hod, mass)
    return numpy.where(n_pair < 1,
                       norm * nu * f_nu(t.m, nu) * n_pair * y / mass,
                       norm * nu * f_nu(t.m, nu) * n_pair * y * y / mass)


def _nbar_integrand(ln_nu, t, norm):                            # :702-707
    nu = numpy.exp(ln_nu)
    mass = mass_of_nu(t.m, nu)
    return norm * nu * zheng_first(t.hod, mass) * f_nu(t.m, nu) / mass


def _n_bar(t):
    """halo.py:674-700."""
    m, hod, prec = t.m, t.hod, t.e.prec
    nu_min = _nu_lo(t, hod.first_moment_zero)
    norm = 1.0
    if (hod.safe_norm != -1 and hod.safe_norm > numpy.exp(m.ln_mass_min) and
            hod.safe_norm < numpy.exp(m.ln_mass_max)):
        inv = _nbar_integrand(numpy.log(nu_of_mass(m, hod.safe_norm)), t, 1.0)
        norm = 1.0 / inv if inv > 1e-16 else 1.0
    t.n_bar_over_rho_bar = _rom(_nbar_integrand, numpy.log(nu_min),
                                numpy.log(m.nu_max), prec["halo_precision"],
                                prec, args=(t, norm)) / norm
    t.n_bar = t.n_bar_over_rho_bar * t.rho_bar


def hod_stats(t):
    """Halo.calculate_bias / calculate_m_eff / calculate_f_sat (halo.py:709-838): HOD-weighted
    integrals over ln nu divided by n_bar / rho_bar.  Returns (bias, m_eff, f_sat)."""
    m, hod, prec = t.m, t.hod, t.e.prec

    def stat(integrand, zero, norm_inside_only):
        nu_min = _nu_lo(t, zero)
        norm = 1.0
        ok = hod.safe_norm != -1
        if norm_inside_only:
            ok = ok and (hod.safe_norm > numpy.exp(m.ln_mass_min) and
                         hod.safe_norm < numpy.exp(m.ln_mass_max))
        if ok:
            inv = integrand(numpy.log(nu_of_mass(m, hod.safe_norm)), 1.0)
            norm = 1.0 / inv if inv > 1e-16 else 1.0
        val = _rom(integrand, numpy.log(nu_min), numpy.log(m.nu_max), prec["halo_precision"],
                   prec, args=(norm,))
        return val / (norm * t.n_bar_over_rho_bar)

    def bias_i(ln_nu, norm):                                     # :745-750
        nu = numpy.exp(ln_nu)
        mass = mass_of_nu(m, nu)
        return norm * nu * zheng_first(hod, mass) * f_nu(m, nu) * bias_nu(m, nu) / mass

    def m_eff_i(ln_nu, norm):                                    # :786-790
        nu = numpy.exp(ln_nu)
        mass = mass_of_nu(m, nu)
        return norm * nu * zheng_first(hod, mass) * f_nu(m, nu)

    def f_sat_i(ln_nu, norm):                                    # :833-838
        nu = numpy.exp(ln_nu)
        mass = mass_of_nu(m, nu)
        return norm * nu * zheng_satellite(hod, mass) * f_nu(m, nu) / mass

    return (stat(bias_i, hod.first_moment_zero, True), stat(m_eff_i, hod.first_moment_zero, True),
            stat(f_sat_i, hod.second_moment_zero, False))


def _knots(t, integrand, ln_nu_lo, safe_norm):
    """The 50-knot loops of halo.py:904-927, 929-969, 971-994, 996-1041,
    1043-1086.  ``safe_norm`` None -> normalise by the integrand at ln nu = 0
    (h_m, pp_mm); else by the integrand at nu(safe_norm) if > 1e-16."""
    m, prec = t.m, t.e.prec
    out = numpy.zeros_like(t.ln_k)
    t.levels = getattr(t, "levels", {})
    lev = []
    for idx, ln_k in enumerate(t.ln_k):
        if safe_norm is None:
            norm = 1.0 / integrand(0.0, t, ln_k, 1.0)
        else:
            norm = 1.0
            if safe_norm != -1:
                inv = integrand(numpy.log(nu_of_mass(m, safe_norm)), t, ln_k, 1.0)
                norm = 1.0 / inv if inv > 1e-16 else 1.0
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always", AccuracyWarning)
            val, level = romberg(integrand, ln_nu_lo, numpy.log(m.nu_max),
                                 args=(t, ln_k, norm), vec_func=True,
                                 tol=prec["global_precision"],
                                 rtol=prec["halo_precision"],
                                 divmax=prec["divmax"], return_level=True)
        DIVMAX_EXCEEDED[0] += sum(1 for w in rec if issubclass(w.category, AccuracyWarning))
        out[idx] = val / norm
        lev.append(level)
    t.levels[integrand.__name__] = lev
    return out


def _ranged(t, spline, k):
    """halo.py:649-672: spline inside [k_min, k_max], zero outside."""
    k = numpy.asarray(k, dtype=float)
    with numpy.errstate(all="ignore"):
        return numpy.where(numpy.logical_and(k >= t.k_min, k <= t.k_max),
                           spline(numpy.log(k)), 0.0)


def halo_power(t, which, k, extrapolate=False):
    """halo.py:266-439.  which: 'lin','mm','gm','gg'.  ``extrapolate`` follows
    Halo(extrapolate=True): above k_max P_mm continues as a rescaled linear spectrum
    (:300-312), P_gm / P_gg as power laws whose slope is the mean log-slope over knots
    -7..-1 (:341-367, 405-431)."""
    k = numpy.asarray(k, dtype=float)
    e = t.e
    if which == "lin":
        return linear_power(e, k)
    if which == "mm":
        ha = hb = t.h_m_spline
        pp = t.pp_mm_spline
    elif which == "gm":
        ha, hb, pp = t.h_g_spline, t.h_m_spline, t.pp_gm_spline
    elif which == "gg":
        ha = hb = t.h_g_spline
        pp = t.pp_gg_spline
    else:
        raise KeyError(which)
    kmin, kmax = t.k_min, t.k_max
    lo = linear_power(e, k) * (
        _ranged(t, ha, kmin) * _ranged(t, hb, kmin) +
        _ranged(t, pp, kmin) / linear_power(e, kmin))
    mid = linear_power(e, k) * _ranged(t, ha, k) * _ranged(t, hb, k) + _ranged(t, pp, k)
    if not extrapolate:
        return numpy.where(k < t.k_min, lo, numpy.where(k <= t.k_max, mid, 0.0))
    if which == "mm":
        hi = linear_power(e, k) * (
            _ranged(t, ha, kmax) * _ranged(t, hb, kmax) +
            _ranged(t, pp, kmax) / linear_power(e, kmax))
    else:
        k_array = numpy.exp(t.ln_k[-7:-1])
        log_values = numpy.log(linear_power(e, k_array) * _ranged(t, ha, k_array) *
                               _ranged(t, hb, k_array) + _ranged(t, pp, k_array))
        slope = numpy.mean((log_values[1:] - log_values[:-1]) /
                           (t.ln_k[-6:-1] - t.ln_k[-7:-2]))
        with numpy.errstate(all="ignore"):
            hi = numpy.power(k / kmax, slope) * (
                linear_power(e, kmax) * _ranged(t, ha, kmax) * _ranged(t, hb, kmax) +
                _ranged(t, pp, kmax))
    return numpy.where(k < t.k_min, lo, numpy.where(k < t.k_max, mid, hi))


def log_slope(t, which):
    """Halo._log_slope_gm / _log_slope_gg (halo.py:343-352, 407-416)."""
    ha, hb, pp = ((t.h_g_spline, t.h_m_spline, t.pp_gm_spline) if which == "gm" else
                  (t.h_g_spline, t.h_g_spline, t.pp_gg_spline))
    k_array = numpy.exp(t.ln_k[-7:-1])
    log_values = numpy.log(linear_power(t.e, k_array) * _ranged(t, ha, k_array) *
                           _ranged(t, hb, k_array) + _ranged(t, pp, k_array))
    return numpy.mean((log_values[1:] - log_values[:-1]) / (t.ln_k[-6:-1] - t.ln_k[-7:-2]))


def halofit_table(t):
    """halo.py:1261-1319 (Takahashi et al. 2012 coefficients)."""
    e, prec = t.e, t.e.prec
    f = Table()
    om = omega_m(e)
    f.f_1, f.f_2, f.f_3 = (numpy.power(om, -0.0307), numpy.power(om, -0.0585),
                           numpy.power(om, 0.0743))
    f.omega_l = omega_l(e)
    f.w = e.cosmo_dict["w0"] + e.cosmo_dict["wa"] * (1 - 1.0 / (1 + e.z))
    f.ln_R = numpy.linspace(numpy.log(0.1), numpy.log(10.0), prec["halo_npoints"])
    f.ln_sigma2 = numpy.empty(prec["halo_npoints"])
    for i, ln_R in enumerate(f.ln_R):
        R = numpy.exp(ln_R)
        s2 = _rom(lambda lk: delta_k(e, numpy.exp(lk)) *
                  numpy.exp(-numpy.exp(lk) * numpy.exp(lk) * R * R),
                  numpy.log(t.k_min), numpy.log(t.k_max),
                  prec["halo_precision"], prec)
        f.ln_sigma2[i] = numpy.log(s2)
    inv = InterpolatedUnivariateSpline(f.ln_sigma2[::-1], f.ln_R[::-1])
    f.k_s = 1.0 / numpy.exp(inv(0.0))
    sp5 = InterpolatedUnivariateSpline(f.ln_R, f.ln_sigma2, k=5)
    dev1, dev2 = sp5.derivatives(numpy.log(1.0 / f.k_s))[1:3]
    n = f.n_eff = -dev1 - 3.0
    C = f.C = -dev2
    f.a_n = numpy.power(10, 1.5222 + 2.8553 * n + 2.3706 * n * n +
                        0.9903 * n * n * n + 0.2250 * n * n * n * n +
                        -0.6038 * C + 0.1749 * f.omega_l * (1 + f.w))
    f.b_n = numpy.power(10, -0.5642 + 0.5864 * n + 0.5716 * n * n +
                        -1.5474 * C + 0.2279 * f.omega_l * (1 + f.w))
    f.c_n = numpy.power(10, 0.3698 + 2.0404 * n + 0.8161 * n * n + 0.5869 * C)
    f.gamma_n = 0.1971 - 0.0843 * n + 0.8460 * C
    f.alpha_n = numpy.fabs(6.0835 + 1.3373 * n - 0.1959 * n * n + -5.5274 * C)
    f.beta_n = (2.0379 - 0.7354 * n + 0.3157 * n * n + 1.2490 * n * n * n +
                0.3980 * n * n * n * n + -0.1682 * C)
    f.mu_n = 0.0
    f.nu_n = numpy.power(10, 5.2105 + 3.6902 * n)
    t.hf = f
    return f


def halofit_power(t, which, k):
    """halo.py:1325-1413."""
    k = numpy.asarray(k, dtype=float)
    f, e = t.hf, t.e
    dk = delta_k(e, k)
    y = k / f.k_s
    d2q = dk * (numpy.power(1 + dk, f.beta_n) / (1 + f.alpha_n * dk) *
                numpy.exp(-(y / 4.0 + y * y / 8.0)))
    d2h = (f.a_n * numpy.power(y, 3.0 * f.f_1) /
           (1.0 + f.b_n * numpy.power(y, f.f_2) +
            numpy.power(f.c_n * f.f_3 * y, 3.0 - f.gamma_n))) / (
               1.0 + f.mu_n / y + f.nu_n / (y * y))
    pmm = 2.0 * numpy.pi * numpy.pi / numpy.power(k, 3) * (d2q + d2h)
    if which == "mm":
        return pmm
    if which == "gm":
        return (pmm * _ranged(t, t.h_g_spline, k) * _ranged(t, t.h_m_spline, k) +
                _ranged(t, t.pp_gm_spline, k))
    if which == "gg":
        return (pmm * _ranged(t, t.h_g_spline, k) * _ranged(t, t.h_g_spline, k) +
                _ranged(t, t.pp_gg_spline, k))
    raise KeyError(which)


# ---------------------------------------------------------------------------
# L1b: multi-epoch cosmology (cosmology.py:731-1164)
# ---------------------------------------------------------------------------
def multi_epoch(z_min, z_max, cosmo_dict=None, limits=None, prec=None, e0=None):
    """cosmology.py:747-817."""
    prec = default_precision if prec is None else prec
    me = Table(z_min=max(z_min, 0.0), z_max=z_max, prec=prec)
    me.e0 = epoch(cosmo_dict, 0.0, limits, prec) if e0 is None else e0
    me.z_arr = numpy.linspace(me.z_min, me.z_max, prec["cosmo_npoints"])
    me.chi_arr = numpy.array([
        _rom(lambda zz: E(me.e0, zz), 0.0, z, prec["cosmo_precision"], prec)
        for z in me.z_arr])
    me.chi_spline = InterpolatedUnivariateSpline(me.z_arr, me.chi_arr)
    me.z_spline = InterpolatedUnivariateSpline(me.chi_arr, me.z_arr)
    me.growth_arr = growth_approx(me.e0, 1. / (1. + me.z_arr)) / me.e0.growth_norm
    me.growth_spline = InterpolatedUnivariateSpline(me.z_arr, me.growth_arr)
    return me


def me_regrid(me, z_min, z_max):
    """copy.copy + set_redshift (kernel.py:296-297, cosmology.py:819-837)."""
    return multi_epoch(z_min, z_max, e0=me.e0, prec=me.prec)


def me_chi(me, z):
    z = numpy.asarray(z, dtype=float)                           # :873-894
    return numpy.where(numpy.logical_and(z <= me.z_max, z >= me.z_min),
                       me.chi_spline(z), 0.0)


def me_growth(me, z):
    z = numpy.asarray(z, dtype=float)                           # :934-953
    return numpy.where(numpy.logical_and(z <= me.z_max, z >= me.z_min),
                       me.growth_spline(z), 1.0)


# ---------------------------------------------------------------------------
# L4: redshift distributions, windows, kernels (kernel.py)
# ---------------------------------------------------------------------------
def dndz_maglim(z_min, z_max, a, z0, b, prec=None):
    """kernel.py:160-179; ``1/b`` is Py2 floor division for an int b (:167)."""
    prec = default_precision if prec is None else prec
    inv_b = (1 // b) if isinstance(b, int) else 1 / b
    tmp = numpy.power(-1 * numpy.log(prec['dNdz_precision']), inv_b) * z0
    if tmp < z_max:
        z_max = tmp
    d = Table(kind="maglim", z_min=z_min, z_max=z_max, a=a, z0=z0, b=b,
              prec=prec, norm=1.0)
    dndz_normalize(d)
    return d


def dndz_gaussian(z_min, z_max, z0, sigma_z, prec=None):
    """kernel.py:100-112."""
    prec = default_precision if prec is None else prec
    if z_min < z0 - 8.0 * sigma_z:
        z_min = z0 - 8.0 * sigma_z
    if z_max > z0 + 8.0 * sigma_z:
        z_max = z0 + 8.0 * sigma_z
    d = Table(kind="gaussian", z_min=z_min, z_max=z_max, z0=z0,
              sigma_z=sigma_z, prec=prec, norm=1.0)
    dndz_normalize(d)
    return d


def dndz_boxcar(z_min, z_max, prec=None):
    """The base class dNdz (kernel.py:26-65): raw_dndz = 1 between z_min and z_max."""
    prec = default_precision if prec is None else prec
    d = Table(kind="boxcar", z_min=z_min, z_max=z_max, prec=prec, norm=1.0)
    dndz_normalize(d)
    return d


def dndz_raw(d, z):
    if d.kind == "boxcar":
        return 1.0 + 0.0 * numpy.asarray(z, dtype=float)
    if d.kind == "maglim":
        return numpy.power(z, d.a) * numpy.exp(-1.0 * numpy.power(z / d.z0, d.b))
    if d.kind == "gaussian":
        return numpy.exp(-1.0 * (z - d.z0) * (z - d.z0) /
                         (2.0 * d.sigma_z * d.sigma_z))
    raise KeyError(d.kind)


def dndz_normalize(d):
    """kernel.py:43-54."""
    norm = _rom(lambda z: dndz_raw(d, z), d.z_min, d.z_max,
                d.prec["dNdz_precision"], d.prec)
    d.norm = 1.0 / norm


def dndz(d, z):
    z = numpy.asarray(z, dtype=float)                           # :67-78
    with numpy.errstate(all="ignore"):
        return numpy.where(numpy.logical_and(z <= d.z_max, z >= d.z_min),
                           d.norm * dndz_raw(d, z), 0.0)


def window_table(kind, dist, me):
    """kernel.py:234-246, 289-313 and the Galaxy (:374-387) / Convergence
    (:429-484) raw window functions.  kind: 'galaxy' or 'convergence'."""
    prec = me.prec
    wp = prec["window_precision"]
    w = Table(kind=kind, dist=dist, prec=prec)
    z_min = dist.z_min if kind == "galaxy" else 0.0
    if z_min < wp:
        z_min = wp
    w.z_min, w.z_max = z_min, dist.z_max
    w.me = me_regrid(me, w.z_min, w.z_max)
    w.chi_min = float(me_chi(w.me, w.z_min))
    if w.chi_min < wp:
        w.chi_min = wp
    w.chi_max = float(me_chi(w.me, w.z_max))
    w.chi_arr = numpy.linspace(w.chi_min, w.chi_max, prec["window_npoints"])
    if kind == "convergence":
        w.g_chi_min = float(me_chi(w.me, dist.z_min))            # :438-441
        if w.g_chi_min < wp:
            w.g_chi_min = wp
    w.wf_arr = numpy.array([_raw_window(w, c) for c in w.chi_arr])
    w.wf_spline = InterpolatedUnivariateSpline(w.chi_arr, w.wf_arr)
    return w


def _raw_window(w, chi):
    me = w.me
    if w.kind == "galaxy":                                       # :382-387
        z = me.z_spline(chi)
        return float(1.0 / E(me.e0, z) * dndz(w.dist, z))
    a = 1.0 / (1.0 + me.z_spline(chi))                           # :443-477
    chi_bound = chi
    if chi_bound < w.g_chi_min:
        chi_bound = w.g_chi_min
    if chi_bound <= w.prec["window_precision"]:
        g = 0.0
    else:
        def lens(c, chi0):                                       # :479-484
            z = me.z_spline(c)
            return 1.0 / E(me.e0, z) * dndz(w.dist, z) * (c - chi0) / c
        g = _rom(lens, chi_bound, w.chi_max, w.prec["window_precision"],
                 w.prec, args=(chi,))
    g *= me.e0.H0 * me.e0.H0 * chi
    return float(3.0 / 2.0 * me.e0.om0 * g / a)


def window(w, chi):
    chi = numpy.asarray(chi, dtype=float)                        # :326-340
    return numpy.where(numpy.logical_and(chi >= w.chi_min, chi <= w.chi_max),
                       w.wf_spline(chi), 0.0)


def kernel_table(ktheta_min, ktheta_max, wa, wb, me, bessel_order=0):
    """kernel.py:584-649 (J0) and 803-839 (J2).  The windows are re-gridded
    against ``me`` as Kernel.__init__ does (:605-608); the files the reference
    writes into the CWD there are deliberately not reproduced."""
    prec = me.prec
    kt = Table(me=me, prec=prec, order=bessel_order)
    kt.ln_kt_min, kt.ln_kt_max = numpy.log(ktheta_min), numpy.log(ktheta_max)
    kt.wa = window_table(wa.kind, wa.dist, me)
    kt.wb = window_table(wb.kind, wb.dist, me)
    kt.z_min = max(kt.wa.z_min, kt.wb.z_min)
    kt.z_max = min(kt.wa.z_max, kt.wb.z_max)
    kt.chi_min = max(prec["window_precision"], float(me_chi(me, kt.z_min)))
    kt.chi_max = float(me_chi(me, kt.z_max))
    kt.ln_kt = numpy.linspace(kt.ln_kt_min, kt.ln_kt_max, prec["kernel_npoints"])
    kt.j_limit = special.jn_zeros(bessel_order, prec["kernel_bessel_limit"])[-1]
    z_arr = numpy.linspace(kt.z_min, kt.z_max, prec["kernel_npoints"])   # :635-639
    # _find_z_bar always uses the base-class J0 integrand (J0(0) = 1), also for
    # the J2 kernel (kernel.py:635-639 is not overridden at 784-839)
    kt.z_bar = z_arr[numpy.argmax(_kernel_integrand(me_chi(me, z_arr), kt, 0.0,
                                                    order=0))]
    kt.levels = []
    kt.k_arr = numpy.array([_raw_kernel(kt, x) for x in kt.ln_kt])
    kt.k_spline = InterpolatedUnivariateSpline(kt.ln_kt, kt.k_arr)
    return kt


def _kernel_integrand(chi, kt, ktheta, order=None):
    D = me_growth(kt.me, kt.me.z_spline(chi))                    # :707-712 / 833-839
    order = kt.order if order is None else order
    bess = special.j0(ktheta * chi) if order == 0 else special.jn(2, ktheta * chi)
    return window(kt.wa, chi) * window(kt.wb, chi) * D * D * bess


def _raw_kernel(kt, ln_ktheta):
    ktheta = numpy.exp(ln_ktheta)                                # :678-705
    chi_max = kt.j_limit / ktheta
    if chi_max >= kt.chi_max:
        chi_max = kt.chi_max
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", AccuracyWarning)
        val, level = romberg(_kernel_integrand, kt.chi_min, chi_max,
                             args=(kt, ktheta), vec_func=True,
                             tol=kt.prec["global_precision"],
                             rtol=kt.prec["kernel_precision"],
                             divmax=kt.prec["divmax"], return_level=True)
    kt.levels.append(level)
    return val


def kernel_eval(kt, ln_ktheta):
    x = numpy.asarray(ln_ktheta, dtype=float)                    # :714-729
    return numpy.where(x < kt.ln_kt_min, kt.k_spline(kt.ln_kt_min),
                       numpy.where(x <= kt.ln_kt_max, kt.k_spline(x), 0.0))


# ---------------------------------------------------------------------------
# L5: observables (correlation.py)
# ---------------------------------------------------------------------------
def xi3d_raw(power, r, k_min, k_max, prec=None, levels=None):
    """Correlation3d.raw_correlation (correlation.py:470-499): int dlnk k^2/(2 pi) P(k) J0(k r)
    -- the cylindrical Bessel function, as the reference has it."""
    prec = default_precision if prec is None else prec
    out = []
    for rv in numpy.atleast_1d(r):
        def integrand(ln_k, rr):
            k = numpy.exp(ln_k)
            return k * k / (2.0 * numpy.pi) * power(k) * special.j0(k * rr)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", AccuracyWarning)
            val, level = romberg(integrand, numpy.log(k_min), numpy.log(k_max),
                                 args=(rv,), vec_func=True, tol=prec["global_precision"],
                                 rtol=prec["corr_precision"], divmax=prec["divmax"],
                                 return_level=True)
        out.append(val)
        if levels is not None:
            levels.append(level)
    return numpy.array(out)


def xi3d(power, r_min, r_max, r, k_min, k_max, prec=None):
    """Correlation3d.__init__/compute_correlation/correlation (correlation.py:414-510):
    corr_npoints log-spaced r, spline in r, zero outside (r_min, r_max]."""
    prec = default_precision if prec is None else prec
    log_r_min, log_r_max = numpy.log10(r_min), numpy.log10(r_max)
    r_array = numpy.logspace(log_r_min, log_r_max, prec["corr_npoints"])
    xi_array = xi3d_raw(power, r_array, k_min, k_max, prec)
    spline = InterpolatedUnivariateSpline(r_array, xi_array)
    r = numpy.asarray(r, dtype=float)
    return (numpy.where(numpy.logical_and(r <= 10.0 ** log_r_max, r > 10.0 ** log_r_min),
                        spline(r), 0.0), r_array, xi_array)


def theta_bins(theta_min_deg, theta_max_deg, bins_per_decade=5.0):
    """correlation.py:69-90."""
    d2r = numpy.pi / 180.0
    lmin = numpy.log10(theta_min_deg * d2r)
    lmax = numpy.log10(theta_max_deg * d2r)
    out = []
    unit_double = numpy.floor(lmin) * bins_per_decade
    theta = numpy.power(10.0, unit_double / (1.0 * bins_per_decade))
    while theta < numpy.power(10.0, lmax):
        if theta >= numpy.power(10.0, lmin) and theta < numpy.power(10.0, lmax):
            out.append(10 ** (0.5 * (numpy.log10(theta) +
                                     (unit_double + 1.0) / (1.0 * bins_per_decade))))
        unit_double += 1.0
        theta = numpy.power(10.0, unit_double / (1.0 * bins_per_decade))
    if theta_min_deg == theta_max_deg:
        return numpy.array([theta_min_deg * d2r])
    return numpy.array(out)


def wtheta(kt, power, theta_rad, k_min, k_max, D_z, prec=None, levels=None):
    """correlation.py:242-275: w(theta) = int dlnk k^2/(2 pi) P(k)/D_z^2 K(ln k theta).
    ``power`` is a callable k -> P(k) (the halo already moved to z_bar)."""
    prec = kt.prec if prec is None else prec
    out = []
    for th in numpy.atleast_1d(theta_rad):
        def integrand(ln_k, theta):
            k = numpy.exp(ln_k)
            return (k * k / (2.0 * numpy.pi) * power(k) / (D_z * D_z) *
                    kernel_eval(kt, numpy.log(k * theta)))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", AccuracyWarning)
            val, level = romberg(integrand, numpy.log(k_min), numpy.log(k_max),
                                 args=(th,), vec_func=True,
                                 tol=prec["global_precision"],
                                 rtol=prec["corr_precision"],
                                 divmax=prec["divmax"], return_level=True)
        out.append(val)
        if levels is not None:
            levels.append(level)
    return numpy.array(out)


def cell(kt, power, ell, D_z, prec=None, levels=None):
    """correlation.py:360-392: Limber C_l over chi."""
    prec = kt.prec if prec is None else prec
    out = []
    for l in numpy.atleast_1d(ell):
        def integrand(chi, ll):
            D = me_growth(kt.me, kt.me.z_spline(chi))
            return (power(ll / chi) / (D_z * D_z) * window(kt.wa, chi) *
                    window(kt.wb, chi) * D * D / (chi * chi))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", AccuracyWarning)
            val, level = romberg(integrand, kt.chi_min, kt.chi_max, args=(l,),
                                 vec_func=True, tol=prec["global_precision"],
                                 rtol=prec["corr_precision"],
                                 divmax=prec["divmax"], return_level=True)
        out.append(val)
        if levels is not None:
            levels.append(level)
    return numpy.array(out)


# ---------------------------------------------------------------------------
# L6: Gaussian covariance of w(theta) (covariance.py), Covariance(corr, corr) with
# nongaussian_cov=False.  In that use the window comparisons of covariance.py:108-115
# come out [F, F, F, F, T, T] whatever the windows are (kernel.py:248-259 compares the
# windows' private MultiEpoch copies by identity), so the shot-noise factors of the G
# integrand (pairs 0 and 2) vanish and only the Poisson term keeps them (pairs 4, 5).
# ---------------------------------------------------------------------------
def annular_bins(theta_min_deg, theta_max_deg, bins_per_decade=5.0):
    """covariance.py:52-75 and AnnulusBin (:1085-1103): arrays inner, outer, center,
    delta (radians)."""
    d2r = numpy.pi / 180.0
    lmin = numpy.log10(theta_min_deg * d2r)
    lmax = numpy.log10(theta_max_deg * d2r)
    inner, outer = [], []
    unit_double = numpy.floor(lmin) * bins_per_decade
    theta = numpy.power(10.0, unit_double / (1.0 * bins_per_decade))
    while theta < numpy.power(10.0, lmax):
        if theta >= numpy.power(10.0, lmin) and theta < numpy.power(10.0, lmax):
            inner.append(theta)
            outer.append(numpy.power(10.0, (unit_double + 1.0) / (1.0 * bins_per_decade)))
        unit_double += 1.0
        theta = numpy.power(10.0, unit_double / (1.0 * bins_per_decade))
    inner, outer = numpy.array(inner), numpy.array(outer)
    center = numpy.power(10.0, 0.5 * (numpy.log10(inner) + numpy.log10(outer)))
    return inner, outer, center, outer - inner


def covariance_table(kt, power, limits=None, levels=None):
    """covariance.py:130-175 + _initialize_halo_splines (:455-543), matching
    correlations: the projected spectrum int dchi P(K/chi) W_a W_b D^2/chi^2 on
    kernel_npoints knots in ln K.  ``power``: k -> P(k), the halo already at z_bar."""
    prec, me = kt.prec, kt.me
    limits = default_limits if limits is None else limits
    cv = Table(kt=kt, prec=prec, limits=limits)
    z_min_a = max(kt.wa.z_min, kt.wb.z_min)
    z_max_a = min(kt.wa.z_max, kt.wb.z_max)
    cv.chi_min = float(me_chi(me, z_min_a))
    if cv.chi_min < prec["window_precision"]:
        cv.chi_min = prec["window_precision"]
    cv.chi_max = float(me_chi(me, z_max_a))
    cv.ln_K_min = numpy.log(limits["k_min"] * cv.chi_min)
    cv.ln_K_max = numpy.log(limits["k_max"] * cv.chi_max)
    cv.ln_K = numpy.linspace(cv.ln_K_min, cv.ln_K_max, prec["kernel_npoints"])
    cv.j0_limit = special.jn_zeros(0, prec["kernel_bessel_limit"])[-1]
    cv.D_z = float(me_growth(me, kt.z_bar))                      # :464
    chi_peak = float(me_chi(me, cv.D_z))                         # :466 (a growth factor as z)

    def integrand(chi, ln_K, norm):                              # :545-552, kernel.py:1066-1071
        D = me_growth(me, me.z_spline(chi))
        return (norm * power(numpy.exp(ln_K) / chi) *
                (window(kt.wa, chi) * window(kt.wb, chi) * D * D / (chi * chi)))

    cv.proj = numpy.empty(cv.ln_K.size)
    for i, ln_K in enumerate(cv.ln_K):
        chi_min = max(numpy.exp(ln_K) / limits["k_max"], cv.chi_min)
        chi_max = min(numpy.exp(ln_K) / limits["k_min"], cv.chi_max)
        norm_int = integrand(chi_peak, ln_K, 1.0)
        norm = 1.0 / norm_int if norm_int > 0.0 else 1.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", AccuracyWarning)
            val, level = romberg(integrand, chi_min, chi_max, args=(ln_K, norm),
                                 vec_func=True, tol=prec["global_precision"],
                                 rtol=prec["corr_precision"], divmax=prec["divmax"],
                                 return_level=True)
        cv.proj[i] = val / norm
        if levels is not None:
            levels.append(level)
    cv.proj_spline = InterpolatedUnivariateSpline(cv.ln_K, cv.proj)
    return cv


def covariance_G_integrand(cv, ln_K, theta_a, theta_b, norm=1.0, poisson=(0.0, 0.0)):
    """covariance.py:397-453 with matching_corrs (two_point_term2 = two_point_term1)."""
    K = numpy.exp(ln_K)
    Pa = cv.proj_spline(numpy.log(K)) / (cv.D_z ** 2)
    Pb = cv.proj_spline(numpy.log(K)) / (cv.D_z ** 2)
    t1 = Pa * Pb + Pa * poisson[1] + Pb * poisson[0]
    return K * K * norm * (t1 + t1) * special.j0(K * theta_a) * special.j0(K * theta_b)


def covariance_G(cv, theta_a, theta_b, area, poisson=(0.0, 0.0), levels=None):
    """covariance.py:361-395."""
    ln_K_max = numpy.log(max(cv.j0_limit / theta_a, cv.j0_limit / theta_b))
    if ln_K_max > cv.ln_K_max:
        ln_K_max = cv.ln_K_max
    elif ln_K_max <= cv.ln_K_min:
        return 0.0
    norm = 1.0 / covariance_G_integrand(cv, 0.0, 0.0, 0.0, 1.0, poisson)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", AccuracyWarning)
        val, level = romberg(lambda x: covariance_G_integrand(cv, x, theta_a, theta_b, norm,
                                                              poisson),
                             cv.ln_K_min, ln_K_max, vec_func=True,
                             tol=cv.prec["global_precision"], rtol=cv.prec["corr_precision"],
                             divmax=cv.prec["divmax"], return_level=True)
    if levels is not None:
        levels.append(level)
    return val / (norm * 2.0 * numpy.pi * area)


def covariance_P(theta, delta, area, n_a, n_b, variance, shear_cross=False):
    """covariance.py:338-359 for Covariance(corr, corr): only window pairs 4 and 5 are
    'equal' (see the section header), so term3 is the whole Poisson term."""
    dens_a, dens_b = n_a / area, n_b / area                      # density[4], density[5]
    term3 = (variance * variance / dens_a) * (variance * variance / dens_b) * (
        1.0 + (1 if shear_cross else 0))
    return term3 / (2.0 * numpy.pi * area * theta * delta)
