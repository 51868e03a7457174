"""CPU tests: the oracle (oracle/chomp_oracle.py) against the reference's own
known-answer pins and against golden vectors produced by running the reference
(tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import json
import os

import numpy
import pytest

from conftest import GOLDEN, load_golden, rel_err
from oracle import chomp_oracle as o
from params import c_dict, c_dict_2, h_dict_2, hod_dict, hod_dict_2

K4 = numpy.logspace(-3, 2, 4)


@pytest.fixture(scope="module")
def pins():
    with open(os.path.join(GOLDEN, "reference_pins.json")) as f:
        return json.load(f)


def _halo(cd, hd=None, hod=None, z=0.0, kind="st", fam=("mm", "gm", "gg"),
          set_halo=None):
    """``set_halo``: emulate Halo.set_halo(dict) of the reference (halo.py:220-235):
    only the mass function sees the new dictionary (stq, st_little_a); the halo
    profile splines (c0, beta) and the Halo's delta_v are NOT rebuilt."""
    e = o.epoch(cd, z)
    m = o.mass_table(e, hd if set_halo is None else set_halo, kind)
    return o.halo_table(e, m, o.zheng(hod), hd, families=fam)


@pytest.mark.parametrize("tag,test,args", [
    ("base", "HaloTest.test_halo", (c_dict, None, hod_dict)),
    ("cosmo2", "HaloTest.test_set_cosmology", (c_dict_2, None, hod_dict)),
    ("halo2", "HaloTest.test_set_halo", (c_dict, h_dict_2, hod_dict)),
    ("hod2", "HaloTest.test_set_hod", (c_dict, None, hod_dict_2)),
])
def test_reference_pins_and_g1(pins, tag, test, args):
    """The reference's own HaloTest literals (unit_test.py:346-407), to the
    reference's stated 4 decimals in ln P, and the same points as computed by
    the reference run here (G1) to round-off."""
    if tag == "halo2":
        t = _halo(args[0], None, args[2], set_halo=args[1])
    else:
        t = _halo(*args)
    g1 = load_golden("g1_unit_points")
    for which in ("mm", "gm", "gg"):
        mine = o.halo_power(t, which, K4)
        assert rel_err(mine, g1["%s_%s" % (tag, which)]) < 1e-10
        key = "ln_power_" + which
        if key in pins[test]:
            for a, b in zip(numpy.log(mine), pins[test][key]):
                assert round(abs(a - b), pins["places_halo"]) == 0
    if "ln_linear_power" in pins[test]:
        for a, b in zip(numpy.log(o.halo_power(t, "lin", K4)),
                        pins[test]["ln_linear_power"]):
            assert round(abs(a - b), pins["places_cosmo"]) == 0


def test_g1_mass_hod_cosmo_points():
    g1 = load_golden("g1_unit_points")
    e = o.epoch(c_dict, 0.0)
    m = o.mass_table(e)
    marr = g1["mass_arr"]
    assert rel_err(o.nu_of_mass(m, marr), g1["mf_nu"]) < 1e-12
    nu = o.nu_of_mass(m, marr)
    assert rel_err(o.f_nu(m, nu), g1["mf_f_m"]) < 1e-12
    assert rel_err(o.bias_nu(m, nu), g1["mf_bias_m"]) < 1e-12
    h = o.zheng(hod_dict)
    assert rel_err(o.zheng_first(h, marr), g1["hod_first"]) < 1e-13
    assert rel_err(o.zheng_second(h, marr), g1["hod_second"]) < 1e-13
    mine = [e.chi, e.growth, o.omega_m(e), o.omega_l(e), o.delta_c(e),
            o.delta_v(e), o.sigma_r(e, 8.0)]
    assert numpy.allclose(mine, g1["cosmo_scalars"], rtol=1e-12, atol=1e-300)


def test_g2_wmap7_z0_full_surface():
    g = load_golden("g2_wmap7_z0")
    t = _halo(None)
    for which in ("lin", "mm", "gm", "gg"):
        assert rel_err(o.halo_power(t, which, g["k"]), g[which]) < 1e-10
        # below k_min, exactly k_min/k_max, 1 ulp above k_max, far above
        assert numpy.allclose(o.halo_power(t, which, g["k_probe"]),
                              g[which + "_probe"], rtol=1e-10, atol=0)
    assert rel_err(t.h_m, g["st_h_m"]) < 1e-12
    assert rel_err(t.pp_mm, g["st_pp_mm"]) < 1e-12
    assert rel_err(t.h_g, g["h_g"]) < 1e-12
    assert rel_err(t.pp_gm, g["pp_gm"]) < 1e-11
    assert rel_err(t.pp_gg, g["pp_gg"]) < 1e-11


@pytest.mark.parametrize("z", [0.0, 0.5, 1.0, 1.5])
def test_g3_stage_tables(z):
    g = load_golden("g3_stages")
    tag = "z%03d_" % round(z * 100)
    e = o.epoch(None, z)
    m = o.mass_table(e)
    t = o.halo_table(e, m, families=("mm",))
    for name, val in (("growth", e.growth), ("chi", e.chi),
                      ("delta_c", o.delta_c(e)), ("delta_v", o.delta_v(e)),
                      ("rho_bar", o.rho_bar(e)), ("sigma_norm", e.sigma_norm),
                      ("f_norm", m.f_norm), ("bias_norm", m.bias_norm),
                      ("m_star", m.m_star), ("n_bar", t.n_bar)):
        assert numpy.isclose(val, float(g[tag + name]), rtol=1e-12, atol=0), name
    assert numpy.array_equal(m.ln_mass, g[tag + "ln_mass"])
    assert rel_err(m.nu_arr, g[tag + "nu"]) < 1e-13
    assert rel_err(t.h_m, g[tag + "h_m"]) < 1e-12
    assert rel_err(t.pp_mm, g[tag + "pp_mm"]) < 1e-12
    sig = [o.sigma_r(e, r) for r in g["R"]]
    assert rel_err(sig, g[tag + "sigma_R"]) < 1e-13


def test_g4_pmm_grid_rows():
    g = load_golden("g4_pmm_grid")
    for i in (0, 9, 31, 63):
        t = _halo(None, z=float(g["z"][i]), fam=("mm",))
        assert rel_err(o.halo_power(t, "mm", g["k"]), g["mm"][i]) < 1e-10


@pytest.mark.parametrize("z", [0.0, 1.0])
def test_g5_tinker_zheng(z):
    g = load_golden("g5_tinker_zheng")
    tag = "z%03d_" % round(z * 100)
    t = _halo(None, z=z, kind="tinker", fam=("mm", "gm"))
    m = t.m
    mine = [m.delta_v, m.t_alpha, m.t_beta, m.t_gamma, m.t_phi, m.t_eta,
            m.bias_norm]
    assert numpy.allclose(mine, g[tag + "tinker"], rtol=1e-12)
    assert abs(t.n_bar / float(g[tag + "n_bar"]) - 1) < 1e-12
    assert rel_err(t.h_g, g[tag + "h_g"]) < 1e-12
    assert rel_err(t.pp_gm, g[tag + "pp_gm"]) < 1e-11
    assert rel_err(o.halo_power(t, "gm", g["k"]), g[tag + "gm"]) < 1e-10
    assert rel_err(o.halo_power(t, "mm", g["k"]), g[tag + "mm"]) < 1e-10


def _projection(ggl):
    me = o.multi_epoch(0.0, 5.0)
    lens = o.dndz_maglim(0.0, 2.0, 2.0, 0.3, 2.0)
    wa = o.window_table("galaxy", lens, me)
    if ggl:
        wb = o.window_table("convergence", o.dndz_gaussian(0.0, 2.0, 1.0, 0.2), me)
    else:
        wb = o.window_table("galaxy", o.dndz_maglim(0.0, 2.0, 2.0, 0.3, 2.0), me)
    d2r = numpy.pi / 180.0
    return me, o.kernel_table(1e-6 * d2r, 100.0 * d2r, wa, wb, me,
                              bessel_order=2 if ggl else 0)


def test_g6_limber_galgal():
    g = load_golden("g6_limber_galgal")
    me, kt = _projection(False)
    assert kt.z_bar == g["z_bar"]
    assert abs(kt.chi_max / float(g["chi_max"]) - 1) < 1e-13
    assert rel_err(kt.wa.wf_arr, g["wa"]) < 1e-12
    assert numpy.allclose(kt.k_arr, g["kernel"], rtol=1e-11, atol=1e-22)
    assert numpy.allclose(o.kernel_eval(kt, g["lnkt_probe"]), g["kernel_probe"],
                          rtol=1e-10, atol=1e-22)
    D_z = float(o.me_growth(me, kt.z_bar))
    assert abs(D_z / float(g["D_z"]) - 1) < 1e-13
    t = _halo(None, z=float(kt.z_bar), fam=("mm", "gg"))
    assert numpy.allclose(o.theta_bins(0.001, 1.0), g["theta_bins"], rtol=1e-14)
    th, ell = g["theta"][::4], g["ell"][::4]
    for ps in ("gg", "mm"):
        power = lambda k, ps=ps: o.halo_power(t, ps, k)
        w = o.wtheta(kt, power, th, t.k_min, t.k_max, D_z)
        assert rel_err(w, g["w_power_" + ps][::4]) < 1e-9
        c = o.cell(kt, power, ell, D_z)
        assert rel_err(c, g["cl_power_" + ps][::4]) < 1e-9


def test_g7_ggl_halofit():
    g = load_golden("g7_ggl_halofit")
    me, kt = _projection(True)
    assert kt.z_bar == g["z_bar"]
    assert rel_err(kt.wb.wf_arr[1:], g["wb"][1:]) < 1e-11
    assert numpy.allclose(kt.k_arr, g["kernel"], rtol=1e-10, atol=1e-22)
    t0 = _halo(None, z=0.0, fam=("mm", "gm", "gg"))
    f = o.halofit_table(t0)
    mine = [f.k_s, f.n_eff, f.C, f.a_n, f.b_n, f.c_n, f.gamma_n, f.alpha_n,
            f.beta_n, f.nu_n]
    assert numpy.allclose(mine, g["hf_z0_pars"], rtol=1e-11)
    for which in ("mm", "gm", "gg"):
        assert rel_err(o.halofit_power(t0, which, g["k"]), g["hf_%s_z0" % which]) < 1e-10
    D_z = float(o.me_growth(me, kt.z_bar))
    t = _halo(None, z=float(kt.z_bar), fam=("gm",))
    # Reference statefulness (halo.py:1254-1259, 1337-1338): HaloFit's f_1..f_3
    # are fixed at construction and its sigma-spline (k_s, n_eff, C) is built at
    # the FIRST power_mm call and never reset by set_redshift.  The fixture called
    # power_mm at z=0 before Correlation moved the halo to z_bar, so the z=0
    # HaloFit coefficients are used with the z_bar halo tables and delta_k.
    assert numpy.allclose(g["hf_zbar_pars"], g["hf_z0_pars"], rtol=0, atol=0)
    t.hf = f
    power = lambda k: o.halofit_power(t, "gm", k)
    th = g["theta"][::4]
    w = o.wtheta(kt, power, th, t.k_min, t.k_max, D_z)
    assert rel_err(w, g["w_ggl"][::4]) < 1e-9
    c = o.cell(kt, power, g["ell"][::8], D_z)
    assert rel_err(c, g["cl_ggl"][::8]) < 1e-9


def test_full_size_vectors_of_configs_2_3_4():
    """G5b / G6b / G7b: the reference run at the FULL sizes of configs[2..4] (64 z x 4096 k P_gm
    with Tinker10 + Zheng07; 1024 theta + 2048 l for the gal-gal and the GGL + HaloFit set-ups).
    The oracle is held to them on a subsample it finishes in seconds (the full vectors are what
    the GPU tests compare against)."""
    g = load_golden("g5b_pgm_grid")
    assert g["gm"].shape == (64, 257) and numpy.array_equal(g["z"], numpy.linspace(0.0, 1.5, 64))
    for i in (0, 37, 63):
        t = _halo(None, z=float(g["z"][i]), kind="tinker", fam=("gm",))
        assert rel_err(o.halo_power(t, "gm", g["k"]), g["gm"][i]) < 1e-10, i
        assert abs(t.n_bar / float(g["n_bar"][i]) - 1) < 1e-12
    g = load_golden("g6b_limber_galgal_full")
    assert g["w_power_gg"].shape == (1024,) and g["cl_power_gg"].shape == (2048,)
    me, kt = _projection(False)
    D_z = float(o.me_growth(me, kt.z_bar))
    assert kt.z_bar == g["z_bar"] and abs(D_z / float(g["D_z"]) - 1) < 1e-13
    t = _halo(None, z=float(kt.z_bar), fam=("mm", "gg"))
    for ps in ("gg", "mm"):
        power = lambda k, ps=ps: o.halo_power(t, ps, k)
        w = o.wtheta(kt, power, g["theta"][5::128], t.k_min, t.k_max, D_z)
        assert rel_err(w, g["w_power_" + ps][5::128]) < 1e-9
        c = o.cell(kt, power, g["ell"][7::128], D_z)
        assert rel_err(c, g["cl_power_" + ps][7::128]) < 1e-9
    g = load_golden("g7b_ggl_halofit_full")
    me, kt = _projection(True)
    D_z = float(o.me_growth(me, kt.z_bar))
    t0 = _halo(None, z=0.0, fam=("mm",))
    t = _halo(None, z=float(kt.z_bar), fam=("gm",))
    t.hf = o.halofit_table(t0)            # (the fixture's call order: sigma spline built at z = 0)
    power = lambda k: o.halofit_power(t, "gm", k)
    w = o.wtheta(kt, power, g["theta"][5::128], t.k_min, t.k_max, D_z)
    assert rel_err(w, g["w_ggl"][5::128]) < 1e-9
    c = o.cell(kt, power, g["ell"][7::128], D_z)
    assert rel_err(c, g["cl_ggl"][7::128]) < 1e-9


def test_g8_extrapolation_and_exclusion():
    """SURVEY 8(f) rank 2 against the reference: Halo(extrapolate=True) above k_max
    (halo.py:300-312, 341-367, 405-431) and HaloExclusion (halo.py:1201-1233)."""
    g = load_golden("g8_extrapolate_exclusion")
    e = o.epoch(None, 0.0)
    m = o.mass_table(e)
    t = o.halo_table(e, m, families=("mm", "gm", "gg"))
    for w in ("mm", "gm", "gg"):
        assert rel_err(o.halo_power(t, w, g["k"], extrapolate=True), g["ext_%s_0" % w]) < 1e-12
    assert abs(o.log_slope(t, "gm") - g["ext_slopes_0"][0]) < 1e-12
    assert abs(o.log_slope(t, "gg") - g["ext_slopes_0"][1]) < 1e-12
    tx = o.halo_table(e, m, families=("mm", "gm", "gg"), exclusion=True)
    assert rel_err(tx.h_m, g["excl_h_m"]) < 1e-11 and rel_err(tx.h_g, g["excl_h_g"]) < 1e-11
    for w in ("mm", "gm", "gg"):
        assert rel_err(o.halo_power(tx, w, g["kx"]), g["excl_" + w]) < 1e-11


def test_g9_correlation3d():
    """Correlation3d xi(r) at a few of the reference's 50 separations
    (the full table takes a minute on the CPU; the GPU test checks all of it)."""
    g = load_golden("g9_correlation3d")
    e = o.epoch(None, 0.0)
    t = o.halo_table(e, o.mass_table(e), families=("mm",))
    idx = [0, 12, 30, 49]
    xi = o.xi3d_raw(lambda k: o.halo_power(t, "mm", k), g["r_array"][idx], 1e-3, 100.0)
    assert rel_err(xi, g["xi_mm"][idx]) < 1e-11
    xi = o.xi3d_raw(lambda k: o.halo_power(t, "mm", k, extrapolate=True), g["r_array"][idx],
                    1e-4, 1e3)
    assert rel_err(xi, g["xi_mm_wide"][idx]) < 1e-11
    assert bool(g["extrap_mm_wide"]) and not bool(g["extrap_mm"])


def test_g10_hod_stats():
    """Halo.calculate_bias / calculate_m_eff (halo.py:709-790) against the reference."""
    g = load_golden("g10_hod_stats")
    for i, z in enumerate(g["z"]):
        for tag in ("st", "tinker"):
            e = o.epoch(None, float(z))
            t = o.halo_table(e, o.mass_table(e, kind=tag), families=())
            bias, m_eff, f_sat = o.hod_stats(t)
            assert numpy.allclose([bias, m_eff, t.n_bar], g["%s_%d" % (tag, i)], rtol=1e-12)
            assert 0.05 < f_sat < 0.5


def test_g11_bao_transfer():
    """SingleEpoch(with_bao=True) (cosmology.py:474-538, 556-572) and a Halo on top of it."""
    g = load_golden("g11_bao")
    e = o.epoch(None, float(g["z"]), with_bao=True)
    assert rel_err(o.transfer_function(e, g["k"]), g["transfer"]) < 1e-13
    assert rel_err(o.linear_power(e, g["k"]), g["linear"]) < 1e-12
    assert abs(e.sigma_norm / float(g["sigma_norm"]) - 1) < 1e-12
    assert rel_err(numpy.array([o.sigma_r(e, x) for x in g["scale"]]), g["sigma_r"]) < 1e-12
    m = o.mass_table(e)
    assert numpy.array_equal(m.ln_mass, g["ln_mass"]) and rel_err(m.nu_arr, g["nu"]) < 1e-12


def test_g12_gaussian_covariance():
    """Covariance(corr, corr, nongaussian_cov=False) (covariance.py:46-543) against the
    reference: the "mag" case of G12 (galaxy x convergence windows, power_mm)."""
    g = load_golden("g12_covariance_gaussian")
    d2r = numpy.pi / 180.0
    me = o.multi_epoch(0.0, 5.0)
    wa = o.window_table("galaxy", o.dndz_maglim(0.0, 2.0, 2.0, 0.3, 2.0), me)
    wb = o.window_table("convergence", o.dndz_gaussian(0.0, 2.0, 1.0, 0.2), me)
    kt = o.kernel_table(1e-6 * d2r, 100 * d2r, wa, wb, me)
    e = o.epoch(None, kt.z_bar)
    t = o.halo_table(e, o.mass_table(e), o.zheng(), families=("mm",))
    cv = o.covariance_table(kt, lambda k: o.halo_power(t, "mm", k))
    inner, outer, center, delta = o.annular_bins(0.01, 1.0, 2.0)
    assert numpy.array_equal(center, g["mag_center"]) and numpy.array_equal(inner, g["mag_inner"])
    assert numpy.array_equal(cv.ln_K, g["mag_ln_K"])
    scale = numpy.max(numpy.abs(g["mag_proj"]))
    assert numpy.max(numpy.abs(cv.proj - g["mag_proj"])) < 1e-13 * scale
    gi = o.covariance_G_integrand(cv, g["mag_K_probe"], center[0], center[-1])
    assert numpy.max(numpy.abs(gi - g["mag_G_integrand"])) < 1e-11 * numpy.max(
        numpy.abs(g["mag_G_integrand"]))        # (the table is ~0 at its ends: absolute scale)
    area = 25.0 * d2r * d2r
    G = numpy.array([[o.covariance_G(cv, a, b, area) for b in center] for a in center])
    assert rel_err(G, g["mag_G"]) < 1e-12
    P = numpy.array([o.covariance_P(c, d, area, 1e10, 1e10, 1.0) for c, d in zip(center, delta)])
    assert rel_err(P, g["mag_P"]) < 1e-14
    assert rel_err(G + numpy.diag(P), g["mag_cov"]) < 1e-12


def test_g14_boxcar_dndz():
    """The base-class dNdz through both windows and the J0 kernel against the reference."""
    g = load_golden("g14_boxcar_dndz")
    d2r = numpy.pi / 180.0
    me = o.multi_epoch(0.0, 5.0)
    wa = o.window_table("galaxy", o.dndz_boxcar(0.2, 0.6), me)
    wb = o.window_table("convergence", o.dndz_boxcar(0.8, 1.2), me)
    assert abs(wa.dist.norm / float(g["wa_norm"]) - 1) < 1e-12
    kt = o.kernel_table(1e-6 * d2r, 100 * d2r, wa, wb, me)
    assert kt.z_bar == float(g["z_bar"])
    assert numpy.allclose(kt.wa.wf_arr, g["wa"], rtol=1e-10, atol=1e-18)
    assert numpy.allclose(kt.wb.wf_arr, g["wb"], rtol=1e-10, atol=1e-18)
    scale = numpy.max(numpy.abs(g["kernel"]))
    assert numpy.max(numpy.abs(kt.k_arr - g["kernel"])) < 1e-10 * scale
