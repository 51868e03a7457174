"""Randomised parity soak of the projection path: w(theta) and C_l of the drop-in classes (device)
against the oracle for random cosmologies, magnitude-limited surveys and HODs -- away from the
golden vectors' one set-up (tests/test_gpu_projection.py runs one seed of four cases).
    python tools/soak_proj.py [seed] [n]
Exit code 1 if any sample differs by more than 1e-6 (the fixtures agree to 1e-9)."""
import os, sys, time, warnings, numpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chomp_amd import correlation, cosmology, halo, hod, kernel
from oracle import chomp_oracle as o


def run(seed=5, n=4, verbose=True):
    """n random cases from `seed`; returns the largest relative difference seen."""
    rng = numpy.random.default_rng(seed)
    d2r = numpy.pi / 180.0
    worst = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for case in range(n):
            worst = max(worst, _case(rng, case, d2r, verbose))
    return worst


def _case(rng, case, d2r, verbose):
    if True:
        c = dict(o.default_cosmo_dict)
        c["omega_m0"] = rng.uniform(0.24, 0.34) - c["omega_r0"]
        c["omega_l0"] = 1.0 - c["omega_m0"] - c["omega_r0"]
        c["omega_b0"] = rng.uniform(0.04, 0.05)
        c["h"] = rng.uniform(0.65, 0.75)
        c["sigma_8"] = rng.uniform(0.75, 0.85)
        c["n_scalar"] = rng.uniform(0.94, 0.99)
        z0, b = rng.uniform(0.2, 0.45), float(rng.choice([1.5, 2.0]))
        hd = dict(o.default_hod_dict)
        hd["log_M_min"] = rng.uniform(11.9, 12.5); hd["log_M_0"] = hd["log_M_min"]
        hd["sigma"] = rng.uniform(0.12, 0.35); hd["log_M_1p"] = hd["log_M_min"] + rng.uniform(1.0, 1.4)
        ggl = case % 2 == 1
        ps = "power_gm" if ggl else "power_gg"
        theta = numpy.array([0.003, 0.05, 0.6]) * d2r
        ell = numpy.array([30.0, 400.0, 5000.0])
        # ---- device, through the drop-in classes
        cm = cosmology.MultiEpoch(0.0, 5.0, c)
        wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, z0, b), cm)
        if ggl:
            wb = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2), cm)
            kern = kernel.GalaxyGalaxyLensingKernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
        else:
            wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, z0, b), cm)
            kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
        hf = case % 4 == 3          # HaloFit spectra (configs[4]'s kind), coefficients from z = 0 as there
        hcls = halo.HaloFit if hf else halo.Halo
        h = hcls(0.0, input_hod=hod.HODZheng(hd), cosmo_single_epoch=cosmology.SingleEpoch(0.0, c))
        if hf:
            h.power_mm(numpy.array([0.1]))      # (halo.py:1337-1338: the sigma spline is built at the first call)
        corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec=ps)
        cf = correlation.CorrelationFourier(10.0, 10000.0, kern, input_halo=h, powSpec=ps)
        w_dev = corr.correlation(theta)
        c_dev = cf.correlation(ell)
        # ---- oracle
        me = o.multi_epoch(0.0, 5.0, c)
        owa = o.window_table("galaxy", o.dndz_maglim(0.0, 2.0, 2.0, z0, b), me)
        owb = (o.window_table("convergence", o.dndz_gaussian(0.0, 2.0, 1.0, 0.2), me) if ggl
               else o.window_table("galaxy", o.dndz_maglim(0.0, 2.0, 2.0, z0, b), me))
        kt = o.kernel_table(1e-6 * d2r, 100.0 * d2r, owa, owb, me, bessel_order=2 if ggl else 0)
        e = o.epoch(c, float(kt.z_bar))
        fam = "gm" if ggl else "gg"
        t = o.halo_table(e, o.mass_table(e), o.zheng(hd), families=(fam,))
        D_z = float(o.me_growth(me, kt.z_bar))
        power = lambda k: o.halo_power(t, fam, k)
        if hf:
            e0 = o.epoch(c, 0.0)
            t0 = o.halo_table(e0, o.mass_table(e0), o.zheng(hd), families=("mm",))
            t.hf = o.halofit_table(t0)
            power = lambda k: o.halofit_power(t, fam, k)
        w_ref = o.wtheta(kt, power, theta, t.k_min, t.k_max, D_z)
        c_ref = o.cell(kt, power, ell, D_z)
        ew = float(numpy.max(numpy.abs(w_dev / w_ref - 1)))
        ec = float(numpy.max(numpy.abs(c_dev / c_ref - 1)))
        if verbose:
            print("case %d %s%s z0=%.3f b=%.1f z_bar %.4f (dev %.4f)  w %.2e  C_l %.2e  status 0x%x" % (
                case, ps, " HaloFit" if hf else "", z0, b, kt.z_bar, kern.z_bar, ew, ec, h.status), flush=True)
        return max(ew, ec)


if __name__ == "__main__":
    t_start = time.time()
    w = run(int(sys.argv[1]) if len(sys.argv) > 1 else 5, int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    print("worst %.3e  (%.0f s)" % (w, time.time() - t_start))
    sys.exit(1 if w > 1e-6 else 0)
