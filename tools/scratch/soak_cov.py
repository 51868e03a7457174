"""Random surveys: the covariance's projected-spectrum table (values and Romberg levels) against the
oracle (scratch soak): python tools/scratch/soak_cov.py [seed] [n]"""
import os, sys, warnings, numpy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from chomp_amd import correlation, covariance, cosmology, halo, hod, kernel
from oracle import chomp_oracle as o
rng = numpy.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 9)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d2r = numpy.pi / 180.0
warnings.simplefilter("ignore")
worst, nlev = 0.0, 0
for case in range(n):
    z0, b = rng.uniform(0.2, 0.45), float(rng.choice([1.5, 2.0]))
    hd = dict(o.default_hod_dict)
    hd["log_M_min"] = rng.uniform(11.9, 12.5); hd["log_M_0"] = hd["log_M_min"]
    hd["sigma"] = rng.uniform(0.12, 0.35); hd["log_M_1p"] = hd["log_M_min"] + rng.uniform(1.0, 1.4)
    ps, fam = ("power_gg", "gg") if case % 2 == 0 else ("power_mm", "mm")
    cm = cosmology.MultiEpoch(0.0, 5.0)
    wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, z0, b), cm)
    kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wa, cm)
    h = halo.Halo(0.0, input_hod=hod.HODZheng(hd))
    corr = correlation.Correlation(0.01, 1.0, kern, input_halo=h, power_spec=ps)
    cv = covariance.Covariance(corr, corr, nongaussian_cov=False, power_spec=ps, bins_per_decade=3.0,
                               survey_area_deg2=100.0, n_a=2.0e6, n_b=2.0e6, variance=0.3)
    cv.get_covariance()
    me = o.multi_epoch(0.0, 5.0)
    ow = o.window_table("galaxy", o.dndz_maglim(0.0, 2.0, 2.0, z0, b), me)
    kt = o.kernel_table(1e-6 * d2r, 100 * d2r, ow, ow, me)
    e = o.epoch(None, kt.z_bar)
    t = o.halo_table(e, o.mass_table(e), o.zheng(hd), families=(fam,))
    lev = []
    ocv = o.covariance_table(kt, lambda k: o.halo_power(t, fam, k), levels=lev)
    big = numpy.abs(ocv.proj) > 1e-6 * numpy.max(numpy.abs(ocv.proj))
    err = float(numpy.max(numpy.abs(cv._halo_a_array[big] / ocv.proj[big] - 1)))
    dl = int(numpy.count_nonzero(numpy.asarray(lev) != cv._halo_a_levels))
    worst = max(worst, err); nlev += dl
    print("case %d %s z0=%.3f b=%.1f  proj %.2e  levels that differ %d" % (case, ps, z0, b, err, dl), flush=True)
print("worst %.3e  levels that differ %d" % (worst, nlev))
sys.exit(1 if worst > 1e-6 or nlev else 0)
