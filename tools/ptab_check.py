"""Development aid (GPU box): error of the 6-point Lagrange interpolation of P(k) on the 8193-point
ln k grid (k_cell_ptab / SpectrumTab) for the spectra of configs[3] / [4]."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from chomp_amd import halo, defaults
warnings.simplefilter("ignore")
N = 8192
x0, x1 = numpy.log(defaults.default_limits["k_min"]), numpy.log(defaults.default_limits["k_max"])
xt = x0 + (x1 - x0) * numpy.arange(N + 1) / N
rng = numpy.random.default_rng(3)
xs = numpy.sort(rng.uniform(x0, x1, 200000))
for name, h, fn in (("halofit gm z=0.3", halo.HaloFit(0.3), "power_gm"), ("halofit mm", halo.HaloFit(0.3), "power_mm"),
                    ("halo gg z=0.3", halo.Halo(0.3), "power_gg"), ("halo mm", halo.Halo(0.3), "power_mm"),
                    ("halo gm", halo.Halo(0.3), "power_gm")):
    kt = numpy.exp(xt); kt[0] = defaults.default_limits["k_min"]; kt[-1] = defaults.default_limits["k_max"]
    pt = getattr(h, fn)(kt)
    ptrue = getattr(h, fn)(numpy.exp(xs))
    u = (xs - x0) / (x1 - x0) * N
    i = numpy.clip(u.astype(int), 2, N - 3)
    t = u - i
    a, b, d, e, f = t + 2, t + 1, t - 1, t - 2, t - 3
    ab, ef, cd = a * b, e * f, t * d
    v = (pt[i - 2] * (b * cd * ef) / -120 + pt[i - 1] * (a * cd * ef) / 24 + pt[i] * (ab * d * ef) / -12 +
         pt[i + 1] * (ab * t * ef) / 12 + pt[i + 2] * (ab * cd * f) / -24 + pt[i + 3] * (ab * cd * e) / 120)
    err = numpy.abs(v / ptrue - 1)
    w = numpy.argsort(err)[-5:]
    print("%-18s max rel err %.2e  (median %.1e); worst at k = %s" % (name, err.max(), numpy.median(err), numpy.exp(xs[w])))
    # error by decade
    for lo in range(-3, 2):
        m = (xs >= numpy.log(10.0 ** lo)) & (xs < numpy.log(10.0 ** (lo + 1)))
        print("    k in 1e%d..1e%d: max %.2e" % (lo, lo + 1, err[m].max()))
