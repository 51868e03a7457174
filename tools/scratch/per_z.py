"""Reference-shaped use of the drop-in classes: one Halo, set_redshift(z) + power_*(k) per z,
host arrays in and out; and a covariance matrix on top of a correlation."""
import os, sys, time, numpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import halo, kernel, correlation, covariance, cosmology
k = numpy.logspace(-3, 2, 4096)
zs = numpy.linspace(0.05, 1.5, 30)
for name in ("power_mm", "power_gm", "power_gg"):
    h = halo.Halo(0.0)
    getattr(h, name)(k)
    t = time.perf_counter()
    for z in zs:
        h.set_redshift(float(z))
        p = getattr(h, name)(k)
    dt = (time.perf_counter() - t) / zs.size
    print("%s: %.3f ms per z  (%.3e samples/s)" % (name, dt * 1e3, k.size / dt), flush=True)
d2r = numpy.pi / 180
cm = cosmology.MultiEpoch(0.0, 5.0)
wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
wb = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2), cm)
kern = kernel.Kernel(1e-6 * d2r, 100 * d2r, wa, wb, cm)
corr = correlation.Correlation(0.001, 1.0, kern, input_halo=halo.Halo(0.0), power_spec="power_mm")
corr.compute_correlation()
cv = covariance.Covariance(corr, corr, bins_per_decade=5.0, survey_area_deg2=25, n_a=[1e10, 1e10],
                           n_b=[1e10, 1e10], variance=1.0, nongaussian_cov=False, power_spec="power_mm")
cv.get_covariance()
t = time.perf_counter()
for _ in range(10):
    cv._initialized_halo_splines = False
    cov = cv.get_covariance()
dt = (time.perf_counter() - t) / 10
print("covariance %dx%d (table + %d pairs): %.3f ms" % (cov.shape[0], cov.shape[1], cov.shape[0] * (cov.shape[0] + 1) // 2, dt * 1e3))
