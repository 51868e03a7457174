import sys, time, numpy
sys.path.insert(0, "/root/repo/tests/golden")
import ref_loader
ref_loader.MODULES = ref_loader.MODULES + ("perturbation_spectra", "halo_trispectrum", "covariance")
ns = ref_loader.load()
k, c, h, cov = ns.kernel, ns.correlation, ns.halo, ns.covariance
deg = numpy.pi/180
t = time.time()
lens = k.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0)
src = k.dNdzGaussian(0.0, 2.0, 1.0, 0.2)
me = ns.cosmology.MultiEpoch(0.0, 5.0)
wl = k.WindowFunctionGalaxy(lens, me)
ws = k.WindowFunctionConvergence(src, me)
kern = k.Kernel(0.001*0.001*deg, 100.0*1.0*deg, wl, ws, me)
hm = h.Halo(0.0)
corr = c.Correlation(0.01, 1.0, kern, input_halo=hm, power_spec='power_mm')
print("corr built", time.time()-t)
cv = cov.Covariance(corr, corr, bins_per_decade=2.0, survey_area_deg2=25, n_a=[1e10,1e10], n_b=[1e10,1e10], variance=1.0, nongaussian_cov=False, power_spec='power_mm')
print("cov built", time.time()-t, len(cv.annular_bins))
cv._initialize_halo_splines()
print("halo splines", time.time()-t)
print(cv._ln_K_array[:3], cv._halo_a_spline(cv._ln_K_array)[:5])
b = cv.annular_bins
for i in range(len(b)):
    for j in range(i, len(b)):
        t1=time.time()
        print(i, j, b[i].center, b[j].center, cv.covariance_G(b[i].center, b[j].center, b[i].delta, b[j].delta), time.time()-t1)
numpy.set_printoptions(linewidth=150)
print(cv._ln_K_array)
print(cv._halo_a_spline(cv._ln_K_array))
print(cv._D_z_a, cv._chi_min_a, cv._chi_max_a, cv._ln_K_min, cv._ln_K_max, cv._j0_limit)
from oracle import romberg as R
import inspect
print(inspect.signature(R.romberg))
norm = 1.0/cv._covariance_G_integrand(0.0, 0.0, 0.0, 1.0, 1.0)
print("norm", norm)
ta, tb = b[0].center, b[3].center
lnKmax = min(numpy.log(max(cv._j0_limit/ta, cv._j0_limit/tb)), cv._ln_K_max)
x = numpy.linspace(cv._ln_K_min, lnKmax, 9)
print(x, cv._covariance_G_integrand(x, ta, tb, b[0].delta, norm))
