"""Levels and values of the covariance's projected-spectrum table against G12 (scratch)."""
import os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy
from conftest import load_golden
from chomp_amd import correlation, covariance, halo, kernel, cosmology
g = load_golden("g12_covariance_gaussian")
warnings.simplefilter("ignore")
wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0))
cm = cosmology.MultiEpoch(0.0, 5.0)
d2r = numpy.pi / 180.0
kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wa, cm)
h = halo.Halo(0.0)
corr = correlation.Correlation(0.01, 1.0, kern, input_halo=h, power_spec="power_gg")
cv = covariance.Covariance(corr, corr, nongaussian_cov=False, power_spec="power_gg", bins_per_decade=3.0,
                           survey_area_deg2=100.0, n_a=2.0e6, n_b=2.0e6, variance=0.3)
cv.get_covariance()
print("levels", list(cv._halo_a_levels))
rel = cv._halo_a_array / g["auto_proj"] - 1
print("rel", " ".join("%.1e" % x for x in rel))
k = numpy.logspace(-3, 2, 41)
print("z_bar", kern.z_bar)
print("pgg", " ".join("%.10e" % x for x in h.power_gg(k)))
