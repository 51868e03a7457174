"""GPU box: the actual maximum relative errors of w(theta) and C_l against the reference's golden
vectors G6 / G7 (the tests only assert < 1e-4)."""
import os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy
from conftest import load_golden, rel_err
from chomp_amd import cosmology, kernel, correlation, halo
warnings.simplefilter("ignore")
D2R = numpy.pi / 180.0
for ggl in (False, True):
    g = load_golden("g7_ggl_halofit" if ggl else "g6_limber_galgal")
    cm = cosmology.MultiEpoch(0.0, 5.0)
    wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    if ggl:
        wb = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2), cm)
        kern = kernel.GalaxyGalaxyLensingKernel(1e-6 * D2R, 100.0 * D2R, wa, wb, cm)
        cases = [("power_gm", halo.HaloFit(0.0), "w_ggl", "cl_ggl")]
    else:
        wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
        kern = kernel.Kernel(1e-6 * D2R, 100.0 * D2R, wa, wb, cm)
        cases = [(ps, halo.Halo(0.0), "w_" + ps, "cl_" + ps) for ps in ("power_gg", "power_mm")]
    for ps, h, kw, kc in cases:
        if ggl:
            h.power_mm(g["k"])
        corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec=ps)
        cf = correlation.CorrelationFourier(10, 1e4, kern, input_halo=h, powSpec=ps)
        print("%s %-9s w(theta) %.2e   C_l %.2e" % ("G7" if ggl else "G6", ps, rel_err(corr.correlation(g["theta"]), g[kw]),
                                                     rel_err(cf.correlation(g["ell"]), g[kc])))
