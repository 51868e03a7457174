"""configs[4]'s tolerance sweep as a measurement (GPU box): w_GGL(theta) with HaloFit power_gm and
the J2 kernel in the four arithmetic modes of chomp_set_precision against the reference's G7
vector; prints one JSON object (tools/profile_round.sh keeps it, tools/collect_profiles.py files
it as profiles/<round>_c5_precision_sweep.json).  The suite's test_c5_precision_sweep asserts the
same numbers' bounds and writes nothing."""
import json, os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy
from chomp_amd import _lib, cosmology, kernel, correlation, halo
g = numpy.load(os.path.join(R, "tests", "golden", "g7_ggl_halofit.npz"))
d2r = numpy.pi / 180.0
warnings.simplefilter("ignore")
import contextlib
cm = cosmology.MultiEpoch(0.0, 5.0)
with contextlib.redirect_stdout(sys.stderr):        # (the reference's z_max warning is a print)
    lens = kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0)
wa = kernel.WindowFunctionGalaxy(lens, cm)
wb = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2), cm)
kern = kernel.GalaxyGalaxyLensingKernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
hf = halo.HaloFit(0.0)
hf.power_mm(g["k"])          # fixture call order: sigma spline built at z = 0
corr = correlation.Correlation(0.001, 1.0, kern, input_halo=hf, power_spec="power_gm")
ctx, _ = corr._prepare()
errs = {}
try:
    for name, mode in (("fp64", _lib.PREC_F64), ("fp32_eval", _lib.PREC_F32_EVAL),
                       ("fp32_tables", _lib.PREC_F32_TABLES), ("fp32_all", _lib.PREC_F32_ALL)):
        ctx.set_precision(mode)
        w = corr.correlation(g["theta"])
        errs[name] = float(numpy.max(numpy.abs(w / g["w_ggl"] - 1)))
finally:
    ctx.set_precision(_lib.PREC_F64)
print(json.dumps({"case": "G7 w_GGL(theta), 33 theta, HaloFit power_gm, J2 kernel",
                  "max_rel_err_vs_reference": errs}, indent=1))
