"""Development aid (GPU box): duration of chomp_cell for the one multipole of C4 that runs
deepest, with divmax capped at 7..14 -- the cost of each further Romberg level of one block."""
import contextlib, os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy, torch
from chomp_amd import cosmology, correlation, defaults, halo, kernel
from cell_probe import timed
d2r = numpy.pi / 180.0
for divmax in (7, 8, 9, 10, 11, 12, 13, 14, 20):
    defaults.default_precision["divmax"] = divmax
    cm = cosmology.MultiEpoch(0.0, 5.0)
    with contextlib.redirect_stdout(sys.stderr):
        wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
        wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
        kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
    h = halo.Halo(0.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec="power_gg")
        ctx, code = corr._prepare()
    ell = torch.logspace(1, 4, 2048, dtype=torch.float64, device="cuda")
    print("divmax %2d  last 1: %7.1f us   last 256: %7.1f us" % (
        divmax, timed(lambda: ctx.cell(code, 0, corr.D_z, ell[-1:])), timed(lambda: ctx.cell(code, 0, corr.D_z, ell[-256:]))))
