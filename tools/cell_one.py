"""Development aid (GPU box, under rocprofv3 --kernel-trace --stats): chomp_cell of C4 for the
deepest multipole alone, 20 times -- per-kernel durations of one block's work."""
import contextlib, os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy, torch
from chomp_amd import cosmology, correlation, halo, kernel
d2r = numpy.pi / 180.0
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
sel = ell[-1:] if len(sys.argv) < 2 else ell[-int(sys.argv[1]):]
for _ in range(20):
    ctx.cell(code, 0, corr.D_z, sel)
torch.cuda.synchronize()
