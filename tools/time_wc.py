"""Development aid: chomp_wtheta alone, chomp_cell alone and chomp_wtheta_cell (C_l beside w(theta)
on the side stream) for configs[3] / configs[4], HIP-event timed; per-kernel averages with
rocprofv3 --kernel-trace --stats -- python3 tools/time_wc.py [ggl]."""
import os, sys, contextlib, warnings
import numpy, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from chomp_amd import cosmology, correlation, halo, kernel
ggl = len(sys.argv) > 1 and sys.argv[1] == "ggl"
d2r = numpy.pi / 180.0
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
cm = cosmology.MultiEpoch(0.0, 5.0)
with contextlib.redirect_stdout(sys.stderr):
    la = kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0); lb = kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0)
wa = kernel.WindowFunctionGalaxy(la, cm)
if ggl:
    wb = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2), cm)
    kern = kernel.GalaxyGalaxyLensingKernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm); h = halo.HaloFit(0.0); spec = "power_gm"
else:
    wb = kernel.WindowFunctionGalaxy(lb, cm)
    kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm); h = halo.Halo(0.0); spec = "power_gg"
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec=spec)
    ctx, code = corr._prepare(defer_status=True)
theta = torch.logspace(-3, 0, 1024, dtype=torch.float64, device="cuda") * d2r
ell = torch.logspace(1, 4, 2048, dtype=torch.float64, device="cuda")
k0, k1, D = corr._k_lim[0], corr._k_lim[1], corr.D_z


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("wtheta alone      %.1f us" % timed(lambda: ctx.wtheta(code, 0, k0, k1, D, theta)))
print("cell alone        %.1f us" % timed(lambda: ctx.cell(code, 0, D, ell)))
if "alone" not in sys.argv:
    print("wtheta_cell       %.1f us" % timed(lambda: ctx.wtheta_cell(code, 0, k0, k1, D, theta, ell)))
