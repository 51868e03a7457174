"""configs[3] steps with / without the status post (argv[1] = 0 / 1): for a kernel-trace look at
the idle gaps of the main stream."""
import contextlib, os, sys, time, warnings
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy, torch
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
from chomp_amd import cosmology, correlation, halo, kernel, _lib
nopost = len(sys.argv) > 1 and sys.argv[1] == "1"
d2r = numpy.pi / 180.0
burn = torch.randn(4096, 4096, device="cuda")
for _ in range(60): burn @ burn
cm = cosmology.MultiEpoch(0.0, 5.0)
with contextlib.redirect_stdout(sys.stderr):
    wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
    wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
h = halo.Halo(0.0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec="power_gg")
    ctx, code = corr._prepare(defer_status=True)
if nopost:
    ctx.status_post = lambda: None
theta = torch.logspace(-3, 0, 1024, dtype=torch.float64, device="cuda") * d2r
ell = torch.logspace(1, 4, 2048, dtype=torch.float64, device="cuda")
def step():
    kern._done.clear(); h._epoch_sig = None; h._nbar_valid = False; h._reset_flags(all_tables=True)
    c, code = corr._prepare(defer_status=True)
    if nopost: h._status_pending = False
    return c.wtheta_cell(code, 0, corr._k_lim[0], corr._k_lim[1], corr.D_z, theta, ell)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(30): step()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    print("nopost" if nopost else "post  ", "%.1f us per step" % ((time.perf_counter() - t0) / n * 1e6))
