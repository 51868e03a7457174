"""Development aid (GPU box): time chomp_cell / chomp_wtheta on subsets of the C4 / C5 sample
arrays, to see what sets a launch's duration.  Not part of the product."""
import contextlib, os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy, torch
from chomp_amd import cosmology, correlation, halo, kernel
d2r = numpy.pi / 180.0


_burn = None


def timed(fn, n=20):
    # (clocks: a few tens of ms of load first, or a short burst is timed at idle clocks)
    global _burn
    if _burn is None:
        _burn = torch.randn(4096, 4096, device="cuda")
    for _ in range(40):
        _burn @ _burn
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
  for ggl in (False, True):
      cm = cosmology.MultiEpoch(0.0, 5.0)
      with contextlib.redirect_stdout(sys.stderr):
          wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
          if ggl:
              wb = kernel.WindowFunctionConvergence(kernel.dNdzGaussian(0.0, 2.0, 1.0, 0.2), cm)
              kern = kernel.GalaxyGalaxyLensingKernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
              h, spec = halo.HaloFit(0.0), "power_gm"
          else:
              wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
              kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
              h, spec = halo.Halo(0.0), "power_gg"
      with warnings.catch_warnings():
          warnings.simplefilter("ignore")
          corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec=spec)
          ctx, code = corr._prepare()
      ell = torch.logspace(1, 4, 2048, dtype=torch.float64, device="cuda")
      theta = torch.logspace(-3, 0, 1024, dtype=torch.float64, device="cuda") * d2r
      print("== c5" if ggl else "== c4")
      with torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream_ptr)) if hasattr(ctx, "stream_ptr") and ctx.stream_ptr else contextlib.nullcontext():
          for name, sub in (("all 2048", ell), ("first 1024", ell[:1024]), ("last 256", ell[-256:]), ("last 32", ell[-32:]),
                            ("last 1", ell[-1:]), ("first 1", ell[:1]), ("reversed", torch.flip(ell, [0]).contiguous())):
              print("  cell %-12s %8.1f us" % (name, timed(lambda: ctx.cell(code, 0, corr.D_z, sub))))
          for name, sub in (("all 1024", theta), ("first 256", theta[:256]), ("last 256", theta[-256:]), ("last 1", theta[-1:])):
              print("  wtheta %-10s %8.1f us" % (name, timed(lambda: ctx.wtheta(code, 0, corr._k_lim[0], corr._k_lim[1], corr.D_z, sub))))


if __name__ == "__main__":
    main()
