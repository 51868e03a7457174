"""Development aid: which Romberg level do the w(theta) and C_l integrals of configs[3] / [4]
stop at?  `build` (here, no GPU) makes build_exp/levels.so = the library whose k_cell and
k_wtheta_fast return the LEVEL instead of the value; `run` (GPU box) prints the histograms.
Not part of the product."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/levels.so")


def build():
    P = os.path.join(R, "chomp_amd/csrc/chomp_proj_kernels.h")
    p0 = open(P).read()
    s = p0
    for old, new in (("  if (threadIdx.x == 0) out[blockIdx.x] = r.value[0];\n}\n\n// ---------------------------------------------------------------------------\n// Gaussian covariance",
                      "  if (threadIdx.x == 0) out[blockIdx.x] = (double)r.level[0];\n}\n\n// ---------------------------------------------------------------------------\n// Gaussian covariance"),
                     ("  if (threadIdx.x == 0) out[blockIdx.x] = R.value[0];", "  if (threadIdx.x == 0) out[blockIdx.x] = (double)R.level[0];")):
        assert old in s, old[:50]
        s = s.replace(old, new, 1)
    try:
        open(P, "w").write(s)
        os.makedirs(os.path.join(R, "build_exp"), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-o", SO, "chomp_capi.hip"], cwd=os.path.join(R, "chomp_amd/csrc"))
    finally:
        open(P, "w").write(p0)
    print("built", SO)


def run():
    sys.path.insert(0, R)
    from chomp_amd import _lib as _l
    _l.LIB_PATH = SO          # (the instrumented build instead of the product library ...)
    _l.build = lambda *a, **k: SO      # (... which must not be rebuilt over it)
    sys.path.insert(0, R)
    import contextlib, warnings
    import numpy
    from chomp_amd import cosmology, correlation, halo, kernel
    d2r = numpy.pi / 180.0
    theta = numpy.logspace(-3, 0, 1024) * d2r
    ell = numpy.logspace(1, 4, 2048)
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
            lw = ctx.wtheta(code, 0, corr._k_lim[0], corr._k_lim[1], corr.D_z, theta).astype(int)
            lc = ctx.cell(code, 0, corr.D_z, ell).astype(int)
        print("c5" if ggl else "c4", "chi range", kern.chi_min, kern.chi_max)
        print("  w(theta) levels:", dict(zip(*numpy.unique(lw, return_counts=True))))
        print("  C_l levels     :", dict(zip(*numpy.unique(lc, return_counts=True))))
        for lo in range(0, 2048, 256):
            print("   ell %8.1f..%8.1f  levels %s" % (ell[lo], ell[lo + 255], numpy.bincount(lc[lo:lo + 256], minlength=21)[8:]))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
