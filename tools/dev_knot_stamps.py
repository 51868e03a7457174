"""Development aid: build build_exp/knot_stamps.so = the library compiled with -DCHOMP_STAMPS
(s_memtime stamps at the phase boundaries of k_halo_knots_fast, first knot each block draws);
`python tools/dev_knot_stamps.py build` here, then `run` on the GPU box prints the
mean phase durations on configs[2] (64 epochs, power_gm, Tinker10) and on one epoch.
Not part of the product."""
import os, subprocess, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/knot_stamps.so")
NB, NS = 2048, 24


def build():
    """The library with -DCHOMP_STAMPS (the KSTAMP macros of chomp_halo_kernels.h)."""
    sys.path.insert(0, R)
    from chomp_amd import _lib
    os.makedirs(os.path.join(R, "build_exp"), exist_ok=True)
    print("built", _lib.build(extra_flags=["-DCHOMP_STAMPS"], out=SO))


def run():
    sys.path.insert(0, R)
    from chomp_amd import _lib as _l
    _l.LIB_PATH = SO          # (the instrumented build instead of the product library ...)
    _l.build = lambda *a, **k: SO      # (... which must not be rebuilt over it)
    sys.path.insert(0, R)
    import numpy, torch
    from chomp_amd import grid, _lib
    L = _lib.lib()
    L.chomp_debug_ks.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, ctypes.c_int]
    numpy.set_printoptions(linewidth=220, precision=1, suppress=True)
    for which, z, mf in (("power_gm", numpy.linspace(0.0, 1.5, 64), "tinker"), ("power_gg", numpy.array([0.3]), "st")):
        hg = grid.HaloGrid(z, mass_function=mf)
        for _ in range(3):
            hg.setup(which)
        torch.cuda.synchronize()
        L.chomp_debug_ks(None, 0, 1)
        hg.setup(which)
        torch.cuda.synchronize()
        out = (ctypes.c_longlong * (NB * NS))()
        L.chomp_debug_ks(out, NB * NS, 0)
        a = numpy.array(out[:], dtype=numpy.int64).reshape(NB, NS)
        a = a[a[:, 0] > 0]
        T = 2.4e3       # ticks per us (shader clock)
        print(which, mf, "blocks with a first knot:", len(a))
        names = ["stage", "coarse", "levels<=LC", "breaks+segments", "self-check"]
        for i, nm in enumerate(names):
            ok = a[:, i + 1] > 0
            d = (a[ok, i + 1] - a[ok, i]) / T
            print("  %-16s n %4d  mean %6.2f  max %6.2f us" % (nm, ok.sum(), d.mean() if ok.sum() else 0, d.max() if ok.sum() else 0))
        for lv in range(1, 5):
            ok = (a[:, 5 + lv] > 0) & (a[:, 4 + lv] > 0)
            if ok.sum():
                d = (a[ok, 5 + lv] - a[ok, 4 + lv]) / T
                print("  round %2d         n %4d  mean %6.2f  max %6.2f us   (nf mean %.1f)" % (lv, ok.sum(), d.mean(), d.max(), a[ok, 22].mean()))
        ok = (a[:, 10] > 0) & (a[:, 12] > 0) & (a[:, 6] > 0)
        if ok.sum():
            print("  inside round 1: stencil pass %.2f  node-by-node %.2f  exchange %.2f  rows %.2f us (mean)" % (
                ((a[ok, 10] - a[ok, 5]) / T).mean(), ((a[ok, 11] - a[ok, 10]) / T).mean(),
                ((a[ok, 12] - a[ok, 11]) / T).mean(), ((a[ok, 6] - a[ok, 12]) / T).mean()))
        ok = a[:, 16] > 0
        tot = (a[ok, 16] - a[ok, 0]) / T
        print("  whole knot       n %4d  mean %6.2f  max %6.2f us;  arrive %5.2f us" % (ok.sum(), tot.mean(), tot.max(), ((a[ok, 17] - a[ok, 16]) / T).mean()))
        print("  final level histogram 11..20:", numpy.bincount(a[ok, 23].astype(int), minlength=21)[11:])
        b = numpy.array(out[:], dtype=numpy.int64).reshape(NB, NS)
        b = b[b[:, 18] > 0]
        t0 = b[:, 18].min()
        print("  blocks %d: start %.1f..%.1f us, end mean %.1f max %.1f us; items per block: %s" % (
            len(b), 0.0, (b[:, 18].max() - t0) / T, ((b[:, 19] - t0) / T).mean(), (b[:, 19].max() - t0) / T,
            numpy.bincount(b[:, 20].astype(int))))
        # (s_memtime is per XCD: only differences inside one block mean anything)
        dur = (b[:, 19] - b[:, 18]) / T
        print("  block lifetimes (entry -> list empty), percentiles 10/50/90/99/100: %s; by items: %s" % (
            numpy.percentile(dur, [10, 50, 90, 99, 100]),
            [round(float(dur[b[:, 20] == k].mean()), 1) for k in range(int(b[:, 20].max()) + 1) if (b[:, 20] == k).any()]))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
