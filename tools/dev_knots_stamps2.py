"""Development aid: per-wavefront timing of k_halo_knots<1> on configs[1] (KNSTAMP macros,
-DCHOMP_STAMPS=2 build).  `build` here, `run` on the GPU box.  Not part of the product."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/knots2_stamps.so")


def build():
    sys.path.insert(0, R)
    from chomp_amd import _lib
    os.makedirs(os.path.join(R, "build_exp"), exist_ok=True)
    print("built", _lib.build(extra_flags=["-DCHOMP_STAMPS=2"], out=SO))


def run():
    sys.path.insert(0, R)
    from chomp_amd import _lib as _l
    _l.LIB_PATH = SO
    _l.build = lambda *a, **k: SO
    import numpy, torch
    from chomp_amd import grid, _lib
    L = _lib.lib()
    L.chomp_debug_ms.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, ctypes.c_int]
    numpy.set_printoptions(linewidth=220, precision=1, suppress=True)
    hg = grid.HaloGrid(numpy.linspace(0.0, 1.5, 64))
    for _ in range(3):
        hg.setup("power_mm")
    torch.cuda.synchronize()
    L.chomp_debug_ms(None, 0, 1)
    hg.setup("power_mm")
    torch.cuda.synchronize()
    n = 64 * 14 * 4 * 4
    out = (ctypes.c_longlong * n)()
    L.chomp_debug_ms(out, n, 0)
    a = numpy.array(out[:], dtype=numpy.int64).reshape(64, 14, 4, 4)     # epoch, block x, wave, slot
    T = 100.0                                    # s_memtime: 100 MHz constant counter? (calibrated below)
    ok = a[..., 2] > 0
    t0 = a[..., 0][a[..., 0] > 0].min()
    start = (a[..., 0] - t0) / T
    staged = (a[..., 1] - a[..., 0]) / T
    dur = (a[..., 2] - a[..., 1]) / T
    end = (a[..., 2] - t0) / T
    lev = a[..., 3] % 100
    print("waves with a knot:", ok.sum(), " launch span (first entry -> last end): %.1f ticks/T" % end[ok].max())
    print("entry time percentiles 0/10/50/90/100:", numpy.percentile(start[ok], [0, 10, 50, 90, 100]))
    print("staging mean %.2f max %.2f" % (staged[ok].mean(), staged[ok].max()))
    for l in range(5, 11):
        m = ok & (lev == l)
        if m.sum():
            print("level %2d: n %4d  romberg mean %6.2f  max %6.2f   end mean %6.2f max %6.2f   entry mean %6.2f" % (
                l, m.sum(), dur[m].mean(), dur[m].max(), end[m].mean(), end[m].max(), start[m].mean()))
    print("end-time percentiles 50/90/99/100:", numpy.percentile(end[ok], [50, 90, 99, 100]))
    # by dispatch order: blockIdx.x (0 = dispatched first = highest k)
    for bx in range(14):
        m = ok[:, bx]
        if m.sum():
            print("block x %2d: entry mean %6.2f  end mean %6.2f max %6.2f" % (bx, start[:, bx][m].mean(), end[:, bx][m].mean(), end[:, bx][m].max()))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
