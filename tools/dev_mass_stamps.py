"""Development aid: phase durations inside k_mass_nodes (MSTAMP macros, -DCHOMP_STAMPS build).
`python tools/dev_mass_stamps.py build` here, `run` on the GPU box.  Not part of the product."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/knot_stamps.so")


def build():
    sys.path.insert(0, R)
    from chomp_amd import _lib
    os.makedirs(os.path.join(R, "build_exp"), exist_ok=True)
    print("built", _lib.build(extra_flags=["-DCHOMP_STAMPS"], out=SO))


def run():
    sys.path.insert(0, R)
    from chomp_amd import _lib as _l
    _l.LIB_PATH = SO
    _l.build = lambda *a, **k: SO
    import numpy, torch
    from chomp_amd import grid, _lib
    L = _lib.lib()
    L.chomp_debug_ms.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, ctypes.c_int]
    numpy.set_printoptions(linewidth=220, precision=2, suppress=True)
    names = ["epoch record staged", "inputs staged", "2 PCR splines", "thread-0 scalars",
             "normalisations", "publish", "halo constants", "node table chunk"]
    for which, z, mf in (("power_mm", numpy.linspace(0.0, 1.5, 64), "st"),
                         ("power_gm", numpy.linspace(0.0, 1.5, 64), "tinker"),
                         ("power_gg", numpy.array([0.3]), "st")):
        hg = grid.HaloGrid(z, mass_function=mf)
        for _ in range(3):
            hg.setup(which)
        torch.cuda.synchronize()
        L.chomp_debug_ms(None, 0, 1)
        hg.setup(which)
        torch.cuda.synchronize()
        n = 64 * 8 * 16
        out = (ctypes.c_longlong * n)()
        L.chomp_debug_ms(out, n, 0)
        a = numpy.array(out[:], dtype=numpy.int64).reshape(64 * 8, 16)
        a = a[a[:, 0] > 0]
        T = 2.4e3      # ticks per us (as tools/dev_knot_stamps.py: shader clock)
        print(which, mf, "blocks:", len(a))
        for i in range(1, 8):
            ok = (a[:, i] > 0) & (a[:, i - 1] > 0)
            if ok.sum():
                d = (a[ok, i] - a[ok, i - 1]) / T
                print("  %-22s mean %6.2f  max %6.2f us" % (names[i], d.mean(), d.max()))
        ok = a[:, 7] > 0
        tot = (a[ok, 7] - a[ok, 0]) / T
        print("  whole block            mean %6.2f  max %6.2f us" % (tot.mean(), tot.max()))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
