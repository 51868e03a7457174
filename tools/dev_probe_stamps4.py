"""Development aid: per-block phase times of k_epoch_probe<., 0> on configs[1] (PSTAMP macros,
-DCHOMP_STAMPS=4 build; s_memrealtime, 100 MHz).  `build` here, `run` on the GPU box."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/probe4_stamps.so")


def build():
    sys.path.insert(0, R)
    from chomp_amd import _lib
    os.makedirs(os.path.join(R, "build_exp"), exist_ok=True)
    print("built", _lib.build(extra_flags=["-DCHOMP_STAMPS=4"], out=SO))


def run():
    sys.path.insert(0, R)
    from chomp_amd import _lib as _l
    _l.LIB_PATH = SO
    _l.build = lambda *a, **k: SO
    import numpy, torch
    from chomp_amd import grid, _lib
    L = _lib.lib()
    L.chomp_debug_ps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, ctypes.c_int]
    numpy.set_printoptions(linewidth=220, precision=1, suppress=True)
    hg = grid.HaloGrid(numpy.linspace(0.0, 1.5, 64))
    for _ in range(3):
        hg.setup("power_mm")
    torch.cuda.synchronize()
    L.chomp_debug_ps(None, 0, 1)
    hg.setup("power_mm")
    torch.cuda.synchronize()
    n = 64 * 8 * 8
    out = (ctypes.c_longlong * n)()
    L.chomp_debug_ps(out, n, 0)
    a = numpy.array(out[:], dtype=numpy.int64).reshape(64, 8, 8).astype(float)   # epoch, blockIdx.y, slot
    a[a == 0] = numpy.nan
    t0 = numpy.nanmin(a[..., 0])
    rel = (a - t0) / 100.0
    print("slots: 0 entry, 1 staged, 2 (probe roles) before plan, 3 planned, 4 probed, 5 arrived, 6 end (last block)")
    print("mean over epochs by blockIdx.y (0..3: mass_max side p = 3..0, 4..7: mass_min side):")
    print(numpy.nanmean(rel, axis=0)[:, :7])
    print("max over epochs:")
    print(numpy.nanmax(rel, axis=0)[:, :7])
    print("span %.1f us" % numpy.nanmax(rel))
    pd = rel[..., 4] - rel[..., 3]
    print("probe durations (planned -> probed) by epoch (rows) and blockIdx.y (columns), every epoch:")
    print(pd)
    sc = hg.ctx.scalars
    print("probe counts n_search by epoch:", [int(sc(i)["n_search"]) for i in range(64)])


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
