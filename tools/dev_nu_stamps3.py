"""Development aid: per-block timing and placement of k_nu_table<., 1> on configs[1] (NUSTAMP
macros, -DCHOMP_STAMPS=3 build).  `build` here, `run` on the GPU box.  Not part of the product."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/nu3_stamps.so")


def build():
    sys.path.insert(0, R)
    from chomp_amd import _lib
    os.makedirs(os.path.join(R, "build_exp"), exist_ok=True)
    print("built", _lib.build(extra_flags=["-DCHOMP_STAMPS=3"], out=SO))


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
    n = 64 * 50 * 4
    out = (ctypes.c_longlong * n)()
    L.chomp_debug_ms(out, n, 0)
    a = numpy.array(out[:], dtype=numpy.int64).reshape(64, 50, 4)     # epoch, blockIdx.y, slot
    os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
    numpy.save(os.path.join(R, "gpurun_out", "nu_stamps.npy"), a)
    T = 100.0
    t0 = a[..., 0].min()
    start = (a[..., 0] - t0) / T
    dur = (a[..., 2] - a[..., 0]) / T
    lev = a[..., 1]
    print('levels by rank (rows: epochs 0, 21, 42, 63; 100+: interpolated integrand):')
    for e in (0, 21, 42, 63):
        print(e, lev[e, ::-1])
    end = (a[..., 2] - t0) / T
    print("span %.1f us; entry percentiles 0/50/90/100: %s; staging mean %.2f" % (
        end.max(), numpy.percentile(start, [0, 50, 90, 100]), 0.0))
    print("integral duration percentiles 0/10/50/90/100:", numpy.percentile(dur, [0, 10, 50, 90, 100]))
    print("by blockIdx.y (0 = largest mass, dispatched first): duration mean / end max")
    for y in range(0, 50, 5):
        print("  y %2d: entry %5.1f  dur mean %5.1f max %5.1f  end max %5.1f" % (y, start[:, y].mean(), dur[:, y].mean(), dur[:, y].max(), end[:, y].max()))
    hw = a[..., 3] & 0xffffffff
    xcc = (a[..., 3] >> 32) & 0xf
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ncu = len(numpy.unique(key))
    ks = key * 4 + simd
    cnt = numpy.bincount(ks.ravel().astype(int))
    cnt = cnt[cnt > 0]
    print("distinct CUs %d, SIMDs used %d; waves per SIMD histogram:" % (ncu, len(cnt)), numpy.bincount(cnt))
    # does duration correlate with the SIMD's load?
    load = numpy.bincount(ks.ravel().astype(int))[ks.astype(int)]
    for l in sorted(set(load.ravel())):
        m = load == l
        print("  SIMDs with %d waves: integrals %4d  dur mean %5.1f  end mean %5.1f max %5.1f" % (l, m.sum(), dur[m].mean(), end[m].mean(), end[m].max()))
    print("xcc histogram:", numpy.bincount(xcc.ravel().astype(int)))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
