import os, sys, numpy, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chomp_amd import grid, _lib
z = numpy.linspace(0.0, 1.5, 64)
hg = grid.HaloGrid(z, device=0)
for it in range(3):
    hg.ctx.epochs_set(hg.cosmo_dicts if hasattr(hg, "cosmo_dicts") else None, z) if False else hg.setup("power_mm")
    hg._tables = 0
hg.ctx.sync()
out = numpy.zeros(64 * 96)
L = _lib.lib()
L.chomp_debug_probe.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_size_t]
L.chomp_debug_probe(hg.ctx._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.size)
out = out.reshape(64, 96)
t = out[:, 24:60].reshape(64, 9, 4)
t0 = t[:, :, 0].min()
start = (t[:, :, 0] - t0) / 100.0
dur = t[:, :, 3] / 100.0
print("block start offsets us: min %.1f median %.1f max %.1f" % (start.min(), numpy.median(start), start.max()))
print("E-ready us median %.1f max %.1f; plan us median %.1f max %.1f" % (numpy.median(t[:, :, 1]) / 100, t[:, :, 1].max() / 100, numpy.median((t[:, :8, 2] - t[:, :8, 1])) / 100, (t[:, :8, 2] - t[:, :8, 1]).max() / 100))
print("probe us (roles 0-3 lower side) median %.1f max %.1f; (roles 4-7 upper) median %.1f max %.1f; chi median %.1f max %.1f" % (
    numpy.median(t[:, 0:4, 3] - t[:, 0:4, 2]) / 100, (t[:, 0:4, 3] - t[:, 0:4, 2]).max() / 100,
    numpy.median(t[:, 4:8, 3] - t[:, 4:8, 2]) / 100, (t[:, 4:8, 3] - t[:, 4:8, 2]).max() / 100,
    numpy.median(t[:, 8, 3] - t[:, 8, 1]) / 100, (t[:, 8, 3] - t[:, 8, 1]).max() / 100))
end_all = (out[:, 60] - t0) / 100.0
print("block end (start+dur) max %.1f us; finisher end: median %.1f max %.1f (epoch %d, role %d)" % ((start + dur).max(), numpy.median(end_all), end_all.max(), end_all.argmax(), out[end_all.argmax(), 61]))
slow = numpy.argsort(-(start + dur).ravel())[:6]
print("slowest blocks (epoch, role, start, dur):", [(int(i // 9), int(i % 9), round(float(start.ravel()[i]), 1), round(float(dur.ravel()[i]), 1)) for i in slow])
