import os, sys, numpy, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chomp_amd import grid, _lib
z = numpy.linspace(0.0, 1.5, 64)
hg = grid.HaloGrid(z, device=0)
for it in range(3):
    hg.setup("power_mm"); hg._tables = 0
hg.ctx.sync()
out = numpy.zeros(64 * 96)
L = _lib.lib()
L.chomp_debug_probe.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_size_t]
L.chomp_debug_probe(hg.ctx._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.size)
out = out.reshape(64, 96)
t = out[:, 24:64].reshape(64, 8, 5)
t0 = t[:, :, 0].min()
start = (t[:, :, 0] - t0) / 100.0
print("block start offsets us: min %.1f median %.1f max %.1f" % (start.min(), numpy.median(start), start.max()))
print("E-ready median %.1f max %.1f | chi (role 4) median %.1f | plan median %.1f max %.1f | probe lower median %.1f max %.1f upper median %.1f max %.1f" % (
    numpy.median(t[:, :, 1]) / 100, t[:, :, 1].max() / 100, numpy.median(t[:, 4, 4]) / 100,
    numpy.median(t[:, :, 2]) / 100, t[:, :, 2].max() / 100,
    numpy.median(t[:, 0:4, 3]) / 100, t[:, 0:4, 3].max() / 100, numpy.median(t[:, 4:8, 3]) / 100, t[:, 4:8, 3].max() / 100))
end_all = (out[:, 64] - t0) / 100.0
tot = start + (t[:, :, 1] + t[:, :, 2] + t[:, :, 3] + t[:, :, 4]) / 100.0
print("probe blocks done: max %.1f us; finisher end: median %.1f max %.1f (epoch %d, role %d)" % (tot.max(), numpy.median(end_all), end_all.max(), end_all.argmax(), out[end_all.argmax(), 65]))
