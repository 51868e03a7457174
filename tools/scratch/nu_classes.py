"""ln R against the Romberg level / path of every integral of configs[1]'s nu table (scratch; the
-DCHOMP_STAMPS=3 build of tools/dev_nu_stamps3.py)."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(R, "build_exp/nu3_stamps.so")
sys.path.insert(0, R)
from chomp_amd import _lib as _l
_l.LIB_PATH = SO
_l.build = lambda *a, **k: SO
import numpy, torch
from chomp_amd import grid, _lib
L = _lib.lib()
L.chomp_debug_ms.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, ctypes.c_int]
z = numpy.linspace(0.0, 1.5, 64)
hg = grid.HaloGrid(z)
for _ in range(2):
    hg.setup("power_mm")
torch.cuda.synchronize()
L.chomp_debug_ms(None, 0, 1)
hg.setup("power_mm")
torch.cuda.synchronize()
n = 64 * 50 * 4
out = (ctypes.c_longlong * n)()
L.chomp_debug_ms(out, n, 0)
a = numpy.array(out[:], dtype=numpy.int64).reshape(64, 50, 4)
NM = 50
def y_to_i(y):
    m0 = NM // 2 - 1
    if y < 2: return m0 + y
    if y >= NM - 2: return m0 + 2 + (y - (NM - 2))
    i = NM - 1 - (y - 2)
    if i <= m0 + 3: i -= 4
    return i
rows = []
for e in range(64):
    sc = hg.ctx.scalars(e)
    lnm = numpy.linspace(sc["ln_mass_min"], sc["ln_mass_max"], NM)
    lnR = (lnm + numpy.log(3.0 / (4.0 * numpy.pi * sc["rho_bar"]))) / 3.0
    for y in range(NM):
        i = y_to_i(y)
        rows.append((e, i, lnR[i], int(a[e, y, 1])))
rows = numpy.array(rows)
numpy.save(os.path.join(R, "gpurun_out", "nu_classes.npy"), rows)
for l in sorted(set(rows[:, 3])):
    m = rows[:, 3] == l
    print(int(l), int(m.sum()), "ln R from %.3f to %.3f" % (rows[m, 2].min(), rows[m, 2].max()))
