import sys, ctypes; sys.path.insert(0,'.')
import numpy as np
from chomp_amd import grid, _lib
hg = grid.HaloGrid(np.linspace(0,1.5,64)); hg.setup('power_mm'); hg.ctx.sync()
hg.setup('power_mm'); hg.ctx.sync()
for e in (0, 21, 42, 63):
    out = np.empty(32)
    hg.ctx._check(hg.ctx._L.chomp_get_table(hg.ctx._h, e, 99, out.ctypes.data_as(_lib.c_double_p), 50))
    for side in (0,1):
        t = out[side*16:side*16+12]; t = t[t>=0]
        print(e, side, np.diff(t).round(0).astype(int)*10//1000, "x10ns->us") 
