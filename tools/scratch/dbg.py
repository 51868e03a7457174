import sys; sys.path.insert(0,'.')
import numpy as np
from chomp_amd import grid
g=np.load('tests/golden/g4_pmm_grid.npz')
hg = grid.HaloGrid(g["z"])
nk=64
got = hg.power("power_mm", g["k"][:nk]); ref=g["mm"][:, :nk]
for r in (0,2,3,4):
    print(r, got[r,:6], ref[r,:6])
# which ref element equals the wrong value?
v = got[3,0]; d = np.abs(g["mm"]/v-1); print(v, np.unravel_index(d.argmin(), d.shape), d.min())
v = got[3,2]; d = np.abs(g["mm"]/v-1); print(v, np.unravel_index(d.argmin(), d.shape), d.min())
