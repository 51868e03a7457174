import sys; sys.path.insert(0,'.')
import numpy as np
from chomp_amd import halo, _lib
g=np.load('tests/golden/g3_stages.npz')
h=halo.Halo(0.0); ctx=h._sync(_lib.FAM_MM)
x=ctx.table("ln_mass"); r=g['z000_ln_mass']
d=x-r; print(np.nonzero(d)[0], d[np.nonzero(d)[0]][:5])
