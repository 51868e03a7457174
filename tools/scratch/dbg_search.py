import os, sys, numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chomp_amd import grid
z = numpy.linspace(0.0, 1.5, 64)
hg = grid.HaloGrid(z, device=0)
hg.setup("power_mm")
print([int(hg.ctx.scalars(i)["n_search"]) for i in range(64)])
