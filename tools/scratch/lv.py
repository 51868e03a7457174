import sys, os, numpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import grid
hg = grid.HaloGrid(numpy.linspace(0, 1.5, 64)); hg.setup("power_mm")
hg.power("power_mm", numpy.logspace(-3, 2, 8))
L = numpy.array([hg.ctx.table("levels", i).reshape(5, -1)[[0, 1]] for i in range(64)])   # [64, 2, 50]
pair = L.max(axis=1)     # the pair walks to the deeper of the two
print("pair level histogram 5..10:", numpy.bincount(pair.astype(int).ravel(), minlength=11)[5:])
print("per knot index, mean pair level:"); numpy.set_printoptions(linewidth=200, precision=1)
print(pair.mean(axis=0))
print("deep (>7) knots per epoch:", (pair > 7).sum(axis=1))
print("fraction of knots > 7 by knot index:"); print((pair > 7).mean(axis=0))
