import numpy, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import grid
hg = grid.HaloGrid(numpy.linspace(0, 1.5, 64), mass_function="tinker")
hg.setup("power_gm")
hg.ctx.sync()
names = ("h_m", "pp_mm", "h_g", "pp_gm", "pp_gg")
L = numpy.array([hg.ctx.table("levels", i).reshape(5, -1) for i in range(64)])
for f in (0, 2, 3):
    print(names[f], numpy.bincount(L[:, f].astype(int).ravel(), minlength=21)[5:])
# pair level = max over (h_g, pp_gm)
pair = numpy.maximum(L[:, 2], L[:, 3]).astype(int)
print("pair(h_g,pp_gm)", numpy.bincount(pair.ravel(), minlength=21)[5:])
print("listed pairs (level>10):", (pair > 10).sum(), " converged at 9-10:", ((pair >= 9) & (pair <= 10)).sum(), " at 7-8:", ((pair>=7)&(pair<=8)).sum(), " at <=6:", (pair<=6).sum())
pm = numpy.maximum(L[:, 0], L[:, 1]).astype(int)
print("pair(h_m,pp_mm)", numpy.bincount(pm.ravel(), minlength=21)[5:])
