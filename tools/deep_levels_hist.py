"""Development aid (GPU box): at which Romberg level do the knots of configs[2] (power_gm,
Tinker10, 64 redshifts) and of one z = 0.3 epoch of power_gg stop?  Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from chomp_amd import grid
for which, z, mf in (("power_gm", numpy.linspace(0.0, 1.5, 64), "tinker"), ("power_gg", numpy.array([0.3]), "st"),
                     ("power_gm", numpy.array([0.3]), "st")):
    hg = grid.HaloGrid(z, mass_function=mf)
    hg.setup(which)
    names = ("h_m", "pp_mm", "h_g", "pp_gm", "pp_gg")
    tot = numpy.zeros((5, 21), dtype=int)
    for e in range(len(z)):
        lev = hg.ctx.table("levels", e).reshape(5, -1).astype(int)
        for f in range(5):
            tot[f] += numpy.bincount(numpy.clip(lev[f], 0, 20), minlength=21)
    print(which, mf, "epochs", len(z), "deep stats", hg.ctx.deep_stats(), hg.ctx.deep_detail)
    for f in range(5):
        if tot[f].sum():
            print("  %-6s levels 5..20: %s" % (names[f], tot[f][5:]))
