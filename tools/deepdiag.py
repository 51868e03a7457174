import os, sys, numpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import grid, _lib
z = numpy.array([0.0, 0.1, 0.55, 1.2, 1.5])
k = numpy.logspace(-3, 2, 50)
hods = [dict(log_M_min=12.14, sigma=0.15, log_M_0=12.14, log_M_1p=13.43, alpha=1.0),
        dict(log_M_min=11.9, sigma=0.35, log_M_0=12.3, log_M_1p=13.1, alpha=0.8),
        dict(log_M_min=12.5, sigma=0.0, log_M_0=12.0, log_M_1p=13.6, alpha=1.2)]
for mf in ("st", "tinker"):
    for hd in hods:
        for which in ("power_gm", "power_gg"):
            g = grid.HaloGrid(z, mass_function=mf, hod_dict=hd)
            g.power(which, k)
            print(mf, hd["alpha"], hd["sigma"], which, g.ctx.deep_stats(), g.ctx.deep_detail)
