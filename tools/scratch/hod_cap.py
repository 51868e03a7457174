"""Development aid (GPU box): configs[2]'s step and one epoch of power_gg against the level at
which the HOD knots leave the node table (CHOMP_TUNE_HOD_CAP)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from chomp_amd import grid, _lib
k = torch.logspace(-3, 2, 4096, dtype=torch.float64, device="cuda")
for which, z, mf in (("power_gm", numpy.linspace(0.0, 1.5, 64), "tinker"), ("power_gg", numpy.array([0.3]), "st")):
    hg = grid.HaloGrid(z, mass_function=mf)
    out = torch.empty((len(z), 4096), dtype=torch.float64, device="cuda")
    for cap in (8, 9, 10):
        hg.ctx.set_tuning(_lib.TUNE_HOD_CAP, cap)
        for _ in range(5):
            hg.setup(which); hg.power(which, k, out=out)
        torch.cuda.synchronize()
        f0 = hg.ctx.deep_stats()[0]
        t0 = time.perf_counter()
        for _ in range(40):
            hg.setup(which); hg.power(which, k, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print("%s %2d epochs cap %2d: %.4f ms per step, %d knots listed per step" % (
            which, len(z), cap, dt * 1e3, (hg.ctx.deep_stats()[0] - f0) // 40))
