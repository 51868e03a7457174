"""Development aid (GPU box): a configs[2]-sized batch with a Zheng HOD whose satellite slope is
not 1 (the deep-level sums then evaluate the margin above the singular onset node by node: the
EVAL instance of k_halo_knots_fast as the main pass), timed; and one epoch of the same."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from chomp_amd import grid, defaults
hod = dict(defaults.default_hod_dict, alpha=0.9)
k = torch.logspace(-3, 2, 4096, dtype=torch.float64, device="cuda")
for z in (numpy.linspace(0.0, 1.5, 64), numpy.array([0.3])):
    hg = grid.HaloGrid(z, mass_function="tinker", hod_dict=hod)
    out = torch.empty((len(z), 4096), dtype=torch.float64, device="cuda")
    for _ in range(5):
        hg.setup("power_gm"); hg.power("power_gm", k, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        hg.setup("power_gm"); hg.power("power_gm", k, out=out)
    torch.cuda.synchronize()
    print("alpha = 0.9, %2d epochs: %.4f ms per step; deep stats %s %s" % (
        len(z), (time.perf_counter() - t0) / 30 * 1e3, hg.ctx.deep_stats(), hg.ctx.deep_detail))
