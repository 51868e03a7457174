import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from chomp_amd import grid, defaults
def run(z, tag):
    hg = grid.HaloGrid(np.array([z]))
    hg.ctx.epochs_set(hg.cosmo, hg.z); hg.ctx.sync()
    t0=time.perf_counter()
    for _ in range(50): hg.ctx.epochs_set(hg.cosmo, hg.z)
    hg.ctx.sync(); t1=time.perf_counter()
    print(tag, z, "epochs_set %.1f us" % ((t1-t0)/50*1e6), int(hg.ctx.scalars(0)['n_search']))
for z in (0.0, 1.5): run(z, "search ")
defaults.default_limits["mass_min"]=1e9; defaults.default_limits["mass_max"]=1e16
for z in (0.0, 1.5): run(z, "fixed  ")
