"""Where do the device's Romberg stopping levels differ from the oracle's, and by how much do
the values?    Run on the GPU box."""
import os, sys, numpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import halo, grid, _lib
from oracle import chomp_oracle as o
import warnings
warnings.simplefilter("ignore")
names = ("_h_m_integrand", "_pp_mm_integrand", "_h_g_integrand", "_pp_gm_integrand", "_pp_gg_integrand")
tabs = ("h_m", "pp_mm", "h_g", "pp_gm", "pp_gg")
for z in (0.0, 0.5, 1.0):
    for mf in ("st", "tinker"):
        hg = grid.HaloGrid(numpy.array([z]), mass_function=mf)
        hg.ctx.epochs_set(hg._c_cosmo, hg._z)
        hg.ctx.stage_k(hg._c_halo, hg.kind, hg._c_halo, hg._c_hod, 31)      # all five tables
        lev = hg.ctx.table("levels").reshape(5, -1)
        e = o.epoch(None, z)
        t = o.halo_table(e, o.mass_table(e, kind=mf), o.zheng(), families=("mm", "gm", "gg"))
        for i, n in enumerate(names):
            ol = numpy.array(t.levels[n])
            d = numpy.nonzero(lev[i] != ol)[0]
            dv = hg.ctx.table(tabs[i])
            ov = {"h_m": t.h_m, "pp_mm": t.pp_mm, "h_g": t.h_g, "pp_gm": t.pp_gm, "pp_gg": t.pp_gg}[tabs[i]]
            rel = numpy.abs(dv / ov - 1)
            print("z=%.1f %-6s %-6s equal levels %2d/50  max rel diff (all knots) %.1e" % (z, mf, tabs[i], 50 - d.size, rel.max()), end="")
            if d.size:
                print("  differ at", [(int(k), int(lev[i][k]), int(ol[k]), "%.1e" % rel[k]) for k in d[:8]], end="")
            print()
