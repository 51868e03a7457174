import os, sys, numpy, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import grid, _lib
hg = grid.HaloGrid(numpy.linspace(0.0, 1.5, 6), mass_function="tinker")
hg.setup("power_gg")
k = torch.logspace(-3, 2, 1 << 16, dtype=torch.float64, device="cuda")
bad = k.clone(); bad[::1000] = float("nan"); bad[1::1000] = 0.0; bad[2::1000] = -1.0; bad[3::1000] = float("inf")
for name, kk in (("sorted", k), ("bad", bad)):
    for which in ("power_mm",):
        hg.ctx.set_tuning(_lib.TUNE_E_STREAM_MIN, 0); a = hg.power(which, kk).clone()
        hg.ctx.set_tuning(_lib.TUNE_E_STREAM_MIN, 1 << 62); b = hg.power(which, kk).clone()
        torch.cuda.synchronize()
        d = (torch.nan_to_num(a, nan=-7.0) != torch.nan_to_num(b, nan=-7.0)).nonzero()
        print(name, which, "mismatches", d.shape[0], d[:10].tolist())
        for r, c in d[:6].tolist():
            print("   ", r, c, float(kk[c]), float(a[r, c]), float(b[r, c]))
