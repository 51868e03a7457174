"""One case of tools/soak.py looked at closely (scratch): python tools/scratch/soak_case.py seed n case"""
import os, sys, numpy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from chomp_amd import grid, _lib
from oracle import chomp_oracle as o
seed, n, case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = numpy.random.default_rng(seed)
k = numpy.logspace(-3, 2, 40)
cos, zs, hods = [], [], []
for i in range(n):
    c = dict(o.default_cosmo_dict)
    c["omega_m0"] = rng.uniform(0.2, 0.4) - c["omega_r0"]
    c["omega_l0"] = 1.0 - c["omega_m0"] - c["omega_r0"]
    c["omega_b0"] = rng.uniform(0.035, 0.055)
    c["h"] = rng.uniform(0.6, 0.8)
    c["sigma_8"] = rng.uniform(0.7, 0.9)
    c["n_scalar"] = rng.uniform(0.92, 1.0)
    cos.append(c); zs.append(rng.uniform(0.0, 1.5))
    h = dict(o.default_hod_dict)
    h["log_M_min"] = rng.uniform(11.8, 12.6); h["log_M_0"] = h["log_M_min"]
    h["sigma"] = rng.uniform(0.1, 0.4); h["log_M_1p"] = h["log_M_min"] + rng.uniform(1.0, 1.5)
    hods.append(h)
e = o.epoch(cos[case], float(zs[case]))
lo, hi, _ = o.mass_limits(e)
m = o.mass_table(e)
tb = o.halo_table(e, m, o.zheng(hods[case]), families=("mm",))
ref = o.halo_power(tb, "mm", k)
print("oracle ln M limits", lo, hi, "sigma_norm", e.sigma_norm)
for label, idx, g in (("batch of %d" % n, case, grid.HaloGrid(numpy.array(zs), cosmo_dict=cos, hod_dict=hods)),
                      ("alone", 0, grid.HaloGrid(numpy.array([zs[case]]), cosmo_dict=[cos[case]], hod_dict=[hods[case]])),
                      ("batch of 100", case, grid.HaloGrid(numpy.array(zs[:100]), cosmo_dict=cos[:100], hod_dict=hods[:100]) if case < 100 else None)):
    if g is None:
        continue
    p = g.power("power_mm", k)
    sc = g.ctx.scalars(idx)
    st = g.status()[idx]
    nu = g.ctx.table("nu", idx)
    print("%-14s status 0x%x  ln M %r %r  (d lo %.3g d hi %.3g)  sigma_norm rel %.2e  nu rel %.2e  P rel %.2e" % (
        label, st, sc["ln_mass_min"], sc["ln_mass_max"], sc["ln_mass_min"] - lo, sc["ln_mass_max"] - hi,
        sc["sigma_norm"] / e.sigma_norm - 1, numpy.max(numpy.abs(nu / m.nu_arr - 1)), numpy.max(numpy.abs(p[idx] / ref - 1))))
    hm, pp = g.ctx.table("h_m", idx), g.ctx.table("pp_mm", idx)
    print("   h_m rel %.2e  pp_mm rel %.2e; levels h_m %s" % (numpy.max(numpy.abs(hm / tb.h_m - 1)) if hasattr(tb, "h_m") else -1,
          numpy.max(numpy.abs(pp / tb.pp_mm - 1)) if hasattr(tb, "pp_mm") else -1, ""))
