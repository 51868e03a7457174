import os, sys, numpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import grid
from oracle import chomp_oracle as o
rng = numpy.random.default_rng(11)
n = 40
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
sel = [0, 7, 13, 1]
g = grid.HaloGrid(numpy.array([zs[i] for i in sel]), cosmo_dict=[cos[i] for i in sel])
g.setup("power_mm")
for q, i in enumerate(sel):
    sc = g.ctx.scalars(q)
    e = o.epoch(cos[i], float(zs[i]))
    m = o.mass_table(e)
    print("case", i, "z %.3f s8 %.3f om %.3f" % (zs[i], cos[i]["sigma_8"], cos[i]["omega_m0"]))
    print("   device ln_mass_min %.6f max %.6f" % (sc["ln_mass_min"], sc["ln_mass_max"]))
    print("   oracle ln_mass_min %.6f max %.6f" % (m.ln_mass[0], m.ln_mass[-1]), " steps apart: %.2f %.2f" % ((sc["ln_mass_min"]-m.ln_mass[0])/numpy.log(1.05), (sc["ln_mass_max"]-m.ln_mass[-1])/numpy.log(1.05)))
