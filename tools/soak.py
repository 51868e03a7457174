"""Randomised parity soak: device P_mm / P_gm against the oracle for random cosmologies,
HODs and redshifts (not part of the test suite: the oracle takes seconds per case).
    python tools/soak.py [seed] [n] [n_gm] [st|tinker] [alpha]   (n_gm: how many of the cases also compare P_gm and P_gg,
                                                          default 6; the mass function, default st;
                                                          "alpha": the satellites' power-law index drawn from [0.8, 1.3] too)
Exit code 1 if any epoch that the status word does not flag differs by more than 1e-4."""
import os, sys, time, numpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import grid
from oracle import chomp_oracle as o
rng = numpy.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
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
if len(sys.argv) > 5 and sys.argv[5] == "alpha":
    rng_a = numpy.random.default_rng(1000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 7))
    for h in hods:
        h["alpha"] = float(rng_a.uniform(0.8, 1.3))
t = time.time()
kind = sys.argv[4] if len(sys.argv) > 4 else "st"
g = grid.HaloGrid(numpy.array(zs), cosmo_dict=cos, hod_dict=hods, mass_function=kind)
pm = g.power("power_mm", k)
from chomp_amd import _lib
status = g.status()
n_gm = min(n, int(sys.argv[3]) if len(sys.argv) > 3 else 6)
pg = g.power("power_gm", k)
pgg = g.power("power_gg", k)
worst, n_flag, n_unflagged_bad = 0.0, 0, 0
for i in range(n):
    e = o.epoch(cos[i], float(zs[i]))
    fam = ("mm", "gm", "gg") if i < n_gm else ("mm",)
    tb = o.halo_table(e, o.mass_table(e, kind=kind), o.zheng(hods[i]), families=fam)
    err = numpy.max(numpy.abs(pm[i] / o.halo_power(tb, "mm", k) - 1))
    msg = "case %2d z=%.3f  mm %.2e" % (i, zs[i], err)
    if i < n_gm:
        eg = numpy.max(numpy.abs(pg[i] / o.halo_power(tb, "gm", k) - 1))
        egg = numpy.max(numpy.abs(pgg[i] / o.halo_power(tb, "gg", k) - 1))
        msg += "  gm %.2e  gg %.2e" % (eg, egg)
        err = max(err, eg, egg)
    flagged = bool(status[i] & (_lib.ST_SATURATED | _lib.ST_MASS_SEARCH_EXHAUSTED))
    if flagged:
        msg += "  FLAGGED (status 0x%x: saturated mass-limit search)" % int(status[i])
        n_flag += 1
    else:
        worst = max(worst, err)
        if err > 1e-4:
            n_unflagged_bad += 1
    print(msg, flush=True)
print("worst unflagged %.3e; %d flagged; %d UNFLAGGED mismatches > 1e-4  (%.0f s)" % (
    worst, n_flag, n_unflagged_bad, time.time() - t))
sys.exit(1 if n_unflagged_bad else 0)
