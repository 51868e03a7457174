"""Random cosmologies / redshifts with the wiggle transfer function: Halo on SingleEpoch(with_bao=True)
P_mm, P_gm against the oracle (scratch soak): python tools/scratch/soak_bao.py [seed] [n]"""
import os, sys, warnings, numpy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from chomp_amd import cosmology, halo, hod
from oracle import chomp_oracle as o
rng = numpy.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
k = numpy.logspace(-3, 2, 60)
warnings.simplefilter("ignore")
worst = 0.0
for i in range(n):
    c = dict(o.default_cosmo_dict)
    c["omega_m0"] = rng.uniform(0.2, 0.4) - c["omega_r0"]
    c["omega_l0"] = 1.0 - c["omega_m0"] - c["omega_r0"]
    c["omega_b0"] = rng.uniform(0.035, 0.055)
    c["h"] = rng.uniform(0.6, 0.8)
    c["sigma_8"] = rng.uniform(0.7, 0.9)
    c["n_scalar"] = rng.uniform(0.92, 1.0)
    z = float(rng.uniform(0.0, 1.2))
    h = halo.Halo(z, cosmo_single_epoch=cosmology.SingleEpoch(z, c, with_bao=True))
    pm, pg = h.power_mm(k), h.power_gm(k)
    st = h.status
    e = o.epoch(c, z, with_bao=True)
    t = o.halo_table(e, o.mass_table(e), o.zheng(), families=("mm", "gm"))
    em = float(numpy.max(numpy.abs(pm / o.halo_power(t, "mm", k) - 1)))
    eg = float(numpy.max(numpy.abs(pg / o.halo_power(t, "gm", k) - 1)))
    flagged = bool(st & 0x9)
    print("case %2d z=%.3f  mm %.2e gm %.2e status 0x%x%s" % (i, z, em, eg, st, "  FLAGGED" if flagged else ""), flush=True)
    if not flagged:
        worst = max(worst, em, eg)
print("worst unflagged %.3e" % worst)
sys.exit(1 if worst > 1e-6 else 0)
