"""Random cosmology / HOD / z: Correlation3d.raw_correlation (xi(r), the reference's cylindrical-J0 form)
and Halo(extrapolate=True) spectra beyond the k limits against the oracle (scratch soak)."""
import os, sys, warnings, numpy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from chomp_amd import correlation, cosmology, halo, hod
from oracle import chomp_oracle as o
rng = numpy.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
warnings.simplefilter("ignore")
r = numpy.logspace(-1, numpy.log10(50.0), 6)
worst = 0.0
for i in range(n):
    c = dict(o.default_cosmo_dict)
    c["omega_m0"] = rng.uniform(0.24, 0.34) - c["omega_r0"]
    c["omega_l0"] = 1.0 - c["omega_m0"] - c["omega_r0"]
    c["h"] = rng.uniform(0.65, 0.75)
    c["sigma_8"] = rng.uniform(0.75, 0.85)
    z = float(rng.uniform(0.0, 1.0))
    hd = dict(o.default_hod_dict)
    hd["log_M_min"] = rng.uniform(11.9, 12.5); hd["log_M_0"] = hd["log_M_min"]
    hd["sigma"] = rng.uniform(0.12, 0.35); hd["log_M_1p"] = hd["log_M_min"] + rng.uniform(1.0, 1.4)
    ps, fam = (("power_mm", "mm"), ("power_gm", "gm"), ("power_gg", "gg"))[i % 3]
    h = halo.Halo(z, input_hod=hod.HODZheng(hd), cosmo_single_epoch=cosmology.SingleEpoch(z, c))
    c3 = correlation.Correlation3d(0.1, 50.0, redshift=z, input_halo=h, powSpec=ps)
    got = c3.raw_correlation(r)
    e = o.epoch(c, z)
    t = o.halo_table(e, o.mass_table(e), o.zheng(hd), families=(fam,))
    ref = o.xi3d_raw(lambda k: o.halo_power(t, fam, k), r, t.k_min, t.k_max)
    err = float(numpy.max(numpy.abs(got - ref)) / numpy.max(numpy.abs(ref)))
    worst = max(worst, err)
    print("case %2d %s z=%.3f  xi %.2e  status 0x%x" % (i, ps, z, err, h.status), flush=True)
print("worst %.3e" % worst)
sys.exit(1 if worst > 1e-6 else 0)
