"""Where a reference-shaped step (set_redshift + power_mm, host arrays) spends its time (scratch)."""
import os, sys, time, warnings, cProfile, pstats, io
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy, torch
from chomp_amd import halo
k = numpy.logspace(-3, 2, 4096)
z = numpy.linspace(0.0, 1.5, 64)
h = halo.Halo(1.5)
warnings.simplefilter("ignore")
for zz in z[:16]:
    h.set_redshift(float(zz)); h.power_mm(k)
ts, tp = 0.0, 0.0
for rep in range(3):
    for zz in z:
        t0 = time.perf_counter(); h.set_redshift(float(zz)); t1 = time.perf_counter(); h.power_mm(k); t2 = time.perf_counter()
        ts += t1 - t0; tp += t2 - t1
n = 3 * 64
print("set_redshift %.1f us  power_mm %.1f us  total %.1f us per z" % (ts / n * 1e6, tp / n * 1e6, (ts + tp) / n * 1e6))
# the device side alone: set-up queued back to back, one sync
torch.cuda.synchronize()
t0 = time.perf_counter()
for zz in z:
    h.set_redshift(float(zz)); h._sync(1) if hasattr(h, "_sync") else None
torch.cuda.synchronize()
print("set_redshift + _sync(FAM_MM) queued back to back: %.1f us per z" % ((time.perf_counter() - t0) / 64 * 1e6))
pr = cProfile.Profile(); pr.enable()
for zz in z:
    h.set_redshift(float(zz)); h.power_mm(k)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
