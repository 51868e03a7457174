import os, sys, ctypes, numpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from chomp_amd import grid, _lib
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
hg = grid.HaloGrid(numpy.linspace(0, 1.5, 64), stream=s.cuda_stream)
k = torch.logspace(-3, 2, 4096, dtype=torch.float64, device="cuda")
for _ in range(5):
    hg.setup("power_mm"); hg.power("power_mm", k)
torch.cuda.synchronize()
L = _lib.lib()
n = 64 * 8 * 8
out = (ctypes.c_longlong * n)()
L.chomp_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
print("rc", L.chomp_debug_stamps(out, n))
a = numpy.array(out[:], dtype=numpy.int64).reshape(64, 8, 8)
t0 = a[:, :, 0].min()
GHZ = 2.1
rel = numpy.where(a > 0, (a - t0) / (GHZ * 1e3), numpy.nan)      # shader cycles -> us (approx)
numpy.set_printoptions(linewidth=200, precision=1, suppress=True)
print("phases: 0 start, 1 staged, 2 planned, 3 probed, 4 arrived, 5 certified (last block), 6 searched, 7 end")
print("mean over blocks:", numpy.nanmean(rel, axis=(0, 1)))
print("max over blocks: ", numpy.nanmax(rel, axis=(0, 1)))
print("per role, mean over epochs (phases 0-4):"); print(numpy.nanmean(rel[:, :, :5], axis=0))
d = numpy.diff(rel[:, :, :5], axis=2)
print("phase durations, mean:", numpy.nanmean(d, axis=(0, 1)), " max:", numpy.nanmax(d, axis=(0, 1)))
print("last-block: certified-arrived mean %.1f  searched-certified mean %.1f max %.1f  end-searched %.1f" % (
    numpy.nanmean(rel[:, :, 5] - rel[:, :, 4]), numpy.nanmean(rel[:, :, 6] - rel[:, :, 5]),
    numpy.nanmax(rel[:, :, 6] - rel[:, :, 5]), numpy.nanmean(rel[:, :, 7] - rel[:, :, 6])))
pd = rel[:, :, 3] - rel[:, :, 2]
print("probe duration per role: mean", numpy.nanmean(pd, axis=0), "max", numpy.nanmax(pd, axis=0))
print("probe duration per epoch (max over roles), every 4th epoch:", numpy.nanmax(pd, axis=1)[::4])
print("side-0 probes per epoch (mean):", numpy.nanmean(pd[:, :4], axis=1)[::4])
print("side-1 probes per epoch (mean):", numpy.nanmean(pd[:, 4:], axis=1)[::4])
pl = rel[:, :, 2] - rel[:, :, 1]
print("plan duration per role mean", numpy.nanmean(pl, axis=0))
