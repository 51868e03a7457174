"""Development aid (GPU box): is the configs[1] step host- or device-bound?  Host time to queue a
step (no synchronisation) against the time the device takes, and the cost of each host call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from chomp_amd import grid
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
hg = grid.HaloGrid(numpy.linspace(0.0, 1.5, 64), device=0, stream=stream.cuda_stream)
k = torch.logspace(-3, 2, 4096, dtype=torch.float64, device=dev)
out = torch.empty((64, 4096), dtype=torch.float64, device=dev)
def step():
    hg.setup("power_mm"); hg.power("power_mm", k, out=out)
for _ in range(20): step()
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host queues a step in %.1f us; device finishes %.1f us per step" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
for name, fn in (("epochs_set", lambda: hg.ctx.epochs_set(hg._c_cosmo, hg._z)),
                 ("stage_k", lambda: hg.ctx.stage_k(hg._c_halo, hg.kind, hg._c_halo, hg._c_hod, 3)),
                 ("power", lambda: hg.power("power_mm", k, out=out))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("  %-10s host %.1f us per call" % (name, (t1 - t0) / 200 * 1e6))
