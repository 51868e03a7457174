"""Experiment: split the 64 epochs of C2 over L independent contexts on L streams."""
import sys, time, numpy, torch
sys.path.insert(0, ".")
from chomp_amd import grid
z = numpy.linspace(0.0, 1.5, 64)
k = torch.logspace(-3, 2, 4096, dtype=torch.float64, device="cuda")
for lanes in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    parts = [list(range(i, 64, lanes)) for i in range(lanes)]
    grids = [grid.HaloGrid(z[p], stream=s.cuda_stream) for p, s in zip(parts, streams)]
    outs = [torch.zeros((len(p), 4096), dtype=torch.float64, device="cuda") for p in parts]
    torch.cuda.synchronize()
    def step():
        for g, o in zip(grids, outs):
            g.setup("power_mm")
        for g, o in zip(grids, outs):
            g.power("power_mm", k, out=o)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 50
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("lanes %d: %.4f ms/step  %.3e samples/s" % (lanes, dt * 1e3, 64 * 4096 / dt), flush=True)
