"""Experiment: split the 64 epochs of C2 over L independent contexts on L streams, fenced
against the caller's stream with events (what a HaloGrid with lanes would have to do)."""
import os, sys, time, numpy, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import grid
z = numpy.linspace(0.0, 1.5, 64)
main = torch.cuda.Stream()
torch.cuda.set_stream(main)
k = torch.logspace(-3, 2, 4096, dtype=torch.float64, device="cuda")
out = torch.zeros((64, 4096), dtype=torch.float64, device="cuda")
for lanes in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    bounds = numpy.linspace(0, 64, lanes + 1).astype(int)
    grids = [grid.HaloGrid(z[a:b], stream=s.cuda_stream) for a, b, s in zip(bounds[:-1], bounds[1:], streams)]
    ev_in = torch.cuda.Event()
    ev_out = [torch.cuda.Event() for _ in range(lanes)]
    torch.cuda.synchronize()
    def step():
        ev_in.record(main)
        for g, s, a, b, e in zip(grids, streams, bounds[:-1], bounds[1:], ev_out):
            s.wait_event(ev_in)
            g.setup("power_mm")
        for g, s, a, b, e in zip(grids, streams, bounds[:-1], bounds[1:], ev_out):
            g.power("power_mm", k, out=out[a:b])
            e.record(s)
            main.wait_event(e)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t = time.perf_counter(); n = 100
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("lanes %d: %.4f ms/step  %.3e samples/s" % (lanes, dt * 1e3, 64 * 4096 / dt), flush=True)
