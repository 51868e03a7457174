"""Stage K + E time per step against the number of epochs in the batch (one cosmology)."""
import sys, time, numpy, torch
sys.path.insert(0, ".")
from chomp_amd import grid
k = torch.logspace(-3, 2, 4096, dtype=torch.float64, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for n in (16, 32, 64, 128, 256, 512, 1024, 2048):
        z = numpy.linspace(0.0, 1.5, n)
        g = grid.HaloGrid(z, stream=s.cuda_stream)
        out = torch.zeros((n, 4096), dtype=torch.float64, device="cuda")
        def step():
            g.setup("power_mm"); g.power("power_mm", k, out=out)
        for _ in range(3): step()
        torch.cuda.synchronize()
        t = time.perf_counter(); reps = 20
        for _ in range(reps): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        print("n_epoch %5d: %.4f ms/step  %.3f us/epoch  %.3e samples/s" % (n, dt*1e3, dt*1e6/n, n*4096/dt), flush=True)
        del g, out
