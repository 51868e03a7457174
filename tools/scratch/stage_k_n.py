import sys, numpy, torch
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import grid
n = int(sys.argv[1])
k = torch.logspace(-3, 2, 4096, dtype=torch.float64, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g = grid.HaloGrid(numpy.linspace(0.0, 1.5, n), stream=s.cuda_stream)
    out = torch.zeros((n, 4096), dtype=torch.float64, device="cuda")
    for _ in range(12):
        g.setup("power_mm"); g.power("power_mm", k, out=out)
    torch.cuda.synchronize()
