"""Experiment: capture one C2 step (Stage K + E) into a HIP graph and replay it."""
import os, sys, time, numpy, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from chomp_amd import grid
z = numpy.linspace(0.0, 1.5, 64)
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
k = torch.logspace(-3, 2, 4096, dtype=torch.float64, device="cuda")
out = torch.zeros((64, 4096), dtype=torch.float64, device="cuda")
hg = grid.HaloGrid(z, stream=s.cuda_stream)
def step():
    hg.setup("power_mm"); hg.power("power_mm", k, out=out)
for _ in range(5): step()
torch.cuda.synchronize()
ref = out.clone()
t = time.perf_counter(); n = 200
for _ in range(n): step()
torch.cuda.synchronize()
print("eager : %.4f ms/step" % ((time.perf_counter() - t) / n * 1e3), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step()      # two steps per graph: the Stage E slow-list counters ping-pong on a host-side
    step()      # parity bit, so one captured step would replay the same parity for ever
torch.cuda.synchronize()
out.zero_()
g.replay(); torch.cuda.synchronize()
print("graph result equal:", bool(torch.equal(out, ref)), flush=True)
t = time.perf_counter()
for _ in range(n // 2): g.replay()
torch.cuda.synchronize()
print("graph : %.4f ms/step" % ((time.perf_counter() - t) / n * 1e3), flush=True)
print("still equal:", bool(torch.equal(out, ref)))
