"""Stage-E timing sweep on the 2^20 x 64 grid (scratch)."""
import os, sys, numpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chomp_amd import grid
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
z = numpy.linspace(0.0, 1.5, 64)
hg = grid.HaloGrid(z, device=0, stream=stream.cuda_stream)
hg.setup("power_mm")
nk = int(os.environ.get("STAGE_E_NK", 1 << 20))
k = torch.logspace(-3, 2, nk, dtype=torch.float64, device=dev)
buf = torch.empty((64, nk), dtype=torch.float64, device=dev)
def timed(reps=200):
    buf.zero_()
    hg.power("power_mm", k, out=buf); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps): hg.power("power_mm", k, out=buf)
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
by = 8.0 * nk + 8.0 * 64 * nk
os.environ["CHOMP_E_STREAM_MIN"] = str(1 << 62)
us = timed(); ref = buf.clone()
print("row-walk path      %7.1f us  %7.1f GB/s" % (us, by / us / 1e3), flush=True)
os.environ["CHOMP_E_STREAM_MIN"] = "0"
pers = [2] if os.environ.get("STAGE_E_ONE") else [1, 2, 4, 2, 4, 2, 4]
for per in pers:
    os.environ["CHOMP_E_PER"] = str(per)
    us = timed()
    ok = bool((buf == ref).all())
    print("stream per %2d  %7.1f us  %7.1f GB/s  same=%s  maxrel=%.2e" % (per, us, by / us / 1e3, ok, float(((buf - ref).abs() / ref.abs().clamp_min(1e-300)).max())), flush=True)
bad = (buf != ref).nonzero()
print("n bad", bad.shape[0])
if bad.shape[0]:
    rows = torch.unique(bad[:, 0]); cols = torch.unique(bad[:, 1])
    print("rows", rows[:20].tolist(), "ncols", cols.numel(), "cols", cols[:10].tolist(), cols[-5:].tolist())
    i, j = bad[0].tolist(); print(i, j, float(buf[i, j]), float(ref[i, j]))
