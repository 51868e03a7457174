"""Development aid: capture the bench's projection step into a HIP graph, full traceback."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
class A: rehearse=False; no_graph=False
D = bench.Dist(A(), 1, 0, 0, torch.device("cuda", 0))
stream = torch.cuda.Stream(D.dev); torch.cuda.set_stream(stream)
os.environ["HIP_LAUNCH_BLOCKING"] = "0"
try:
    print(bench.projection_leg(D, len(sys.argv) > 1, 5, 2))
except Exception:
    traceback.print_exc()
