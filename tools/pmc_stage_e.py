#!/usr/bin/env python3
"""Workload for the PMC passes of the Stage-E kernel: configs[1] tables, then the
enlarged 2^20 k x 64 z grid evaluated 5 times.  Run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_stage_e.py
and again with --pmc WRITE_SIZE (TCC counters do not fit one pass)."""
import os
import sys

import numpy
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from chomp_amd import grid  # noqa: E402

nk = 1 << 20
hg = grid.HaloGrid(numpy.linspace(0.0, 1.5, 64), device=0)
k = torch.logspace(-3, 2, nk, dtype=torch.float64, device="cuda")
out = torch.empty((64, nk), dtype=torch.float64, device="cuda")
hg.setup("power_mm")
for _ in range(5):
    hg.power("power_mm", k, out=out)
hg.ctx.sync()
torch.cuda.synchronize()
print("done", float(out[0, 0]))
