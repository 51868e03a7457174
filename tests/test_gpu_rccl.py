"""The RCCL leg of the multi-GPU path on the hardware a one-GPU box has: ONE rank.  RCCL refuses
two ranks on one device ("Duplicate GPU detected"), so what can run here is a real RCCL
communicator of world size 1 with the collectives of chomp_amd/grid.py sent through it anyway
(collective_at_world_1): librccl loads, the process group comes up on the device, the
all_gather_into_tensor calls take the persistent buffers and the context's stream ordering they
take at N > 1, and the gathered grid equals the rank's own.  The N > 1 index logic is covered by
the gloo tests (tests/test_distributed_cpu.py)."""
import os
import socket
import sys

import numpy
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _rank(rank, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from chomp_amd import grid
    z = numpy.linspace(0.0, 1.5, 6)
    k = torch.logspace(-3, 2, 512, dtype=torch.float64, device="cuda")
    plain = grid.HaloGrid(z, device=0)
    want = plain.power("power_mm", k).clone()
    g = grid.HaloGrid(z, device=0, rank=0, world=1, collective_at_world_1=True)
    got = []
    for _ in range(3):                       # (three steps: both buffer turns, one reused)
        pend = g.power_all_async("power_mm", k)
        g.setup("power_mm")                  # the next step's set-up queued behind the gather
        got.append(pend.wait().clone())
    torch.cuda.synchronize()
    ok = all(t.shape == want.shape and bool(torch.equal(t, want)) for t in got)
    # the 1-D sample axis of the projection workloads
    x = torch.linspace(1.0, 2.0, 33, dtype=torch.float64, device="cuda")
    mine = grid.shard_samples(x, 0, 1)
    full = grid.gather_samples(mine * mine, 33, 1, collective_at_world_1=True)
    ok = ok and bool(torch.equal(full, x * x))
    ret["ok"] = ok
    ret["backend"] = dist.get_backend()
    dist.destroy_process_group()


def test_rccl_single_rank_gathers():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_rank, args=(0, port, ret))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert ret.get("backend") == "nccl" and ret.get("ok"), dict(ret)
