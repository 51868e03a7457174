"""Multi-process CPU test (gloo, world_size 2 and 3) of the N > 1 path: epochs are
dealt to ranks interleaved, each rank fills its rows, ONE all-gather re-assembles
the (z, k) grid in the caller's z order on every rank."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_all, nk, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chomp_amd import grid
    idx = grid.shard_indices(n_all, rank, world)
    rpr = grid.rows_per_rank(n_all, world)
    local = torch.zeros((rpr, nk), dtype=torch.float64)
    for j, i in enumerate(idx):           # row of epoch i holds i*1000 + column
        local[j] = i * 1000.0 + torch.arange(nk, dtype=torch.float64)
    full = grid.gather_rows(local, n_all, world)
    want = (torch.arange(n_all, dtype=torch.float64)[:, None] * 1000.0 +
            torch.arange(nk, dtype=torch.float64)[None, :])
    ok = full.shape == want.shape and bool(torch.equal(full, want))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_all", [(2, 64), (2, 7), (3, 64)])
def test_shard_allgather_unshard(world, n_all):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_all, 33, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _sample_worker(rank, world, port, n_all, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chomp_amd import grid
    x = torch.linspace(1.0, 2.0, n_all, dtype=torch.float64)
    mine = grid.shard_samples(x, rank, world)
    ok = mine.shape[0] == grid.rows_per_rank(n_all, world)
    full = grid.gather_samples(mine * mine + rank * 0.0, n_all, world)     # "evaluate", gather
    ok = ok and full.shape[0] == n_all and bool(torch.equal(full, x * x))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_all", [(2, 1024), (2, 33), (3, 100)])
def test_sample_axis_shard_and_gather(world, n_all):
    """The theta / l axis of the projection workloads: interleaved
    samples, equal counts per rank, one all-gather, caller's order restored."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_sample_worker, args=(r, world, port, n_all, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)
