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


class _StubContext(object):
    """Stands in for the device context on a box without a GPU: same call surface as
    _lib.Context for what HaloGrid uses (epochs_set, stage_k, power, status), with a closed
    form in place of the halo model -- P(k; z, cosmology) = sigma_8 (1 + z) k -- so that the
    host logic of the N > 1 path (which rows a rank owns, what it hands the context, the row
    buffers, the all-gather, the re-ordering) is exercised end to end on CPU tensors."""

    def __init__(self):
        self.calls = []
        self.n_epoch = 0

    @staticmethod
    def pack_cosmo(c, n):
        return [c] * n if isinstance(c, dict) else list(c)

    pack_halo = pack_cosmo

    @staticmethod
    def pack_hod(h, n):
        return [h] * n if not isinstance(h, (list, tuple)) else list(h)

    def epochs_set(self, cosmo, z, with_bao=False):
        self.cosmo, self.z, self.n_epoch = list(cosmo), [float(x) for x in z], len(z)
        self.calls.append(("epochs_set", len(z)))

    def stage_k(self, mass_halo, mf_kind, profile, hods, tables):
        assert len(mass_halo) == len(profile) == len(hods) == self.n_epoch
        self.calls.append(("stage_k", int(tables)))

    def power(self, which, k, epoch0=0, n=None, out=None):
        n = self.n_epoch - epoch0 if n is None else n
        if out is None:
            out = torch.empty((n, k.numel()), dtype=torch.float64)
        for i in range(n):
            out[i] = self.cosmo[epoch0 + i]["sigma_8"] * (1.0 + self.z[epoch0 + i]) * k
        self.calls.append(("power", which, n))
        return out

    def status(self, epoch0=0, n=None):
        import numpy
        return numpy.zeros(self.n_epoch if n is None else n, dtype=numpy.uint32)


def _grid_worker(rank, world, port, n_all, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy
    from chomp_amd import cosmology, defaults, grid
    cosmology._context = lambda stream=None, device=None: _StubContext()
    z = numpy.linspace(0.0, 1.5, n_all)
    cds = [dict(defaults.default_cosmo_dict, sigma_8=0.7 + 0.01 * i) for i in range(n_all)]
    hg = grid.HaloGrid(z, cosmo_dict=cds, rank=rank, world=world)
    k = torch.logspace(-3, 2, 17, dtype=torch.float64)
    ok = hg.idx == list(range(rank, n_all, world))
    for step in range(3):                  # persistent row buffers alternate between calls
        hg.setup("power_mm")
        full = hg.power_all("power_mm", k)
        want = torch.stack([cds[i]["sigma_8"] * (1.0 + z[i]) * k for i in range(n_all)])
        ok = ok and full.shape == (n_all, 17) and bool(torch.equal(full, want))
    # what the context was asked for: this rank's epochs only, in this rank's order
    ok = ok and hg.ctx.z == [float(z[i]) for i in hg.idx]
    ok = ok and [c[0] for c in hg.ctx.calls[:3]] == ["epochs_set", "stage_k", "power"]
    ok = ok and hg.status().shape[0] == len(hg.idx)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_all", [(2, 64), (2, 7), (3, 8)])
def test_halo_grid_power_all_end_to_end(world, n_all):
    """HaloGrid.setup + power_all on every rank of a gloo job, the device context replaced by
    a closed-form stub: the full (z, k) grid arrives on every rank in the caller's z order,
    uneven splits and per-epoch cosmologies included (rows partitioned, one
    all-gather, no other collective)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(r, world, port, n_all, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` in the driver's command form (no launcher): the parent starts N
    rank processes itself and passes their exit code on.  Without a GPU the ranks stop at their
    "needs MI355X" assertion -- which proves that N ranks started, each with its own RANK, and
    that the launcher neither died on a world-size assertion nor reported success."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU rehearsal of the launcher (on a GPU box bench.py itself is run)")
    # (--no-cpu-baseline: rank 0 would otherwise time the oracle first, at every N, and be
    #  torn down by the launcher when rank 1 fails meanwhile)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0
    assert p.stderr.count("bench.py needs MI355X GPUs") >= 2, p.stderr[-2000:]
    assert "launch with torch.distributed.run" not in p.stderr
    assert p.stdout.strip() == ""
    # a WORLD_SIZE that contradicts --gpus is refused with a message, not an assertion
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in p.stderr
