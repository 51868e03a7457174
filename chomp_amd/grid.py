"""Batched (k, z) grids and their multi-GPU sharding.

The reference sweeps a (k, z) grid by calling halo.set_redshift(z) and
power_mm(k_array) once per z; every z rebuilds ~400 Romberg
integrals in Python.  Here the z-axis (or a list of cosmologies: the
SimulationDesign axis, simulation_design.py:116-155) is a batch of "epochs" whose
tables are built by the same kernel launches, and Stage E evaluates the whole grid.

Multi-GPU (one process per GPU, torch.distributed / RCCL): epochs are independent,
so rank r builds and evaluates epochs r, r+W, r+2W, ... (interleaved: the set-up
cost grows with z) and ONE all-gather of the padded row blocks re-assembles the
spectrum on every rank.  No other collective is needed.
"""
import numpy

from . import _lib
from . import cosmology
from . import defaults
from . import hod as hod_mod

_WHICH = {"linear_power": (_lib.P_LIN, 0), "power_mm": (_lib.P_MM, _lib.FAM_MM),
          "power_gm": (_lib.P_GM, _lib.FAM_GM), "power_mg": (_lib.P_GM, _lib.FAM_GM),
          "power_gg": (_lib.P_GG, _lib.FAM_GG)}


def shard_indices(n, rank, world):
    """Epoch indices of `rank`: interleaved so high-z (expensive) rows spread."""
    return list(range(rank, n, world))


def rows_per_rank(n, world):
    return (n + world - 1) // world


def unshard_order(n, world):
    """Positions, in the gathered [world * rows_per_rank] layout, of epochs 0..n-1."""
    rpr = rows_per_rank(n, world)
    return [(i % world) * rpr + i // world for i in range(n)]


class HaloGrid(object):
    """P(k, z) for a batch of epochs on one GPU (or this rank's shard of it).

    z: 1-D array of redshifts (one epoch each); cosmo_dict: one dict or a list of
    dicts (one per epoch); mass_function: 'st' or 'tinker'; halo_dict, hod_dict:
    one dict (or list).  Every epoch follows the reference's construction order
    Halo(z, HODZheng(hod_dict), SingleEpoch(z, cosmo_dict), MassFunction(...)).
    """

    def __init__(self, z, cosmo_dict=None, halo_dict=None, hod_dict=None,
                 mass_function="st", device=None, stream=None, rank=0, world=1,
                 collective_at_world_1=False):
        self.z_all = numpy.atleast_1d(numpy.asarray(z, dtype=numpy.float64))
        self.n_all = self.z_all.size
        self.rank, self.world = rank, world
        # (a single rank normally skips the all-gather; True sends it through the process
        #  group anyway -- how the RCCL path is exercised on a one-GPU box)
        self._collective_1 = bool(collective_at_world_1)
        self.idx = shard_indices(self.n_all, rank, world)
        pick = lambda v: [v[i] for i in self.idx] if isinstance(v, (list, tuple)) else v
        self.cosmo = pick(cosmo_dict if cosmo_dict is not None
                          else defaults.default_cosmo_dict)
        self.halo = pick(halo_dict if halo_dict is not None
                         else defaults.default_halo_dict)
        hd = hod_dict if hod_dict is not None else defaults.default_hod_dict
        self.hod = ([hod_mod.HODZheng(h) for h in pick(hd)]
                    if isinstance(hd, (list, tuple)) else hod_mod.HODZheng(hd))
        self.kind = {"st": _lib.MF_ST, "tinker": _lib.MF_TINKER}[mass_function]
        self.z = self.z_all[self.idx]
        self.ctx = cosmology._context(stream=stream, device=device)
        self._tables = 0
        self._bufs = {}
        n = len(self.idx)
        # parameter blocks packed once (ctypes), not per set-up call
        self._c_cosmo = self.ctx.pack_cosmo(self.cosmo, n) if n else None
        self._c_halo = self.ctx.pack_halo(self.halo, n) if n else None
        self._c_hod = self.ctx.pack_hod(self.hod, n) if n else None
        self._z = numpy.ascontiguousarray(self.z, dtype=numpy.float64)

    def set_parameters(self, cosmo=None, halo=None, hod=None):
        """New parameters for this rank's epochs, taking effect at the next setup(): the step of
        an MCMC / design loop.  Each argument: None (keep), one dict, a list with one dict per
        local epoch, or the packed ctypes array (Context.pack_*; cosmo also as a float64 array
        [n_local, 10]) -- packing once outside the loop keeps the host out of the step."""
        import ctypes
        n = len(self.idx)
        if n == 0:
            return
        if cosmo is not None:
            self._c_cosmo = cosmo if isinstance(cosmo, ctypes.Array) else self.ctx.pack_cosmo(cosmo, n)
        if halo is not None:
            self._c_halo = halo if isinstance(halo, ctypes.Array) else self.ctx.pack_halo(halo, n)
        if hod is not None:
            if not isinstance(hod, ctypes.Array):
                hod = self.ctx.pack_hod([hod_mod.HODZheng(h) for h in hod]
                                        if isinstance(hod, (list, tuple)) else hod_mod.HODZheng(hod), n)
            self._c_hod = hod
        assert len(self._c_cosmo) == n and len(self._c_halo) == n and len(self._c_hod) == n
        self._tables = 0

    def setup(self, which="power_mm"):
        """Stage K for this rank's epochs (asynchronous on the context's stream)."""
        _, need = _WHICH[which]
        if len(self.idx) == 0:
            return
        self.ctx.epochs_set(self._c_cosmo, self._z)
        self.ctx.stage_k(self._c_halo, self.kind, self._c_halo, self._c_hod, need)
        self._tables = need

    def status(self, warn=False):
        """Status words of this rank's epochs (chomp_get_status: saturated mass-limit search,
        Romberg integrals that exhausted divmax, ...); synchronises.  warn=True also raises
        them as Python warnings.  The batch path never checks by itself: a step stays
        asynchronous."""
        if len(self.idx) == 0:
            return numpy.zeros(0, dtype=numpy.uint32)
        return self.ctx.warn_status(0, len(self.idx)) if warn else self.ctx.status(0, len(self.idx))

    def power(self, which, k, out=None):
        """Stage E: [n_local, nk] for numpy or torch-cuda k."""
        code, need = _WHICH[which]
        if (self._tables & need) != need:
            self.setup(which)
        return self.ctx.power(code, k, 0, len(self.idx), out=out)

    def power_all(self, which, k):
        """Stage K + E on this rank's shard, then all-gather (torch.distributed) and
        re-order to the caller's z order.  k: torch cuda tensor.  Returns the full
        [n_all, nk] tensor on every rank."""
        return self.power_all_async(which, k).wait()

    def power_all_async(self, which, k):
        """As power_all, but the all-gather is only launched (RCCL runs it on its own
        stream): the caller may queue the next set-up behind it and call .wait() on the
        returned handle later -- the collective then overlaps that compute.

        The row block this rank fills and the gathered grid live in two alternating
        persistent buffers (no allocation or fill per call): a returned grid is valid
        until the second call after the one that produced it."""
        import torch
        rpr = rows_per_rank(self.n_all, self.world)
        gathers = self.world > 1 or self._collective_1
        key = (rpr, k.numel(), str(k.device))
        if self._bufs.get("key") != key:
            self._bufs = {"key": key, "turn": 0, "local": [], "full": []}
            for _ in range(2):
                self._bufs["local"].append(
                    torch.zeros((rpr, k.numel()), dtype=torch.float64, device=k.device))
                self._bufs["full"].append(
                    torch.empty((self.world * rpr, k.numel()), dtype=torch.float64,
                                device=k.device) if gathers else None)
        turn = self._bufs["turn"]
        self._bufs["turn"] = turn ^ 1
        local = self._bufs["local"][turn]
        if len(self.idx):
            self.power(which, k, out=local[:len(self.idx)])
        return gather_rows_async(local, self.n_all, self.world, out=self._bufs["full"][turn],
                                 collective_at_world_1=self._collective_1)


_ORDER_CACHE = {}


def _order_tensor(n_all, world, device):
    """unshard_order as a device index tensor, built once (a per-step host-to-device copy
    would stall the stream the next set-up is queued on)."""
    import torch
    key = (n_all, world, str(device))
    t = _ORDER_CACHE.get(key)
    if t is None:
        t = _ORDER_CACHE[key] = torch.as_tensor(unshard_order(n_all, world), device=device)
    return t


class PendingRows(object):
    """Handle of an all-gather in flight; wait() returns the [n_all, nk] grid in epoch
    order (stream-ordered: it does not block the host)."""

    def __init__(self, full, work, n_all, world):
        self._full, self._work, self._n_all, self._world = full, work, n_all, world

    def wait(self):
        import torch
        if self._work is not None:
            self._work.wait()
        if self._world == 1:
            return self._full[:self._n_all]
        return self._full.index_select(0, _order_tensor(self._n_all, self._world,
                                                        self._full.device))


def gather_rows_async(local, n_all, world, out=None, collective_at_world_1=False):
    """Launch the all-gather of the per-rank row blocks ([rows_per_rank, nk], zero
    padded) into `out` (allocated if None).  RCCL on GPUs ("nccl" backend), gloo on CPU
    tensors (the multi-process CPU tests).  A single rank has nothing to gather and returns
    its block, unless collective_at_world_1 asks for the collective all the same."""
    import torch
    import torch.distributed as dist
    if world == 1 and not collective_at_world_1:
        return PendingRows(local, None, n_all, world)
    rpr = rows_per_rank(n_all, world)
    assert local.shape[0] == rpr
    full = out if out is not None else torch.empty((world * rpr, local.shape[1]),
                                                   dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(full, local.contiguous(), async_op=True)
    return PendingRows(full, work, n_all, world)


def gather_rows(local, n_all, world):
    """All-gather the per-rank row blocks and put the rows back into epoch order."""
    return gather_rows_async(local, n_all, world).wait()


# -- 1-D sample axes (theta, l): SURVEY 8(e) C4 / C5 -------------------------------------------
def shard_samples(x, rank, world):
    """This rank's share of a 1-D sample array (theta or l): every world-th sample starting at
    `rank`, padded with its last sample to ceil(n / world) so that all ranks gather equal
    counts (the tables the samples are evaluated against are replicated, not sharded)."""
    mine = x[rank::world]
    per = rows_per_rank(x.shape[0], world)
    if mine.shape[0] < per:
        import torch
        pad = mine[-1:] if mine.shape[0] else x[-1:]
        mine = torch.cat([mine, pad.expand(per - mine.shape[0])])
    return mine.contiguous()


def gather_samples(local, n_all, world, collective_at_world_1=False):
    """All-gather the per-rank results of shard_samples and undo the interleaving: the n_all
    values in the caller's sample order on every rank."""
    import torch
    import torch.distributed as dist
    if world == 1 and not collective_at_world_1:
        return local[:n_all]
    per = local.shape[0]
    full = torch.empty(world * per, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous())
    return full.view(world, per).t().reshape(-1)[:n_all]
