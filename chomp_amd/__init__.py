"""chomp_amd -- MI355X-native implementation of CHOMP's halo-model + Limber hot path.

Drop-in mirrors of the reference's modules (same class, keyword and method names):

    from chomp_amd import cosmology, mass_function, hod, halo, kernel, correlation
    from chomp_amd import covariance, simulation_design

plus the batched entry point the reference lacks, ``chomp_amd.grid.HaloGrid``:
P(k, z) on a whole (k, z) grid -- or a batch of cosmologies -- in one set of kernel
launches, sharded over the GPUs of a node with torch.distributed (RCCL).

All numerical work runs in hand-written HIP kernels (chomp_amd/csrc) behind the C
ABI of include/chomp_mi355x.h.  There is no CPU fallback: without the HIP library
and an MI355X, constructing a device context raises.
"""
from . import _lib
from . import defaults
from ._lib import ChompError, ChompScopeError, build

__all__ = ["defaults", "cosmology", "mass_function", "hod", "halo", "kernel",
           "correlation", "covariance", "simulation_design", "grid", "build",
           "ChompError", "ChompScopeError"]


def __getattr__(name):
    if name in ("cosmology", "mass_function", "hod", "halo", "kernel",
                "correlation", "covariance", "simulation_design", "grid"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
