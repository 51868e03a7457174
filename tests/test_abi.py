"""CPU tests of the boundary: the C-ABI library builds for gfx950, loads, exports
every symbol include/chomp_mi355x.h declares, and fails loudly without a GPU."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "chomp_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chomp_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    from chomp_amd import _lib
    _lib.build()
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.EXPORTS) == names


def test_struct_layouts_match_header():
    import ctypes
    from chomp_amd import _lib
    assert ctypes.sizeof(_lib.Cosmo) == 80
    assert ctypes.sizeof(_lib.HaloPar) == 48
    assert ctypes.sizeof(_lib.HodPar) == 40
    assert ctypes.sizeof(_lib.Config) == 12 * 8 + 8 * 4
    assert ctypes.sizeof(_lib.Dndz) == 8 + 16 + 32 + 2 * 8 + 8   # + the dNdzInterpolation spline
    assert ctypes.sizeof(_lib.Window) == 8 + ctypes.sizeof(_lib.Dndz)
    c = _lib.Config()
    _lib.lib().chomp_default_config(ctypes.byref(c))
    from chomp_amd import defaults
    ref = _lib.make_config(defaults.default_limits, defaults.default_precision)
    for name, _ in _lib.Config._fields_:
        assert getattr(c, name) == getattr(ref, name), name


def test_no_cpu_fallback():
    """Without a HIP device the product path must raise, not compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from chomp_amd import halo, _lib
    h = halo.Halo(0.0)
    with pytest.raises(_lib.ChompError):
        h.power_mm(1.0)


def test_sharding_is_a_permutation():
    from chomp_amd import grid
    for n in (1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            rpr = grid.rows_per_rank(n, w)
            slots = [None] * (w * rpr)
            for r in range(w):
                for j, i in enumerate(grid.shard_indices(n, r, w)):
                    slots[r * rpr + j] = i
            order = grid.unshard_order(n, w)
            assert [slots[p] for p in order] == list(range(n))
