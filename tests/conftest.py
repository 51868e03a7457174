import os
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
for _p in (GOLDEN, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return numpy.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(a, b):
    a = numpy.asarray(a, dtype=float)
    b = numpy.asarray(b, dtype=float)
    den = numpy.maximum(numpy.abs(b), 1e-300)
    return float(numpy.max(numpy.abs(a - b) / den))


@pytest.fixture(scope="session")
def golden():
    return load_golden
