"""In-memory loader for the Python-2 reference at /root/reference (dev container only).

Used ONLY by tests/golden/make_golden.py to generate fixtures and by the
oracle-vs-reference cross-check test (which skips where /root/reference does not
exist, i.e. on the GPU box).  Nothing of the reference is copied into this
repository: the sources are read as text at run time, converted in memory and
executed into throw-away module objects.

Recipe:
  1. each module's text goes through the stdlib ``lib2to3`` refactoring tool
     (Py2 print statements, xrange, ``except A, B``);
  2. two textual substitutions restore Python-2 integer-division semantics the
     reference's pinned values depend on: ``(Omb2)**(3/4)`` is ``**0`` under Py2
     (cosmology.py:464) and ``1/b`` floors when ``b`` is an int (kernel.py:167);
  3. ``scipy.integrate.romberg`` (removed in SciPy 1.15) is provided by
     oracle/romberg.py, a restatement of its published algorithm;
  4. ``numpy.float128`` tables are honoured as-is (x86 long double).
"""
import importlib.abc
import importlib.util
import os
import sys
import types
import warnings

REFERENCE_DIR = "/root/reference"
MODULES = ("defaults", "cosmology", "mass_function", "hod", "halo", "kernel",
           "correlation", "perturbation_spectra", "halo_trispectrum", "covariance")

_here = os.path.dirname(os.path.abspath(__file__))
_root = os.path.dirname(os.path.dirname(_here))
if _root not in sys.path:
    sys.path.insert(0, _root)


def available():
    return os.path.isdir(REFERENCE_DIR)


def _convert(name, text):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
        tool = refactor.RefactoringTool(
            refactor.get_fixers_from_package("lib2to3.fixes"))
        out = str(tool.refactor_string(text + "\n", name))
    if name == "cosmology":
        assert "(Omb2)**(3/4)" in out
        out = out.replace("(Omb2)**(3/4)", "(Omb2)**(3//4)")
    if name == "kernel":
        assert "1/b)*" in out
        out = out.replace("1/b)*", "(1//b if isinstance(b, int) else 1/b))*")
    return out


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname in MODULES:
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return types.ModuleType(spec.name)

    def exec_module(self, module):
        name = module.__name__
        fn = os.path.join(REFERENCE_DIR, name + ".py")
        with open(fn) as f:
            text = f.read()
        code = compile(_convert(name, text), "<reference:%s>" % name, "exec")
        module.__file__ = fn
        exec(code, module.__dict__)


_installed = False


def load():
    """Return a namespace with the reference modules (defaults, cosmology, ...)."""
    global _installed
    if not available():
        raise RuntimeError("reference not present at %s" % REFERENCE_DIR)
    sys.dont_write_bytecode = True
    if not _installed:
        from scipy import integrate
        from oracle.romberg import romberg
        if not hasattr(integrate, "romberg"):
            integrate.romberg = romberg
        for m in MODULES:
            if m in sys.modules and not getattr(sys.modules[m], "__file__",
                                                "").startswith(REFERENCE_DIR):
                raise RuntimeError("module name clash: %s" % m)
        sys.meta_path.insert(0, _Finder())
        _installed = True
    ns = types.SimpleNamespace()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for m in MODULES:
            setattr(ns, m, importlib.import_module(m))
    return ns
