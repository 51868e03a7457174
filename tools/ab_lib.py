"""Run bench.py against another build of the library (A/B timing of compiler flags or of an
experimental kernel; development aid, not part of the product):
    python tools/ab_lib.py build_exp/other.so [bench.py arguments]"""
import os, runpy, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
so = os.path.abspath(sys.argv[1])
from chomp_amd import _lib as _l
_l.LIB_PATH = so
_l.build = lambda *a, **k: so
sys.argv = [os.path.join(R, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
