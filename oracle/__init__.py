"""CPU oracle for the CHOMP halo-model + Limber hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a NumPy/SciPy
restatement of the reference's algorithm (adaptive Romberg integrals tabulated on
50-point grids and served through interpolating cubic splines), written so that
every function cites the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline -- never as a
fallback of the product path (``chomp_amd`` raises when its HIP library is
missing).

Pinning: the oracle is checked against (a) the reference's own known-answer
pins for this path (unit_test.py:346-407 and friends, stored as data in
tests/golden/reference_pins.json) and (b) golden vectors produced by running the
reference itself in the development container (tests/golden/make_golden.py).
"""
