"""Romberg quadrature with the semantics of ``scipy.integrate.romberg`` (SciPy < 1.15).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

The reference calls ``scipy.integrate.romberg`` for every tabulated integral
(e.g. /root/reference/cosmology.py:634-639, halo.py:909-915, kernel.py:699-704,
correlation.py:253-259).  SciPy is a third-party dependency of the reference
(README.txt:29-34 lists "python2.7, numpy, scipy" with no pinned version) and
the routine was removed in SciPy 1.15, which is the version in this image, so
its published algorithm is restated here:

* T_0 = (b-a) * (f(a) + f(b))/2, the end points evaluated as scalars;
* level i = 1..divmax adds the 2**(i-1) mid-points of the previous panels,
  ``lox + h*arange(n/2)`` with ``h = (b-a)/(n/2)`` and ``lox = a + h/2``, in ONE
  vectorised call, and forms R[i][0] = (b-a) * ordsum / 2**i;
* Richardson: R[i][k] = (4**k R[i][k-1] - R[i-1][k-1]) / (4**k - 1);
* result R[i][i]; err = |R[i][i] - R[i-1][i-1]|; stop when ``err < tol`` or
  ``err < rtol*|result|``; if the loop runs out, warn and return the last R[i][i].

After level L the integrand has been evaluated at 2**L + 1 points.
"""
import warnings

import numpy


class AccuracyWarning(Warning):
    pass


def romberg(function, a, b, args=(), tol=1.48e-8, rtol=1.48e-8, show=False,
            divmax=10, vec_func=False, return_level=False):
    if numpy.isinf(a) or numpy.isinf(b):
        raise ValueError("Romberg integration only available for finite limits.")
    if vec_func:
        def vfunc(x):
            return function(x, *args)
    else:
        def vfunc(x):
            if numpy.isscalar(x):
                return function(x, *args)
            return numpy.array([function(xi, *args) for xi in x])

    n = 1
    intrange = b - a
    ordsum = 0.5 * (vfunc(a) + vfunc(b))
    result = intrange * ordsum
    last_row = [result]
    err = numpy.inf
    level = 0
    for i in range(1, divmax + 1):
        n *= 2
        numtosum = n // 2
        h = float(b - a) / numtosum
        lox = a + 0.5 * h
        points = lox + h * numpy.arange(numtosum)
        ordsum = ordsum + numpy.sum(vfunc(points), axis=0)
        row = [intrange * ordsum / n]
        for k in range(i):
            tmp = 4.0 ** (k + 1)
            row.append((tmp * row[k] - last_row[k]) / (tmp - 1.0))
        result = row[i]
        lastresult = last_row[i - 1]
        err = abs(result - lastresult)
        level = i
        if err < tol or err < rtol * abs(result):
            break
        last_row = row
    else:
        warnings.warn("divmax (%d) exceeded. Latest difference = %e"
                      % (divmax, err), AccuracyWarning)
    if return_level:
        return result, level
    return result
