"""hod.HOD / hod.HODZheng (hod.py:17-230): parameter holders with the reference's
attribute and method names.  Inside the halo-model integrals the moments are
evaluated on the device (chomp_math.h zheng_*); the public first_moment() /
second_moment() accessors below are the same one-line closed forms."""
import math

import numpy

from . import defaults


class HOD(object):
    """hod.py:17-129 (base class: API only)."""

    def __init__(self, hod_dict):
        self.hod_dict = hod_dict
        self.first_moment_zero = -1
        self.second_moment_zero = -1
        self._safe_norm = -1

    def first_moment(self, mass, z=None):
        """hod.py:40-52 (base class: one galaxy per halo)."""
        return 1.0

    def second_moment(self, mass, z=None):
        """hod.py:54-66."""
        return 1.0

    def nth_moment(self, mass, n=3, z=None):
        """hod.py:68-92: <N(N-1)...(N-n+1)> from the first two moments."""
        if n == 1:
            return self.first_moment(mass, z)
        if n == 2:
            return self.second_moment(mass, z)
        first_mom = self.first_moment(mass, z)
        exp_nth = first_mom ** n
        with numpy.errstate(all="ignore"):
            alpha_m2 = numpy.where(first_mom != 0.0,
                                   self.second_moment(mass, z) / first_mom ** 2, 0.0)
        for j in range(n):
            exp_nth = exp_nth * (j * alpha_m2 - j + 1)
        return exp_nth

    def get_hod(self):
        return self.hod_dict

    def set_hod(self, hod_dict):
        self.__init__(hod_dict)

    def set_halo(self, halo_dict):
        pass

    def write(self, output_file_name):
        """hod.py:112-129 (the reference passes (mass, None, 3) to nth_moment, i.e. n = None:
        written here as the third moment it means)."""
        mass_max, mass_min = 1.0e16, 1.0e9
        dln_mass = (numpy.log(mass_max) - numpy.log(mass_min)) / 200
        ln_mass_array = numpy.arange(numpy.log(mass_min) - dln_mass,
                                     numpy.log(mass_max) + dln_mass + dln_mass, dln_mass)
        with open(output_file_name, "w") as f:
            for ln_mass in ln_mass_array:
                mass = numpy.exp(ln_mass)
                f.write("%1.10f %1.10f %1.10f %1.10f\n" % (
                    mass, self.first_moment(mass), self.second_moment(mass),
                    self.nth_moment(mass, 3)))


def _erfinv(y):
    """Damped Newton on erf/erfc in the tail-accurate form (called once per HOD
    for first_moment_zero, hod.py:172-175)."""
    if y <= -1.0:
        return -math.inf
    if y >= 1.0:
        return math.inf
    x = 0.0
    for _ in range(200):
        if y < -0.5:
            r = math.erfc(-x) - (1.0 + y)
        elif y > 0.5:
            r = (1.0 - y) - math.erfc(x)
        else:
            r = math.erf(x) - y
        d = 2.0 / math.sqrt(math.pi) * math.exp(-x * x)
        dx = r / d
        if abs(dx) > 1.0:
            dx = math.copysign(1.0, dx)
        x -= dx
        if abs(dx) <= 1e-16 * max(abs(x), 1e-300):
            break
    return x


class HODZheng(HOD):
    """Zheng et al. 2007 five-parameter HOD (hod.py:141-230)."""

    def __init__(self, hod_dict=None):
        if hod_dict is None:
            self.log_M_min = 12.14
            self.sigma = 0.15
            self.log_M_0 = 12.14
            self.log_M_1p = 13.43
            self.alpha = 1.0
        else:
            self.log_M_min = hod_dict['log_M_min']
            self.sigma = hod_dict['sigma']
            self.log_M_0 = hod_dict['log_M_0']
            self.log_M_1p = hod_dict['log_M_1p']
            self.alpha = hod_dict['alpha']
        HOD.__init__(self, hod_dict)
        # hod.py:172-186.  The reference's `secon_moment_zero` typo means the clamp
        # of second_moment_zero never takes effect; it is not applied here either.
        self.first_moment_zero = 10.0 ** (
            self.log_M_min + self.sigma * _erfinv(
                2. * defaults.default_precision['halo_precision'] - 1.0))
        self.second_moment_zero = 10.0 ** self.log_M_0
        self._safe_norm = 10.0 ** (self.log_M_min + 1.0 * self.sigma)

    def central_first_moment(self, mass):
        mass = numpy.asarray(mass, dtype=numpy.float64)
        if self.sigma <= 0.0:
            return numpy.where(numpy.log10(mass) > self.log_M_min, 1.0, 0.0)
        erf = numpy.vectorize(math.erf, otypes=[float])
        return 0.5 * (1 + erf((numpy.log10(mass) - self.log_M_min) / self.sigma))

    def satellite_first_moment(self, mass):
        mass = numpy.asarray(mass, dtype=numpy.float64)
        diff = mass - numpy.power(10, self.log_M_0)
        with numpy.errstate(all="ignore"):
            return numpy.where(diff > 0.0,
                               self.central_first_moment(mass) *
                               numpy.power(diff / (10 ** self.log_M_1p), self.alpha),
                               0.0)

    def first_moment(self, mass, z=None):
        return self.central_first_moment(mass) + self.satellite_first_moment(mass)

    def second_moment(self, mass, z=None):
        n_sat = self.satellite_first_moment(mass)
        return (2 + n_sat) * n_sat


class HODPoisson(HOD):
    """Empty in the reference too (hod.py:131-134)."""

    def __init__(self):
        pass


class HODBinomial(HOD):
    """Empty in the reference too (hod.py:136-139)."""

    def __init__(self, n_max, min_mass, mass_max, p_m_spline):
        pass
