"""cosmology.SingleEpoch / MultiEpoch with the reference's constructor and method
surface (cosmology.py:25-728, 731-1164), backed by the HIP library.

Numbers with an integral behind them (chi, sigma_8 normalisation, sigma_r, the
linear spectrum) come from the device; closed-form background functions of an
arbitrary redshift argument (E, E0, w) are one-line formulas kept on the host, as
they are in the reference.
"""
import numpy

from . import _lib
from . import defaults


def _context(stream=None, device=None):
    """New device context snapshotting defaults.default_limits/_precision now."""
    cfg = _lib.make_config(defaults.default_limits, defaults.default_precision)
    if device is None:
        device = _lib.current_device()
    return _lib.Context(cfg, device=device, stream=stream)


class SingleEpoch(object):
    """cosmology.py:25-728.  w0/wa != -1/0 are outside the accelerated scope.  As in the
    reference, set_cosmology() re-runs __init__ without with_bao, i.e. switches it off."""

    def __init__(self, redshift, cosmo_dict=None, with_bao=False, **kws):
        if redshift < 0.0:
            redshift = 0.0
        self._redshift = redshift
        if cosmo_dict is None:
            cosmo_dict = defaults.default_cosmo_dict
        self.cosmo_dict = cosmo_dict
        for attr, key in (("_omega_m0", "omega_m0"), ("_omega_b0", "omega_b0"),
                          ("_omega_l0", "omega_l0"), ("_omega_r0", "omega_r0"),
                          ("_cmb_temp", "cmb_temp"), ("_h", "h"),
                          ("_sigma_8", "sigma_8"), ("_n", "n_scalar"),
                          ("_w0", "w0"), ("_wa", "wa")):
            setattr(self, attr, cosmo_dict[key])          # KeyError like the reference
        self.H0 = 100.0 / (2.998 * 10 ** 5)
        self._with_bao = with_bao
        self._k_min = defaults.default_limits['k_min']
        self._k_max = defaults.default_limits['k_max']
        self._ctx = None
        self._sc = None

    # -- device state ----------------------------------------------------------
    def _dev(self):
        if self._ctx is None:
            self._ctx = _context()
        if self._sc is None:
            self._ctx.epochs_set(self.cosmo_dict, [self._redshift], self._with_bao)
            self._sc = self._ctx.scalars(0)
        return self._ctx

    def _scalar(self, name):
        self._dev()
        return float(self._sc[name])

    _chi = property(lambda self: self._scalar("chi"))
    _growth = property(lambda self: self._scalar("growth"))
    growth_norm = property(lambda self: self._scalar("growth_norm"))
    _sigma_norm = property(lambda self: self._scalar("sigma_norm"))
    delta_H = property(lambda self: self._scalar("delta_H"))

    # -- reference surface -----------------------------------------------------
    def set_redshift(self, redshift):
        if redshift != self._redshift:
            self._redshift = redshift
            self._sc = None

    def get_cosmology(self):
        return self.cosmo_dict

    def set_cosmology(self, cosmo_dict, redshift=None):
        if redshift is None:
            redshift = self._redshift
        ctx = self._ctx
        self.__init__(redshift, cosmo_dict)
        self._ctx = ctx

    def E0(self, redshift):
        a = 1.0 / (1.0 + redshift)
        return (self._omega_l0 + self._omega_m0 / (a * a * a) +
                self._omega_r0 / (a * a * a * a))

    def E(self, redshift):
        return 1.0 / (self.H0 * numpy.sqrt(self.E0(redshift)))

    def w(self, redshift):
        a = 1.0 / (1 + redshift)
        return self._w0 + self._wa * (1 - a)

    def comoving_distance(self):
        return self._chi

    def luminosity_distance(self):
        return (1.0 + self._redshift) * self._chi

    def angular_diameter_distance(self):
        return self._chi / (1.0 + self._redshift)

    def redshift(self):
        return self._redshift

    def growth_factor(self):
        return self._growth

    def omega_m(self):
        return self._scalar("omega_m")

    def omega_l(self):
        return self._scalar("omega_l")

    def delta_c(self):
        return self._scalar("delta_c")

    def delta_v(self):
        return self._scalar("delta_v")

    def rho_crit(self):
        return self.rho_bar() / self.omega_m()

    def rho_bar(self):
        return self._scalar("rho_bar")

    def delta_k(self, k):
        return self._dev().eval("delta_k", numpy.asarray(k, dtype=numpy.float64))

    def linear_power(self, k):
        ka = numpy.asarray(k, dtype=numpy.float64)
        return self._dev().power(_lib.P_LIN, ka, 0, 1).reshape(ka.shape)

    def sigma_r(self, scale):
        s = numpy.asarray(scale, dtype=numpy.float64)
        out = self._dev().sigma_r(0, s).reshape(s.shape)
        return float(out) if s.ndim == 0 else out

    def sigma_m(self, mass):
        scale = (3.0 * numpy.asarray(mass, dtype=numpy.float64) /
                 (4.0 * numpy.pi * self.rho_bar())) ** (1.0 / 3.0)
        return self.sigma_r(scale)

    def nu_r(self, scale):
        sqrt_nu = self.delta_c() / self.sigma_r(scale)
        return sqrt_nu * sqrt_nu

    def nu_m(self, mass):
        sqrt_nu = self.delta_c() / self.sigma_m(mass)
        return sqrt_nu * sqrt_nu

    def growth_factor_eval(self, a):
        """cosmology.py:305-326: always the Carroll et al. approximation (the return of
        _growth_approx, :215-231, with its Omega_m * (4/7) term)."""
        a = numpy.asarray(a, dtype=numpy.float64)
        om = self._omega_m0 / (a * a * a)
        denom = self._omega_l0 + om
        Omega_m, Omega_L = om / denom, self._omega_l0 / denom
        coeff = 5.0 * Omega_m / (2.0 / a)
        return coeff / (Omega_m * (4.0 / 7.0) - Omega_L +
                        (1.0 + 0.5 * Omega_m) * (1.0 + Omega_L / 70.0))

    def transfer_function(self, k):
        """cosmology.py:540-572 (the no-wiggle E&H fit as the reference evaluates it), from
        the device's Delta^2(k) = delta_H^2 (k/H0)^(3+n) T^2 / h * D^2 sigma_norm^2."""
        ka = numpy.asarray(k, dtype=numpy.float64)
        d2 = self.delta_k(ka)
        amp = (self.delta_H ** 2 * numpy.power(ka / self.H0, 3.0 + self._n) / self._h *
               self._growth ** 2 * self._sigma_norm ** 2)
        return numpy.sqrt(d2 / amp)

    def write(self, output_power_file_name=None):
        """cosmology.py:700-728."""
        print("z = %1.4f" % self._redshift)
        print("Comoving distance = %1.4f" % self._chi)
        print("Growth factor = %1.4f" % self._growth)
        print("Omega_m(z) = %1.4f" % self.omega_m())
        print("Omega_l(z) = %1.4f" % self.omega_l())
        print("DE w(z)    = %1.4f" % self.w(self._redshift))
        print("Delta_V(z) = %1.4f" % self.delta_v())
        print("delta_c(z) = %1.4f" % self.delta_c())
        print("sigma_8(z) = %1.4f" % self.sigma_r(8.0))
        if output_power_file_name is not None:
            dln_k = (numpy.log(self._k_max) - numpy.log(self._k_min)) / 200
            ln_k = numpy.arange(numpy.log(self._k_min) - dln_k,
                                numpy.log(self._k_max) + dln_k + dln_k, dln_k)
            k = numpy.exp(ln_k)
            with open(output_power_file_name, "w") as f:
                f.write("#ttype1 = k [Mpc/h]\n#ttype2 = P(k) [(Mpc/h)^3]\n")
                for row in zip(k, self.linear_power(k)):
                    f.write("%1.10f %1.10f\n" % row)


class MultiEpoch(object):
    """cosmology.MultiEpoch (cosmology.py:731-1164): chi(z), z(chi), D(z) on a
    50-point grid over [z_min, z_max], tabulated and splined on the device."""

    def __init__(self, z_min, z_max, cosmo_dict=None, with_bao=False, **kws):
        self.z_min = z_min
        self.z_max = z_max
        if self.z_min < 0.0:
            self.z_min = 0.0
        if cosmo_dict is None:
            cosmo_dict = defaults.default_cosmo_dict
        self.epoch0 = SingleEpoch(0.0, cosmo_dict, with_bao, **kws)
        self.cosmo_dict = cosmo_dict
        e0 = self.epoch0
        self._omega_m0, self._omega_b0, self._omega_l0 = e0._omega_m0, e0._omega_b0, e0._omega_l0
        self._h, self.H0, self._omega_r0 = e0._h, e0.H0, e0._omega_r0
        self._sigma_8, self._w0, self._wa, self._n = e0._sigma_8, e0._w0, e0._wa, e0._n
        self._k_min, self._k_max = e0._k_min, e0._k_max
        self._ctx = None
        self._sig = None

    def _dev(self):
        if self._ctx is None:
            self._ctx = _context()
        sig = (tuple(sorted(self.cosmo_dict.items())), self.z_min, self.z_max)
        if sig != self._sig:
            self._ctx.multi_epoch_setup(self.cosmo_dict, self.z_min, self.z_max)
            self._sig = sig
        return self._ctx

    _z_array = property(lambda self: self._dev().kernel_table("me_z"))
    _chi_array = property(lambda self: self._dev().kernel_table("me_chi"))
    _growth_array = property(lambda self: self._dev().kernel_table("me_growth"))
    growth_norm = property(lambda self: self.epoch0.growth_norm)

    def set_redshift(self, z_min, z_max):
        self.z_max = z_max
        self.z_min = z_min
        if self.z_min < 0.0:
            self.z_min = 0.0

    def get_cosmology(self):
        return self.epoch0.get_cosmology()

    def set_cosmology(self, cosmo_dict, z_min=None, z_max=None):
        if z_min is None:
            z_min = self.z_min
        if z_max is None:
            z_max = self.z_max
        ctx = self._ctx
        self.__init__(z_min, z_max, cosmo_dict)
        self._ctx = ctx

    def E(self, redshift):
        return self.epoch0.E(redshift)

    def comoving_distance(self, redshift):
        return self._dev().me_eval("chi_of_z", redshift)

    def luminosity_distance(self, redshift):
        return (1.0 + redshift) * self.comoving_distance(redshift)

    def angular_diameter_distance(self, redshift):
        return self.comoving_distance(redshift) / (1.0 + redshift)

    def redshift(self, comoving_distance):
        return self._dev().me_eval("z_of_chi", comoving_distance)

    def growth_factor(self, redshift):
        return self._dev().me_eval("growth_of_z", redshift)

    def omega_m(self, redshift=None):
        if redshift is None:
            redshift = 0.0
        return self._omega_m0 * (1.0 + redshift) ** 3 / self.epoch0.E0(redshift)

    def omega_l(self, redshift=None):
        if redshift is None:
            redshift = 0.0
        return self._omega_l0 / self.epoch0.E0(redshift)

    def linear_power(self, k, redshift=None):
        p = self.epoch0.linear_power(k)
        if redshift is not None:
            p = p * self.growth_factor(redshift) ** 2
        return p

    def sigma_r(self, scale, redshift=None):
        sigma = self.epoch0.sigma_r(scale)
        if redshift is not None:
            sigma = sigma * self.growth_factor(redshift)
        return sigma

    # -- cosmology.py:977-1110: redshift-argument versions of the SingleEpoch scalars ----
    def _flags(self):
        p = defaults.default_precision["cosmo_precision"]
        tot = self._omega_m0 + self._omega_l0 + self._omega_r0
        return (tot <= 1.0 + p and tot >= 1.0 - p), tot <= 1.0 - p      # flat, open

    def delta_c(self, redshift=None):
        flat, open_ = self._flags()
        delta_c = 0.15 * (12.0 * numpy.pi) ** (2.0 / 3.0)
        if open_:
            delta_c *= self.omega_m(redshift) ** 0.0185
        if flat and self._omega_m0 < 1.0001:
            delta_c *= self.omega_m(redshift) ** 0.0055
        if redshift is None:
            return delta_c
        return delta_c / self.growth_factor(redshift)

    def delta_v(self, redshift=None):
        flat, open_ = self._flags()
        delta_v = 178.0
        if open_:
            delta_v /= self.omega_m(redshift) ** 0.7
        if flat and self._omega_m0 < 1.0001:
            delta_v /= self.omega_m(redshift) ** 0.55
        if redshift is None:
            return delta_v
        return delta_v / self.growth_factor(redshift)

    def rho_crit(self, redshift=None):
        if redshift is None:
            redshift = 0.0
        return 1.879 / (1.989) * 3.086 ** 3 * 1e10 * self.epoch0.E0(redshift)

    def rho_bar(self, redshift=None):
        return self.rho_crit(redshift) * self.omega_m(redshift)

    def delta_k(self, k, redshift=None):
        delta_k = self.epoch0.delta_k(k)
        if redshift is not None:
            delta_k = delta_k * self.growth_factor(redshift) ** 2
        return delta_k

    def sigma_m(self, mass, redshift=None):
        scale = (3.0 * numpy.asarray(mass, dtype=numpy.float64) /
                 (4.0 * numpy.pi * self.rho_bar(redshift))) ** (1.0 / 3.0)
        return self.sigma_r(scale, redshift)

    def nu_r(self, scale, redshift=None):
        sqrt_nu = self.delta_c(redshift) / self.sigma_r(scale, redshift)
        return sqrt_nu * sqrt_nu

    def nu_m(self, mass, redshift=None):
        sqrt_nu = self.delta_c(redshift) / self.sigma_m(mass, redshift)
        return sqrt_nu * sqrt_nu

    def write(self, output_file_name, output_power_file_name=None):
        """cosmology.py:1112-1164."""
        with open(output_file_name, "w") as f:
            f.write("#ttype1 = z\n#ttype2 = chi [Mpc/h]\n#ttype3 = growth\n"
                    "#ttype4 = omega_m\n#ttype5 = omega_l\n#ttype6 = delta_c\n"
                    "#ttype7 = delta_v\n#ttype8 = sigma_8\n")
            for z, chi, growth in zip(self._z_array, self._chi_array, self._growth_array):
                if z <= self.z_max:
                    f.write("%1.10f %1.10f %1.10f %1.10f %1.10f %1.10f %1.10f %1.10f\n" % (
                        z, chi, growth, self.omega_m(z), self.omega_l(z), self.delta_c(z),
                        self.delta_v(z), self.sigma_r(8.0, z)))
        if output_power_file_name is not None:
            ln_k_array = numpy.linspace(numpy.log(self._k_min), numpy.log(self._k_max), 100)
            k = numpy.exp(ln_k_array)
            with open(output_power_file_name, "w") as f:
                f.write("#ttype1 = k [Mpc/h]\n#ttype2 = linear_power [Mpc/h]^3\n")
                for row in zip(k, self.linear_power(k)):
                    f.write("%1.10f %1.10f\n" % row)
