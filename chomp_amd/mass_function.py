"""mass_function.MassFunction / TinkerMassFunction (mass_function.py:25-362,
436-564) over the HIP library: the mass-limit search, the 50 sigma(M) integrals,
the nu <-> ln M splines and the normalisations all run on the device."""
import numpy

from . import _lib
from . import cosmology
from . import defaults


class MassFunction(object):
    """Sheth-Tormen mass function and bias (mass_function.py:25-362)."""
    _kind = _lib.MF_ST

    def __init__(self, redshift=0.0, cosmo_single_epoch=None, halo_dict=None, **kws):
        self._redshift = redshift
        if cosmo_single_epoch is None:
            cosmo_single_epoch = cosmology.SingleEpoch(self._redshift)
        self.cosmo = cosmo_single_epoch
        self.cosmo.set_redshift(self._redshift)       # mutates the caller's object (:45)
        if halo_dict is None:
            halo_dict = defaults.default_halo_dict
        self.halo_dict = halo_dict
        self.stq = halo_dict["stq"]
        self.st_little_a = halo_dict["st_little_a"]
        self.c0 = halo_dict["c0"] / (1.0 + redshift)
        self._ctx = None
        self._sig = None
        self._sc = None

    # -- device state ----------------------------------------------------------
    def _signature(self):
        return (tuple(sorted(self.cosmo.cosmo_dict.items())), self.cosmo._redshift,
                tuple(sorted(self.halo_dict.items())), self._kind)

    def _dev(self):
        sig = self._signature()
        if self._ctx is None:
            self._ctx = cosmology._context()
        if sig != self._sig:
            self._ctx.epochs_set(self.cosmo.cosmo_dict, [self.cosmo._redshift],
                                 getattr(self.cosmo, "_with_bao", False))
            self._ctx.mass_setup(self.halo_dict, self._kind)
            self._sc = self._ctx.scalars(0)
            self._sig = sig
            self._ctx.warn_status(0, 1, stacklevel=4)   # saturated search, exhausted divmax
        return self._ctx

    def _scalar(self, name):
        self._dev()
        return float(self._sc[name])

    delta_c = property(lambda self: self._scalar("delta_c"))
    delta_v = property(lambda self: self._scalar("mf_delta_v"))
    ln_mass_min = property(lambda self: self._scalar("ln_mass_min"))
    ln_mass_max = property(lambda self: self._scalar("ln_mass_max"))
    nu_min = property(lambda self: self._scalar("nu_min"))
    nu_max = property(lambda self: self._scalar("nu_max"))
    m_star = property(lambda self: self._scalar("m_star"))
    f_norm = property(lambda self: self._scalar("f_norm"))
    bias_norm = property(lambda self: self._scalar("bias_norm"))
    _ln_mass_array = property(lambda self: self._dev().table("ln_mass"))
    _nu_array = property(lambda self: self._dev().table("nu"))

    # -- reference surface -----------------------------------------------------
    def get_redshift(self):
        return self._redshift

    def set_redshift(self, redshift):
        self._redshift = redshift
        self.cosmo.set_redshift(redshift)
        self.c0 = self.halo_dict["c0"] / (1.0 + redshift)

    def get_cosmology(self):
        return self.cosmo.get_cosmology()

    def set_cosmology(self, cosmo_dict, redshift=None):
        if redshift is None:
            redshift = self._redshift
        self._redshift = redshift
        self.cosmo.set_cosmology(cosmo_dict, redshift)
        self.c0 = self.halo_dict["c0"] / (1.0 + redshift)

    def set_cosmology_object(self, cosmo_single_epoch):
        self._redshift = cosmo_single_epoch.redshift()
        self.cosmo = cosmo_single_epoch
        self.c0 = self.halo_dict["c0"] / (1.0 + self._redshift)

    def get_halo(self):
        return self.halo_dict

    def set_halo(self, halo_dict):
        self.halo_dict = halo_dict
        self.stq = self.halo_dict["stq"]
        self.st_little_a = self.halo_dict["st_little_a"]
        self.c0 = self.halo_dict["c0"] / (1.0 + self._redshift)

    def f_nu(self, nu):
        return self._dev().eval("f_nu", nu)

    def f_m(self, mass):
        return self.f_nu(self.nu(mass))

    def bias_nu(self, nu):
        return self._dev().eval("bias_nu", nu)

    def bias_m(self, mass):
        return self.bias_nu(self.nu(mass))

    def nu(self, mass):
        return self._dev().eval("nu_of_mass", mass)

    def ln_mass(self, nu):
        return self._dev().eval("ln_mass_of_nu", nu)

    def mass(self, nu):
        return numpy.exp(self.ln_mass(nu))

    def dndm(self, mass):
        """mass_function.py:268-287: 0.5 rho_bar / M^2 f(nu(M)) dnu/dlnM, the derivative
        taken from the nu(ln M) spline."""
        m = numpy.atleast_1d(numpy.asarray(mass, dtype=numpy.float64))
        dev = self._dev()
        dnu = dev.spline_eval(self._ln_mass_array, self._nu_array, numpy.log(m), deriv=1)
        out = 0.5 * (self.cosmo.rho_bar() / (m * m) * self.f_m(m) * dnu)
        return out if numpy.ndim(mass) else float(out[0])

    def write(self, output_file_name):
        """mass_function.py:289-302."""
        print("M* = 10^%1.4f M_sun" % numpy.log10(self.m_star))
        nu = self._nu_array
        cols = (numpy.exp(self._ln_mass_array), nu, self.f_nu(nu), self.bias_nu(nu))
        with open(output_file_name, "w") as f:
            f.write("#ttype1 = mass [M_solar/h]\n#ttype2 = nu\n"
                    "#ttype3 = f(nu)\n#ttype4 = bias(nu)\n")
            for row in zip(*cols):
                f.write("%1.10f %1.10f %1.10f %1.10f\n" % row)


class TinkerMassFunction(MassFunction):
    """Tinker et al. 2010 (mass_function.py:436-564)."""
    _kind = _lib.MF_TINKER

    def _alpha(self):
        return self._scalar("t_alpha")

    def _beta(self):
        return self._scalar("t_beta")

    def _gamma(self):
        return self._scalar("t_gamma")

    def _phi(self):
        return self._scalar("t_phi")

    def _eta(self):
        return self._scalar("t_eta")
