"""correlation.Correlation / CorrelationFourier / Correlation3d with the reference's
constructors and methods (correlation.py:33-289, 297-405, 408-510): one Romberg integral
per theta or r (over ln k) or per multipole (Limber, over chi), each on its own group of
wavefronts.
"""
import numpy

from . import _lib
from . import halo as halo_mod

speed_of_light = 3 * 10 ** 5
deg_to_rad = numpy.pi / 180.0
rad_to_deg = 180.0 / numpy.pi

_POWER = {"linear_power": (_lib.P_LIN, 0), "power_mm": (_lib.P_MM, _lib.FAM_MM),
          "power_gm": (_lib.P_GM, _lib.FAM_GM), "power_mg": (_lib.P_GM, _lib.FAM_GM),
          "power_gg": (_lib.P_GG, _lib.FAM_GG)}


class Correlation(object):
    """w(theta) (correlation.py:33-289)."""

    def __init__(self, theta_min_deg, theta_max_deg, input_kernel,
                 bins_per_decade=5.0, input_halo=None, power_spec=None,
                 k_min=None, k_max=None, keep_halo_z_bar=False, **kws):
        self.log_theta_min = numpy.log10(theta_min_deg * deg_to_rad)
        self.log_theta_max = numpy.log10(theta_max_deg * deg_to_rad)
        theta_array = []
        unit_double = numpy.floor(self.log_theta_min) * bins_per_decade
        theta = numpy.power(10.0, unit_double / (1.0 * bins_per_decade))
        while theta < numpy.power(10.0, self.log_theta_max):
            if (theta >= numpy.power(10.0, self.log_theta_min) and
                    theta < numpy.power(10.0, self.log_theta_max)):
                theta_array.append(10 ** (0.5 * (
                    numpy.log10(theta) + (unit_double + 1.0) / (1.0 * bins_per_decade))))
            unit_double += 1.0
            theta = numpy.power(10.0, unit_double / (1.0 * bins_per_decade))
        self.theta_array = numpy.array(theta_array)
        if theta_min_deg == theta_max_deg:
            self.theta_array = numpy.array([theta_min_deg * deg_to_rad])
        self.wtheta_array = numpy.zeros(self.theta_array.size)
        self.kernel = input_kernel
        if input_halo is None:
            input_halo = halo_mod.Halo(self.kernel.z_bar)
        self.halo = input_halo
        self.D_z = self._growth_at_z_bar()
        if not keep_halo_z_bar:
            self.halo.set_redshift(self.kernel.z_bar)
        if ((k_min is not None and k_min < self.halo._k_min) or
                (k_max is not None and k_max > self.halo._k_max)):
            self.halo.set_extrapolation(True)
        if k_min is None:
            k_min = self.halo._k_min
        self._ln_k_min = numpy.log(k_min)
        if k_max is None:
            k_max = self.halo._k_max
        self._ln_k_max = numpy.log(k_max)
        self._k_lim = (float(k_min), float(k_max))
        self.set_power_spectrum(power_spec)

    def _growth_at_z_bar(self):
        if self.kernel._z_bar_override is None:
            return self.kernel._get("D_zbar")
        return float(self.kernel.cosmo.growth_factor(self.kernel.z_bar))

    def get_redshift(self):
        return self.kernel.z_bar

    def set_redshift(self, redshift):
        self.kernel.z_bar = redshift
        self.D_z = self._growth_at_z_bar()
        self.halo.set_redshift(self.kernel.z_bar)

    def get_cosmology(self):
        return self.kernel.get_cosmology()

    def set_cosmology(self, cosmo_dict):
        self.kernel.set_cosmology(cosmo_dict)
        self.D_z = self._growth_at_z_bar()
        self.halo.set_cosmology(cosmo_dict, self.kernel.z_bar)

    def get_power_spectrum(self):
        return self._power_name

    def set_power_spectrum(self, powSpec):
        if powSpec is None:
            powSpec = 'linear_power'
        if powSpec not in _POWER or not hasattr(self.halo, powSpec):
            print("WARNING: Invalid input for power spectra variable,")
            print("\t setting to linear_power")
            powSpec = 'linear_power'
        self._power_name = powSpec
        self.power_spec = getattr(self.halo, powSpec)

    def get_halo(self):
        return self.halo.get_halo()

    def set_halo(self, halo_dict):
        self.halo.set_halo(halo_dict)

    def get_hod(self, return_object=False):
        return self.halo.get_hod(return_object)

    def set_hod(self, hod_dict):
        self.halo.set_hod(hod_dict)

    def set_hod_object(self, input_hod):
        self.halo.set_hod_object(input_hod)

    def _prepare(self, power_name=None, defer_status=False):
        """Halo tables and projection tables in ONE device context.  defer_status: see
        Halo._sync (the device-resident evaluation path synchronises nowhere)."""
        code, need = _POWER[self._power_name if power_name is None else power_name]
        # The projection tables first: their set-up runs on the context's side stream, beside
        # the halo set-up queued next (neither needs anything of the other).
        self.kernel._setup_on(self.halo._context())
        if isinstance(self.halo, halo_mod.HaloFit) and code != _lib.P_LIN:
            code |= _lib.P_HALOFIT
            if (code & 15) == _lib.P_MM:
                need = 0
            ctx = self.halo._ensure_halofit(need, defer_status=defer_status)
        else:
            ctx = self.halo._sync(need, defer_status=defer_status)
        return ctx, self.halo._power_code(code)

    def compute_correlation(self):
        self.wtheta_array = numpy.asarray(self.correlation(self.theta_array))

    def correlation(self, theta_rad):
        th = numpy.asarray(theta_rad, dtype=numpy.float64)
        ctx, code = self._prepare(defer_status=True)
        out = ctx.wtheta(code, 0, self._k_lim[0], self._k_lim[1], self.D_z,
                         numpy.ascontiguousarray(th).ravel())
        self.halo._resolve_status()            # (the result is on the host: nothing to hide)
        return float(out[0]) if th.ndim == 0 else out.reshape(th.shape)

    def write(self, output_file_name):
        with open(output_file_name, "w") as f:
            f.write("#ttype1 = theta [deg]\n#ttype2 = wtheta\n")
            for theta, wtheta in zip(self.theta_array, self.wtheta_array):
                f.write("%1.10g %1.10g\n" % (theta / deg_to_rad, wtheta))


class CorrelationProjectedComoving(Correlation):
    """Empty in the reference too (correlation.py:291-295)."""

    def __init__(self, r_min_Mpc, r_max_Mpc, input_kernel,
                 bins_per_decade=5.0, input_halo=None, power_spec=None, **kws):
        pass


class CorrelationFourier(Correlation):
    """Limber C_l (correlation.py:297-405)."""

    def __init__(self, l_min, l_max, input_kernel, input_halo=None, powSpec=None,
                 **kws):
        from . import defaults
        self.log_l_min = numpy.log10(l_min)
        self.log_l_max = numpy.log10(l_max)
        self.l_array = numpy.logspace(self.log_l_min, self.log_l_max,
                                      defaults.default_precision["corr_npoints"])
        if l_min == l_max:
            self.l_array = numpy.array([l_min])
        self.power_array = numpy.zeros(self.l_array.size, dtype='float64')
        self.kernel = input_kernel
        if input_halo is None:
            input_halo = halo_mod.Halo(self.kernel.z_bar)
        self.halo = input_halo
        self.D_z = self._growth_at_z_bar()
        self.halo.set_redshift(self.kernel.z_bar)
        self.set_power_spectrum(powSpec)

    def compute_correlation(self):
        self.power_array = numpy.asarray(self.correlation(self.l_array))

    def correlation(self, l):
        la = numpy.asarray(l, dtype=numpy.float64)
        ctx, code = self._prepare(defer_status=True)
        out = ctx.cell(code, 0, self.D_z, numpy.ascontiguousarray(la).ravel())
        self.halo._resolve_status()
        return float(out[0]) if la.ndim == 0 else out.reshape(la.shape)

    def write(self, output_file_name):
        with open(output_file_name, "w") as f:
            f.write("#ttype1 = l [deg]\n#ttype2 = power\n")
            for l, power in zip(self.l_array, self.power_array):
                f.write("%1.10f %1.10f\n" % (l, power))


class Correlation3d(Correlation):
    """xi(r) from a halo-model spectrum (correlation.py:408-510): corr_npoints
    log-spaced separations, one Romberg integral of k^2/(2 pi) P(k) J0(k r) each -- the
    cylindrical J0, as the reference has it -- and a cubic spline in r."""

    def __init__(self, r_min, r_max, redshift=0.0, input_halo=None, powSpec=None,
                 k_min=None, k_max=None):
        from . import defaults
        self.log_r_min = numpy.log10(r_min)
        self.log_r_max = numpy.log10(r_max)
        self.r_array = numpy.logspace(self.log_r_min, self.log_r_max,
                                      defaults.default_precision["corr_npoints"])
        if r_min == r_max:
            self.r_array = numpy.array([r_min])
        self.xi_array = numpy.zeros(self.r_array.size)
        if input_halo is None:
            input_halo = halo_mod.Halo(redshift)
        self.halo = input_halo
        self.halo.set_redshift(redshift)
        if ((k_min is not None or k_max is not None) and
                not self.halo.get_extrapolation() and
                (k_min < self.halo._k_min or k_max > self.halo._k_max)):
            self.halo.set_extrapolation(True)
        if k_min is None:
            k_min = self.halo._k_min
        self._ln_k_min = numpy.log(k_min)
        if k_max is None:
            k_max = self.halo._k_max
        self._ln_k_max = numpy.log(k_max)
        self._k_lim = (float(k_min), float(k_max))
        self.set_power_spectrum(powSpec)
        self.initialized_spline = False

    def _prepare(self):
        code, need = _POWER[self._power_name]
        if isinstance(self.halo, halo_mod.HaloFit) and code != _lib.P_LIN:
            self.halo._ensure_halofit()     # (xi(r) returns to the host: nothing to defer)
            code |= _lib.P_HALOFIT
            if (code & 15) == _lib.P_MM:
                need = 0
        return self.halo._sync(need), self.halo._power_code(code)

    def compute_correlation(self):
        self.xi_array = numpy.asarray(self.raw_correlation(self.r_array))
        self.initialized_spline = True

    def raw_correlation(self, r):
        ra = numpy.asarray(r, dtype=numpy.float64)
        ctx, code = self._prepare()
        out = ctx.xi3d(code, 0, self._k_lim[0], self._k_lim[1],
                       numpy.ascontiguousarray(ra).ravel())
        return float(out[0]) if ra.ndim == 0 else out.reshape(ra.shape)

    def correlation(self, r):
        if not self.initialized_spline:
            self.compute_correlation()
        ra = numpy.asarray(r, dtype=numpy.float64)
        r_min, r_max = 10. ** self.log_r_min, 10. ** self.log_r_max
        inside = numpy.logical_and(ra <= r_max, ra > r_min)
        out = numpy.zeros(ra.shape)
        if numpy.any(inside):
            ctx, _ = self._prepare()
            out[inside] = ctx.spline_eval(self.r_array, self.xi_array, ra[inside])
        return out

    def write(self, output_file_name):
        with open(output_file_name, "w") as f:
            f.write("#ttype1 = r [Mpc/h]\n#ttype2 = xi\n")
            for r, xi in zip(self.r_array, self.xi_array):
                f.write("%1.10g %1.10g\n" % (r, xi))
