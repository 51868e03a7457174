"""kernel.dNdz*, WindowFunction*, Kernel, GalaxyGalaxyLensingKernel with the
reference's constructors and methods (kernel.py:26-208, 211-484, 559-839), computed
on the MI355X: dN/dz normalisation, the windows' chi tables (with the lensing-
efficiency integral per knot), z_bar and the 50 Bessel-weighted kernel knots.

Deliberate deviation: Kernel.__init__ of the reference writes two debug files
('test_window_before' / 'test_window_after', kernel.py:606-608) into the current
directory; that side effect is not reproduced.
"""
import numpy

from . import _lib
from . import cosmology
from . import defaults


class dNdz(object):
    """Base redshift distribution (kernel.py:26-86): a boxcar between z_min and z_max.  The
    normalisation integral is done on the device as part of a Kernel / window set-up;
    ``norm`` is filled in from there."""
    _kind = _lib.DNDZ_BOXCAR

    def raw_dndz(self, redshift):
        return 1.0 + 0.0 * numpy.asarray(redshift, dtype=numpy.float64)

    def _params(self):
        return ()

    def __init__(self, z_min, z_max):
        self.z_min = z_min
        self.z_max = z_max
        self.norm = 1.0

    def normalize(self):
        self.norm = _norm_of(self)

    def set_limits(self, z_min=None, z_max=None, calc_norm=False):
        if z_min is not None:
            self.z_min = z_min
        if z_max is not None:
            self.z_max = z_max
        if calc_norm:
            self.normalize()

    def dndz(self, redshift):
        z = numpy.asarray(redshift, dtype=numpy.float64)
        self.normalize()
        with numpy.errstate(all="ignore"):
            return numpy.where(numpy.logical_and(z <= self.z_max, z >= self.z_min),
                               self.norm * self.raw_dndz(z), 0.0)

    def _struct(self):
        if self._kind is None or (self._kind == _lib.DNDZ_BOXCAR and type(self).raw_dndz
                                  is not dNdz.raw_dndz):
            raise _lib.ChompScopeError(
                "%s is outside the accelerated scope (dNdz, dNdzMagLim, dNdzGaussian and "
                "dNdzInterpolation are)" % type(self).__name__)
        d = _lib.Dndz()
        d.kind = self._kind
        d.z_min, d.z_max = float(self.z_min), float(self.z_max)
        for i, v in enumerate(self._params()):
            d.p[i] = float(v)
        return d


class dNdzGaussian(dNdz):
    """kernel.py:89-112."""
    _kind = _lib.DNDZ_GAUSSIAN

    def __init__(self, z_min, z_max, z0, sigma_z):
        if z_min < z0 - 8.0 * sigma_z:
            z_min = z0 - 8.0 * sigma_z
        if z_max > z0 + 8.0 * sigma_z:
            z_max = z0 + 8.0 * sigma_z
        self.z0 = z0
        self.sigma_z = sigma_z
        dNdz.__init__(self, z_min, z_max)

    def _params(self):
        return (self.z0, self.sigma_z)

    def raw_dndz(self, redshift):
        return numpy.exp(-1.0 * (redshift - self.z0) * (redshift - self.z0) /
                         (2.0 * self.sigma_z * self.sigma_z))


class dNdzMagLim(dNdz):
    """kernel.py:148-179.  ``1/b`` is Python-2 integer division when b is an int
    (kernel.py:167): dNdzMagLim(0, 2, 2, 0.3, 2) and (..., 2.0) differ, as there."""
    _kind = _lib.DNDZ_MAGLIM

    def __init__(self, z_min, z_max, a, z0, b):
        self.a = a
        self.z0 = z0
        self.b = b
        inv_b = (1 // b) if isinstance(b, int) else 1 / b
        tmp_zmax = numpy.power(
            -1 * numpy.log(defaults.default_precision['dNdz_precision']), inv_b) * z0
        if tmp_zmax < z_max:
            print("WARNING:: z_max requested could result in failed normalization...")
            print("\tReseting z_max from %.2f to %.2f..." % (z_max, tmp_zmax))
            z_max = tmp_zmax
        dNdz.__init__(self, z_min, z_max)

    def _params(self):
        return (self.a, self.z0, self.b)

    def raw_dndz(self, redshift):
        return (numpy.power(redshift, self.a) *
                numpy.exp(-1.0 * numpy.power(redshift / self.z0, self.b)))


class dNdzInterpolation(dNdz):
    """p(z) tabulated by the caller (kernel.py:181-208).  The constructor fits the same FITPACK
    spline as the reference (order 2 through the points, or a smoothing spline) -- host work,
    done once; the device evaluates it as a piecewise polynomial."""
    _kind = _lib.DNDZ_PPOLY

    def __init__(self, z_array, p_array, weights=None, interpolation_order=2,
                 smoothing=None):
        from scipy.interpolate import (InterpolatedUnivariateSpline, PPoly,
                                       UnivariateSpline)
        if smoothing is None:
            self._p_of_z = InterpolatedUnivariateSpline(z_array, p_array, w=weights,
                                                        k=interpolation_order)
        else:
            self._p_of_z = UnivariateSpline(z_array, p_array, w=weights,
                                            k=interpolation_order, s=smoothing)
        dNdz.__init__(self, z_array[0], z_array[-1])
        pp = PPoly.from_spline(self._p_of_z._eval_args)
        keep = numpy.diff(pp.x) > 0.0                 # (the repeated end knots: empty pieces)
        self._breaks = numpy.ascontiguousarray(
            numpy.concatenate([pp.x[:-1][keep], pp.x[-1:]]), dtype=numpy.float64)
        # PPoly: c[m, i] multiplies (z - x_i)^(k - m); the ABI wants ascending powers per piece
        self._coef = numpy.ascontiguousarray(pp.c[::-1, keep].T, dtype=numpy.float64)

    def raw_dndz(self, redshift):
        return self._p_of_z(redshift)

    def _params(self):
        return (self._breaks.tobytes(), self._coef.tobytes())

    def _struct(self):
        import ctypes
        d = _lib.Dndz()
        d.kind = self._kind
        d.z_min, d.z_max = float(self.z_min), float(self.z_max)
        d.pp_breaks = self._breaks.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        d.pp_coef = self._coef.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        d.pp_n = int(self._coef.shape[0])
        d.pp_order = int(self._coef.shape[1] - 1)
        return d


class dNdChiGaussian(dNdzInterpolation):
    """Gaussian in comoving distance (kernel.py:114-145): raw_dndz(z) = exp(-(chi(z) - chi0)^2
    / (2 sigma_chi^2)) with chi(z) from the distribution's own MultiEpoch (0 <= z <= 5 unless
    one is given).  chi(z) lives on the device; the distribution is handed to the projection
    kernels as a piecewise polynomial through 4097 points of it (cubic, relative error of the
    tabulation < 1e-9: far inside the Romberg tolerances that consume it)."""
    _n_tab = 4097

    def __init__(self, chi_min, chi_max, chi0, sigma_chi, cosmo_multi_epoch=None):
        from scipy.interpolate import InterpolatedUnivariateSpline, PPoly
        if cosmo_multi_epoch is None:
            cosmo_multi_epoch = cosmology.MultiEpoch(0.0, 5.0)
        self.cosmo = cosmo_multi_epoch
        z_min = float(self.cosmo.redshift(chi_min))
        z_max = float(self.cosmo.redshift(chi_max))
        self.chi0 = chi0
        self.sigma_chi = sigma_chi
        dNdz.__init__(self, z_min, z_max)
        z_tab = numpy.linspace(z_min, z_max, self._n_tab)
        self._p_of_z = InterpolatedUnivariateSpline(z_tab, self.raw_dndz(z_tab), k=3)
        pp = PPoly.from_spline(self._p_of_z._eval_args)
        keep = numpy.diff(pp.x) > 0.0
        self._breaks = numpy.ascontiguousarray(
            numpy.concatenate([pp.x[:-1][keep], pp.x[-1:]]), dtype=numpy.float64)
        self._coef = numpy.ascontiguousarray(pp.c[::-1, keep].T, dtype=numpy.float64)

    def raw_dndz(self, redshift):
        chi = self.cosmo.comoving_distance(redshift)
        return numpy.exp(-1.0 * (chi - self.chi0) * (chi - self.chi0) /
                         (2.0 * self.sigma_chi * self.sigma_chi))


def _norm_of(dist):
    """dNdz.normalize (kernel.py:43-54) on the device."""
    w = WindowFunctionGalaxy(dist)
    return float(w._dev().kernel_info()["norm_a"])


class WindowFunction(object):
    """kernel.py:211-355 (base)."""
    _kind = None

    def __init__(self, z_min, z_max, cosmo_multi_epoch=None, **kws):
        if z_min < defaults.default_precision['window_precision']:
            z_min = defaults.default_precision['window_precision']
        self.z_min = z_min
        self.z_max = z_max
        if cosmo_multi_epoch is None:
            cosmo_multi_epoch = cosmology.MultiEpoch(z_min, z_max)
        self.cosmo = cosmo_multi_epoch
        self._ctx = None
        self._sig = None

    def _struct(self):
        if self._kind is None:
            raise _lib.ChompScopeError(
                "%s is outside the accelerated scope (WindowFunctionGalaxy and "
                "WindowFunctionConvergence are)" % type(self).__name__)
        w = _lib.Window()
        w.kind = self._kind
        w.dist = self._redshift_dist._struct()
        return w

    def _signature(self):
        d = self._redshift_dist
        return (self._kind, d._kind, d.z_min, d.z_max, tuple(d._params()),
                tuple(sorted(self.cosmo.cosmo_dict.items())))

    def _dev(self):
        """A stand-alone window is tabulated as a (window x window, J0) kernel."""
        if self._ctx is None:
            self._ctx = cosmology._context()
        sig = self._signature()
        if sig != self._sig:
            s = self._struct()
            self._ctx.kernel_setup(self.cosmo.cosmo_dict, self.z_min, self.z_max,
                                   1e-6, 1.0, s, s, 0)
            self._sig = sig
        return self._ctx

    chi_min = property(lambda self: float(self._dev().kernel_info()["wa_chi_min"]))
    chi_max = property(lambda self: float(self._dev().kernel_info()["wa_chi_max"]))
    _chi_array = property(lambda self: self._dev().kernel_table("wa_chi"))
    _wf_array = property(lambda self: self._dev().kernel_table("wa"))

    def get_cosmology(self):
        return self.cosmo.get_cosmology()

    def set_cosmology_object(self, cosmo_multi_epoch):
        self.cosmo = cosmo_multi_epoch

    def set_cosmology(self, cosmo_dict):
        self.cosmo = cosmology.MultiEpoch(self.z_min, self.z_max, cosmo_dict)

    def write(self, output_file_name):
        """kernel.py:342-355 (header text as in the reference, "/n" included)."""
        with open(output_file_name, "w") as f:
            f.write("#ttype1 = chi [Mpc/h]/n#ttype2 = window function value\n")
            for chi, wf in zip(self._chi_array, self._wf_array):
                f.write("%1.10f %1.10f\n" % (chi, wf))

    def window_function(self, chi):
        return self._dev().window_eval(0, numpy.asarray(chi, dtype=numpy.float64))


class WindowFunctionGalaxy(WindowFunction):
    """W(chi) = dN/dz dz/dchi (kernel.py:358-387)."""
    _kind = _lib.WINDOW_GALAXY

    def __init__(self, redshift_dist, cosmo_multi_epoch=None, **kws):
        self._redshift_dist = redshift_dist
        WindowFunction.__init__(self, redshift_dist.z_min, redshift_dist.z_max,
                                cosmo_multi_epoch)


class WindowFunctionConvergence(WindowFunction):
    """Lensing convergence window (kernel.py:410-484)."""
    _kind = _lib.WINDOW_CONVERGENCE

    def __init__(self, redshift_dist, cosmo_multi_epoch=None, **kws):
        self._redshift_dist = redshift_dist
        WindowFunction.__init__(self, 0.0, redshift_dist.z_max, cosmo_multi_epoch, **kws)


class WindowFunctionFlatConvergence(WindowFunction):
    """Constant lensing window between z_min and z_max (kernel.py:487-513)."""
    _kind = _lib.WINDOW_FLAT_CONVERGENCE

    def __init__(self, z_min, z_max, cosmo_multi_epoch=None, **kws):
        WindowFunction.__init__(self, z_min, z_max, cosmo_multi_epoch, **kws)
        self._redshift_dist = dNdz(self.z_min, self.z_max)     # (carries the limits only)


class WindowFunctionConvergenceDelta(WindowFunction):
    """Lensing window of sources on one plane (kernel.py:516-556)."""
    _kind = _lib.WINDOW_CONVERGENCE_DELTA

    def __init__(self, redshift, cosmo_multi_epoch=None, **kws):
        self._redshift = redshift
        self._g_chi_min = 0.0
        WindowFunction.__init__(self, 0.0, redshift, cosmo_multi_epoch, **kws)
        self._redshift_dist = dNdz(0.0, redshift)              # (carries the limits only)


class Kernel(object):
    """K(k theta) = int dchi W_a W_b D^2 J0(k theta chi) (kernel.py:559-781)."""
    _order = 0

    def __init__(self, ktheta_min, ktheta_max, window_function_a, window_function_b,
                 cosmo_multi_epoch=None, force_quad=False, **kws):
        if force_quad:
            raise _lib.ChompScopeError(
                "force_quad=True (scipy.integrate.quad, kernel.py:692-697) is outside "
                "the accelerated scope")
        self.ln_ktheta_min = numpy.log(ktheta_min)
        self.ln_ktheta_max = numpy.log(ktheta_max)
        self._ktheta = (float(ktheta_min), float(ktheta_max))
        self.window_function_a = window_function_a
        self.window_function_b = window_function_b
        self.z_min = numpy.max([window_function_a.z_min, window_function_b.z_min])
        self.z_max = numpy.min([window_function_a.z_max, window_function_b.z_max])
        if cosmo_multi_epoch is None:
            cosmo_multi_epoch = cosmology.MultiEpoch(self.z_min, self.z_max)
        self.cosmo = cosmo_multi_epoch
        self._force_quad = force_quad
        self._own = None
        self._done = {}
        self._info = None
        self._info_sig = None
        self._info_ctx = None
        self._z_bar_override = None

    # -- device state: the tables can live in several contexts (the kernel's own,
    #    and the halo's when a Correlation joins them) --------------------------
    def _signature(self):
        return (self.window_function_a._signature(), self.window_function_b._signature(),
                tuple(sorted(self.cosmo.cosmo_dict.items())), self.cosmo.z_min,
                self.cosmo.z_max, self._ktheta, self._order)

    def _setup_on(self, ctx):
        sig = self._signature()
        mine = (id(self), sig)           # (another Kernel may have used this context since)
        if self._done.get(id(ctx)) != sig or getattr(ctx, "_proj_owner", None) != mine:
            ctx.kernel_setup(self.cosmo.cosmo_dict, self.cosmo.z_min, self.cosmo.z_max,
                             self._ktheta[0], self._ktheta[1],
                             self.window_function_a._struct(),
                             self.window_function_b._struct(), self._order)
            self._done[id(ctx)] = sig
            ctx._proj_owner = mine
            self._info = None            # (read back -- a synchronisation -- when first asked for)
            self._info_ctx = ctx
        return ctx

    def _dev(self):
        if self._own is None:
            self._own = cosmology._context()
        return self._setup_on(self._own)

    def _get(self, name):
        """Host scalars of the projection set-up (z_bar, chi limits, D(z_bar), ...), read back
        once per set-up.  A read-back made for other windows / another cosmology is dropped:
        set_cosmology re-runs _find_z_bar and the growth factor in the reference
        (kernel.py:657-676), so must every scalar served from here."""
        sig = self._signature()
        if self._info is not None and self._info_sig != sig:
            self._info = None
        if self._info is None:
            ctx = self._info_ctx if self._info_ctx is not None else self._dev()
            self._info = self._setup_on(ctx).kernel_info()
            self._info_sig = sig
        return float(self._info[name])

    chi_min = property(lambda self: self._get("chi_min"))
    chi_max = property(lambda self: self._get("chi_max"))
    _j0_limit = property(lambda self: self._get("j_limit"))
    _ln_ktheta_array = property(lambda self: self._dev().kernel_table("ln_ktheta"))
    _kernel_array = property(lambda self: self._dev().kernel_table("kernel"))

    @property
    def z_bar(self):
        if self._z_bar_override is not None:
            return self._z_bar_override
        return self._get("z_bar")

    @z_bar.setter
    def z_bar(self, value):          # Correlation.set_redshift assigns it (correlation.py:151)
        self._z_bar_override = value

    def get_cosmology(self):
        return self.cosmo.get_cosmology()

    def set_cosmology(self, cosmo_dict):
        """kernel.py:657-676: new MultiEpoch tables, windows, chi limits, and _find_z_bar again
        (which also replaces a z_bar assigned through Correlation.set_redshift)."""
        self.cosmo.set_cosmology(cosmo_dict)
        self.window_function_a.set_cosmology_object(self.cosmo)
        self.window_function_b.set_cosmology_object(self.cosmo)
        self._z_bar_override = None

    def write(self, output_file_name):
        """kernel.py:765-781."""
        with open(output_file_name, "w") as f:
            f.write("#ttype1 = k*theta [h/Mpc*Radians]\n"
                    "#ttype2 = kernel [(h/Mpc)^2]\n")
            for ln_ktheta, kernel in zip(self._ln_ktheta_array, self._kernel_array):
                f.write("%1.10g %1.10g\n" % (numpy.exp(ln_ktheta), kernel))

    def kernel(self, ln_ktheta):
        return self._dev().kernel_eval(numpy.asarray(ln_ktheta, dtype=numpy.float64))

    def raw_kernel(self, ln_ktheta):
        """kernel.py:678-704: the chi integral at ln(k theta) itself (no spline)."""
        x = numpy.asarray(ln_ktheta, dtype=numpy.float64)
        out = self._dev().kernel_raw(numpy.ascontiguousarray(x).ravel())
        return float(out[0]) if x.ndim == 0 else out.reshape(x.shape)

    def kernel_weighted_mean(self, function):
        raise _lib.ChompScopeError(
            "kernel_weighted_mean (kernel.py:731-753) integrates a Python callable: "
            "outside the accelerated scope")


class GalaxyGalaxyLensingKernel(Kernel):
    """J2 variant (kernel.py:784-839)."""
    _order = 2
    _j2_limit = property(lambda self: self._get("j_limit"))
