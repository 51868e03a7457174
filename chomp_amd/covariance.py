"""covariance.Covariance for the Gaussian part of the w(theta) covariance
(covariance.py:23-543, 778-793, 1085-1103), the consumer of P(k) and the windows that
SURVEY.md 8(f) ranks fourth.

Accelerated: ``Covariance(corr, corr, nongaussian_cov=False)`` -- the use of
examples/example_covariance_script.py: the projected spectrum over ln K
(``_initialize_halo_splines``), ``covariance_G`` for every pair of bins in one launch,
the Poisson term, ``get_covariance`` and ``write``.  Outside the scope (ChompScopeError):
the trispectrum terms (``nongaussian_cov=True``, ``ssc_cov=True``: halo_trispectrum.py,
perturbation_spectra.py), ``CovarianceMulti`` and ``CovarianceFourier``.  Two different
correlation objects cannot be given to the reference's Covariance either (its constructor
raises ValueError comparing them), so that branch is not built.
"""
import numpy
from scipy import special

from . import _lib
from . import defaults
from . import kernel as kernel_mod
from .correlation import _POWER

deg_to_rad = numpy.pi / 180.0
rad_to_deg = 180.0 / numpy.pi
deg2_to_strad = deg_to_rad * deg_to_rad
strad_to_deg2 = rad_to_deg * rad_to_deg


class AnnulusBin(object):
    """covariance.py:1085-1103."""

    def __init__(self, inner, outer):
        self.inner = inner
        self.outer = outer
        self.center = numpy.power(10.0, 0.5 * (numpy.log10(inner) + numpy.log10(outer)))
        self.delta = outer - inner


class KernelCovariance(object):
    """The part of kernel.KernelCovariance (kernel.py:864-960) the Gaussian term uses:
    the four windows, the MultiEpoch and the common redshift / distance range.  Its own
    kernels (kernel_NG, kernel_ssc) belong to the trispectrum terms."""

    def __init__(self, ktheta_min, ktheta_max, window_function_a1, window_function_a2,
                 window_function_b1, window_function_b2, cosmo_multi_epoch,
                 force_quad=False):
        if force_quad:
            raise _lib.ChompScopeError("force_quad=True is outside the accelerated scope")
        self.ln_ktheta_min = numpy.log(ktheta_min)
        self.ln_ktheta_max = numpy.log(ktheta_max)
        self.window_function_a1 = window_function_a1
        self.window_function_a2 = window_function_a2
        self.window_function_b1 = window_function_b1
        self.window_function_b2 = window_function_b2
        ws = (window_function_a1, window_function_a2, window_function_b1, window_function_b2)
        self.z_min = numpy.max([w.z_min for w in ws])
        self.z_max = numpy.min([w.z_max for w in ws])
        self.cosmo = cosmo_multi_epoch

    def get_cosmology(self):
        return self.cosmo.get_cosmology()

    def kernel(self, ln_ktheta_a, ln_ktheta_b):
        raise _lib.ChompScopeError(
            "KernelCovariance.kernel_NG / kernel_ssc (kernel.py:987-1111) serve the "
            "trispectrum terms: outside the accelerated scope")

    kernel_NG = raw_kernel = raw_kernel_NG = kernel_ssc = kernel


class Covariance(object):
    """covariance.py:23-200."""

    def __init__(self, input_correlation_a, input_correlation_b,
                 bins_per_decade=5.0, survey_area_deg2=20,
                 n_a=1.0e4, n_b=1.0e4, variance=1.0, nongaussian_cov=True,
                 input_halo_trispectrum=None, power_spec='power_mm',
                 poisson_noise_only=False, ssc_cov=False, **kws):
        if nongaussian_cov or ssc_cov or input_halo_trispectrum is not None:
            raise _lib.ChompScopeError(
                "the trispectrum terms of the covariance (covariance_NG, covariance_ssc; "
                "halo_trispectrum.py) are outside the accelerated scope: pass "
                "nongaussian_cov=False, ssc_cov=False")
        if input_correlation_a is not input_correlation_b:
            # The reference cannot get here either: covariance.py:60 compares the two
            # correlations with Correlation.__eq__ (correlation.py:119-131), which compares
            # their attribute dictionaries -- numpy arrays included -- and raises
            # "ValueError: The truth value of an array ... is ambiguous" for two different
            # objects.  The four-spectra branch (covariance.py:497-532) is dead code as shipped.
            raise _lib.ChompScopeError(
                "Covariance of two different correlation objects: the reference raises "
                "ValueError at covariance.py:60 for them (Correlation.__eq__ compares numpy "
                "arrays); only Covariance(corr, corr) can run there, and that is what is "
                "accelerated")
        self.annular_bins = []
        self.log_theta_min = input_correlation_a.log_theta_min
        self.log_theta_max = input_correlation_a.log_theta_max
        unit_double = numpy.floor(self.log_theta_min) * bins_per_decade
        theta = numpy.power(10.0, unit_double / (1.0 * bins_per_decade))
        self.bins_per_decade = bins_per_decade
        self.corr_a = input_correlation_a
        self.corr_b = input_correlation_b
        self.matching_corrs = True
        while theta < numpy.power(10.0, self.log_theta_max):
            if (theta >= numpy.power(10.0, self.log_theta_min) and
                    theta < numpy.power(10.0, self.log_theta_max)):
                self.annular_bins.append(AnnulusBin(
                    theta, numpy.power(10.0, (unit_double + 1.0) / (1.0 * bins_per_decade))))
            unit_double += 1.0
            theta = numpy.power(10.0, unit_double / (1.0 * bins_per_decade))

        self.area = survey_area_deg2 * deg2_to_strad
        try:
            self.n_a1, self.n_a2 = n_a[0], n_a[1]
        except (TypeError, IndexError):
            self.n_a1 = self.n_a2 = n_a
        try:
            self.n_b1, self.n_b2 = n_b[0], n_b[1]
        except (TypeError, IndexError):
            self.n_b1 = self.n_b2 = n_b
        self.nongaussian_cov = False
        self.ssc_cov = False
        self.poisson_noise_only = poisson_noise_only

        kern = input_correlation_a.kernel
        self.kernel = KernelCovariance(
            numpy.power(10.0, self.log_theta_min) * defaults.default_limits["k_min"],
            numpy.power(10.0, self.log_theta_max) * defaults.default_limits["k_max"],
            kern.window_function_a, kern.window_function_b,
            kern.window_function_a, kern.window_function_b, kern.cosmo)
        # covariance.py:108-115.  The reference's Kernel holds *copies* of its two
        # windows, each with a private copy of the MultiEpoch, and WindowFunction.__eq__
        # (kernel.py:248-259) compares those by identity: two windows are "equal" only
        # when they are the same object.  With one correlation given twice that is the
        # case for the pairs (a1, b1) and (a2, b2) and for no other, whatever the windows.
        self.equal_windows = [False, False, False, False, True, True]
        self.density = [self.n_a1 / self.area, self.n_a2 / self.area,
                        self.n_b1 / self.area, self.n_b2 / self.area,
                        self.n_a1 / self.area, self.n_a2 / self.area]
        self.variance = variance
        self.cosmic_shear = self._identify_cosmic_shear()

        self.halo_a = input_correlation_a.halo
        self.halo_b = input_correlation_b.halo
        self._initialized_halo_splines = False
        self._table_key = None
        self._ln_k_min = numpy.log(defaults.default_limits['k_min'])
        self._ln_k_max = numpy.log(defaults.default_limits['k_max'])
        self._j0_limit = special.jn_zeros(
            0, defaults.default_precision["kernel_bessel_limit"])[-1]
        if power_spec is None:
            power_spec = 'linear_power'
        if power_spec not in _POWER or not hasattr(self.halo_a, power_spec):
            print("WARNING: Invalid input for power spectra variable,")
            print("\t setting to linear_power")
            power_spec = 'linear_power'
        self.power_spec = power_spec

    def _identify_cosmic_shear(self):
        shear = [isinstance(w, kernel_mod.WindowFunctionConvergence) for w in (
            self.kernel.window_function_a1, self.kernel.window_function_a2,
            self.kernel.window_function_b1, self.kernel.window_function_b2)]
        return [shear[0] * shear[1] or shear[2] * shear[3],
                shear[0] * shear[3] or shear[1] * shear[2]]

    # -- device tables -----------------------------------------------------------
    def _table(self):
        """Projected spectrum over ln K in the correlation's device context
        (covariance.py:455-543); rebuilt when anything it was built from has changed."""
        ctx, code = self.corr_a._prepare(self.power_spec)
        h = self.halo_a
        hod = h.get_hod_object()
        key = (id(ctx), self.corr_a.kernel._signature(), code, h._epoch_sig, h._mass_sig,
               tuple(getattr(hod, a, None) for a in ("log_M_min", "sigma", "log_M_0",
                                                     "log_M_1p", "alpha")),
               repr(sorted(h._profile_dict.items())), h.get_extrapolation(),
               self.corr_a.kernel.z_bar)
        if key != self._table_key or not self._initialized_halo_splines:
            self._z_bar_G_a = self._z_bar_G_b = self.corr_a.kernel.z_bar
            self._D_z_a = self._D_z_b = self.corr_a._growth_at_z_bar()
            self._ln_K_array, self._halo_a_array, self._halo_a_levels = \
                ctx.covariance_table(code, 0, self._D_z_a)
            self._ln_K_min, self._ln_K_max = self._ln_K_array[0], self._ln_K_array[-1]
            self._table_key = key
            self._initialized_halo_splines = True
        return ctx

    def _initialize_halo_splines(self):
        self._initialized_halo_splines = False
        self._table()

    def _projected_halo_a(self, K):
        ctx = self._table()
        return ctx.spline_eval(self._ln_K_array, self._halo_a_array, numpy.log(K))

    _projected_halo_b = _projected_halo_a

    def set_cosmology(self, cosmo_dict):
        self.corr_a.set_cosmology(cosmo_dict)
        self.halo_a = self.halo_b = self.corr_a.halo
        self._initialized_halo_splines = False

    def get_cosmology(self):
        return self.kernel.get_cosmology()

    # -- covariance --------------------------------------------------------------
    def get_covariance(self):
        """covariance.py:297-317; the Gaussian term of all bin pairs is one launch."""
        nb = len(self.annular_bins)
        self.covar = numpy.zeros((nb, nb))
        iu = numpy.triu_indices(nb)
        centers = numpy.array([b.center for b in self.annular_bins])
        if not self.poisson_noise_only and nb:
            vals = self._covariance_G_pairs(centers[iu[0]], centers[iu[1]])
            self.covar[iu] = vals
            self.covar[(iu[1], iu[0])] = vals
        for i, b in enumerate(self.annular_bins):
            self.covar[i, i] += self.covariance_P(b.delta, b.center)
        return self.covar

    def covariance(self, annular_bin_a, annular_bin_b):
        """covariance.py:319-351."""
        cov_P = 0.0
        if annular_bin_a is annular_bin_b and self.matching_corrs:
            cov_P = self.covariance_P(annular_bin_a.delta, annular_bin_a.center)
        if self.poisson_noise_only:
            return cov_P
        return self.covariance_G(annular_bin_a.center, annular_bin_b.center,
                                 annular_bin_a.delta, annular_bin_b.delta) + cov_P

    def covariance_P(self, delta, theta, window_1=0, window_2=1):
        """covariance.py:338-359."""
        term1 = (self.proj_power_poisson(0) * self.proj_power_poisson(2) *
                 (1. + self.cosmic_shear[0]))
        term2 = (self.proj_power_poisson(3) * self.proj_power_poisson(1) *
                 (1. + self.cosmic_shear[1]))
        term3 = (self.proj_power_poisson(4) * self.proj_power_poisson(5) *
                 (1. + self.cosmic_shear[1]))
        return (term1 + term2 + term3) / (2. * numpy.pi * self.area * theta * delta)

    def proj_power_poisson(self, window_pair=0):
        if self.equal_windows[window_pair]:
            return self.variance * self.variance / self.density[window_pair]
        return 0.0

    def _covariance_G_pairs(self, theta_a, theta_b):
        ctx = self._table()
        return ctx.covariance_gaussian(self._j0_limit, self.area,
                                       self.proj_power_poisson(0),
                                       self.proj_power_poisson(2), theta_a, theta_b)

    def covariance_G(self, theta_a, theta_b, delta_a=None, delta_b=None):
        """covariance.py:361-395 (the bin widths do not enter the integrand as shipped)."""
        ta = numpy.asarray(theta_a, dtype=numpy.float64)
        out = self._covariance_G_pairs(ta.ravel(), numpy.asarray(theta_b,
                                                                 dtype=numpy.float64).ravel())
        return float(out[0]) if ta.ndim == 0 else out.reshape(ta.shape)

    def covariance_NG(self, theta_a_rad, theta_b_rad):
        raise _lib.ChompScopeError("covariance_NG (halo trispectrum) is outside the "
                                   "accelerated scope")

    covariance_ssc = covariance_NG

    def write(self, file_name):
        """covariance.py:778-793."""
        with open(file_name, 'w') as f:
            f.write("#ttype1 = theta_a [deg]\n#ttype2 = theta_b [deg]\n" +
                    "#ttype3 = covariance\n")
            for idx_a, bin_a in enumerate(self.annular_bins):
                for idx_b, bin_b in enumerate(self.annular_bins):
                    f.writelines('%1.16f %1.16f %1.16f\n' % (
                        bin_a.center * rad_to_deg, bin_b.center * rad_to_deg,
                        self.covar[idx_a, idx_b]))


class FiniteAreaEffect(object):
    """Fitting formula for the finite-survey-area correction of Sato et al. 2011, App. A
    (covariance.py:1106-1139): two power laws in the source redshift; host arithmetic."""

    def __init__(self):
        self.alpha1 = 3.2952
        self.alpha2 = -0.316369
        self.beta1 = 0.170708
        self.beta2 = -0.349913

    def alpha(self, zs):
        return self.alpha1 * zs ** self.alpha2

    def beta(self, zs):
        return self.beta1 * zs ** self.beta2

    def area_scaling(self, area, zs):
        return self.alpha(zs) / area ** self.beta(zs)
