"""halo.Halo / halo.HaloFit with the reference's constructor, setters and power_*
methods (halo.py:23-1086, 1236-1412), every integral and spline on the MI355X.

The object keeps the reference's lazy-initialisation flags (halo.py:93-104) and
their exact invalidation rules, including the ones that look like oversights,
because they change the numbers a drop-in user gets:

* set_halo() hands the new dictionary to the mass function only; the profile
  splines (c0, beta) and the Halo's delta_v keep the constructor's values and the
  h_m / pp_mm flags are NOT reset (halo.py:220-235);
* HaloFit fixes f_1..f_3 at construction and builds its sigma-spline once, at the
  first power_mm call; set_redshift()/set_cosmology() never refresh either
  (halo.py:1254-1266, 1337-1338).
"""
import numpy

from . import _lib
from . import cosmology
from . import defaults
from . import hod
from . import mass_function

_FLAG_BITS = (("_initialized_h_m", _lib.T_H_M), ("_initialized_pp_mm", _lib.T_PP_MM),
              ("_initialized_h_g", _lib.T_H_G), ("_initialized_pp_gm", _lib.T_PP_GM),
              ("_initialized_pp_gg", _lib.T_PP_GG))


class Halo(object):
    """Seljak (2000) halo model (halo.py:23-1086)."""

    def __init__(self, redshift=0.0, input_hod=None, cosmo_single_epoch=None,
                 mass_func=None, halo_dict=None, extrapolate=False, **kws):
        self._k_min = defaults.default_limits['k_min']
        self._k_max = defaults.default_limits['k_max']
        self._ln_k_max = numpy.log(self._k_max)
        self._ln_k_min = numpy.log(self._k_min)
        self._ln_k_array = numpy.linspace(
            self._ln_k_min, self._ln_k_max,
            defaults.default_precision["halo_npoints"])
        self._redshift = redshift
        if cosmo_single_epoch is None:
            cosmo_single_epoch = cosmology.SingleEpoch(redshift)
        self.cosmo = cosmo_single_epoch
        if halo_dict is None:
            halo_dict = defaults.default_halo_dict
        self.halo_dict = halo_dict
        if mass_func is None:
            mass_func = mass_function.MassFunction(
                self._redshift, self.cosmo, self.halo_dict)
        self.mass = mass_func
        self.c0 = halo_dict["c0"] / (1.0 + self._redshift)
        self.beta = halo_dict["beta"]
        self.alpha = halo_dict["alpha"]
        if self.alpha != -1.0:
            raise _lib.ChompScopeError(
                "halo alpha != -1 (y_general, halo.py:491-559) is outside the "
                "hot-path scope: NFW only")
        self._h = self.cosmo._h
        if input_hod is None:
            input_hod = hod.HODZheng()
        self.local_hod = input_hod
        self._extrapolate = extrapolate
        # the dictionary the PROFILE sees: fixed here, untouched by set_halo
        self._profile_dict = dict(halo_dict)
        self._ctx = None
        self._epoch_sig = None
        self._mass_sig = None
        self._nbar_valid = False
        self._reset_flags(all_tables=True)

    # -- device orchestration --------------------------------------------------
    def _reset_flags(self, all_tables):
        if all_tables:
            self._initialized_h_m = False
            self._initialized_pp_mm = False
        self._initialized_h_g = False
        self._initialized_pp_gm = False
        self._initialized_pp_gg = False

    def _profile(self):
        # delta_v of the Halo is re-read from self.halo_dict on set_cosmology
        # (halo.py:151-153) but c0/beta stay with the splines built at __init__ /
        # set_cosmology from self.halo_dict as well; set_halo changes neither.
        return self._profile_dict

    _status_pending = False
    _status_word = 0

    def _resolve_status(self, stacklevel=4):
        """The status word of the last halo set-up -- posted to pinned host memory right behind
        the set-up's kernels (chomp_status_post) -- turned into warnings the first time it is
        looked at: what scipy's AccuracyWarning (divmax exceeded) told the reference's user, plus
        the saturated mass-limit search (include/chomp_mi355x.h, chomp_get_status).  Waits for
        that copy, not for work enqueued after it."""
        if self._status_pending and self._ctx is not None:
            try:
                word = int(self._ctx.warn_status(0, 1, stacklevel=stacklevel, posted=True)[0])
            except _lib.ChompError as exc:
                # (the context's stream is being captured into a HIP graph: the host cannot wait
                #  inside a capture -- the word stays pending and is read after a replay)
                if "captured" not in str(exc):
                    raise
                return self._status_word
            self._status_pending = False
            self._status_word = word
        return self._status_word

    status = property(lambda self: self._resolve_status())

    def _context(self):
        """The device context of this object (created on first use)."""
        if self._ctx is None:
            self._ctx = cosmology._context()
        return self._ctx

    def _sync(self, need_tables, defer_status=False):
        """Bring the device tables named by the CHOMP_T_* mask up to date.  defer_status: leave
        the status word on the device (no synchronisation here); it is read by whoever next
        looks at .status, brings a result to the host, or rebuilds the tables."""
        ctx = self._context()
        bao = bool(getattr(self.cosmo, "_with_bao", False))
        esig = (tuple(sorted(self.cosmo.cosmo_dict.items())), self.cosmo._redshift, bao)
        if esig != self._epoch_sig:
            self._resolve_status(stacklevel=6)     # (the set-up below clears the device's words)
            self._before_epochs_set()
            ctx.epochs_set(self.cosmo.cosmo_dict, [self.cosmo._redshift], bao)
            self._epoch_sig = esig
            self._mass_sig = None
            self._nbar_valid = False
            self._after_epochs_set()
        msig = (tuple(sorted(self.mass.halo_dict.items())), self.mass._kind)
        new_mass = msig != self._mass_sig
        if new_mass:
            self._nbar_valid = False
        build = 0
        for flag, bit in _FLAG_BITS:
            if (need_tables & bit) and not getattr(self, flag):
                build |= bit
        if build or not self._nbar_valid:
            self._resolve_status(stacklevel=6)     # (the next set-up clears the device's words)
            tables = build | (_lib.T_EXCLUSION if self._exclusion else 0)
            if new_mass:       # mass function and halo model in one call: one launch fewer
                # the signature is recorded only once the set-up has returned: after a raise
                # (a status warning turned error above, a HIP / scope error in the set-up)
                # the retry must run the mass function again, not halo_setup on old nu tables
                self._mass_sig = None
                self._stage_k(ctx, tables)
                self._mass_sig = msig
            else:
                ctx.halo_setup(self._profile(), self.local_hod, tables)
            for flag, bit in _FLAG_BITS:
                if build & bit:
                    setattr(self, flag, True)
            self._nbar_valid = True
            ctx.status_post()
            self._status_pending = True
        if not defer_status:
            self._resolve_status(stacklevel=5)
        return ctx

    def _stage_k(self, ctx, tables):
        ctx.stage_k(self.mass.halo_dict, self.mass._kind, self._profile(), self.local_hod, tables)

    def _before_epochs_set(self):
        pass

    def _after_epochs_set(self):
        pass

    def _scalars(self):
        return self._sync(0).scalars(0)

    n_bar = property(lambda self: float(self._scalars()["n_bar"]))
    n_bar_over_rho_bar = property(
        lambda self: float(self._scalars()["n_bar_over_rho_bar"]))
    rho_bar = property(lambda self: float(self._scalars()["rho_bar"]))

    @property
    def delta_v(self):
        dv = self._profile()['delta_v']
        return float(self._scalars()["delta_v"]) if dv == -1 else dv

    _exclusion = False          # HaloExclusion sets it: CHOMP_T_EXCLUSION on every build

    def _power_code(self, which):
        """CHOMP_P_* code of a spectrum of this object (extrapolation flag included)."""
        if self._extrapolate and which != _lib.P_LIN and not (which & _lib.P_HALOFIT):
            which |= _lib.P_EXTRAPOLATE
        return which

    def _power(self, which, need, k):
        ka = numpy.asarray(k, dtype=numpy.float64)
        # the evaluation is queued behind the set-up BEFORE the host looks at the set-up's status
        # words: a host array comes back through a synchronising copy anyway, after which the
        # words are there (no wait of their own, and the evaluation's launch does not wait for
        # the host to wake up in between)
        ctx = self._sync(need, defer_status=True)
        out = ctx.power(self._power_code(which), ka, 0, 1).reshape(ka.shape)
        self._resolve_status(stacklevel=5)
        return out

    # -- reference surface -----------------------------------------------------
    def get_extrapolation(self):
        return self._extrapolate

    def set_extrapolation(self, boolean):
        self._extrapolate = boolean

    def get_cosmology(self):
        return self.cosmo.get_cosmology()

    def get_cosmology_object(self):
        return self.cosmo

    def set_cosmology(self, cosmo_dict, redshift=None):
        """halo.py:135-173."""
        if redshift is None:
            redshift = self._redshift
        self.cosmo_dict = cosmo_dict
        self._redshift = redshift
        self.cosmo = cosmology.SingleEpoch(redshift, cosmo_dict)
        self._h = self.cosmo._h
        self.c0 = self.halo_dict["c0"] / (1.0 + redshift)
        self.mass.set_cosmology_object(self.cosmo)
        # _initialize_halo_splines is re-run here (halo.py:162) with the Halo's OWN
        # dictionary for c0 / delta_v (:151-157) and the current self.beta.
        self._profile_dict = dict(self._profile_dict, c0=self.halo_dict["c0"],
                                  beta=self.beta,
                                  delta_v=self.halo_dict["delta_v"])
        self._nbar_valid = False
        self._reset_flags(all_tables=True)

    def get_hod(self, return_object=False):
        return self.local_hod.get_hod()

    def get_hod_object(self):
        return self.local_hod

    def set_hod(self, hod_dict):
        """halo.py:181-192."""
        self.local_hod.set_hod(hod_dict)
        self._nbar_valid = False
        self._reset_flags(all_tables=False)

    def set_hod_object(self, input_hod):
        """halo.py:194-212."""
        self.local_hod = input_hod
        self._nbar_valid = False
        self._reset_flags(all_tables=False)

    def get_halo(self):
        return self.halo_dict

    def set_halo(self, halo_dict=None):
        """halo.py:220-235: only the mass function sees the new dictionary."""
        self.c0 = halo_dict["c0"] / (1.0 + self._redshift)
        self.beta = halo_dict["beta"]
        self.alpha = -1.0
        self.mass.set_halo(halo_dict)
        self.set_hod_object(self.local_hod)

    def get_mass(self):
        return self.mass

    def get_redshift(self):
        return self._redshift

    def set_redshift(self, redshift):
        if redshift != self._redshift:
            self.set_cosmology(self.cosmo.cosmo_dict, redshift)

    def linear_power(self, k):
        return self._power(_lib.P_LIN, 0, k)

    def power_mm(self, k):
        return self._power(_lib.P_MM, _lib.FAM_MM, k)

    def power_gm(self, k):
        return self._power(_lib.P_GM, _lib.FAM_GM, k)

    def power_mg(self, k):
        return self.power_gm(k)

    def power_gg(self, k):
        return self._power(_lib.P_GG, _lib.FAM_GG, k)

    def virial_radius(self, mass):
        return self._sync(0).eval("virial_radius", mass)

    def concentration(self, mass):
        return self._sync(0).eval("concentration", mass)

    def y(self, ln_k, mass):
        m = numpy.asarray(mass, dtype=numpy.float64)
        out = self._sync(0).y_nfw(0, ln_k, m)
        return out.reshape(numpy.broadcast(numpy.asarray(ln_k), m).shape)

    y_nfw = y

    # knot accessors (the reference's *_spline evaluated inside [k_min, k_max])
    def _knots(self, name, need):
        return self._sync(need).table(name, 0)

    def _h_m(self, k):
        return self._ranged("h_m", _lib.T_H_M, k)

    def _pp_mm(self, k):
        return self._ranged("pp_mm", _lib.T_PP_MM, k)

    def _h_g(self, k):
        return self._ranged("h_g", _lib.T_H_G, k)

    def _pp_gm(self, k):
        return self._ranged("pp_gm", _lib.T_PP_GM, k)

    def _pp_gg(self, k):
        return self._ranged("pp_gg", _lib.T_PP_GG, k)

    def _ranged(self, name, bit, k):
        """halo.py:649-672 -- only exact at the knots here: served from the knot
        table (used by write_power_components-style callers at the knot k's)."""
        ka = numpy.asarray(k, dtype=numpy.float64)
        knots = self._knots(name, bit)
        idx = numpy.rint((numpy.log(ka) - self._ln_k_min) /
                         (self._ln_k_array[1] - self._ln_k_array[0])).astype(int)
        ok = (ka >= self._k_min) & (ka <= self._k_max)
        idx = numpy.clip(idx, 0, knots.size - 1)
        if not numpy.allclose(numpy.log(ka[ok]), self._ln_k_array[idx[ok]], atol=1e-12):
            raise _lib.ChompScopeError(
                "_%s(k) is only served at the spline knots" % name)
        return numpy.where(ok, knots[idx], 0.0)


    # -- HOD summary integrals (halo.py:709-838) -----------------------------------
    def calculate_bias(self):
        self.bias = float(self._sync(0).hod_stats(0, 1)[0, 0])
        return self.bias

    def calculate_m_eff(self):
        self.m_eff = float(self._sync(0).hod_stats(0, 1)[0, 1])
        return self.m_eff

    def calculate_f_sat(self):
        """halo.py:792-838.  The reference hands the redshift to
        satellite_first_moment(mass, z=...) (:836); HODZheng's takes none, so with the shipped
        Zheng HOD the reference raises TypeError -- as does this -- and the integral is only
        reachable with an HOD whose satellite_first_moment accepts z."""
        import inspect
        fn = getattr(self.local_hod, "satellite_first_moment", None)
        if fn is None:
            raise AttributeError("the HOD object has no satellite_first_moment")
        if "z" not in inspect.signature(fn).parameters:
            raise TypeError("satellite_first_moment() got an unexpected keyword argument 'z'")
        self.f_sat = float(self._sync(0).hod_stats(0, 1)[0, 2])
        return self.f_sat

    # -- ASCII writers (halo.py:587-647) -----------------------------------------
    def write(self, output_file_name):
        """halo.py:587-603.  The last knot is exp(ln k_max) = 100.00000000000004 > k_max,
        so -- as in the reference -- its row holds zeros unless extrapolating."""
        k = numpy.exp(self._ln_k_array)
        cols = (k, self.linear_power(k), self.power_mm(k), self.power_gg(k), self.power_gm(k))
        with open(output_file_name, "w") as f:
            f.write("#ttype1 = k [Mpc/h]\n#ttype2 = linear_power [(Mpc/h)^3]\n"
                    "#ttype3 = power_mm\n#ttype4 = power_gg\n"
                    "#ttype5 = power_gm\n")
            for row in zip(*cols):
                f.write("%1.10f %1.10f %1.10f %1.10f %1.10f\n" % row)

    def write_halo(self, output_file_name, k=None):
        """halo.py:605-627 (the halo_normalization column needs scipy's hyp2f1 and is
        unused by the NFW path: written as nan)."""
        if k is None:
            k = 0.01
        ln_k = numpy.log(k)
        mass = self.mass.mass(self.mass._nu_array)
        cols = (mass, self.y(ln_k, mass), self.concentration(mass),
                numpy.full(mass.shape, numpy.nan), self.virial_radius(mass))
        with open(output_file_name, "w") as f:
            f.write("#ttype1 = mass [M_solar/h]\n"
                    "#ttype2 = y(k, M), NFW Fourier Transform\n"
                    "#ttype3 = concentration\n#ttype4 = halo_norm\n"
                    "#ttype4 = halo_normalization"
                    "#ttype5 = virial_radius [M_solar/h]\n")
            for row in zip(*cols):
                f.write("%1.10f %1.10f %1.10f %1.10f %1.10f\n" % row)

    def write_power_components(self, output_file_name):
        """halo.py:629-647."""
        k = numpy.exp(self._ln_k_array)
        cols = (k, self._h_m(k), self._pp_mm(k), self._h_g(k), self._pp_gm(k), self._pp_gg(k))
        with open(output_file_name, "w") as f:
            f.write("#ttype1 = k [Mpc/h]\n#ttype2 = 2 halo dark matter component\n"
                    "#ttype3 = dark matter poisson component\n"
                    "#ttype4 = 2 halo galaxy component\n"
                    "#ttype5 = matter-galaxy poisson component\n"
                    "#ttype6 = galaxy-galaxy poisson component\n")
            for row in zip(*cols):
                f.write("%1.10f %1.10f %1.10f %1.10f %1.10f %1.10f\n" % row)


class HaloExclusion(Halo):
    """halo.py:1201-1233: the 2-halo integrands h_m, h_g carry the transform of a window
    that excludes halo pairs closer than two virial radii."""
    _exclusion = True

    def __init__(self, redshift=0.0, input_hod=None, cosmo_single_epoch=None,
                 mass_func=None, halo_dict=None, **kws):
        Halo.__init__(self, redshift, input_hod, cosmo_single_epoch,
                      mass_func, halo_dict, **kws)


class HaloFit(Halo):
    """HALOFIT with the Takahashi et al. 2012 coefficients (halo.py:1236-1412)."""

    def __init__(self, redshift=0.0, input_hod=None, cosmo_single_epoch=None,
                 mass_func=None, halo_dict=None, **kws):
        Halo.__init__(self, redshift, input_hod, cosmo_single_epoch, mass_func,
                      halo_dict)
        self._initialize_halo_fit()
        self._initialized_sigma_spline = False
        self._hf_coef = None

    def _initialize_halo_fit(self):
        om = self.cosmo.omega_m()
        self._f_1 = numpy.power(om, -0.0307)
        self._f_2 = numpy.power(om, -0.0585)
        self._f_3 = numpy.power(om, 0.0743)
        self._omega_l = self.cosmo.omega_l()
        self._w = self.cosmo.w(self._redshift)

    def _hf_values(self):
        """The HaloFit coefficient block of the device epoch, read back (a synchronisation) the
        first time anything on the host asks for it."""
        if not self._initialized_sigma_spline:
            self._ensure_halofit()
        if self._hf_coef is None:
            self._hf_coef = self._ctx.halofit_get(0)
        return self._hf_coef

    def _before_epochs_set(self):
        # the block about to be overwritten is the one _after_epochs_set re-installs
        if self._initialized_sigma_spline and self._hf_coef is None and self._ctx is not None:
            self._hf_coef = self._ctx.halofit_get(0)

    def _after_epochs_set(self):
        # the device epoch was rebuilt: re-install the (possibly stale, as in the
        # reference) HaloFit coefficient block
        if self._initialized_sigma_spline and self._hf_coef is not None:
            self._ctx.halofit_put(self._hf_coef, 0)

    def _stage_k(self, ctx, tables):
        """The first set-up of a HaloFit object builds its sigma spline as well: one call, the
        HaloFit kernels beside the halo model's knot integrals (chomp_stage_k_halofit)."""
        # (only on the way to a spectrum: the reference builds the spline at its first power_*
        #  call, halo.py:1337-1338 -- a set-up made for n_bar or a table must not fix its redshift)
        if self._initialized_sigma_spline or not getattr(self, "_building_halofit", False):
            return Halo._stage_k(self, ctx, tables)
        ctx.stage_k_halofit(self.mass.halo_dict, self.mass._kind, self._profile(), self.local_hod,
                            tables, 0, float(self._f_1), float(self._f_2), float(self._f_3),
                            float(self._omega_l), float(self._w))
        self._hf_coef = None
        self._initialized_sigma_spline = True

    def _ensure_halofit(self, need=0, defer_status=False):
        """The sigma spline (built once, at whatever redshift the object then has: halo.py:
        1337-1338) and the knot tables of `need`, in one pass over the device where both are
        missing."""
        self._building_halofit = True
        try:
            ctx = self._sync(need, defer_status=defer_status)
        finally:
            self._building_halofit = False
        if not self._initialized_sigma_spline:
            ctx.halofit_setup(0, 0, float(self._f_1), float(self._f_2),
                              float(self._f_3), float(self._omega_l), float(self._w))
            self._hf_coef = None
            self._initialized_sigma_spline = True
        return ctx


    def power_mm(self, k):
        self._ensure_halofit()
        return self._power(_lib.P_MM | _lib.P_HALOFIT, 0, k)

    def power_gm(self, k):
        self._ensure_halofit(_lib.FAM_GM)
        return self._power(_lib.P_GM | _lib.P_HALOFIT, _lib.FAM_GM, k)

    def power_gg(self, k):
        self._ensure_halofit(_lib.FAM_GG)
        return self._power(_lib.P_GG | _lib.P_HALOFIT, _lib.FAM_GG, k)


# k_s, n_eff, C and the Takahashi et al. coefficients (halo.py:1289-1330) as the reference's
# attribute names
for _i, _name in enumerate(("_k_s", "_n_eff", "_C", "_a_n", "_b_n", "_c_n", "_gamma_n",
                            "_alpha_n", "_beta_n", "_mu_n", "_nu_n")):
    setattr(HaloFit, _name, property(lambda self, _i=_i: float(self._hf_values()[3 + _i])))
del _i, _name
