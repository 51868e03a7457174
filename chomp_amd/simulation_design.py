"""simulation_design.SimulationDesign with the reference's interface
(simulation_design.py:36-234): Latin-hypercube design points in parameter space, one model
evaluation per point, results in a pandas DataFrame.

The reference evaluates the points one after another through the object's
set_cosmology / set_halo / set_hod and the requested method.  That generic loop is kept
(it works with every chomp_amd object).  When the object is a Halo and the method one of
its spectra, the design is the native batch axis of the device library: every design
point becomes one epoch of a HaloGrid and all of them are built and evaluated by the same
kernel launches.
"""
import copy
import warnings

import numpy
import pandas

from . import _lib
from . import defaults
from . import grid
from . import halo as halo_mod

default_parameter_dict = {"cosmo_dict": defaults.default_cosmo_dict,
                          "halo_dict": defaults.default_halo_dict,
                          "hod_dict": defaults.default_hod_dict}

_BATCHED_METHODS = ("linear_power", "power_mm", "power_gm", "power_mg", "power_gg")


def _hod_dict_of(h):
    """The parameter dictionary of an HODZheng; one built without a dictionary keeps None
    (hod.py:157-165) and answers from its attributes."""
    d = h.get_hod()
    if d is None:
        d = {key: getattr(h, key) for key in ("log_M_min", "sigma", "log_M_0", "log_M_1p", "alpha")}
    return dict(d)


def random_lhs(n, k):
    """n points of a random Latin hypercube in k variables (what simulation_design.py:17-33 takes
    from the R package 'lhs'): every variable's range is cut into n strata, each stratum is used
    by exactly one point, and a point sits uniformly inside its stratum.  Draws from numpy's
    global generator in the reference's order (the k orderings first, then the positions), so a
    seeded design is the reference's design."""
    strata = numpy.empty((n, k), dtype=numpy.float64)
    for col in range(k):
        strata[:, col] = numpy.random.permutation(n)
    inside = numpy.random.uniform(size=(n, k))
    return (strata + inside) / float(n)


class _Recorder(object):
    """Stands in for the model object while the design's setters run: keeps the
    dictionaries they hand over instead of rebuilding a model per point."""

    def __init__(self):
        self.cosmo = self.halo = self.hod = None

    def set_cosmology(self, cosmo_dict, *args, **kws):
        self.cosmo = dict(cosmo_dict)

    def set_halo(self, halo_dict, *args, **kws):
        self.halo = dict(halo_dict)

    def set_hod(self, hod_dict, *args, **kws):
        self.hod = dict(hod_dict)


class SimulationDesign(object):
    """simulation_design.py:36-234.  params: {name: [center, min, max]} with names from
    the cosmology, halo or HOD dictionaries of defaults.py."""

    def __init__(self, input_chomp_object, method_name, params, n_design=100,
                 independent_var=None, default_param_dict=None):
        self._input_object = input_chomp_object
        self._method = method_name
        self.params = pandas.DataFrame(params, index=['center', 'min', 'max'])
        self.n_design = n_design
        self._ind_var = independent_var
        self._initialized_design = False
        if default_param_dict is None:
            default_param_dict = default_parameter_dict
        # (the reference mutates the caller's dictionaries point after point; a private
        # copy keeps the defaults of this process intact)
        self._default_param_dict = copy.deepcopy(default_param_dict)
        self._vary_cosmology = False
        self._vary_halo = False
        self._vary_hod = False
        self._param_types = []
        for key in self.params.keys():
            for kind, flag in (("cosmo_dict", "_vary_cosmology"), ("halo_dict", "_vary_halo"),
                               ("hod_dict", "_vary_hod")):
                if key in self._default_param_dict[kind]:
                    self._param_types.append(kind)
                    setattr(self, flag, True)
                    break

    def _init_design_points(self):
        """simulation_design.py:101-114."""
        diff = (self.params.xs('max') - self.params.xs('min')).rename('diff')
        self.params = pandas.concat([self.params, diff.to_frame().transpose()])
        points = pandas.DataFrame(random_lhs(self.n_design, self.params.shape[1]),
                                  columns=self.params.columns)
        self.lhs = points
        self.points = points * self.params.xs('diff') + self.params.xs('min')
        self._initialized_design = True

    def _apply_point(self, point):
        if self._vary_cosmology:
            self.set_cosmology(self._default_param_dict['cosmo_dict'], point)
        if self._vary_halo:
            self.set_halo(self._default_param_dict['halo_dict'], point)
        if self._vary_hod:
            self.set_hod(self._default_param_dict['hod_dict'], point)

    def _run_des_point(self, point):
        """simulation_design.py:116-138: one design point through the object's setters."""
        self._apply_point(point)
        if self._ind_var is None:
            values = getattr(self._input_object, self._method)()
        else:
            values = getattr(self._input_object, self._method)(self._ind_var)
        return pandas.Series(numpy.asarray(values).flatten())

    def _batched(self):
        obj = self._input_object
        # Halo.set_halo only reaches the mass function and leaks into later points through
        # set_cosmology (halo.py:151-162, 220-235): designs over halo parameters keep the
        # point-by-point loop, which reproduces that.
        return (type(obj) is halo_mod.Halo and self._method in _BATCHED_METHODS and
                self._ind_var is not None and not self._vary_halo and
                not obj.get_extrapolation())

    def _run_batched(self):
        """Every design point = one epoch of one HaloGrid."""
        real, rec = self._input_object, _Recorder()
        obj = real
        cosmos, halos, hods = [], [], []
        self._input_object = rec
        try:
            for _, point in self.points.iterrows():
                rec.cosmo = rec.halo = rec.hod = None
                self._apply_point(point)
                cosmos.append(rec.cosmo if rec.cosmo is not None else dict(obj.cosmo.cosmo_dict))
                halos.append(dict(obj.mass.halo_dict))
                hods.append(rec.hod if rec.hod is not None else _hod_dict_of(obj.local_hod))
        finally:
            self._input_object = real
        kind = "tinker" if getattr(obj.mass, "_kind", 0) else "st"
        hg = grid.HaloGrid(numpy.full(len(cosmos), obj.get_redshift()), cosmo_dict=cosmos,
                           halo_dict=halos, hod_dict=hods, mass_function=kind)
        k = numpy.asarray(self._ind_var, dtype=numpy.float64)
        out = hg.power(self._method, k.ravel())
        # The result is on the host: read the status words now and say which design points they
        # concern.  What the point-by-point loop reports through Halo._sync (an exhausted divmax =
        # scipy's AccuracyWarning; a saturated mass-limit search, where the reference's own answer
        # is decided by rounding noise and may differ by percents) must not get lost in a batch.
        words = hg.status()
        self.design_status = pandas.Series([int(w) for w in words], index=self.points.index,
                                           name="status", dtype="int64")
        for label, w in self.design_status.items():
            if w:
                category = (_lib.ChompParityWarning
                            if w & (_lib.ST_SATURATED | _lib.ST_MASS_SEARCH_EXHAUSTED)
                            else _lib.ChompAccuracyWarning)
                warnings.warn("design point %s: %s" % (label, "; ".join(_lib.describe_status(w))),
                              category, stacklevel=3)
        return pandas.DataFrame(numpy.asarray(out).T, columns=self.points.index)

    def run_design(self, batched=None, with_status=False):
        """simulation_design.py:140-155.  Returns a DataFrame with one column per design
        point holding the flattened output of the method.  batched=None picks the
        one-launch path when the object / method allow it.

        The batched path keeps the status word of every design point (chomp_get_status bits) in
        `self.design_status` (an int64 Series indexed like the design points) and raises them as
        warnings naming the point; with_status=True returns the pair (frame, design_status) --
        the words are never mixed into the float frame, whose rows stay the method's output."""
        if not self._initialized_design:
            self._init_design_points()
        if batched is None:
            batched = self._batched()
        self.design_status = None
        if batched:
            self.design_values = self._run_batched()
        else:
            self.design_values = self.points.transpose().apply(self._run_des_point)
        self.values_frame = self.design_values
        if with_status:
            return self.design_values, self.design_status
        return self.design_values

    def set_cosmology(self, cosmo_dict=None, values=None):
        """simulation_design.py:157-175."""
        if cosmo_dict is None:
            cosmo_dict = self._default_param_dict['cosmo_dict']
        for key in self.params.keys():
            if key in cosmo_dict and key in values:
                cosmo_dict[key] = values[key]
        self._input_object.set_cosmology(cosmo_dict)

    def set_halo(self, halo_dict=None, values=None):
        """simulation_design.py:177-193."""
        if halo_dict is None:
            halo_dict = self._default_param_dict['halo_dict']
        for key in self.params.keys():
            if key in halo_dict and key in values:
                halo_dict[key] = values[key]
        self._input_object.set_halo(halo_dict)

    def set_hod(self, hod_dict=None, values=None):
        """simulation_design.py:195-211."""
        if hod_dict is None:
            hod_dict = self._default_param_dict['hod_dict']
        for key in self.params.keys():
            if key in hod_dict and key in values:
                hod_dict[key] = values[key]
        self._input_object.set_hod(hod_dict)

    def write(self, output_name):
        """simulation_design.py:213-220."""
        self.values_frame.to_csv(output_name, index=False, sep=',')


class SimulationDesignFlatUniverse(SimulationDesign):
    """simulation_design.py:223-241: omega_l0 = 1 - omega_m0 - omega_r0 at every point."""

    def __init__(self, input_chomp_object, method_name, params, n_design=100,
                 independent_var=None, default_param_dict=None):
        SimulationDesign.__init__(self, input_chomp_object, method_name, params,
                                  n_design, independent_var, default_param_dict)
        self.set_cosmology(self._default_param_dict['cosmo_dict'], self.params.xs('center'))

    def set_cosmology(self, cosmo_dict=None, values=None):
        if cosmo_dict is None:
            cosmo_dict = self._default_param_dict['cosmo_dict']
        for key in self.params.keys():
            if key in cosmo_dict and key in values:
                cosmo_dict[key] = values[key]
        cosmo_dict['omega_l0'] = 1.0 - cosmo_dict['omega_m0'] - cosmo_dict['omega_r0']
        self._input_object.set_cosmology(cosmo_dict)


class SimulationDesignHubbleNormalizedDensities(SimulationDesign):
    """simulation_design.py:244-267: parameters omega_mh2 = Omega_m h^2, omega_bh2."""

    def __init__(self, input_chomp_object, method_name, params, n_design=100,
                 independent_var=None, default_param_dict=None):
        SimulationDesign.__init__(self, input_chomp_object, method_name, params,
                                  n_design, independent_var, default_param_dict)
        self._vary_cosmology = True
        self.set_cosmology(self._default_param_dict['cosmo_dict'], self.params.xs('center'))

    def set_cosmology(self, cosmo_dict=None, values=None):
        if cosmo_dict is None:
            cosmo_dict = self._default_param_dict['cosmo_dict']
        for key in self.params.keys():
            if key in cosmo_dict and key in values:
                cosmo_dict[key] = values[key]
        cosmo_dict['omega_m0'] = values['omega_mh2'] / cosmo_dict['h'] ** 2
        cosmo_dict['omega_b0'] = values['omega_bh2'] / cosmo_dict['h'] ** 2
        cosmo_dict['omega_l0'] = 1.0 - cosmo_dict['omega_m0'] - cosmo_dict['omega_r0']
        self._input_object.set_cosmology(cosmo_dict)
