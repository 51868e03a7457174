"""ctypes binding of libchomp_mi355x.so (include/chomp_mi355x.h).

The HIP library is the only compute path of this package: if it cannot be loaded
(or built in-tree with hipcc) importing the binding raises -- there is no CPU
fallback.
"""
import ctypes
import os
import subprocess
import threading

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libchomp_mi355x.so")

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_SCOPE = 0, -1, -2, -3, -4
HOST, DEVICE = 0, 1
MF_ST, MF_TINKER = 0, 1
T_H_M, T_PP_MM, T_H_G, T_PP_GM, T_PP_GG = 1, 2, 4, 8, 16
T_EXCLUSION = 64
FAM_MM, FAM_GM, FAM_GG = T_H_M | T_PP_MM, T_H_M | T_H_G | T_PP_GM, T_H_G | T_PP_GG
P_LIN, P_MM, P_GM, P_GG, P_HALOFIT, P_EXTRAPOLATE = 0, 1, 2, 3, 16, 32
PREC_F64, PREC_F32_EVAL, PREC_F32_TABLES, PREC_F32_ALL = 0, 1, 2, 3
DNDZ_MAGLIM, DNDZ_GAUSSIAN, DNDZ_BOXCAR, DNDZ_PPOLY = 0, 1, 2, 3
WINDOW_GALAXY, WINDOW_CONVERGENCE, WINDOW_FLAT_CONVERGENCE, WINDOW_CONVERGENCE_DELTA = 0, 1, 2, 3

SC = {name: i for i, name in enumerate([
    "z", "chi", "growth", "omega_m", "omega_l", "delta_c", "delta_v", "rho_bar",
    "sigma_norm", "ln_mass_min", "ln_mass_max", "nu_min", "nu_max", "m_star",
    "f_norm", "bias_norm", "n_bar", "n_bar_over_rho_bar", "n_search",
    "mf_delta_v", "t_alpha", "t_beta", "t_gamma", "t_phi", "t_eta",
    "growth_norm", "delta_H", "hf_k_s", "hf_n_eff", "hf_C"])}
SC_COUNT = 30
TAB = {"ln_mass": 0, "nu": 1, "h_m": 2, "pp_mm": 3, "h_g": 4, "pp_gm": 5,
       "pp_gg": 6, "levels": 7, "hf_ln_sigma2": 8}
EV = {"nu_of_mass": 0, "ln_mass_of_nu": 1, "f_nu": 2, "bias_nu": 3,
      "hod_first": 4, "hod_second": 5, "hod_central": 6, "hod_satellite": 7,
      "virial_radius": 8, "concentration": 9, "delta_k": 10}
HF_COUNT = 14
KI = {name: i for i, name in enumerate([
    "z_bar", "chi_min", "chi_max", "z_min", "z_max", "D_zbar", "norm_a", "norm_b",
    "wa_chi_min", "wa_chi_max", "wb_chi_min", "wb_chi_max", "j_limit"])}
KI_COUNT = 13
ME = {"chi_of_z": 0, "z_of_chi": 1, "growth_of_z": 2}
KTAB = {"ln_ktheta": 0, "kernel": 1, "wa_chi": 2, "wa": 3, "wb_chi": 4, "wb": 5,
        "me_z": 6, "me_chi": 7, "me_growth": 8, "levels": 9}

c_double_p = ctypes.POINTER(ctypes.c_double)


class Cosmo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in (
        "omega_m0", "omega_b0", "omega_l0", "omega_r0", "cmb_temp", "h",
        "sigma_8", "n_scalar", "w0", "wa")]


class HaloPar(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in (
        "stq", "st_little_a", "c0", "beta", "alpha", "delta_v")]


class HodPar(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in (
        "log_M_min", "sigma", "log_M_0", "log_M_1p", "alpha")]


class Config(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_double) for n in (
        "k_min", "k_max", "mass_min", "mass_max", "corr_precision",
        "cosmo_precision", "dNdz_precision", "halo_precision",
        "kernel_precision", "mass_precision", "window_precision",
        "global_precision")] +
        [(n, ctypes.c_int) for n in (
            "corr_npoints", "cosmo_npoints", "halo_npoints", "kernel_npoints",
            "kernel_bessel_limit", "mass_npoints", "window_npoints", "divmax")])


class Dndz(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("pad_", ctypes.c_int),
                ("z_min", ctypes.c_double), ("z_max", ctypes.c_double),
                ("p", ctypes.c_double * 4),
                ("pp_breaks", ctypes.POINTER(ctypes.c_double)),
                ("pp_coef", ctypes.POINTER(ctypes.c_double)),
                ("pp_n", ctypes.c_int), ("pp_order", ctypes.c_int)]


class Window(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("pad_", ctypes.c_int), ("dist", Dndz)]


EXPORTS = [
    "chomp_default_config", "chomp_ctx_create", "chomp_ctx_destroy",
    "chomp_last_error", "chomp_sync", "chomp_epochs_set", "chomp_mass_setup",
    "chomp_halo_setup", "chomp_halofit_setup", "chomp_power", "chomp_power_range",
    "chomp_sigma_r", "chomp_y_nfw", "chomp_get_scalars", "chomp_get_table",
    "chomp_eval", "chomp_halofit_get", "chomp_halofit_put",
    "chomp_multi_epoch_setup", "chomp_me_eval",
    "chomp_kernel_setup", "chomp_kernel_info", "chomp_kernel_table",
    "chomp_kernel_eval", "chomp_window_eval", "chomp_wtheta", "chomp_cell",
    "chomp_set_precision", "chomp_xi3d", "chomp_spline_eval", "chomp_hod_stats",
    "chomp_set_transfer", "chomp_kernel_raw",
    "chomp_covariance_table", "chomp_covariance_gaussian",
    "chomp_set_timing", "chomp_get_timing", "chomp_get_status", "chomp_status_post",
    "chomp_status_wait", "chomp_set_tuning",
    "chomp_get_deep_stats", "chomp_stage_k", "chomp_power_plan", "chomp_get_stream",
    "chomp_wtheta_cell", "chomp_stage_k_halofit",
]

# chomp_get_status bits (include/chomp_mi355x.h)
ST_MASS_MIN_SATURATED, ST_MASS_MAX_SATURATED, ST_MASS_SEARCH_EXHAUSTED, ST_SIGMA_DIVMAX = 1, 2, 4, 8
ST_HALO_DIVMAX = {"h_m": 0x100, "pp_mm": 0x200, "h_g": 0x400, "pp_gm": 0x800, "pp_gg": 0x1000}
ST_NONFINITE = 0x10000
ST_SATURATED = ST_MASS_MIN_SATURATED | ST_MASS_MAX_SATURATED
TUNE_E_STREAM_MIN, TUNE_E_ROWS, TUNE_DEEP_LITERAL, TUNE_ROCTX, TUNE_WTHETA_DIRECT = 0, 1, 2, 3, 4
TUNE_CELL_ONE_KERNEL = 5
TUNE_DEEP_TOL, TUNE_DEEP_MAX_BREAKS, TUNE_DEEP_MAX_FINE, TUNE_HOD_CAP, TUNE_DEEP_SLOTS = 6, 7, 8, 9, 10


class ChompAccuracyWarning(UserWarning):
    """What scipy.integrate.romberg's AccuracyWarning (divmax exceeded) was in the reference."""


class ChompParityWarning(UserWarning):
    """The result is well defined but the reference's own answer for this input is decided by
    rounding noise (saturated mass-limit search): the two may differ by percents."""


def describe_status(word):
    """Human-readable list of the bits of a chomp_get_status word."""
    out = []
    if word & ST_MASS_MIN_SATURATED:
        out.append("mass_min search ended in the saturated regime of sigma_r (k R < 0.2 over "
                   "its whole k range, cosmology.py:627-632): the reference's own limit is "
                   "decided by rounding error there")
    if word & ST_MASS_MAX_SATURATED:
        out.append("mass_max search ended in the saturated regime of sigma_r "
                   "(k range clamped at k_min / 100, cosmology.py:617-622)")
    if word & ST_MASS_SEARCH_EXHAUSTED:
        out.append("mass-limit search did not end within 2047 steps of 5 %")
    if word & ST_SIGMA_DIVMAX:
        out.append("a sigma(R) Romberg of the nu table exhausted divmax")
    for name, bit in ST_HALO_DIVMAX.items():
        if word & bit:
            out.append("%s: Romberg exhausted divmax at some knots (last row kept)" % name)
    if word & ST_NONFINITE:
        out.append("a knot table holds NaN / infinity")
    return out


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))
            if f.endswith((".hip", ".h", ".inc"))] + [
        os.path.join(HERE, "..", "include", "chomp_mi355x.h")]


HASH_PATH = LIB_PATH + ".srchash"


# Compiler flags of the device code.  -disable-machine-licm: LLVM's machine-level loop-invariant
# code motion hoists the materialisation of every double constant of the inlined libm / special-
# function polynomials (a v_mov pair each) out of the kernels' outer loops, where they stay live
# across everything and -- the scalar registers being exhausted -- end up in VGPRs or even
# spilled to scratch and re-loaded INSIDE the dependent FMA chains (k_halo_knots_fast: 256 VGPRs +
# 652 B of scratch per lane).  Without it: k_halo_knots 231 -> 168 VGPRs, k_epoch_probe 215 -> 152,
# k_nu_table 128 -> 98, k_power_grid 166 -> 106, no spills left in k_cell / k_cell_deep / k_wtheta
# (tools/kernel_regs.py); same arithmetic, bit-identical results.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
NO_LICM = ["-mllvm", "-disable-machine-licm"]
# translation units: (source, extra flags).  k_epoch_probe is the one kernel that is faster
# with machine LICM (38.5 against 41.8 us per C2 launch) and has a unit of its own.
UNITS = [("chomp_capi.hip", NO_LICM), ("chomp_probe.hip", [])]


def source_hash():
    """Content hash of everything the library is compiled from (mtimes do not survive a
    copy of the tree to another box; contents do), compiler flags included."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((HIPCC_FLAGS, UNITS)).encode())
    for path in sources():
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def built_hash():
    """The source hash the shipped library was built from (None if not recorded)."""
    try:
        with open(HASH_PATH) as f:
            return f.read().strip() or None
    except OSError:
        return None


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a
    GPU).  Rebuilds when the sources differ from the ones the .so was built from (their
    hash is kept beside it).  extra_flags / out: a development build somewhere else (e.g.
    -DCHOMP_STAMPS into build_exp/), leaving the product library alone."""
    want = source_hash()
    if out is None and not force and os.path.exists(LIB_PATH):
        try:
            with open(HASH_PATH) as f:
                if f.read().strip() == want:
                    return LIB_PATH
        except OSError:
            pass
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = LIB_PATH + ".tmp%d" % os.getpid()
    import tempfile
    with tempfile.TemporaryDirectory(prefix="chomp_build_") as objdir:
        procs = []
        for src, extra in UNITS:                      # (the units compile side by side)
            obj = os.path.join(objdir, src.replace(".hip", ".o"))
            cmd = [hipcc] + HIPCC_FLAGS + list(extra) + list(extra_flags) + ["-c", "-o", obj,
                                                                             os.path.join(CSRC, src)]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, obj, subprocess.Popen(cmd, cwd=CSRC)))
        for cmd, obj, pr in procs:
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [o for _, o, _ in procs]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link, cwd=CSRC)
    if out is not None:
        os.replace(tmp, out)
        return out
    os.replace(tmp, LIB_PATH)
    with open(HASH_PATH, "w") as f:
        f.write(want + "\n")
    return LIB_PATH


_lib = None
_lock = threading.Lock()


def _preload_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm wheels bundle their own
    libamdhip64.so (same SONAME as /opt/rocm's).  If our library were loaded first it
    would bind the system runtime and a later `import torch` would bring in a second
    one, which then fails to open the GPU.  Loading torch's copy first (without
    importing torch) makes both resolve to the same runtime, in either import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    """Load (building if necessary) the HIP library.  Raises if impossible."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # build() is a no-op when the library was built from exactly these sources (content
        # hash beside it).  A library that was built from OTHER sources is never loaded
        # silently: a test run would certify a binary that is not the tree.
        try:
            build()
        except Exception as exc:   # noqa: BLE001
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    "chomp_amd: libchomp_mi355x.so is missing and could not be "
                    "built with hipcc (%s). This package has no CPU fallback."
                    % exc) from exc
            have = built_hash()
            msg = ("chomp_amd: libchomp_mi355x.so was built from other sources (library %s, "
                   "tree %s) and the rebuild failed (%s)"
                   % (have or "unknown", source_hash(), exc))
            if os.environ.get("CHOMP_ALLOW_STALE_LIB") != "1":
                raise ImportError(msg + "; set CHOMP_ALLOW_STALE_LIB=1 to load it all the "
                                  "same (a box without hipcc)") from exc
            import warnings
            warnings.warn(msg + "; CHOMP_ALLOW_STALE_LIB=1: using the existing library")
        _preload_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name in EXPORTS:
            if not hasattr(L, name):
                raise ImportError("chomp_amd: %s lacks symbol %s" % (LIB_PATH, name))
        vp, sz, i, d = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double
        L.chomp_default_config.argtypes = [ctypes.POINTER(Config)]
        L.chomp_default_config.restype = None
        L.chomp_ctx_create.argtypes = [ctypes.POINTER(Config), i, vp,
                                       ctypes.POINTER(vp)]
        L.chomp_ctx_destroy.argtypes = [vp]
        L.chomp_ctx_destroy.restype = None
        L.chomp_last_error.argtypes = [vp]
        L.chomp_last_error.restype = ctypes.c_char_p
        L.chomp_sync.argtypes = [vp]
        L.chomp_get_stream.argtypes = [vp, ctypes.POINTER(vp)]
        L.chomp_epochs_set.argtypes = [vp, sz, ctypes.POINTER(Cosmo), c_double_p]
        L.chomp_mass_setup.argtypes = [vp, ctypes.POINTER(HaloPar), i]
        L.chomp_halo_setup.argtypes = [vp, ctypes.POINTER(HaloPar),
                                       ctypes.POINTER(HodPar), ctypes.c_uint]
        L.chomp_stage_k.argtypes = [vp, ctypes.POINTER(HaloPar), i, ctypes.POINTER(HaloPar),
                                    ctypes.POINTER(HodPar), ctypes.c_uint]
        L.chomp_halofit_setup.argtypes = [vp, sz, sz, d, d, d, d, d]
        L.chomp_stage_k_halofit.argtypes = [vp, ctypes.POINTER(HaloPar), i, ctypes.POINTER(HaloPar),
                                            ctypes.POINTER(HodPar), ctypes.c_uint, sz, d, d, d, d, d]
        L.chomp_power.argtypes = [vp, i, vp, sz, vp, i]
        L.chomp_power_range.argtypes = [vp, i, sz, sz, vp, sz, vp, i]
        L.chomp_power_plan.argtypes = [vp, sz, vp, sz]
        L.chomp_sigma_r.argtypes = [vp, sz, c_double_p, sz, c_double_p]
        L.chomp_y_nfw.argtypes = [vp, sz, c_double_p, c_double_p, sz, c_double_p]
        L.chomp_get_scalars.argtypes = [vp, sz, c_double_p]
        L.chomp_get_table.argtypes = [vp, sz, i, c_double_p, sz]
        L.chomp_eval.argtypes = [vp, sz, i, vp, sz, vp, i]
        L.chomp_halofit_get.argtypes = [vp, sz, c_double_p]
        L.chomp_halofit_put.argtypes = [vp, sz, c_double_p]
        L.chomp_kernel_setup.argtypes = [vp, ctypes.POINTER(Cosmo), d, d, d, d,
                                         ctypes.POINTER(Window),
                                         ctypes.POINTER(Window), i]
        L.chomp_multi_epoch_setup.argtypes = [vp, ctypes.POINTER(Cosmo), d, d]
        L.chomp_me_eval.argtypes = [vp, i, vp, sz, vp, i]
        L.chomp_kernel_info.argtypes = [vp, c_double_p]
        L.chomp_kernel_table.argtypes = [vp, i, c_double_p, sz]
        L.chomp_kernel_eval.argtypes = [vp, vp, sz, vp, i]
        L.chomp_kernel_raw.argtypes = [vp, vp, sz, vp, i]
        L.chomp_window_eval.argtypes = [vp, i, vp, sz, vp, i]
        L.chomp_wtheta.argtypes = [vp, i, sz, d, d, d, vp, sz, vp, i]
        L.chomp_cell.argtypes = [vp, i, sz, d, vp, sz, vp, i]
        L.chomp_wtheta_cell.argtypes = [vp, i, sz, d, d, d, vp, sz, vp, vp, sz, vp, i]
        L.chomp_set_precision.argtypes = [vp, i]
        L.chomp_hod_stats.argtypes = [vp, sz, sz, c_double_p]
        L.chomp_set_transfer.argtypes = [vp, i]
        L.chomp_set_timing.argtypes = [vp, i]
        L.chomp_get_timing.argtypes = [vp, c_double_p, sz]
        L.chomp_get_status.argtypes = [vp, sz, sz, ctypes.POINTER(ctypes.c_uint)]
        L.chomp_status_post.argtypes = [vp]
        L.chomp_status_wait.argtypes = [vp, sz, sz, ctypes.POINTER(ctypes.c_uint)]
        L.chomp_set_tuning.argtypes = [vp, i, ctypes.c_longlong]
        L.chomp_get_deep_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
        L.chomp_covariance_table.argtypes = [vp, i, sz, d, c_double_p, c_double_p,
                                             c_double_p, sz]
        L.chomp_covariance_gaussian.argtypes = [vp, d, d, d, d, vp, sz, vp, i]
        L.chomp_xi3d.argtypes = [vp, i, sz, d, d, vp, sz, vp, i]
        L.chomp_spline_eval.argtypes = [vp, c_double_p, c_double_p, sz, c_double_p, sz, i,
                                        c_double_p]
        for name in EXPORTS:
            if name not in ("chomp_default_config", "chomp_ctx_destroy",
                            "chomp_last_error"):
                getattr(L, name).restype = i
        _lib = L
        return _lib


def current_device():
    """HIP ordinal for new contexts: CHOMP_DEVICE, else LOCAL_RANK (one process
    per GPU under torch.distributed.run), else 0."""
    for key in ("CHOMP_DEVICE", "LOCAL_RANK"):
        if key in os.environ:
            return int(os.environ[key])
    return 0


class ChompError(RuntimeError):
    pass


class ChompScopeError(NotImplementedError):
    """Feature of the reference that is outside the accelerated hot path."""


def make_config(limits, precision):
    """Snapshot defaults.default_limits / default_precision into a Config."""
    c = Config()
    for k in ("k_min", "k_max", "mass_min", "mass_max"):
        setattr(c, k, float(limits[k]))
    for k in ("corr_precision", "cosmo_precision", "dNdz_precision",
              "halo_precision", "kernel_precision", "mass_precision",
              "window_precision", "global_precision"):
        setattr(c, k, float(precision[k]))
    for k in ("corr_npoints", "cosmo_npoints", "halo_npoints", "kernel_npoints",
              "kernel_bessel_limit", "mass_npoints", "window_npoints", "divmax"):
        setattr(c, k, int(precision[k]))
    return c


def cosmo_struct(cosmo_dict):
    """KeyError on a missing key, like the reference (cosmology.py:49-58)."""
    return Cosmo(*[float(cosmo_dict[k]) for k in (
        "omega_m0", "omega_b0", "omega_l0", "omega_r0", "cmb_temp", "h",
        "sigma_8", "n_scalar", "w0", "wa")])


def halo_struct(halo_dict):
    return HaloPar(*[float(halo_dict[k]) for k in (
        "stq", "st_little_a", "c0", "beta", "alpha", "delta_v")])


def hod_struct(hod):
    return HodPar(float(hod.log_M_min), float(hod.sigma), float(hod.log_M_0),
                  float(hod.log_M_1p), float(hod.alpha))


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch"


def _torch_current_stream(device):
    """torch's current HIP stream on `device`, if torch is in use in this process (never
    imports torch or initialises the GPU through it by itself)."""
    import sys
    torch = sys.modules.get("torch")
    if torch is None or not hasattr(torch, "cuda") or not torch.cuda.is_initialized():
        return None
    return torch.cuda.current_stream(device)


class Context(object):
    """Owns one chomp_ctx.  device: HIP ordinal; stream: raw hipStream_t or None.

    stream=None: torch's current stream of the device when torch is already driving the GPU
    in this process (so that tensors handed to power() / wtheta() / ... are produced and
    consumed in stream order with the caller's other work), else a stream of the library's
    own.  Calls that pass torch tensors from ANOTHER current stream are ordered against it
    with events on both sides (_torch_enter / _torch_leave)."""

    def __init__(self, config, device=0, stream=None):
        self._L = lib()
        self._h = ctypes.c_void_p()
        self.config = config
        self.device = int(device)
        if stream is None:
            cur = _torch_current_stream(self.device)
            if cur is not None:
                stream = cur.cuda_stream
        rc = self._L.chomp_ctx_create(ctypes.byref(config), self.device,
                                      ctypes.c_void_p(stream or 0),
                                      ctypes.byref(self._h))
        if rc != OK:
            raise ChompError("chomp_ctx_create failed (%d): no usable MI355X / HIP "
                             "device; chomp_amd has no CPU fallback" % rc)
        sp = ctypes.c_void_p()
        self._L.chomp_get_stream(self._h, ctypes.byref(sp))
        self.stream_ptr = sp.value or 0
        self.n_epoch = 0
        self._plan_k = None

    # -- ordering against the caller's torch stream ------------------------------------
    def _torch_enter(self):
        """Before a call that reads torch tensors: the context's stream waits for what the
        caller's current stream has queued.  Returns the pair of streams for _torch_leave
        (None when both are the same stream: nothing to do)."""
        import torch
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == self.stream_ptr:
            return None
        mine = torch.cuda.ExternalStream(self.stream_ptr, device=self.device)
        mine.wait_stream(cur)
        return cur, mine

    @staticmethod
    def _torch_leave(pair):
        """After it: the caller's stream waits for the context's (the outputs)."""
        if pair is not None:
            pair[0].wait_stream(pair[1])

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.chomp_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()
            self._plan_k = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass

    def _check(self, rc):
        if rc == OK:
            return
        msg = self._L.chomp_last_error(self._h).decode()
        if rc == ERR_SCOPE:
            raise ChompScopeError(msg)
        if rc == ERR_ARG:
            raise ValueError(msg)
        raise ChompError("%s (code %d)" % (msg, rc))

    # -- Stage K -------------------------------------------------------------
    # Each call takes either Python dictionaries / objects (converted here) or the
    # ctypes arrays built once by pack_* (the batched path does that: converting 64
    # dictionaries per step costs more host time than the kernels take).
    @staticmethod
    def pack_cosmo(cosmo_dicts, n):
        """One dict (for all n epochs), a list of n dicts, or a float64 array [n, 10] in the
        field order of chomp_cosmo (omega_m0, omega_b0, omega_l0, omega_r0, cmb_temp, h,
        sigma_8, n_scalar, w0, wa) -- what a sampler that draws thousands of points per step
        hands over without building dictionaries."""
        if isinstance(cosmo_dicts, numpy.ndarray):
            a = numpy.ascontiguousarray(cosmo_dicts, dtype=numpy.float64)
            if a.shape != (n, 10):
                raise ValueError("cosmology array must be [%d, 10], got %r" % (n, a.shape))
            return (Cosmo * n).from_buffer_copy(a.tobytes())
        if isinstance(cosmo_dicts, dict):
            cosmo_dicts = [cosmo_dicts] * n
        return (Cosmo * n)(*[cosmo_struct(c) for c in cosmo_dicts])

    @staticmethod
    def pack_halo(halo_dicts, n):
        if isinstance(halo_dicts, dict):
            halo_dicts = [halo_dicts] * n
        return (HaloPar * n)(*[halo_struct(h) for h in halo_dicts])

    @staticmethod
    def pack_hod(hods, n):
        if not isinstance(hods, (list, tuple)):
            hods = [hods] * n
        return (HodPar * n)(*[hod_struct(h) for h in hods])

    def epochs_set(self, cosmo, z, with_bao=False):
        """with_bao: SingleEpoch(with_bao=True), the E&H transfer function with wiggles."""
        self._check(self._L.chomp_set_transfer(self._h, 1 if with_bao else 0))
        z = numpy.ascontiguousarray(numpy.atleast_1d(z), dtype=numpy.float64)
        n = z.size
        arr = cosmo if isinstance(cosmo, ctypes.Array) else self.pack_cosmo(cosmo, n)
        assert len(arr) == n
        self._check(self._L.chomp_epochs_set(self._h, n, arr,
                                             z.ctypes.data_as(c_double_p)))
        self.n_epoch = n

    def mass_setup(self, halo, mf_kind):
        arr = halo if isinstance(halo, ctypes.Array) else self.pack_halo(halo, self.n_epoch)
        assert len(arr) == self.n_epoch
        self._check(self._L.chomp_mass_setup(self._h, arr, int(mf_kind)))

    def halo_setup(self, profile, hods, tables):
        n = self.n_epoch
        pa = profile if isinstance(profile, ctypes.Array) else self.pack_halo(profile, n)
        ha = hods if isinstance(hods, ctypes.Array) else self.pack_hod(hods, n)
        assert len(pa) == n and len(ha) == n
        self._check(self._L.chomp_halo_setup(self._h, pa, ha, int(tables)))

    def stage_k(self, mass_halo, mf_kind, profile, hods, tables):
        """mass_setup + halo_setup in one call (chomp_stage_k)."""
        n = self.n_epoch
        ma = mass_halo if isinstance(mass_halo, ctypes.Array) else self.pack_halo(mass_halo, n)
        pa = profile if isinstance(profile, ctypes.Array) else self.pack_halo(profile, n)
        ha = hods if isinstance(hods, ctypes.Array) else self.pack_hod(hods, n)
        assert len(ma) == n and len(pa) == n and len(ha) == n
        self._check(self._L.chomp_stage_k(self._h, ma, int(mf_kind), pa, ha, int(tables)))

    def stage_k_halofit(self, mass_halo, mf_kind, profile, hods, tables, epoch, f1, f2, f3,
                        omega_l, w):
        """stage_k + halofit_setup(epoch, epoch, ...) in one call (chomp_stage_k_halofit)."""
        n = self.n_epoch
        ma = mass_halo if isinstance(mass_halo, ctypes.Array) else self.pack_halo(mass_halo, n)
        pa = profile if isinstance(profile, ctypes.Array) else self.pack_halo(profile, n)
        ha = hods if isinstance(hods, ctypes.Array) else self.pack_hod(hods, n)
        assert len(ma) == n and len(pa) == n and len(ha) == n
        self._check(self._L.chomp_stage_k_halofit(self._h, ma, int(mf_kind), pa, ha, int(tables),
                                                  epoch, f1, f2, f3, omega_l, w))

    def halofit_setup(self, dst, src, f1, f2, f3, omega_l, w):
        self._check(self._L.chomp_halofit_setup(self._h, dst, src, f1, f2, f3,
                                                omega_l, w))

    # -- Stage E -------------------------------------------------------------
    def power(self, which, k, epoch0=0, n=None, out=None):
        """k: numpy array (host path) or torch cuda tensor (device path, async on
        the context's stream).  Returns [n, nk] in the same kind of container."""
        n = self.n_epoch - epoch0 if n is None else n
        if _is_torch(k):
            import torch
            assert k.is_cuda and k.dtype == torch.float64 and k.is_contiguous()
            if out is None:
                out = torch.empty((n, k.numel()), dtype=torch.float64, device=k.device)
            pair = self._torch_enter()
            self._check(self._L.chomp_power_range(
                self._h, which, epoch0, n, ctypes.c_void_p(k.data_ptr()), k.numel(),
                ctypes.c_void_p(out.data_ptr()), DEVICE))
            self._torch_leave(pair)
            return out
        k = numpy.ascontiguousarray(k, dtype=numpy.float64).ravel()
        if out is None:
            out = numpy.empty((n, k.size), dtype=numpy.float64)
        if k.size:
            self._check(self._L.chomp_power_range(
                self._h, which, epoch0, n, ctypes.c_void_p(k.ctypes.data), k.size,
                ctypes.c_void_p(out.ctypes.data), HOST))
        return out

    def power_plan(self, k, epoch0=0):
        """Register a torch cuda k grid for repeated power() calls (chomp_power_plan): the
        k-only work is done once; the caller keeps k unchanged meanwhile."""
        assert _is_torch(k) and k.is_cuda and k.is_contiguous()
        self._check(self._L.chomp_power_plan(self._h, epoch0, ctypes.c_void_p(k.data_ptr()),
                                             k.numel()))
        # The library recognises the registered grid by (address, length, cosmology).  Keeping
        # the tensor alive for as long as the context may hold that registration means the
        # caching allocator can never hand the same address to a different k grid.
        self._plan_k = k

    def sigma_r(self, epoch, scale):
        s = numpy.ascontiguousarray(numpy.atleast_1d(scale), dtype=numpy.float64)
        out = numpy.empty_like(s)
        self._check(self._L.chomp_sigma_r(self._h, epoch, s.ctypes.data_as(c_double_p),
                                          s.size, out.ctypes.data_as(c_double_p)))
        return out

    def y_nfw(self, epoch, ln_k, mass):
        a, b = numpy.broadcast_arrays(numpy.asarray(ln_k, dtype=numpy.float64),
                                      numpy.asarray(mass, dtype=numpy.float64))
        a = numpy.ascontiguousarray(a).ravel()
        b = numpy.ascontiguousarray(b).ravel()
        out = numpy.empty_like(a)
        self._check(self._L.chomp_y_nfw(self._h, epoch, a.ctypes.data_as(c_double_p),
                                        b.ctypes.data_as(c_double_p), a.size,
                                        out.ctypes.data_as(c_double_p)))
        return out

    def scalars(self, epoch=0):
        out = numpy.empty(SC_COUNT)
        self._check(self._L.chomp_get_scalars(self._h, epoch,
                                              out.ctypes.data_as(c_double_p)))
        return {name: out[i] for name, i in SC.items()}

    def table(self, name, epoch=0):
        cfg = self.config
        n = {"ln_mass": cfg.mass_npoints, "nu": cfg.mass_npoints,
             "levels": 5 * cfg.halo_npoints}.get(name, cfg.halo_npoints)
        out = numpy.empty(n)
        self._check(self._L.chomp_get_table(self._h, epoch, TAB[name],
                                            out.ctypes.data_as(c_double_p), n))
        return out

    def eval(self, what, x, epoch=0):
        """Element-wise lookup; x numpy (any shape) or torch cuda tensor."""
        if _is_torch(x):
            return self._map1(self._L.chomp_eval, x, epoch, EV[what])
        xa = numpy.asarray(x, dtype=numpy.float64)
        out = self._map1(self._L.chomp_eval, numpy.ascontiguousarray(xa).ravel(),
                         epoch, EV[what])
        return out.reshape(xa.shape)

    def halofit_get(self, epoch=0):
        out = numpy.empty(HF_COUNT)
        self._check(self._L.chomp_halofit_get(self._h, epoch,
                                              out.ctypes.data_as(c_double_p)))
        return out

    def halofit_put(self, coef, epoch=0):
        c = numpy.ascontiguousarray(coef, dtype=numpy.float64)
        assert c.size == HF_COUNT
        self._check(self._L.chomp_halofit_put(self._h, epoch,
                                              c.ctypes.data_as(c_double_p)))

    def sync(self):
        self._check(self._L.chomp_sync(self._h))

    def status(self, epoch0=0, n=None):
        """Per-epoch status words (uint32 array; bits ST_*, describe_status).  Synchronises."""
        n = self.n_epoch - epoch0 if n is None else n
        out = numpy.zeros(n, dtype=numpy.uint32)
        if n:
            self._check(self._L.chomp_get_status(
                self._h, epoch0, n, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint))))
        return out

    def status_post(self):
        """Enqueue a copy of the status words to pinned host memory (chomp_status_post): no
        synchronisation; status_wait / warn_status(posted=True) pick it up."""
        self._check(self._L.chomp_status_post(self._h))

    def status_wait(self, epoch0=0, n=None):
        """The words of the last status_post (waits for that copy only, not for the stream)."""
        n = self.n_epoch - epoch0 if n is None else n
        out = numpy.zeros(n, dtype=numpy.uint32)
        if n:
            self._check(self._L.chomp_status_wait(
                self._h, epoch0, n, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint))))
        return out

    def warn_status(self, epoch0=0, n=None, stacklevel=3, posted=False):
        """Turn the status words of the epochs into Python warnings, as the reference's
        scipy.integrate.romberg did for an exhausted divmax (AccuracyWarning); a saturated
        mass-limit search gets a ChompParityWarning.  Returns the words.  posted: the words of
        the last status_post instead of a fresh (synchronising) read."""
        import warnings
        words = self.status_wait(epoch0, n) if posted else self.status(epoch0, n)
        for i, w in enumerate(words):
            w = int(w)
            if not w:
                continue
            kind = (ChompParityWarning if w & (ST_SATURATED | ST_MASS_SEARCH_EXHAUSTED)
                    else ChompAccuracyWarning)
            warnings.warn("epoch %d: %s" % (epoch0 + i, "; ".join(describe_status(w))), kind,
                          stacklevel=stacklevel)
        return words

    def deep_stats(self):
        """(knots done by the fast deep-level sums, knots done by literal evaluation) so far."""
        out = (ctypes.c_longlong * 7)()
        self._check(self._L.chomp_get_deep_stats(self._h, out))
        self.deep_detail = {"too_many_breaks": int(out[2]), "too_many_fine": int(out[3]),
                            "self_check": int(out[4]), "worst_estimate": out[5] * 1e-15,
                            "needs_node_evaluation": int(out[6])}
        return int(out[0]), int(out[1])

    def set_tuning(self, what, value):
        """Test / tuning hook (chomp_set_tuning); value None or < 0 restores the default."""
        self._check(self._L.chomp_set_tuning(self._h, int(what), -1 if value is None else int(value)))

    # -- projection ------------------------------------------------------------
    def kernel_setup(self, cosmo_dict, me_z_min, me_z_max, ktheta_min, ktheta_max,
                     wa, wb, bessel_order):
        c = cosmo_struct(cosmo_dict)
        self._check(self._L.chomp_kernel_setup(
            self._h, ctypes.byref(c), me_z_min, me_z_max, ktheta_min, ktheta_max,
            ctypes.byref(wa), ctypes.byref(wb), int(bessel_order)))

    def multi_epoch_setup(self, cosmo_dict, z_min, z_max):
        c = cosmo_struct(cosmo_dict)
        self._check(self._L.chomp_multi_epoch_setup(self._h, ctypes.byref(c),
                                                    float(z_min), float(z_max)))

    def me_eval(self, what, x):
        if _is_torch(x):
            return self._map1(self._L.chomp_me_eval, x, ME[what])
        xa = numpy.asarray(x, dtype=numpy.float64)
        out = self._map1(self._L.chomp_me_eval, numpy.ascontiguousarray(xa).ravel(),
                         ME[what])
        return out.reshape(xa.shape)

    def kernel_info(self):
        out = numpy.empty(KI_COUNT)
        self._check(self._L.chomp_kernel_info(self._h, out.ctypes.data_as(c_double_p)))
        return {name: out[i] for name, i in KI.items()}

    def kernel_table(self, name):
        cfg = self.config
        n = {"ln_ktheta": cfg.kernel_npoints, "kernel": cfg.kernel_npoints,
             "levels": cfg.kernel_npoints, "me_z": cfg.cosmo_npoints,
             "me_chi": cfg.cosmo_npoints, "me_growth": cfg.cosmo_npoints}.get(
                 name, cfg.window_npoints)
        out = numpy.empty(n)
        self._check(self._L.chomp_kernel_table(self._h, KTAB[name],
                                               out.ctypes.data_as(c_double_p), n))
        return out

    def _map1(self, fn, x, *pre):
        """Apply an (in, n, out, mem) entry point to numpy or torch input."""
        if _is_torch(x):
            import torch
            assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()
            out = torch.empty_like(x)
            pair = self._torch_enter()
            self._check(fn(self._h, *pre, ctypes.c_void_p(x.data_ptr()), x.numel(),
                           ctypes.c_void_p(out.data_ptr()), DEVICE))
            self._torch_leave(pair)
            return out
        x = numpy.ascontiguousarray(x, dtype=numpy.float64)
        out = numpy.empty_like(x)
        if x.size:
            self._check(fn(self._h, *pre, ctypes.c_void_p(x.ctypes.data), x.size,
                           ctypes.c_void_p(out.ctypes.data), HOST))
        return out

    def kernel_eval(self, ln_ktheta):
        return self._map1(self._L.chomp_kernel_eval, ln_ktheta)

    def kernel_raw(self, ln_ktheta):
        return self._map1(self._L.chomp_kernel_raw, ln_ktheta)

    def window_eval(self, which, chi):
        return self._map1(self._L.chomp_window_eval, chi, int(which))

    def wtheta(self, which, epoch, k_min, k_max, D_z, theta):
        return self._map1(self._L.chomp_wtheta, theta, int(which), epoch,
                          float(k_min), float(k_max), float(D_z))

    def wtheta_cell(self, which, epoch, k_min, k_max, D_z, theta, ell):
        """(w(theta), C_l) of one set-up in one call (chomp_wtheta_cell): with torch cuda
        tensors C_l is computed beside w(theta) on the context's side stream; same numbers as
        wtheta() and cell()."""
        if _is_torch(theta):
            import torch
            assert _is_torch(ell)
            for x in (theta, ell):
                assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()
            w, c = torch.empty_like(theta), torch.empty_like(ell)
            pair = self._torch_enter()
            self._check(self._L.chomp_wtheta_cell(
                self._h, int(which), epoch, float(k_min), float(k_max), float(D_z),
                ctypes.c_void_p(theta.data_ptr()), theta.numel(), ctypes.c_void_p(w.data_ptr()),
                ctypes.c_void_p(ell.data_ptr()), ell.numel(), ctypes.c_void_p(c.data_ptr()), DEVICE))
            self._torch_leave(pair)
            return w, c
        th = numpy.ascontiguousarray(theta, dtype=numpy.float64).ravel()
        el = numpy.ascontiguousarray(ell, dtype=numpy.float64).ravel()
        w, c = numpy.empty_like(th), numpy.empty_like(el)
        if th.size and el.size:
            self._check(self._L.chomp_wtheta_cell(
                self._h, int(which), epoch, float(k_min), float(k_max), float(D_z),
                ctypes.c_void_p(th.ctypes.data), th.size, ctypes.c_void_p(w.ctypes.data),
                ctypes.c_void_p(el.ctypes.data), el.size, ctypes.c_void_p(c.ctypes.data), HOST))
        return w, c

    def set_timing(self, on=True):
        self._check(self._L.chomp_set_timing(self._h, int(bool(on))))

    def get_timing(self):
        """Microseconds of (k_power_prep, k_power_stream, k_power_grid_lanes) of the last
        timed streaming power() call."""
        out = numpy.empty(3)
        self._check(self._L.chomp_get_timing(self._h, out.ctypes.data_as(c_double_p), 3))
        return out

    def covariance_table(self, which, epoch, D_z):
        """(ln_K, projected spectrum, Romberg levels), each [kernel_npoints]."""
        n = self.config.kernel_npoints
        ln_K, proj, lev = numpy.empty(n), numpy.empty(n), numpy.empty(n)
        self._check(self._L.chomp_covariance_table(
            self._h, int(which), epoch, float(D_z), ln_K.ctypes.data_as(c_double_p),
            proj.ctypes.data_as(c_double_p), lev.ctypes.data_as(c_double_p), n))
        return ln_K, proj, lev.astype(int)

    def covariance_gaussian(self, j0_limit, area, poisson_a, poisson_b, theta_a, theta_b):
        ta = numpy.ascontiguousarray(theta_a, dtype=numpy.float64).ravel()
        tb = numpy.ascontiguousarray(theta_b, dtype=numpy.float64).ravel()
        assert ta.size == tb.size
        th = numpy.concatenate([ta, tb])
        out = numpy.empty(ta.size)
        if ta.size:
            self._check(self._L.chomp_covariance_gaussian(
                self._h, float(j0_limit), float(area), float(poisson_a), float(poisson_b),
                ctypes.c_void_p(th.ctypes.data), ta.size, ctypes.c_void_p(out.ctypes.data),
                HOST))
        return out

    def hod_stats(self, epoch0=0, n=None):
        """[n, 3]: effective bias, effective halo mass, satellite fraction."""
        n = self.n_epoch - epoch0 if n is None else n
        out = numpy.empty((n, 3))
        self._check(self._L.chomp_hod_stats(self._h, epoch0, n, out.ctypes.data_as(c_double_p)))
        return out

    def xi3d(self, which, epoch, k_min, k_max, r):
        return self._map1(self._L.chomp_xi3d, r, int(which), epoch, float(k_min), float(k_max))

    def spline_eval(self, xk, yk, x, deriv=0):
        """Not-a-knot cubic spline through (xk, yk) at x (FITPACK k=3, s=0); deriv=1: its
        first derivative."""
        xk = numpy.ascontiguousarray(xk, dtype=numpy.float64)
        yk = numpy.ascontiguousarray(yk, dtype=numpy.float64)
        xa = numpy.ascontiguousarray(numpy.atleast_1d(x), dtype=numpy.float64).ravel()
        out = numpy.empty_like(xa)
        if xa.size:
            self._check(self._L.chomp_spline_eval(
                self._h, xk.ctypes.data_as(c_double_p), yk.ctypes.data_as(c_double_p), xk.size,
                xa.ctypes.data_as(c_double_p), xa.size, int(deriv),
                out.ctypes.data_as(c_double_p)))
        return out

    def set_precision(self, mode):
        """Arithmetic of the w(theta) integral: PREC_F64 (default, the only mode held to
        the parity bar) or one of the narrowed modes of the configs[4] precision sweep."""
        self._check(self._L.chomp_set_precision(self._h, int(mode)))

    def cell(self, which, epoch, D_z, ell):
        return self._map1(self._L.chomp_cell, ell, int(which), epoch, float(D_z))
