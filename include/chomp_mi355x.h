/*
 * chomp_mi355x.h -- C ABI of libchomp_mi355x.so, the MI355X (gfx950) implementation
 * of CHOMP's halo-model + Limber-projection hot path.
 *
 * The reference (morriscb/chomp) has no FFI: its boundary for this path is the
 * Python method surface of halo.Halo / kernel.Kernel / correlation.Correlation.
 * Each entry point below names the reference interface it replaces (file:line in
 * /root/reference).  The chomp_amd Python package binds these with ctypes and mirrors the
 * reference classes; INTEGRATION.md shows the stub a reference maintainer would
 * add.  All functions return CHOMP_OK (0) or a negative error code; the message
 * is available from chomp_last_error().  Plain pointers and sizes only.
 *
 * Memory-space convention: every array argument is paired with (or covered by) a
 * `mem` argument: CHOMP_HOST (pointer is host memory; the library stages it) or
 * CHOMP_DEVICE (pointer is HBM on the context's device, e.g. torch
 * tensor.data_ptr(); no copy, the call is asynchronous on the context's stream).
 *
 * Threading: one HIP stream per context; calls on one context must be serialised
 * by the caller; distinct contexts are independent.
 */
#ifndef CHOMP_MI355X_H
#define CHOMP_MI355X_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHOMP_OK 0
#define CHOMP_ERR_ARG (-1)      /* bad argument (KeyError/ValueError analogue)   */
#define CHOMP_ERR_HIP (-2)      /* HIP runtime failure                            */
#define CHOMP_ERR_STATE (-3)    /* stage called before its prerequisite           */
#define CHOMP_ERR_SCOPE (-4)    /* feature outside the hot-path scope (w != -1)   */

#define CHOMP_HOST 0
#define CHOMP_DEVICE 1

typedef struct chomp_ctx chomp_ctx;

/* defaults.default_cosmo_dict (defaults.py:6-18); w0/wa must be -1/0. */
typedef struct chomp_cosmo {
  double omega_m0, omega_b0, omega_l0, omega_r0, cmb_temp, h, sigma_8, n_scalar,
      w0, wa;
} chomp_cosmo;

/* defaults.default_halo_dict (defaults.py:21-29). */
typedef struct chomp_halo_par {
  double stq, st_little_a, c0, beta, alpha, delta_v;
} chomp_halo_par;

/* hod.HODZheng parameters (hod.py:156-186, defaults.py:33-39). */
typedef struct chomp_hod_par {
  double log_M_min, sigma, log_M_0, log_M_1p, alpha;
} chomp_hod_par;

/* Snapshot of defaults.default_limits + defaults.default_precision
 * (defaults.py:42-51, 62-92), taken when the context is created. */
typedef struct chomp_config {
  double k_min, k_max, mass_min, mass_max;
  double corr_precision, cosmo_precision, dNdz_precision, halo_precision,
      kernel_precision, mass_precision, window_precision, global_precision;
  int corr_npoints, cosmo_npoints, halo_npoints, kernel_npoints,
      kernel_bessel_limit, mass_npoints, window_npoints, divmax;
} chomp_config;

/* Mass functions: mass_function.MassFunction (Sheth-Tormen, mass_function.py:25)
 * and mass_function.TinkerMassFunction (:436). */
#define CHOMP_MF_ST 0
#define CHOMP_MF_TINKER 1

/* Knot tables built by chomp_halo_setup (bit mask): one bit per lazily
 * initialised spline of the reference's Halo (halo.py:96-101), so a caller can
 * mirror which of them a setter invalidates (halo.py:165-170, 186-189). */
#define CHOMP_T_H_M 1u    /* Halo._initialize_h_m    halo.py:904-927   */
#define CHOMP_T_PP_MM 2u  /* Halo._initialize_pp_mm  halo.py:971-994   */
#define CHOMP_T_H_G 4u    /* Halo._initialize_h_g    halo.py:929-969   */
#define CHOMP_T_PP_GM 8u  /* Halo._initialize_pp_gm  halo.py:1043-1086 */
#define CHOMP_T_PP_GG 16u /* Halo._initialize_pp_gg  halo.py:996-1041  */
/* OR-ed in: the object is a HaloExclusion (halo.py:1201-1233): h_m and h_g are built
 * with the halo-exclusion mass window in their integrands. */
#define CHOMP_T_EXCLUSION 64u
#define CHOMP_FAM_MM (CHOMP_T_H_M | CHOMP_T_PP_MM)               /* power_mm */
#define CHOMP_FAM_GM (CHOMP_T_H_M | CHOMP_T_H_G | CHOMP_T_PP_GM) /* power_gm */
#define CHOMP_FAM_GG (CHOMP_T_H_G | CHOMP_T_PP_GG)               /* power_gg */

/* Power spectra served by chomp_power. */
#define CHOMP_P_LIN 0 /* Halo.linear_power  halo.py:266-275 */
#define CHOMP_P_MM 1  /* Halo.power_mm      halo.py:277-320 */
#define CHOMP_P_GM 2  /* Halo.power_gm/mg   halo.py:322-389 */
#define CHOMP_P_GG 3  /* Halo.power_gg      halo.py:391-439 */
#define CHOMP_P_HALOFIT 16 /* OR-ed in: HaloFit.power_* halo.py:1325-1413 */
/* OR-ed in: Halo(extrapolate=True) -- above k_max P_mm continues as a rescaled linear
 * spectrum, P_gm / P_gg as power laws (halo.py:300-312, 341-367, 405-431); HaloFit
 * ignores it, as in the reference. */
#define CHOMP_P_EXTRAPOLATE 32

void chomp_default_config(chomp_config* cfg);

/* Create a context on `device` using HIP stream `hip_stream` (NULL -> a new
 * stream owned by the context).  Replaces the import-time snapshot of
 * defaults.py that every reference constructor reads. */
int chomp_ctx_create(const chomp_config* cfg, int device, void* hip_stream,
                     chomp_ctx** out);
void chomp_ctx_destroy(chomp_ctx* ctx);
const char* chomp_last_error(chomp_ctx* ctx);
/* Measurement aid (no counterpart in the reference).  With timing on, the streaming shape
 * of chomp_power / chomp_power_range (large grid of one cosmology) brackets its three
 * launches -- k_power_prep, k_power_stream, k_power_grid_lanes -- with HIP events on the
 * context's stream; chomp_get_timing waits for the last such call and returns their
 * durations in microseconds (us[3], n = 3).  CHOMP_ERR_STATE when the last call took
 * another launch shape or timing is off. */
int chomp_set_timing(chomp_ctx* ctx, int on);
int chomp_get_timing(chomp_ctx* ctx, double* us, size_t n);

/* The HIP stream every call of this context is queued on (the one given to chomp_ctx_create,
 * or the one it created): a caller that works on another stream orders the two with events
 * (hipEventRecord / hipStreamWaitEvent) around calls that pass device buffers. */
int chomp_get_stream(chomp_ctx* ctx, void** out);

/* Block until everything queued on the context's stream has finished. */
int chomp_sync(chomp_ctx* ctx);

/* ---- Stage K: per-(cosmology, z) tables -------------------------------------
 * An "epoch" is one (cosmology, redshift) pair = one cosmology.SingleEpoch
 * (cosmology.py:39-119).  A batch of n epochs is set up together; this is the
 * z-axis of the (k, z) grid and the design-point axis of SimulationDesign. */

/* SingleEpoch.__init__/_initialize_defaults for every epoch (cosmology.py:39-119):
 * flatness flags, delta_H, chi(z), growth, sigma_8 normalisation. */
int chomp_epochs_set(chomp_ctx* ctx, size_t n_epoch, const chomp_cosmo* cosmo,
                     const double* z);

/* MassFunction.__init__ (mass_function.py:38-61; Tinker :448-492): mass-limit
 * search (:160-203), nu table + splines (:205-223), normalisation (:225-241).
 * `par[i]` are the halo_dict values the mass function sees for epoch i. */
int chomp_mass_setup(chomp_ctx* ctx, const chomp_halo_par* par, int mf_kind);

/* Halo.__init__ + lazy initialisers (halo.py:41-104, 674-707, 839-1086): n_bar,
 * then the 50-knot tables selected by `tables` (CHOMP_T_* bits) and their
 * splines; tables not selected keep their previous contents.
 * `profile[i]` are the halo_dict values the profile (c0, beta, delta_v) sees;
 * they differ from chomp_mass_setup's only after Halo.set_halo (halo.py:220-235,
 * which does not rebuild the profile splines). */
int chomp_halo_setup(chomp_ctx* ctx, const chomp_halo_par* profile,
                     const chomp_hod_par* hod, unsigned tables);

/* chomp_mass_setup followed by chomp_halo_setup in fewer launches (the mass function's tail
 * and the halo model's node tables share a kernel): what a batch that always builds both --
 * the (k, z) grid, a SimulationDesign -- calls per set-up.  Same results, bit for bit. */
int chomp_stage_k(chomp_ctx* ctx, const chomp_halo_par* mass_par, int mf_kind,
                  const chomp_halo_par* profile, const chomp_hod_par* hod,
                  unsigned tables);

/* chomp_stage_k followed by chomp_halofit_setup(epoch, epoch, ...) in one call -- a HaloFit
 * object's first set-up (halo.py:1236-1266 then 1268-1319).  Same results, bit for bit; the
 * HaloFit sigma table and fit run on a second HIP stream beside the halo model's knot
 * integrals (neither needs the other's results) and are joined before the call returns. */
int chomp_stage_k_halofit(chomp_ctx* ctx, const chomp_halo_par* mass_par, int mf_kind,
                          const chomp_halo_par* profile, const chomp_hod_par* hod,
                          unsigned tables, size_t epoch, double f_1, double f_2, double f_3,
                          double omega_l, double w);

/* HaloFit._initialize_sigma_spline (halo.py:1268-1319) for epoch `src_epoch`,
 * stored as the HaloFit coefficient set of epoch `dst_epoch`; f_1..f_3, omega_l
 * and w are passed explicitly because the reference fixes them at construction
 * (halo.py:1261-1266) and never refreshes them on set_redshift. */
int chomp_halofit_setup(chomp_ctx* ctx, size_t dst_epoch, size_t src_epoch,
                        double f_1, double f_2, double f_3, double omega_l,
                        double w);

/* ---- Stage E: grid evaluation ------------------------------------------------
 * Halo.linear_power / power_mm / power_gm / power_gg (halo.py:266-439) for every
 * epoch of the batch: out[i*nk + j] = P_which(k[j]; epoch i), row-major (z-major).
 * k in h/Mpc, P in (Mpc/h)^3.  extrapolate=False semantics: k < k_min -> scaled
 * linear spectrum, k > k_max -> 0. */
int chomp_power(chomp_ctx* ctx, int which, const double* k, size_t nk,
                double* out, int mem);
/* Same for the epoch range [epoch0, epoch0 + n). */
int chomp_power_range(chomp_ctx* ctx, int which, size_t epoch0, size_t n,
                      const double* k, size_t nk, double* out, int mem);

/* Register a k grid (device memory, 16-byte aligned, even length) for repeated chomp_power /
 * chomp_power_range calls over the same cosmology: everything that depends on k alone -- ln k,
 * its knot interval, the Eisenstein-Hu shape of the linear spectrum (halo.py:649-672,
 * cosmology.py:449-472) -- is tabulated now, and calls that pass the same pointer and length
 * for epochs of the same cosmology skip that step.  The caller promises not to modify k[]
 * while the registration lasts; it ends with the next chomp_power call on another grid (or a
 * host buffer) that takes the streaming launch shape, or with the context.  No counterpart in
 * the reference, where every power_mm(k) call recomputes log(k) and the linear spectrum. */
int chomp_power_plan(chomp_ctx* ctx, size_t epoch0, const double* k, size_t nk);

/* SingleEpoch.sigma_r (cosmology.py:602-642) at n scales for one epoch (host). */
int chomp_sigma_r(chomp_ctx* ctx, size_t epoch, const double* scale, size_t n,
                  double* out);
/* Halo.y (NFW, halo.py:561-585) at (ln_k[i], mass[i]) pairs for one epoch (host). */
int chomp_y_nfw(chomp_ctx* ctx, size_t epoch, const double* ln_k,
                const double* mass, size_t n, double* out);

/* Element-wise lookups of one epoch (host or device buffers):
 * MassFunction.nu / ln_mass / f_nu / bias_nu (mass_function.py:243-346),
 * HODZheng moments (hod.py:189-230), Halo.virial_radius / concentration
 * (halo.py:441-463), SingleEpoch.delta_k (cosmology.py:574-587). */
#define CHOMP_EV_NU_OF_MASS 0
#define CHOMP_EV_LN_MASS_OF_NU 1
#define CHOMP_EV_F_NU 2
#define CHOMP_EV_BIAS_NU 3
#define CHOMP_EV_HOD_FIRST 4
#define CHOMP_EV_HOD_SECOND 5
#define CHOMP_EV_HOD_CENTRAL 6
#define CHOMP_EV_HOD_SATELLITE 7
#define CHOMP_EV_VIRIAL_RADIUS 8
#define CHOMP_EV_CONCENTRATION 9
#define CHOMP_EV_DELTA_K 10
int chomp_eval(chomp_ctx* ctx, size_t epoch, int what, const double* x, size_t n,
               double* out, int mem);

/* HaloFit coefficient block of one epoch (f_1, f_2, f_3, k_s, n_eff, C, a_n, b_n,
 * c_n, gamma_n, alpha_n, beta_n, mu_n, nu_n): read it back / write it.  The
 * reference keeps these across set_redshift (halo.py:1254-1259 never resets
 * _initialized_sigma_spline), so a caller mirroring that re-installs them after
 * chomp_epochs_set. */
#define CHOMP_HF_COUNT 14
int chomp_halofit_get(chomp_ctx* ctx, size_t epoch, double* out);
int chomp_halofit_put(chomp_ctx* ctx, size_t epoch, const double* in);

/* ---- Introspection (per-stage parity tests, write() mirrors) ---------------- */
#define CHOMP_SC_Z 0
#define CHOMP_SC_CHI 1
#define CHOMP_SC_GROWTH 2
#define CHOMP_SC_OMEGA_M 3
#define CHOMP_SC_OMEGA_L 4
#define CHOMP_SC_DELTA_C 5
#define CHOMP_SC_DELTA_V 6
#define CHOMP_SC_RHO_BAR 7
#define CHOMP_SC_SIGMA_NORM 8
#define CHOMP_SC_LN_MASS_MIN 9
#define CHOMP_SC_LN_MASS_MAX 10
#define CHOMP_SC_NU_MIN 11
#define CHOMP_SC_NU_MAX 12
#define CHOMP_SC_M_STAR 13
#define CHOMP_SC_F_NORM 14
#define CHOMP_SC_BIAS_NORM 15
#define CHOMP_SC_N_BAR 16
#define CHOMP_SC_N_BAR_OVER_RHO_BAR 17
#define CHOMP_SC_N_SEARCH 18
#define CHOMP_SC_MF_DELTA_V 19
#define CHOMP_SC_T_ALPHA 20
#define CHOMP_SC_T_BETA 21
#define CHOMP_SC_T_GAMMA 22
#define CHOMP_SC_T_PHI 23
#define CHOMP_SC_T_ETA 24
#define CHOMP_SC_GROWTH_NORM 25
#define CHOMP_SC_DELTA_H 26
#define CHOMP_SC_HF_K_S 27
#define CHOMP_SC_HF_N_EFF 28
#define CHOMP_SC_HF_C 29
#define CHOMP_SC_COUNT 30
/* out[CHOMP_SC_COUNT] <- scalars of one epoch (SingleEpoch / MassFunction / Halo
 * attributes: _chi, _growth, omega_m(), delta_c(), ..., f_norm, n_bar). */
int chomp_get_scalars(chomp_ctx* ctx, size_t epoch, double* out);

/* Per-epoch status word: what the reference would have told its user through a warning (or
 * through never returning), reported instead of computed around.  Bits accumulate over the
 * stages of an epoch; chomp_epochs_set clears the word, chomp_halo_setup the HALO / NONFINITE
 * bits.  out[n] <- status of epochs [epoch0, epoch0 + n) (host buffer; synchronises).
 *
 * MASS_MIN_SATURATED: MassFunction._set_mass_limits' 5 % walk (mass_function.py:171-181) ended
 *   at a mass so small that k R < 0.2 over the whole k range of sigma_r (clamped at 100 k_max,
 *   cosmology.py:627-632).  There nu(M) has converged to a constant above the band -- in exact
 *   arithmetic the walk would never end -- and what ends the reference's walk is the rounding
 *   error of 3 (sin x / x^3 - cos x / x^2) at x << 1 (its variance biases sigma^2 upwards like
 *   eps^2 / x^4): the step it stops at is a property of the libm in use, not of the model.
 *   Results for such an epoch can differ from the reference's by percents (either answer is
 *   equally arbitrary); low sigma_8 / Omega_m at z >~ 0.8, M_min ~ 1e-8 M_sun/h.
 * MASS_MAX_SATURATED: the mass_max walk ended with sigma_r's k range clamped at k_min / 100
 *   (cosmology.py:617-622, behind the reference's commented-out extrapolation warning).
 * MASS_SEARCH_EXHAUSTED: the walk did not end within 2047 steps (the reference loops on).
 * SIGMA_DIVMAX: a sigma(R) Romberg of the nu table exhausted divmax (scipy: AccuracyWarning).
 * HALO_DIVMAX_*: some knot of that table exhausted divmax (halo.py:909-915, 951-957, 976-982,
 *   1018-1024, 1065-1071; scipy returns the last row with an AccuracyWarning -- with the
 *   default precision the discontinuous HOD integrands of pp_gm / pp_gg do this routinely).
 * NONFINITE: a knot table holds a NaN or an infinity. */
#define CHOMP_ST_MASS_MIN_SATURATED 1u
#define CHOMP_ST_MASS_MAX_SATURATED 2u
#define CHOMP_ST_MASS_SEARCH_EXHAUSTED 4u
#define CHOMP_ST_SIGMA_DIVMAX 8u
#define CHOMP_ST_HALO_DIVMAX_H_M 0x100u   /* << 0..4: H_M, PP_MM, H_G, PP_GM, PP_GG */
#define CHOMP_ST_HALO_DIVMAX_PP_MM 0x200u
#define CHOMP_ST_HALO_DIVMAX_H_G 0x400u
#define CHOMP_ST_HALO_DIVMAX_PP_GM 0x800u
#define CHOMP_ST_HALO_DIVMAX_PP_GG 0x1000u
#define CHOMP_ST_NONFINITE 0x10000u
int chomp_get_status(chomp_ctx* ctx, size_t epoch0, size_t n, unsigned* out);
/* The same words without draining the stream.  chomp_status_post makes every epoch's word, as
 * it is behind the work enqueued so far (call it right after a set-up), available in pinned host
 * memory; chomp_status_wait blocks until THOSE words have landed -- not until the stream is idle
 * -- and returns them as they were then, whatever has been set up since.  Behind a halo set-up
 * (chomp_stage_k, chomp_halo_setup) a post puts nothing on the stream: the set-up's finalising
 * blocks have written the words, tagged with the set-up's sequence number, to the pinned words
 * themselves, and the wait polls for that number (an event record alone costs a step ~10 us on
 * this stack).  Behind any other set-up it is a copy and an event.  A caller that keeps its
 * samples on the device posts after each set-up and waits whenever it next has a reason to look
 * (the reference printed its AccuracyWarning at the time of the integral; here the time of
 * looking is the caller's choice).
 * ERR_STATE: wait before any post; wait while the stream is being captured; wait for a set-up
 * that never finalised its epochs (it failed). */
int chomp_status_post(chomp_ctx* ctx);
int chomp_status_wait(chomp_ctx* ctx, size_t epoch0, size_t n, unsigned* out);

/* Test / tuning hooks (no counterpart in the reference; not needed by a caller): override a
 * launch-shape decision of this context.  value < 0 restores the default.
 *   CHOMP_TUNE_E_STREAM_MIN  samples from which chomp_power takes the streaming launch shape
 *   CHOMP_TUNE_E_ROWS        rows per block of the streaming kernel (1, 2 or 4)
 *   CHOMP_TUNE_DEEP_LITERAL  1: knots beyond the node tables by literal evaluation of every
 *                            Romberg node (the checker of the fast deep-level sums)
 *   CHOMP_TUNE_DEEP_TOL      self-check threshold of the fast deep-level sums, in units of 1e-15
 *                            relative (default 1 000 000 = 1e-9): a knot whose estimate is above
 *                            it is handed to the literal evaluation.  0 sends every knot that
 *                            reaches the self-check there
 *   CHOMP_TUNE_DEEP_MAX_BREAKS  break points of the integrand (changes of its discrete state
 *                            along ln nu) a knot may have and still take the fast sums (default
 *                            and maximum 8); a knot with more goes to the literal evaluation
 *   CHOMP_TUNE_DEEP_MAX_FINE coarse intervals a knot may evaluate node by node (break points,
 *                            the margin above a singular satellite onset, segments shorter
 *                            than a stencil) and still take the fast sums (default and
 *                            maximum 64)
 *   CHOMP_TUNE_HOD_CAP       Romberg level (6..10; default 9 for a set-up of one or a few
 *                            epochs, 10 for a batch) up to which a knot of the HOD integrands
 *                            walks the node table before it is listed for the fast deep-level
 *                            sums (10: the whole table, as for the smooth pair)
 *   CHOMP_TUNE_DEEP_SLOTS    slots of the sample buffer of the listed knots (default: one per
 *                            knot that can be listed, up to 1 GiB): with fewer slots than
 *                            listed knots the sampling and summing launches work the list off
 *                            in rounds (the path of a batch of hundreds of epochs, for a test)
 *   CHOMP_TUNE_WTHETA_DIRECT 1: w(theta) by evaluating the kernel spline at every Romberg node
 *                            (the checker of the moment route of chomp_wtheta)
 *   CHOMP_TUNE_CELL_ONE_KERNEL 1: C_l with every Romberg level in the per-multipole kernel (the
 *                            checker of the hand-over to k_cell_deep)
 *   CHOMP_TUNE_ROCTX         1: roctx ranges around the stages on the host timeline (one per
 *                            entry point: "chomp:epochs_set", "chomp:stage_k", "chomp:power",
 *                            "chomp:wtheta", ...; rocprofv3 --marker-trace shows them over
 *                            the kernels they queued).  The marker library is looked up with
 *                            dlopen; CHOMP_ERR_STATE if none is installed. */
#define CHOMP_TUNE_E_STREAM_MIN 0
#define CHOMP_TUNE_E_ROWS 1
#define CHOMP_TUNE_DEEP_LITERAL 2
#define CHOMP_TUNE_ROCTX 3
#define CHOMP_TUNE_WTHETA_DIRECT 4
#define CHOMP_TUNE_CELL_ONE_KERNEL 5
#define CHOMP_TUNE_DEEP_TOL 6
#define CHOMP_TUNE_DEEP_MAX_BREAKS 7
#define CHOMP_TUNE_DEEP_MAX_FINE 8
#define CHOMP_TUNE_HOD_CAP 9
#define CHOMP_TUNE_DEEP_SLOTS 10
#define CHOMP_TUNE_COUNT 11
int chomp_set_tuning(chomp_ctx* ctx, int what, long long value);
/* Measurement aid: out[7] <- knots beyond the node tables done so far (since the context was
 * created) by [0] the fast deep-level sums, [1] literal evaluation of every node; why literal:
 * [2] too many break points, [3] too many node-by-node intervals, [4] the self-check of the
 * interpolation; [5] the largest self-check error estimate seen, in units of 1e-15; [6] why
 * literal, continued: the knot needed the integrand at a node off the coarse grid in a set-up
 * whose kernel instance carries none (every HOD with alpha = 1). */
int chomp_get_deep_stats(chomp_ctx* ctx, long long* out);

#define CHOMP_TAB_LN_MASS 0 /* MassFunction._ln_mass_array  [mass_npoints] */
#define CHOMP_TAB_NU 1      /* MassFunction._nu_array       [mass_npoints] */
#define CHOMP_TAB_H_M 2     /* knots of Halo._h_m_spline    [halo_npoints] */
#define CHOMP_TAB_PP_MM 3
#define CHOMP_TAB_H_G 4
#define CHOMP_TAB_PP_GM 5
#define CHOMP_TAB_PP_GG 6
#define CHOMP_TAB_LEVELS 7  /* Romberg levels reached, 5 x halo_npoints, as doubles */
#define CHOMP_TAB_HF_LN_SIGMA2 8 /* HaloFit._ln_sigma2_array [halo_npoints] */
int chomp_get_table(chomp_ctx* ctx, size_t epoch, int table, double* out,
                    size_t n);

/* ---- Projection: MultiEpoch, windows, kernel, correlation --------------------
 * One projection set-up per context. */

/* kernel.dNdz family (kernel.py:26-179). kind: CHOMP_DNDZ_*; p[] per kind:
 * MAGLIM {a, z0, b}; GAUSSIAN {z0, sigma_z}; BOXCAR {} -- the base class dNdz, whose
 * raw_dndz is 1 (kernel.py:56-65).  z_min/z_max are the values AFTER the constructor's
 * clipping (kernel.py:101-104, 164-173), done by the caller.
 * PPOLY: dNdzInterpolation (kernel.py:181-208), a p(z) tabulated by the caller.  The
 * reference fits a FITPACK spline of order 2 (or a smoothing spline) to the table in its
 * constructor; the caller does the same and hands the spline over as a piecewise
 * polynomial: pp_n pieces, piece i on [pp_breaks[i], pp_breaks[i + 1]] with value
 * sum_m pp_coef[i (pp_order + 1) + m] (z - pp_breaks[i])^m (host pointers, copied by
 * chomp_kernel_setup; pp_order <= 5); z_min / z_max = the table's first / last z. */
#define CHOMP_DNDZ_MAGLIM 0
#define CHOMP_DNDZ_GAUSSIAN 1
#define CHOMP_DNDZ_BOXCAR 2
#define CHOMP_DNDZ_PPOLY 3
typedef struct chomp_dndz {
  int kind;
  int pad_;
  double z_min, z_max;
  double p[4];
  const double* pp_breaks;   /* [pp_n + 1] */
  const double* pp_coef;     /* [pp_n][pp_order + 1] */
  int pp_n, pp_order;
} chomp_dndz;

/* kernel.WindowFunctionGalaxy (kernel.py:358-387) / WindowFunctionConvergence
 * (:410-484). */
#define CHOMP_WINDOW_GALAXY 0
#define CHOMP_WINDOW_CONVERGENCE 1
/* WindowFunctionFlatConvergence (kernel.py:487-513): constant 3/2 Omega_m H0^2 1907.71
 * between dist.z_min and dist.z_max; WindowFunctionConvergenceDelta (:516-556): sources on
 * one plane at z = dist.z_max.  Both take only z_min / z_max from `dist`. */
#define CHOMP_WINDOW_FLAT_CONVERGENCE 2
#define CHOMP_WINDOW_CONVERGENCE_DELTA 3
typedef struct chomp_window {
  int kind;
  int pad_;
  chomp_dndz dist;
} chomp_window;

/* cosmology.MultiEpoch(z_min, z_max, cosmo) (cosmology.py:747-817) +
 * kernel.Kernel / GalaxyGalaxyLensingKernel.__init__ and _initialize_spline
 * (kernel.py:584-649, 803-839): window tables, z_bar, 50 kernel knots.
 * bessel_order 0 (J0) or 2 (J2). */
int chomp_kernel_setup(chomp_ctx* ctx, const chomp_cosmo* cosmo, double me_z_min,
                       double me_z_max, double ktheta_min, double ktheta_max,
                       const chomp_window* a, const chomp_window* b,
                       int bessel_order);

/* cosmology.MultiEpoch alone (cosmology.py:747-817): the 50-point chi(z), z(chi),
 * D(z) tables and splines, without windows or kernel. */
int chomp_multi_epoch_setup(chomp_ctx* ctx, const chomp_cosmo* cosmo, double z_min,
                            double z_max);
/* MultiEpoch.comoving_distance / redshift / growth_factor (cosmology.py:873-953)
 * of the context's MultiEpoch (after chomp_multi_epoch_setup or chomp_kernel_setup). */
#define CHOMP_ME_CHI_OF_Z 0
#define CHOMP_ME_Z_OF_CHI 1
#define CHOMP_ME_GROWTH_OF_Z 2
int chomp_me_eval(chomp_ctx* ctx, int what, const double* x, size_t n, double* out,
                  int mem);

#define CHOMP_KI_Z_BAR 0
#define CHOMP_KI_CHI_MIN 1
#define CHOMP_KI_CHI_MAX 2
#define CHOMP_KI_Z_MIN 3
#define CHOMP_KI_Z_MAX 4
#define CHOMP_KI_D_ZBAR 5 /* MultiEpoch.growth_factor(z_bar), correlation.py:94 */
#define CHOMP_KI_NORM_A 6 /* dNdz.norm of window a's distribution */
#define CHOMP_KI_NORM_B 7
#define CHOMP_KI_WA_CHI_MIN 8 /* WindowFunction.chi_min / chi_max of window a, b */
#define CHOMP_KI_WA_CHI_MAX 9
#define CHOMP_KI_WB_CHI_MIN 10
#define CHOMP_KI_WB_CHI_MAX 11
#define CHOMP_KI_J_LIMIT 12   /* Kernel._j0_limit / _j2_limit */
#define CHOMP_KI_COUNT 13
int chomp_kernel_info(chomp_ctx* ctx, double* out);

#define CHOMP_KTAB_LN_KTHETA 0 /* Kernel._ln_ktheta_array [kernel_npoints]   */
#define CHOMP_KTAB_KERNEL 1    /* Kernel._kernel_array    [kernel_npoints]   */
#define CHOMP_KTAB_WA_CHI 2    /* window a _chi_array     [window_npoints]   */
#define CHOMP_KTAB_WA 3        /* window a _wf_array                          */
#define CHOMP_KTAB_WB_CHI 4
#define CHOMP_KTAB_WB 5
#define CHOMP_KTAB_ME_Z 6      /* MultiEpoch._z_array     [cosmo_npoints]    */
#define CHOMP_KTAB_ME_CHI 7
#define CHOMP_KTAB_ME_GROWTH 8
#define CHOMP_KTAB_LEVELS 9
int chomp_kernel_table(chomp_ctx* ctx, int table, double* out, size_t n);

/* Kernel.raw_kernel(ln_ktheta) (kernel.py:678-704): the projection integral itself, not
 * its 50-knot spline. */
int chomp_kernel_raw(chomp_ctx* ctx, const double* ln_ktheta, size_t n,
                     double* out, int mem);
/* Kernel.kernel(ln_ktheta) (kernel.py:714-729); argument is ln(k*theta). */
int chomp_kernel_eval(chomp_ctx* ctx, const double* ln_ktheta, size_t n,
                      double* out, int mem);
/* WindowFunction.window_function(chi) (kernel.py:326-340) of window 0 (a) / 1 (b). */
int chomp_window_eval(chomp_ctx* ctx, int which_window, const double* chi,
                      size_t n, double* out, int mem);

/* Correlation.correlation(theta_rad) (correlation.py:242-275):
 * w(theta) = int dlnk k^2/(2 pi) P(k)/D_z^2 K(ln k theta), one wavefront-group per
 * theta.  P is `which` of halo epoch `epoch` (the caller has moved the halo to
 * z_bar as Correlation.__init__ does, correlation.py:102-103). */
int chomp_wtheta(chomp_ctx* ctx, int which, size_t epoch, double k_min,
                 double k_max, double D_z, const double* theta, size_t n,
                 double* out, int mem);
/* Gaussian covariance of w(theta), Covariance(corr, corr) with nongaussian_cov=False.
 *
 * chomp_covariance_table replaces Covariance._initialize_halo_splines (covariance.py:455-543,
 * the matching_corrs branch): on kernel_npoints knots in ln K, from ln(k_min chi_min) to
 * ln(k_max chi_max) (covariance.py:159-175), the projected spectrum
 *   int dchi P(K/chi) W_a(chi) W_b(chi) D(chi)^2 / chi^2
 * of `which` of halo epoch `epoch` (moved to the kernel's z_bar by the caller, :460) with
 * the reference's limits, normalisation and Romberg tolerances, and its spline.  D_z is
 * MultiEpoch.growth_factor(z_bar) (:464).  ln_K / proj / levels (each [n = kernel_npoints],
 * host, may be NULL) receive the knots, the table and the Romberg levels reached.
 *
 * chomp_covariance_gaussian replaces Covariance.covariance_G (covariance.py:361-453) for n
 * pairs of bin centres: theta holds theta_a[n] then theta_b[n] (radians).  j0_limit is
 * Covariance._j0_limit (:188-189), area the survey area in steradians, poisson_a /
 * poisson_b the shot-noise terms proj_power_poisson(0) / (2) of the integrand (:427-432). */
int chomp_covariance_table(chomp_ctx* ctx, int which, size_t epoch, double D_z,
                           double* ln_K, double* proj, double* levels, size_t n);
int chomp_covariance_gaussian(chomp_ctx* ctx, double j0_limit, double area,
                              double poisson_a, double poisson_b,
                              const double* theta, size_t n, double* out, int mem);

/* CorrelationFourier.correlation(l) (correlation.py:360-392): Limber C_l. */
int chomp_cell(chomp_ctx* ctx, int which, size_t epoch, double D_z,
               const double* ell, size_t n, double* out, int mem);

/* Both observables of one survey set-up -- Correlation.correlation(theta) (correlation.py:
 * 242-275) and CorrelationFourier.correlation(l) (correlation.py:360-392) on the same kernel,
 * halo and spectrum -- in one call: the results of chomp_wtheta and chomp_cell, bit for bit,
 * with C_l computed beside w(theta) on a second HIP stream when the buffers are device
 * memory (neither integral needs anything of the other).  In the order of the context's
 * stream the call is complete when it returns. */
int chomp_wtheta_cell(chomp_ctx* ctx, int which, size_t epoch, double k_min, double k_max,
                      double D_z, const double* theta, size_t n_theta, double* w_out,
                      const double* ell, size_t n_ell, double* c_out, int mem);

/* SingleEpoch(..., with_bao=...) (cosmology.py:39, 87, 556-572): which Eisenstein & Hu
 * transfer function the context's epochs use -- the no-wiggle fit (default,
 * cosmology.py:449-472) or the one with baryon wiggles (cosmology.py:474-538).  Call it
 * before chomp_epochs_set; changing it invalidates every table. */
#define CHOMP_TRANSFER_EH 0
#define CHOMP_TRANSFER_EH_BAO 1
int chomp_set_transfer(chomp_ctx* ctx, int kind);

/* Halo.calculate_bias / calculate_m_eff / calculate_f_sat (halo.py:709-838) of epochs
 * [epoch0, epoch0 + n): out[3 i + {0, 1, 2}] = effective bias, effective halo mass,
 * satellite fraction (host buffer).  Needs chomp_halo_setup (n_bar). */
int chomp_hod_stats(chomp_ctx* ctx, size_t epoch0, size_t n, double* out);

/* Correlation3d.raw_correlation(r) (correlation.py:470-499): xi(r) = int dlnk k^2/(2 pi)
 * P(k) J0(k r) over [k_min, k_max] -- the cylindrical J0, as the reference has it.  One
 * wavefront-group per r; P is `which` of halo epoch `epoch`. */
int chomp_xi3d(chomp_ctx* ctx, int which, size_t epoch, double k_min, double k_max,
               const double* r, size_t n, double* out, int mem);
/* scipy InterpolatedUnivariateSpline(xk, yk)(x) (k = 3, not-a-knot), host buffers: the
 * 50-knot xi(r) spline of Correlation3d.compute_correlation / correlation
 * (correlation.py:459-468, 501-510).  x outside [xk[0], xk[nk-1]] extrapolates the end
 * pieces, as FITPACK does.  deriv = 0: values; 1: first derivative (the
 * InterpolatedUnivariateSpline.derivatives(x)[1] of MassFunction.dndm,
 * mass_function.py:268-287). */
int chomp_spline_eval(chomp_ctx* ctx, const double* xk, const double* yk, size_t nk,
                      const double* x, size_t n, int deriv, double* out);

/* Arithmetic of the w(theta) integral (BASELINE.json configs[4]: "mixed fp32/fp64 with
 * tolerance sweep").  The reference computes everything in fp64; F64 is the
 * default and the only mode held to the 1e-4 parity bar -- the others exist so that the
 * cost of each narrowing can be measured against the same golden vectors
 * (tests/test_gpu_projection.py::test_c5_precision_sweep). */
enum {
  CHOMP_PREC_F64 = 0,        /* tables, evaluation and sums in fp64 */
  CHOMP_PREC_F32_EVAL = 1,   /* fp32 integrand evaluation, fp64 tables and sums */
  CHOMP_PREC_F32_TABLES = 2, /* spline coefficients rounded to fp32, fp64 evaluation and sums */
  CHOMP_PREC_F32_ALL = 3     /* fp32 tables, evaluation, sums and Richardson extrapolation */
};
int chomp_set_precision(chomp_ctx* ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif /* CHOMP_MI355X_H */
