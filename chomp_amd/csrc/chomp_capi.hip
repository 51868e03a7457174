// chomp_capi.hip -- C ABI (include/chomp_mi355x.h) over the HIP kernels.
// Host side only orchestrates: parameter upload, kernel launches on the context's
// stream, staging of host buffers.  No numerical work of the hot path runs on the
// CPU (the only host arithmetic is parameter preparation: the 1.05-step candidate
// mass sequence, erfinv for HODZheng.first_moment_zero, the Tinker parameter
// splines of a fixed 9-row table).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/chomp_mi355x.h"
#include "chomp_power_kernels.h"
#include "chomp_proj_kernels.h"
#include "chomp_probe_kernel.h"

using namespace chomp;

// A parameter block of the epoch batch on its way to the device: the host shadow says what
// the device holds (an unchanged block is not uploaded again -- MCMC-style loops re-run the
// set-up with mostly identical inputs); a changed block goes through one of two pinned
// staging buffers, each guarded by an event, so the copy is a true asynchronous DMA and the
// host never drains the stream (SimulationDesign / MCMC loops change a block on every call).
struct StagedBlock {
  std::vector<char> shadow;
  void* pin[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  bool used[2] = {false, false};
  size_t cap = 0;
  int turn = 0;
  void reset() { shadow.clear(); }
  void release() {
    for (int i = 0; i < 2; ++i) {
      if (pin[i]) (void)hipHostFree(pin[i]);
      if (done[i]) (void)hipEventDestroy(done[i]);
      pin[i] = nullptr; done[i] = nullptr; used[i] = false;
    }
    cap = 0;
  }
};

// roctx ranges around the stages (chomp_set_tuning CHOMP_TUNE_ROCTX; rocprofv3 --marker-trace):
// the marker library is looked up at run time, so nothing links against a profiler.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  bool load() {
    if (push) return true;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push && pop) return true;
      push = nullptr; pop = nullptr;
    }
    return false;
  }
};

struct chomp_ctx {
  chomp_config cfg;
  RoctxApi roctx;
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  TabLayout L;
  std::string err;

  // constant tables
  SiCiTab* d_sici = nullptr;
  BesselTab* d_j0 = nullptr;
  BesselTab* d_j2 = nullptr;
  TinkerTab* d_tinker = nullptr;
  double* d_gl16 = nullptr;
  double* d_cand = nullptr;

  // epoch batch
  size_t n_epoch = 0, cap_epoch = 0;
  chomp_cosmo* d_cosmo = nullptr;
  double* d_z = nullptr;
  Epoch* d_epochs = nullptr;
  double* d_search = nullptr;
  double* d_probe = nullptr;       // k_epoch_init: certifying probes of the mass-limit search
  int* d_count = nullptr;          // k_epoch_init: arrivals per epoch (reset by the kernel)
  int* d_pending = nullptr;        // work list of k_halo_knots_deep (cleared by k_halo_finalize)
  double* d_tab = nullptr;
  chomp_halo_par* d_mass_par = nullptr;
  chomp_halo_par* d_profile = nullptr;
  HodDev* d_hod = nullptr;
  double* d_nodes = nullptr;       // node tables of the halo integrals
  double* d_snodes = nullptr;      // node tables of the sigma(R) integrals (per cosmology slot)
  int* d_slot = nullptr;           // epoch -> cosmology slot
  int* d_first = nullptr;          // slot -> an epoch with that cosmology
  unsigned* d_status = nullptr;    // per-epoch status word (chomp_get_status)
  double* d_endp = nullptr;        // integrand pairs of the knots at the upper end point
  int* d_npend = nullptr;          // per epoch: listed knots + 1 token (k_halo_knots_fast)
  long long tune[CHOMP_TUNE_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};   // chomp_set_tuning
  bool have_epochs = false, have_mass = false, have_halo = false;
  unsigned fam_mask = 0;          // families (F_* bits) with valid splines
  std::vector<char> have_halofit;
  StagedBlock sh_cosmo, sh_z, sh_mass, sh_profile, sh_hod, sh_slot, sh_first;
  StagedBlock sh_proj, sh_pp[2];   // projection scalars, tabulated redshift distributions
  // pinned host mirrors of the staging buffers of host-pointer calls (chomp_power)
  unsigned* h_status = nullptr;    // chomp_status_post: pinned copy of d_status ...
  size_t cap_hstatus = 0, n_hstatus = 0;
  hipEvent_t ev_status = nullptr;  // ... complete when this event is
  double* h_stage_in = nullptr;
  double* h_stage_out = nullptr;
  size_t cap_hin = 0, cap_hout = 0;
  // Buffers a captured HIP graph may still reference are never freed before the context
  // is destroyed (graph_seen: a call of this context has been captured at least once).
  bool graph_seen = false;
  std::vector<void*> graveyard;
  std::vector<void*> host_graveyard;   // (pinned words a graph's kernels or copies may write)

  // staging for host-pointer calls
  double* d_stage_in = nullptr;
  double* d_stage_in2 = nullptr;
  double* d_kcache = nullptr;      // device copy of the last host k grid of chomp_power
  const double* kcache_ptr = nullptr;   // (== d_kcache while kcache_shadow describes its contents)
  size_t cap_kcache = 0;
  std::vector<double> kcache_shadow;
  double* d_stage_out = nullptr;
  int* d_slow = nullptr;           // Stage E: 2 counters + list of k groups for the per-lane pass
  int* d_winfo = nullptr;          // Stage E: per k group knot interval / flags (k_power_prep)
  double* d_ktab = nullptr;        // Stage E: per-k (offset, shape) table (k_power_prep)
  double* d_wnodes = nullptr;      // w(theta): theta-independent integrand factor on the Romberg nodes
  double* d_cnodes = nullptr;      // C_l: chi-only factors on the Romberg nodes
  size_t cap_cnodes = 0;
  double* d_samples = nullptr;     // coarse samples of the listed knots, a slot each (k_halo_knots_samples -> _fast)
  double* d_psum = nullptr;        // ... and their per-level sums
  size_t cap_samples = 0, cap_psum = 0, cap_plan = 0;
  char* d_plan = nullptr;          // per (epoch, group): the break-point plan (DeepPlan; k_halo_knots -> _fast)
  double* d_deepw = nullptr;       // k_halo_knots_fast: level weights (deep_weights_host)
  double* d_hf_ainv = nullptr;     // k_halofit_finalize: inverse collocation matrix of the ln R grid
  int* d_deepstat = nullptr;       // k_halo_knots_fast: knots done by the fast / literal path
  size_t cap_slow = 0, cap_winfo = 0, cap_ktab = 0, cap_wnodes = 0;
  // chomp_power_plan: the k-only table of a registered k grid, kept across chomp_power calls
  struct PowerPlan {
    bool valid = false;
    const double* k = nullptr;
    size_t nk = 0;
    chomp_cosmo cosmo;             // the cosmology the shape column was computed for
    int bao = 0, parity = 0, n_slow = -1;
  } plan;
  int slow_parity = 0;
  // opt-ins for more than 64 KiB of dynamic LDS, done once per context (hipFuncSetAttribute
  // applies to the current device: a process-wide flag would skip a second device)
  bool lds_knots_set = false, lds_lns_set = false;
  unsigned lds_cell_mask = 0;      // k_cell_deep<HF, BAO>: bit 2 HF + BAO
  bool slow_by_memset = false;     // set once a Stage E call has been captured into a HIP graph
  int precision = CHOMP_PREC_F64;  // chomp_set_precision
  int with_bao = 0;                // chomp_set_transfer
  int timing = 0;                  // chomp_set_timing: HIP events around the Stage E launches
  bool timing_valid = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<int> slot;           // host copy: epoch -> cosmology slot
  size_t n_slots = 1;              // distinct cosmologies of the batch
  size_t cap_in = 0, cap_in2 = 0, cap_out = 0;

  // projection
  ProjState proj;
  // The side stream.  The projection set-up (chomp_kernel_setup / chomp_multi_epoch_setup:
  // a chain of six small launches) depends on nothing the halo set-up produces, and C_l on
  // nothing w(theta) produces: they run here, beside the context's stream, ordered against it
  // by events -- behind everything queued on `stream` when they start, and joined back before
  // anything reads what they wrote (proj_pending: lazily, at the next call that looks at the
  // projection tables).  Under stream capture the fork and the join are captured with the rest
  // (the side stream joins the capture through ev_side_go and is joined back before the
  // capture ends: every call that forks also joins, the lazy projection join at the first
  // reader); what must not happen is a join, inside a capture, of side work queued before it.
  hipStream_t side = nullptr;
  hipEvent_t ev_side_go = nullptr, ev_proj_ready = nullptr, ev_side_done = nullptr;
  bool proj_pending = false;
  bool proj_pending_captured = false;   // ... and it was queued inside the capture in progress
  bool status_mirrored = false;    // the last set-up ended with a halo set-up: its finalising blocks
                                   // wrote the status words to the pinned host words themselves
  // (see ensure_hstatus / mirror_next / chomp_status_post)
  unsigned long long* h_mirror = nullptr;
  unsigned setup_seq = 0, posted_seq = 0;
  int mirror_half = 0, mirror_region = 0, posted_half = 0;
  int posted_kind = 0;             // 0 none, 1 the mirror, 2 the copy, 3 collected into posted_saved
  bool posted_in_capture = false;
  std::vector<unsigned> posted_saved;
};

namespace {

// Scope of one stage on the host timeline (a no-op unless ranges are switched on).
struct StageRange {
  chomp_ctx* ctx;
  StageRange(chomp_ctx* c, const char* name) : ctx(c && c->roctx.push ? c : nullptr) {
    if (ctx) ctx->roctx.push(name);
  }
  ~StageRange() { if (ctx) ctx->roctx.pop(); }
};

int fail(chomp_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

#define HIPCHK(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess)                                                         \
      return fail(ctx, CHOMP_ERR_HIP,                                             \
                  std::string(#call) + ": " + hipGetErrorString(e_));             \
  } while (0)

bool capturing(chomp_ctx* ctx) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess) return false;
  if (st == hipStreamCaptureStatusActive) ctx->graph_seen = true;
  return st == hipStreamCaptureStatusActive;
}

// Free a device buffer, unless a captured graph may still hold its address.
int release_device(chomp_ctx* ctx, void* p) {
  if (!p) return CHOMP_OK;
  if (ctx->graph_seen) { ctx->graveyard.push_back(p); return CHOMP_OK; }
  HIPCHK(hipFree(p));
  return CHOMP_OK;
}

// Upload `bytes` from `src` to `dst` unless the shadow says the device already holds
// exactly these bytes.  No host synchronisation with the stream (see StagedBlock).
// Under stream capture a changed block is an error: the graph would bake in the staging
// buffer's contents of capture time (include/chomp_mi355x.h, "HIP graphs").
int upload(chomp_ctx* ctx, void* dst, const void* src, size_t bytes, StagedBlock& b) {
  if (b.shadow.size() == bytes && std::memcmp(b.shadow.data(), src, bytes) == 0) return CHOMP_OK;
  if (capturing(ctx))
    return fail(ctx, CHOMP_ERR_STATE,
                "a parameter block changed while the context's stream is being captured: run the "
                "set-up once with these parameters before capturing (HIP graphs replay the "
                "captured parameters)");
  if (bytes > b.cap) {
    // (buffers still in flight are drained by their events first)
    for (int i = 0; i < 2; ++i)
      if (b.used[i]) HIPCHK(hipEventSynchronize(b.done[i]));
    const size_t cap = bytes + bytes / 2;
    for (int i = 0; i < 2; ++i) {
      if (b.pin[i]) HIPCHK(hipHostFree(b.pin[i]));
      b.pin[i] = nullptr;
      HIPCHK(hipHostMalloc(&b.pin[i], cap, hipHostMallocDefault));
      if (!b.done[i]) HIPCHK(hipEventCreateWithFlags(&b.done[i], hipEventDisableTiming));
      b.used[i] = false;
    }
    b.cap = cap;
  }
  const int t = b.turn;
  if (b.used[t]) HIPCHK(hipEventSynchronize(b.done[t]));   // the copy of two uploads ago: long done
  std::memcpy(b.pin[t], src, bytes);
  HIPCHK(hipMemcpyAsync(dst, b.pin[t], bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(b.done[t], ctx->stream));
  b.used[t] = true;
  b.turn = t ^ 1;
  b.shadow.assign(static_cast<const char*>(src), static_cast<const char*>(src) + bytes);
  return CHOMP_OK;
}

// -- the side stream ------------------------------------------------------------------
int side_create(chomp_ctx* ctx) {
  if (ctx->side) return CHOMP_OK;
  HIPCHK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
  for (hipEvent_t* e : {&ctx->ev_side_go, &ctx->ev_proj_ready, &ctx->ev_side_done})
    HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  return CHOMP_OK;
}

// While alive, everything the context queues goes to the side stream, which first waits for
// what `stream` holds now; on destruction `done` (if any) is recorded there and the context's
// stream restored.  on() is false (nothing redirected) under stream capture or on failure.
struct SideScope {
  chomp_ctx* ctx;
  hipStream_t saved = nullptr;
  hipEvent_t* done;                // (a member of *ctx: the events are created by begin())
  bool active = false;
  bool marks_proj;                 // the work is a projection set-up: proj_pending on EVERY way out
  SideScope(chomp_ctx* c, hipEvent_t* done_, bool marks_proj_ = false)
      : ctx(c), done(done_), marks_proj(marks_proj_) {}
  int begin() {
    // (the side stream and its events exist before any capture: an eager call of the same kind
    //  comes first -- buffers and parameters have to be in place for a capture anyway)
    if (capturing(ctx) && !ctx->side)
      return fail(ctx, CHOMP_ERR_STATE,
                  "the first call that runs work beside the context's stream came while the stream "
                  "is being captured: run the step once eagerly before capturing it");
    const int rc = side_create(ctx);
    if (rc) return rc;
    HIPCHK(hipEventRecord(ctx->ev_side_go, ctx->stream));
    HIPCHK(hipStreamWaitEvent(ctx->side, ctx->ev_side_go, 0));
    saved = ctx->stream;
    ctx->stream = ctx->side;
    active = true;
    return CHOMP_OK;
  }
  bool on() const { return active; }
  void end() {
    if (!active) return;
    (void)hipEventRecord(*done, ctx->side);
    if (marks_proj) {              // (also on an error return: what was queued is still in flight)
      ctx->proj_pending = true;
      ctx->proj_pending_captured = capturing(ctx);
    }
    ctx->stream = saved;
    active = false;
  }
  ~SideScope() { end(); }
};

// Before anything on the context's stream reads (or rewrites) the projection tables: wait for
// a projection set-up still in flight on the side stream.
int proj_join(chomp_ctx* ctx) {
  if (!ctx->proj_pending) return CHOMP_OK;
  if (capturing(ctx) && !ctx->proj_pending_captured)
    return fail(ctx, CHOMP_ERR_STATE,
                "a projection set-up queued BEFORE the capture is still in flight beside the "
                "stream being captured: make any projection call (or chomp_sync) before the "
                "capture begins");
  HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_proj_ready, 0));   // (captured under capture)
  ctx->proj_pending = false;
  ctx->proj_pending_captured = false;
  return CHOMP_OK;
}

template <class T>
int ensure(chomp_ctx* ctx, T** p, size_t* cap, size_t n) {
  if (n <= *cap && *p) return CHOMP_OK;
  if (capturing(ctx))
    return fail(ctx, CHOMP_ERR_STATE,
                "a work buffer would have to grow while the context's stream is being captured: "
                "run the call once at this size before capturing");
  if (*p) { const int rc = release_device(ctx, *p); if (rc) return rc; }
  *p = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  *cap = n;
  return CHOMP_OK;
}

// Pinned host mirror of a staging buffer (host-pointer calls).
int ensure_host(chomp_ctx* ctx, double** p, size_t* cap, size_t n) {
  if (n <= *cap && *p) return CHOMP_OK;
  if (*p) HIPCHK(hipHostFree(*p));
  *p = nullptr;
  HIPCHK(hipHostMalloc(reinterpret_cast<void**>(p), n * sizeof(double), hipHostMallocDefault));
  *cap = n;
  return CHOMP_OK;
}

// erfinv(y) for y in (-1, 1): Newton on erf/erfc (used once per HOD, hod.py:172-175).
double erfinv_host(double y) {
  if (y <= -1.0) return -INFINITY;
  if (y >= 1.0) return INFINITY;
  if (y == 0.0) return 0.0;
  const double a = 0.147;
  const double ln1 = std::log1p(-y * y);
  const double t = 2.0 / (M_PI * a) + 0.5 * ln1;
  double x = std::copysign(std::sqrt(std::sqrt(t * t - ln1 / a) - t), y);
  for (int it = 0; it < 60; ++it) {
    // residual computed in the tail-accurate form
    double r;
    if (y < -0.5) r = std::erfc(-x) - (1.0 + y);
    else if (y > 0.5) r = (1.0 - y) - std::erfc(x);
    else r = std::erf(x) - y;
    const double d = 2.0 / std::sqrt(M_PI) * std::exp(-x * x);
    const double dx = r / d;
    x -= dx;
    if (std::fabs(dx) <= 1e-16 * std::fabs(x)) break;
  }
  return x;
}

int setup_constants(chomp_ctx* ctx) {
  SiCiTab s;
  BesselTab j0, j2;
  fill_tables(&s, &j0, &j2);
  HIPCHK(hipMalloc(&ctx->d_sici, sizeof(SiCiTab)));
  HIPCHK(hipMalloc(&ctx->d_j0, sizeof(BesselTab)));
  HIPCHK(hipMalloc(&ctx->d_j2, sizeof(BesselTab)));
  HIPCHK(hipMemcpy(ctx->d_sici, &s, sizeof(s), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_j0, &j0, sizeof(j0), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_j2, &j2, sizeof(j2), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&ctx->d_gl16, 32 * sizeof(double)));
  HIPCHK(hipMemcpy(ctx->d_gl16, CHOMP_GL16, 32 * sizeof(double), hipMemcpyHostToDevice));
  // Tinker et al. 2010 parameter table (mass_function.py:450-459), splined in
  // ln(Delta) exactly as the reference does (:461-470).
  static const double delta[9] = {200, 300, 400, 600, 800, 1200, 1600, 2400, 3200};
  static const double par[5][9] = {
      {0.368, 0.363, 0.385, 0.389, 0.393, 0.365, 0.379, 0.355, 0.327},       // alpha
      {0.589, 0.585, 0.544, 0.543, 0.564, 0.632, 0.637, 0.673, 0.702},       // beta
      {0.864, 0.922, 0.987, 1.09, 1.20, 1.34, 1.50, 1.68, 1.81},             // gamma
      {-0.729, -0.789, -0.910, -1.05, -1.20, -1.26, -1.45, -1.50, -1.49},    // phi
      {-0.243, -0.261, -0.261, -0.273, -0.278, -0.301, -0.301, -0.319, -0.336}};  // eta
  TinkerTab tt;
  double work[18];
  for (int i = 0; i < 9; ++i) tt.x[i] = std::log(delta[i]);
  for (int q = 0; q < 5; ++q) spline_build(tt.x, par[q], 9, tt.c[q], work);
  HIPCHK(hipMalloc(&ctx->d_tinker, sizeof(TinkerTab)));
  HIPCHK(hipMemcpy(ctx->d_tinker, &tt, sizeof(tt), hipMemcpyHostToDevice));
  // Candidate masses of the 5 % walk (mass_function.py:161-193), generated by the
  // same repeated multiply/divide so every candidate is bit-identical to the value
  // the reference's loop would hold after j steps.
  std::vector<double> cand(4 * kSearchJ);
  const double start[4] = {1.0e9, 1.0e9, 1.0e16, 1.0e16};
  const bool mul[4] = {false, true, true, false};
  for (int t = 0; t < 4; ++t) {
    double m = start[t];
    for (int j = 0; j < kSearchJ; ++j) {
      cand[t * kSearchJ + j] = m;
      m = mul[t] ? m * 1.05 : m / 1.05;
    }
  }
  {   // level weights of the fast deep-knot sums
    const int top = ctx->cfg.divmax;
    std::vector<double> w((size_t)(top > kDeepCoarse ? top - kDeepCoarse : 1) * kDeepWStride, 0.0);
    if (top > kDeepCoarse) deep_weights_host(kDeepCoarse, top, w.data());
    HIPCHK(hipMalloc(&ctx->d_deepw, w.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_deepw, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&ctx->d_deepstat, 8 * sizeof(int)));
    HIPCHK(hipMemset(ctx->d_deepstat, 0, 8 * sizeof(int)));
  }
  if (ctx->cfg.halo_npoints >= 12) {   // (fewer points: the serial quintic solve of k_halofit_finalize)
    const int nk = ctx->cfg.halo_npoints;
    std::vector<double> ai((size_t)nk * nk);
    halofit_collocation_inverse_host(nk, ai.data());
    HIPCHK(hipMalloc(&ctx->d_hf_ainv, ai.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_hf_ainv, ai.data(), ai.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  HIPCHK(hipMalloc(&ctx->d_cand, cand.size() * sizeof(double)));
  HIPCHK(hipMemcpy(ctx->d_cand, cand.data(), cand.size() * sizeof(double),
                   hipMemcpyHostToDevice));
  return CHOMP_OK;
}

int alloc_epochs(chomp_ctx* ctx, size_t n) {
  if (n <= ctx->cap_epoch) return CHOMP_OK;
  if (capturing(ctx))
    return fail(ctx, CHOMP_ERR_STATE,
                "the epoch batch would have to grow while the context's stream is being captured");
  void* old[] = {ctx->d_cosmo, ctx->d_z, ctx->d_epochs, ctx->d_search, ctx->d_probe, ctx->d_count, ctx->d_pending, ctx->d_tab,
                 ctx->d_mass_par, ctx->d_profile, ctx->d_hod, ctx->d_nodes, ctx->d_snodes,
                 ctx->d_slot, ctx->d_first, ctx->d_status, ctx->d_endp, ctx->d_npend};
  for (void* p : old) {
    const int rc = release_device(ctx, p);
    if (rc) return rc;
  }
  HIPCHK(hipMalloc(&ctx->d_cosmo, n * sizeof(chomp_cosmo)));
  HIPCHK(hipMalloc(&ctx->d_z, n * sizeof(double)));
  HIPCHK(hipMalloc(&ctx->d_epochs, n * sizeof(Epoch)));
  HIPCHK(hipMalloc(&ctx->d_search, n * 4 * sizeof(double)));
  HIPCHK(hipMalloc(&ctx->d_probe, n * kProbeStride * sizeof(double)));
  HIPCHK(hipMalloc(&ctx->d_count, n * sizeof(int)));
  HIPCHK(hipMemsetAsync(ctx->d_count, 0, n * sizeof(int), ctx->stream));
  HIPCHK(hipMalloc(&ctx->d_pending, pending_ints(n, ctx->L.NK) * sizeof(int)));
  HIPCHK(hipMemsetAsync(ctx->d_pending, 0, kPendingHead * sizeof(int), ctx->stream));
  HIPCHK(hipMalloc(&ctx->d_endp, n * 3 * 2 * (size_t)ctx->L.NK * sizeof(double)));
  HIPCHK(hipMalloc(&ctx->d_npend, n * sizeof(int)));
  HIPCHK(hipMemsetAsync(ctx->d_npend, 0, n * sizeof(int), ctx->stream));
  HIPCHK(hipMalloc(&ctx->d_tab, n * (size_t)ctx->L.stride * sizeof(double)));
  HIPCHK(hipMalloc(&ctx->d_mass_par, n * sizeof(chomp_halo_par)));
  HIPCHK(hipMalloc(&ctx->d_profile, n * sizeof(chomp_halo_par)));
  HIPCHK(hipMalloc(&ctx->d_hod, n * sizeof(HodDev)));
  HIPCHK(hipMalloc(&ctx->d_nodes, n * 3 * (size_t)kNodeStride * sizeof(double)));
  HIPCHK(hipMalloc(&ctx->d_snodes, n * (size_t)kSigmaStride * sizeof(double)));
  // (holds the arrival counters of k_sigma_nodes: zero once, the kernel resets them)
  HIPCHK(hipMemsetAsync(ctx->d_snodes, 0, n * (size_t)kSigmaStride * sizeof(double), ctx->stream));
  HIPCHK(hipMalloc(&ctx->d_slot, n * sizeof(int)));
  HIPCHK(hipMalloc(&ctx->d_first, n * sizeof(int)));
  HIPCHK(hipMalloc(&ctx->d_status, n * sizeof(unsigned)));
  ctx->cap_epoch = n;
  ctx->sh_cosmo.reset(); ctx->sh_z.reset(); ctx->sh_mass.reset();
  ctx->sh_profile.reset(); ctx->sh_hod.reset(); ctx->sh_slot.reset(); ctx->sh_first.reset();
  return CHOMP_OK;
}

}  // namespace

extern "C" {

void chomp_default_config(chomp_config* c) {
  // defaults.py:42-51 and 62-92
  c->k_min = 0.001; c->k_max = 100.0; c->mass_min = -1; c->mass_max = -1;
  c->corr_precision = 1.48e-6; c->cosmo_precision = 1.48e-8; c->dNdz_precision = 1.48e-8;
  c->halo_precision = 1.48e-5; c->kernel_precision = 1.48e-6; c->mass_precision = 1.48e-8;
  c->window_precision = 1.48e-6; c->global_precision = 1.48e-32;
  c->corr_npoints = 50; c->cosmo_npoints = 50; c->halo_npoints = 50;
  c->kernel_npoints = 50; c->kernel_bessel_limit = 8; c->mass_npoints = 50;
  c->window_npoints = 100; c->divmax = 20;
}

int chomp_ctx_create(const chomp_config* cfg, int device, void* hip_stream,
                     chomp_ctx** out) {
  if (!out) return CHOMP_ERR_ARG;
  *out = nullptr;
  chomp_ctx* ctx = new chomp_ctx();
  if (cfg) ctx->cfg = *cfg; else chomp_default_config(&ctx->cfg);
  const chomp_config& c = ctx->cfg;
  auto bad = [&](const char* m) {
    delete ctx;
    fprintf(stderr, "chomp_ctx_create: %s\n", m);
    return CHOMP_ERR_ARG;
  };
  if (c.mass_npoints < 8 || c.mass_npoints > 256) return bad("mass_npoints out of [8,256]");
  if (c.halo_npoints < 6 || c.halo_npoints > 256) return bad("halo_npoints out of [6,256]");
  if (c.kernel_npoints < 4 || c.kernel_npoints > 512) return bad("kernel_npoints");
  if (c.window_npoints < 4 || c.window_npoints > 1024) return bad("window_npoints");
  if (c.cosmo_npoints < 4 || c.cosmo_npoints > 512) return bad("cosmo_npoints");
  if (c.divmax < 1 || c.divmax > 30) return bad("divmax out of [1,30]");
  if (!(c.k_min > 0.0) || !(c.k_max > c.k_min)) return bad("k limits");
  ctx->device = device;
  ctx->L = make_layout(c.mass_npoints, c.halo_npoints);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    delete ctx;
    fprintf(stderr, "chomp_ctx_create: no usable HIP device %d (%s) -- this library has "
                    "no CPU fallback\n", device, hipGetErrorString(e));
    return CHOMP_ERR_HIP;
  }
  if (hipSetDevice(device) != hipSuccess) { delete ctx; return CHOMP_ERR_HIP; }
  if (hip_stream) {
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  } else {
    if (hipStreamCreate(&ctx->stream) != hipSuccess) { delete ctx; return CHOMP_ERR_HIP; }
    ctx->owns_stream = true;
  }
  int rc = setup_constants(ctx);
  if (rc != CHOMP_OK) {
    fprintf(stderr, "chomp_ctx_create: %s\n", ctx->err.c_str());
    delete ctx;
    return rc;
  }
  *out = ctx;
  return CHOMP_OK;
}

void chomp_ctx_destroy(chomp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->side) (void)hipStreamSynchronize(ctx->side);
  (void)hipStreamSynchronize(ctx->stream);
  void* ptrs[] = {ctx->d_sici, ctx->d_j0, ctx->d_j2, ctx->d_tinker, ctx->d_gl16,
                  ctx->d_cand, ctx->d_cosmo, ctx->d_z, ctx->d_epochs, ctx->d_search, ctx->d_probe, ctx->d_count, ctx->d_pending,
                  ctx->d_tab, ctx->d_mass_par, ctx->d_profile, ctx->d_hod, ctx->d_nodes, ctx->d_snodes, ctx->d_slot, ctx->d_first, ctx->d_status, ctx->d_endp, ctx->d_npend,
                  ctx->d_stage_in, ctx->d_stage_in2, ctx->d_kcache, ctx->d_stage_out, ctx->d_slow, ctx->d_wnodes, ctx->d_cnodes, ctx->d_deepw, ctx->d_deepstat,
                  ctx->d_winfo, ctx->d_ktab, ctx->d_samples, ctx->d_psum, ctx->d_plan, ctx->d_hf_ainv};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (void* p : ctx->graveyard) (void)hipFree(p);
  for (void* p : ctx->host_graveyard) (void)hipHostFree(p);
  for (StagedBlock* b : {&ctx->sh_cosmo, &ctx->sh_z, &ctx->sh_mass, &ctx->sh_profile, &ctx->sh_hod,
                         &ctx->sh_slot, &ctx->sh_first, &ctx->sh_proj, &ctx->sh_pp[0],
                         &ctx->sh_pp[1]})
    b->release();
  if (ctx->h_status) (void)hipHostFree(ctx->h_status);
  if (ctx->h_mirror) (void)hipHostFree(ctx->h_mirror);
  if (ctx->ev_status) (void)hipEventDestroy(ctx->ev_status);
  if (ctx->h_stage_in) (void)hipHostFree(ctx->h_stage_in);
  if (ctx->h_stage_out) (void)hipHostFree(ctx->h_stage_out);
  proj_free(ctx->proj);
  for (hipEvent_t e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : {ctx->ev_side_go, ctx->ev_proj_ready, ctx->ev_side_done})
    if (e) (void)hipEventDestroy(e);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* chomp_last_error(chomp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int chomp_set_timing(chomp_ctx* ctx, int on) {
  if (!ctx) return CHOMP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  if (on)
    for (hipEvent_t& e : ctx->ev)
      if (!e) HIPCHK(hipEventCreate(&e));
  ctx->timing = on ? 1 : 0;
  ctx->timing_valid = false;
  return CHOMP_OK;
}

int chomp_get_timing(chomp_ctx* ctx, double* us, size_t n) {
  if (!ctx || !us || n != 3) return fail(ctx, CHOMP_ERR_ARG, "get_timing: us[3]");
  if (!ctx->timing || !ctx->timing_valid)
    return fail(ctx, CHOMP_ERR_STATE,
                "get_timing: no timed streaming chomp_power call (chomp_set_timing, large grid)");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipEventSynchronize(ctx->ev[3]));
  for (int i = 0; i < 3; ++i) {
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
    us[i] = 1e3 * (double)ms;
  }
  return CHOMP_OK;
}

int chomp_get_stream(chomp_ctx* ctx, void** out) {
  if (!ctx || !out) return CHOMP_ERR_ARG;
  *out = reinterpret_cast<void*>(ctx->stream);
  return CHOMP_OK;
}

int chomp_sync(chomp_ctx* ctx) {
  if (!ctx) return CHOMP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  { const int rcj = proj_join(ctx); if (rcj) return rcj; }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return CHOMP_OK;
}

#ifdef CHOMP_STAMPS
// (development builds only: read / clear the stamps of k_halo_knots_fast)
int chomp_debug_ks(long long* out, int n, int clear) {
  if (clear) {
    static long long z[kStampBlocks * kStampSlots];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(chomp::g_ks), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chomp::g_ks), (size_t)n * sizeof(long long));
}
// (... and of k_mass_nodes)
int chomp_debug_ms(long long* out, int n, int clear) {
  if (clear) {
    static long long z[64 * 16 * kMStampSlots];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(chomp::g_ms), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chomp::g_ms), (size_t)n * sizeof(long long));
}
#endif

int chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {
  if (!ctx) return CHOMP_ERR_ARG;
  if (what < 0 || what >= CHOMP_TUNE_COUNT) return fail(ctx, CHOMP_ERR_ARG, "set_tuning: unknown knob");
  ctx->tune[what] = value < 0 ? -1 : value;
  if (what == CHOMP_TUNE_ROCTX) {
    if (value > 0) {
      if (!ctx->roctx.load()) return fail(ctx, CHOMP_ERR_STATE, "set_tuning: no roctx library found");
    } else {
      ctx->roctx.push = nullptr;
    }
  }
  return CHOMP_OK;
}

int chomp_get_deep_stats(chomp_ctx* ctx, long long* out) {
  if (!ctx || !out) return fail(ctx, CHOMP_ERR_ARG, "get_deep_stats: bad args");
  HIPCHK(hipSetDevice(ctx->device));
  int v[8] = {0};
  HIPCHK(hipMemcpyAsync(v, ctx->d_deepstat, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 5; ++i) out[i] = v[i];
  float worst;
  std::memcpy(&worst, &v[5], sizeof worst);
  out[5] = (long long)((double)worst * 1e15);      // largest self-check estimate, in 1e-15
  out[6] = v[6];
  return CHOMP_OK;
}

int chomp_get_status(chomp_ctx* ctx, size_t epoch0, size_t n, unsigned* out) {
  if (!ctx || !out || n == 0) return fail(ctx, CHOMP_ERR_ARG, "get_status: bad args");
  if (!ctx->have_epochs) return fail(ctx, CHOMP_ERR_STATE, "get_status before epochs_set");
  if (epoch0 + n > ctx->n_epoch) return fail(ctx, CHOMP_ERR_ARG, "get_status: epoch range");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(out, ctx->d_status + epoch0, n * sizeof(unsigned), hipMemcpyDeviceToHost,
                        ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return CHOMP_OK;
}

// The pinned host words of the status posts, large enough for the batch: h_mirror, three regions
// the finalising blocks of the halo set-ups write (sequence number << 32 | word) into in turn
// (known to the kernels through ctx->L), and h_status, where a post without a mirror is copied.
// Not under stream capture (an allocation): returns OK with the words as they are.
static int mirror_collect(chomp_ctx* ctx, int half, unsigned seq, size_t epoch0, size_t n,
                          unsigned* out);
static int ensure_hstatus(chomp_ctx* ctx) {
  if (ctx->n_epoch <= ctx->cap_hstatus && ctx->h_status && ctx->h_mirror) return CHOMP_OK;
  if (capturing(ctx)) return CHOMP_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));       // (a copy or a kernel's mirror may be in flight)
  // (a post nobody has waited for yet survives the reallocation: its words are there now)
  if (ctx->posted_kind == 1 || ctx->posted_kind == 2) {
    const size_t np = ctx->n_hstatus;
    std::vector<unsigned> keep(np);
    if (ctx->posted_kind == 1) {
      const int rcc = mirror_collect(ctx, ctx->posted_half, ctx->posted_seq, 0, np, keep.data());
      if (rcc) return rcc;
    } else {
      std::memcpy(keep.data(), ctx->h_status, np * sizeof(unsigned));
    }
    ctx->posted_saved.swap(keep);
    ctx->posted_kind = 3;
  }
  const int kept_kind = ctx->posted_kind;
  const size_t kept_n = ctx->n_hstatus;
  if (ctx->graph_seen) {
    if (ctx->h_status) ctx->host_graveyard.push_back(ctx->h_status);
    if (ctx->h_mirror) ctx->host_graveyard.push_back(ctx->h_mirror);
  } else {
    if (ctx->h_status) HIPCHK(hipHostFree(ctx->h_status));
    if (ctx->h_mirror) HIPCHK(hipHostFree(ctx->h_mirror));
  }
  ctx->h_status = nullptr;
  ctx->h_mirror = nullptr;
  ctx->L.h_status = nullptr;
  ctx->cap_hstatus = 0;
  ctx->posted_kind = kept_kind == 3 ? 3 : 0;
  ctx->n_hstatus = kept_kind == 3 ? kept_n : 0;
  HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_status), ctx->n_epoch * sizeof(unsigned),
                       hipHostMallocDefault));
  HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_mirror),
                       3 * ctx->n_epoch * sizeof(unsigned long long), hipHostMallocDefault));
  std::memset(ctx->h_mirror, 0, 3 * ctx->n_epoch * sizeof(unsigned long long));
  ctx->cap_hstatus = ctx->n_epoch;
  ctx->mirror_half = 0;
  ctx->mirror_region = 0;
  ctx->L.h_status = ctx->h_mirror;
  ctx->L.h_seq = 0u;
  return CHOMP_OK;
}

// The words of a mirrored post, once every one of them carries the post's sequence number
// (the set-up's finalising blocks have written them).  Spins on the pinned words; gives up with
// an error when the stream has drained and they still do not (a set-up that failed).
static int mirror_collect(chomp_ctx* ctx, int half, unsigned seq, size_t epoch0, size_t n,
                          unsigned* out) {
  const volatile unsigned long long* w = ctx->h_mirror + (size_t)half * ctx->cap_hstatus + epoch0;
  for (size_t i = 0; i < n; ++i) {
    bool drained = false;
    for (;;) {
      const unsigned long long v = w[i];
      if ((unsigned)(v >> 32) == seq) { out[i] = (unsigned)v; break; }
      if (drained)
        return fail(ctx, CHOMP_ERR_STATE, "status_wait: the posted set-up never finalised its epochs");
      if (hipStreamQuery(ctx->stream) == hipSuccess) drained = true;   // (one more look, then give up)
    }
  }
  return CHOMP_OK;
}

// In front of a halo set-up's kernels: the region its finalising blocks will mirror into and its
// sequence number.  Eager set-ups take regions 0 and 1 in turn; a post of the region about to be
// written again that nobody has waited for yet is collected first -- it is the set-up of two
// set-ups ago.  A set-up under stream capture gets region 2 and a number of its own, both fixed
// in the graph's kernel arguments: replays never write where an eager set-up's words are.
static int mirror_next(chomp_ctx* ctx) {
  if (!ctx->h_mirror) return CHOMP_OK;
  int next = 2;
  if (!capturing(ctx)) {
    next = ctx->mirror_half ^ 1;
    ctx->mirror_half = next;
  }
  if (ctx->posted_kind == 1 && ctx->posted_half == next && !capturing(ctx)) {
    ctx->posted_saved.resize(ctx->n_hstatus);
    const int rc = mirror_collect(ctx, next, ctx->posted_seq, 0, ctx->n_hstatus, ctx->posted_saved.data());
    if (rc) return rc;
    ctx->posted_kind = 3;
  }
  ctx->mirror_region = next;
  ctx->L.h_status = ctx->h_mirror + (size_t)next * ctx->cap_hstatus;
  ctx->L.h_seq = ++ctx->setup_seq;
  return CHOMP_OK;
}

int chomp_status_post(chomp_ctx* ctx) {
  if (!ctx) return CHOMP_ERR_ARG;
  if (!ctx->have_epochs) return fail(ctx, CHOMP_ERR_STATE, "status_post before epochs_set");
  HIPCHK(hipSetDevice(ctx->device));
  const bool cap = capturing(ctx);
  if (cap && (ctx->n_epoch > ctx->cap_hstatus || !ctx->h_status))
    return fail(ctx, CHOMP_ERR_STATE,
                "status_post during stream capture before any eager set-up of this size (the "
                "pinned words would have to be allocated)");
  { const int rce = ensure_hstatus(ctx); if (rce) return rce; }
  if (ctx->status_mirrored) {
    // Behind a halo set-up the words are on their way already (its finalising blocks write them,
    // with the set-up's sequence number): the post puts nothing on the stream -- it names the
    // region and the number to wait for.
    ctx->posted_kind = 1;
    ctx->posted_half = ctx->mirror_region;
    ctx->posted_seq = ctx->L.h_seq;
  } else {
    if (!ctx->ev_status) {
      if (cap) return fail(ctx, CHOMP_ERR_STATE, "status_post during stream capture before any eager post");
      HIPCHK(hipEventCreateWithFlags(&ctx->ev_status, hipEventDisableTiming));
    }
    if (ctx->posted_kind == 2 && !cap && !ctx->posted_in_capture)
      HIPCHK(hipEventSynchronize(ctx->ev_status));       // (the previous copy into the same words)
    HIPCHK(hipMemcpyAsync(ctx->h_status, ctx->d_status, ctx->n_epoch * sizeof(unsigned),
                          hipMemcpyDeviceToHost, ctx->stream));
    // (under capture the copy is a node of the graph, and whoever looks at the words waits for
    //  the stream the graph was launched on, not for an event)
    if (!cap) HIPCHK(hipEventRecord(ctx->ev_status, ctx->stream));
    ctx->posted_kind = 2;
  }
  ctx->posted_in_capture = cap;
  ctx->n_hstatus = ctx->n_epoch;
  return CHOMP_OK;
}

int chomp_status_wait(chomp_ctx* ctx, size_t epoch0, size_t n, unsigned* out) {
  if (!ctx || !out || n == 0) return fail(ctx, CHOMP_ERR_ARG, "status_wait: bad args");
  if (!ctx->n_hstatus || !ctx->posted_kind) return fail(ctx, CHOMP_ERR_STATE, "status_wait before status_post");
  if (epoch0 + n > ctx->n_hstatus) return fail(ctx, CHOMP_ERR_ARG, "status_wait: epoch range");
  HIPCHK(hipSetDevice(ctx->device));
  if (capturing(ctx))
    return fail(ctx, CHOMP_ERR_STATE,
                "status_wait while the context's stream is being captured (a wait on the host "
                "cannot be part of a graph): look at the status before the capture or after a replay");
  if (ctx->posted_kind == 3) {                       // (collected when its region was needed again)
    std::memcpy(out, ctx->posted_saved.data() + epoch0, n * sizeof(unsigned));
    return CHOMP_OK;
  }
  if (ctx->posted_in_capture) HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->posted_kind == 1) return mirror_collect(ctx, ctx->posted_half, ctx->posted_seq, epoch0, n, out);
  if (!ctx->posted_in_capture) HIPCHK(hipEventSynchronize(ctx->ev_status));
  std::memcpy(out, ctx->h_status + epoch0, n * sizeof(unsigned));
  return CHOMP_OK;
}

int chomp_epochs_set(chomp_ctx* ctx, size_t n_epoch, const chomp_cosmo* cosmo,
                     const double* z) {
  StageRange range_(ctx, "chomp:epochs_set (Stage K: sigma tables, mass-limit search)");
  if (!ctx || !cosmo || !z || n_epoch == 0) return fail(ctx, CHOMP_ERR_ARG, "epochs_set: bad args");
  for (size_t i = 0; i < n_epoch; ++i) {
    if (cosmo[i].w0 != -1.0 || cosmo[i].wa != 0.0)
      return fail(ctx, CHOMP_ERR_SCOPE,
                  "w0 != -1 or wa != 0: dynamical dark energy (cosmology.py:96-104, "
                  "odeint growth) is outside the hot-path scope");
    if (!(cosmo[i].omega_m0 > 0.0) || !(cosmo[i].h > 0.0) || !(cosmo[i].sigma_8 > 0.0))
      return fail(ctx, CHOMP_ERR_ARG, "epochs_set: omega_m0, h, sigma_8 must be > 0");
  }
  HIPCHK(hipSetDevice(ctx->device));
  int rc = alloc_epochs(ctx, n_epoch);
  if (rc) return rc;
  ctx->n_epoch = n_epoch;
  ctx->status_mirrored = false;     // (until a halo set-up of these epochs has finalised them)
  { const int rch = ensure_hstatus(ctx); if (rch) return rch; }
  ctx->have_mass = ctx->have_halo = false;
  ctx->fam_mask = 0;
  ctx->have_halofit.assign(n_epoch, 0);
  rc = upload(ctx, ctx->d_cosmo, cosmo, n_epoch * sizeof(chomp_cosmo), ctx->sh_cosmo);
  if (rc) return rc;
  rc = upload(ctx, ctx->d_z, z, n_epoch * sizeof(double), ctx->sh_z);
  if (rc) return rc;
  // cosmology-only work (sigma node table, sigma_8 integral) is shared by the epochs of
  // one cosmology: slot = index of the first epoch with identical parameters
  // (hashed: a design of a thousand distinct cosmologies must not cost n^2 comparisons per step)
  std::vector<int> slot(n_epoch), first;
  {
    std::unordered_multimap<uint64_t, int> seen;     // hash of the parameter bytes -> slot
    seen.reserve(n_epoch * 2);
    for (size_t i = 0; i < n_epoch; ++i) {
      uint64_t h = 1469598103934665603ull;           // FNV-1a over the 10 doubles
      const unsigned char* b = reinterpret_cast<const unsigned char*>(&cosmo[i]);
      for (size_t j = 0; j < sizeof(chomp_cosmo); ++j) { h ^= b[j]; h *= 1099511628211ull; }
      int s_found = -1;
      auto range = seen.equal_range(h);
      for (auto it = range.first; it != range.second; ++it)
        if (std::memcmp(&cosmo[first[it->second]], &cosmo[i], sizeof(chomp_cosmo)) == 0) {
          s_found = it->second;
          break;
        }
      if (s_found < 0) {
        s_found = (int)first.size();
        first.push_back((int)i);
        seen.emplace(h, s_found);
      }
      slot[i] = s_found;
    }
  }
  const size_t n_slots = first.size();
  ctx->n_slots = n_slots;
  ctx->slot = slot;
  first.resize(n_epoch, 0);
  rc = upload(ctx, ctx->d_slot, slot.data(), n_epoch * sizeof(int), ctx->sh_slot);
  if (rc) return rc;
  rc = upload(ctx, ctx->d_first, first.data(), n_epoch * sizeof(int), ctx->sh_first);
  if (rc) return rc;
  // (one cosmology or a few: most blocks, shortest launch; a batch of many: four nodes per
  //  thread and the ln S integrals one per wavefront -- see k_sigma_nodes)
  const unsigned gy = (unsigned)(n_slots + (n_epoch + 255) / 256);
#define CHOMP_SIGMA_NODES(BAO, NPT)                                                              \
  hipLaunchKernelGGL((k_sigma_nodes<BAO, NPT>),                                                   \
                     dim3(sigma_node_blocks<NPT>() + sigma_lns_blocks<NPT>() + sigma_gtab_blocks<NPT>(), gy), \
                     dim3(256), 0, ctx->stream, ctx->cfg, ctx->d_cosmo, ctx->d_z, ctx->d_first,    \
                     ctx->d_slot, (int)n_slots, (int)n_epoch, ctx->d_epochs, ctx->d_snodes,         \
                     ctx->d_status)
  if (n_slots >= 16) {
    if (ctx->with_bao) CHOMP_SIGMA_NODES(true, 4); else CHOMP_SIGMA_NODES(false, 4);
    // ... and the aiming tables behind them (the g table of a cosmology staged in LDS)
    const size_t shl = (size_t)kGTabCount * sizeof(double);
    if (!ctx->lds_lns_set) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sigma_lns<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shl));
      ctx->lds_lns_set = true;
    }
    if (ctx->with_bao)
      hipLaunchKernelGGL(k_sigma_lns<true>, dim3((unsigned)n_slots), dim3(kLnsThreads), 0, ctx->stream,
                         ctx->cfg, ctx->d_cosmo, ctx->d_z, ctx->d_first, ctx->d_snodes);
    else
      hipLaunchKernelGGL(k_sigma_lns<false>, dim3((unsigned)n_slots), dim3(kLnsThreads), shl, ctx->stream,
                         ctx->cfg, ctx->d_cosmo, ctx->d_z, ctx->d_first, ctx->d_snodes);
  } else {
    if (ctx->with_bao) CHOMP_SIGMA_NODES(true, 1); else CHOMP_SIGMA_NODES(false, 1);
  }
#undef CHOMP_SIGMA_NODES
  // (k_epoch_probe lives in chomp_probe.hip: the one kernel that is faster WITH machine LICM --
  //  but for the probing phase of a large batch, one wavefront per probe, compiled here)
  if (n_epoch >= 128) {
    const dim3 pgrid((unsigned)n_epoch, 2 * kProbes);
    if (ctx->with_bao)
      hipLaunchKernelGGL((k_epoch_probe<true, 1, 1>), pgrid, dim3(64), 0, ctx->stream, ctx->cfg,
                         ctx->d_epochs, ctx->d_search, ctx->d_cand, ctx->d_snodes, ctx->d_probe,
                         ctx->d_count, ctx->d_status);
    else
      hipLaunchKernelGGL((k_epoch_probe<false, 1, 1>), pgrid, dim3(64), 0, ctx->stream, ctx->cfg,
                         ctx->d_epochs, ctx->d_search, ctx->d_cand, ctx->d_snodes, ctx->d_probe,
                         ctx->d_count, ctx->d_status);
  }
  chomp::launch_epoch_probe(ctx->with_bao != 0, (unsigned)n_epoch, ctx->stream, ctx->cfg,
                            ctx->d_epochs, ctx->d_search, ctx->d_cand, ctx->d_snodes,
                            ctx->d_probe, ctx->d_count, ctx->d_status);
  HIPCHK(hipGetLastError());
  ctx->have_epochs = true;
  return CHOMP_OK;
}

// Host half of a halo set-up: argument checks, the HOD-derived constants (hod.py:172-186; one
// erfinv per epoch), the parameter uploads, and the integration groups the requested
// families need.
struct HaloPlan {
  unsigned fam = 0, kmask = 0;
  int groups[3] = {-1, -1, -1};
  int ng = 0;
  int want_nbar = 1;
  bool eval = false;     // some epoch's HOD has alpha != 1: the deep-level sums evaluate nodes
};
static int halo_prepare(chomp_ctx* ctx, const chomp_halo_par* profile, const chomp_hod_par* hod,
                        unsigned tables, HaloPlan* P) {
  const size_t n = ctx->n_epoch;
  for (size_t i = 0; i < n; ++i)
    if (profile[i].alpha != -1.0)
      return fail(ctx, CHOMP_ERR_SCOPE,
                  "halo alpha != -1: the general-profile transform y_general "
                  "(halo.py:491-559) is outside the hot-path scope (NFW only)");
  std::vector<HodDev> hd(n);
  for (size_t i = 0; i < n; ++i) {
    const chomp_hod_par& h = hod[i];
    HodDev& d = hd[i];
    d.log_M_min = h.log_M_min; d.sigma = h.sigma; d.log_M_0 = h.log_M_0;
    d.log_M_1p = h.log_M_1p; d.alpha = h.alpha;
    // hod.py:172-186; the `secon_moment_zero` typo there means second_moment_zero
    // is never clamped to first_moment_zero.
    d.first_zero = std::pow(10.0, h.log_M_min +
                                      h.sigma * erfinv_host(2.0 * ctx->cfg.halo_precision - 1.0));
    d.second_zero = std::pow(10.0, h.log_M_0);
    d.safe_norm = std::pow(10.0, h.log_M_min + 1.0 * h.sigma);
    if (h.alpha != 1.0) P->eval = true;
  }
  int rcu = upload(ctx, ctx->d_profile, profile, n * sizeof(chomp_halo_par), ctx->sh_profile);
  if (rcu) return rcu;
  rcu = upload(ctx, ctx->d_hod, hd.data(), n * sizeof(HodDev), ctx->sh_hod);
  if (rcu) return rcu;
  // header bits CHOMP_T_* are (1 << F_*) by construction
  P->fam = tables & 31u;
  P->kmask = P->fam | ((tables & CHOMP_T_EXCLUSION) ? kMaskExclusion : 0u);
  P->ng = 0;
  if (P->fam & ((1u << F_HM) | (1u << F_PPMM))) P->groups[P->ng++] = 0;
  if (P->fam & ((1u << F_HG) | (1u << F_PPGM))) P->groups[P->ng++] = 1;
  if (P->fam & (1u << F_PPGG)) P->groups[P->ng++] = 2;
  // integrands that can run beyond the node tables (the HOD ones): one level more in the tables
  if ((P->groups[0] > 0 || P->groups[1] > 0 || P->groups[2] > 0) && ctx->cfg.divmax > kNodeLevel)
    P->kmask |= kMaskDeepNodes;
  return CHOMP_OK;
}

// k_nu_table, then k_mass_nodes; plan != nullptr: the halo node tables in the same launch.
static int launch_nu_mass(chomp_ctx* ctx, int mf_kind, const HaloPlan* plan) {
  ctx->status_mirrored = false;    // (the mass function's status bits: mirrored by the halo set-up behind it, if any)
  const size_t n = ctx->n_epoch;
  const TabLayout& L = ctx->L;
  // (one cosmology or a few: epochs fastest in dispatch order, the longest integrals first;
  //  a cosmology per epoch: an epoch's masses side by side -- see k_nu_table)
  //  (the epochs as the grid's y axis: at most 65535 of them)
  const int ef = (ctx->n_slots < 16 || n > 65535) ? 1 : 0;
#define CHOMP_NU_TABLE(BAO, NW)                                                                 \
  hipLaunchKernelGGL((k_nu_table<BAO, NW>), ef ? dim3((unsigned)n, L.NM) : dim3(L.NM, (unsigned)n), \
                     dim3(64 * NW), 0, ctx->stream, ctx->cfg, L, ctx->d_epochs, ctx->d_search,   \
                     ctx->d_snodes, ctx->d_tab, ctx->d_status, ef)
  if ((size_t)L.NM * n <= 512) {      // (fewer integrals than SIMDs to put them on)
    if (ctx->with_bao) CHOMP_NU_TABLE(true, 4); else CHOMP_NU_TABLE(false, 4);
  } else {
    if (ctx->with_bao) CHOMP_NU_TABLE(true, 1); else CHOMP_NU_TABLE(false, 1);
  }
#undef CHOMP_NU_TABLE
  const size_t sh = (size_t)mass_lds_doubles(L.NM) * sizeof(double);
  const int ng = plan && plan->ng > 0 ? plan->ng : 1;
  // (node-table chunks: as many blocks per (epoch, group) as keep the launch under ~2 blocks
  //  per CU -- below that the chip is idle anyway and each block's node loop gets shorter)
  //  (measured: twice the chunks for the tables that are one level deeper -- 1024 blocks, each
  //   repeating the mass function part, for 768 resident -- 58 against 37 us on configs[2])
  unsigned chunks = plan ? (unsigned)(512 / (n * ng)) : 1u;
  chunks = chunks < 1 ? 1 : (chunks > 12 ? 12 : chunks);
  hipLaunchKernelGGL(k_mass_nodes, dim3((unsigned)n, (unsigned)ng, chunks), dim3(256), sh, ctx->stream,
                     ctx->cfg, L, ctx->d_epochs, ctx->d_search, ctx->d_tab, ctx->d_mass_par, mf_kind,
                     ctx->d_tinker, ctx->d_gl16, plan ? 1 : 0, ctx->d_profile, ctx->d_hod,
                     ctx->d_sici, ctx->d_nodes, ctx->d_endp, plan ? plan->groups[0] : -1,
                     plan ? plan->groups[1] : -1, plan ? plan->groups[2] : -1,
                     plan ? plan->kmask : 0u, ctx->d_status, ctx->d_npend, ctx->d_pending);
  HIPCHK(hipGetLastError());
  return CHOMP_OK;
}

// The knot integrals and everything after them (k_halo_knots, k_halo_knots_fast with the
// per-epoch finalisation).
static int launch_halo_knots(chomp_ctx* ctx, const HaloPlan& P) {
  { const int rcm = mirror_next(ctx); if (rcm) return rcm; }
  const size_t n = ctx->n_epoch;
  const TabLayout& L = ctx->L;
  const int ng = P.ng > 0 ? P.ng : 1;
  // (a block per knot pair when the whole launch is a few hundred knots: one or a few epochs)
  const bool wide = (size_t)L.NK * n * ng <= 512;
  // (... single-wavefront blocks when the knots outnumber the chip's wavefront slots about twice:
  //  a finished knot then frees its slot at once -- C3 89 -> 78 us; below that, four knots to a
  //  256-thread block, whose n_bar block also keeps its four wavefronts -- C2 37 vs 39 us)
  const bool lone = !wide && (size_t)L.NK * n * ng > 4096;
  const unsigned kb = (wide || lone) ? (unsigned)L.NK : (unsigned)((L.NK + 3) / 4);
  const size_t shk = (size_t)(L.NM + 8 * (L.NM - 1) + kKnotScratch) * sizeof(double);
  // The knots of the HOD groups that do not converge within the node tables are listed; every
  // listed knot gets a slot of the sample buffer (k_halo_knots_samples fills it, k_halo_knots_fast
  // sums the knot's levels from it).  The buffer holds every knot that CAN be listed while that
  // stays under the budget; beyond it the two kernels work the list off in rounds of `slots`.
  const bool hod_groups = P.groups[0] > 0 || P.groups[1] > 0 || P.groups[2] > 0;
  const bool deep_route = hod_groups && ctx->cfg.divmax > kNodeLevel;
  size_t slots = 0;
  int rounds = 1;
  if (deep_route) {
    const size_t worst = (size_t)ng * n * (size_t)L.NK;
    const size_t budget = ((size_t)1 << 30) / ((size_t)kDeepSlot * sizeof(double));
    slots = worst < budget ? worst : budget;
    if (ctx->tune[CHOMP_TUNE_DEEP_SLOTS] > 0 && (size_t)ctx->tune[CHOMP_TUNE_DEEP_SLOTS] < slots)
      slots = (size_t)ctx->tune[CHOMP_TUNE_DEEP_SLOTS];
    if ((worst + slots - 1) / slots > (size_t)kPendingRounds)
      slots = (worst + kPendingRounds - 1) / kPendingRounds;
    rounds = (int)((worst + slots - 1) / slots);
    int rck = ensure(ctx, &ctx->d_samples, &ctx->cap_samples, slots * (size_t)kDeepSlot);
    if (rck) return rck;
    rck = ensure(ctx, &ctx->d_psum, &ctx->cap_psum, slots * (size_t)(kDeepChunks * kDeepPsum));
    if (rck) return rck;
  }
  // (a set-up of one or a few epochs -- every launch lasts as long as its slowest unit -- leaves
  //  the table at level 9; a batch, where the ~11 % of knots that converge AT level 10 would each
  //  take a slot, a sampling work item and a summing block, walks the whole table: 888 against
  //  1001 listed knots and -6.6 us per configs[2] step, tools/scratch/hod_cap.py)
  int hod_cap = ctx->tune[CHOMP_TUNE_HOD_CAP] >= 0 ? (int)ctx->tune[CHOMP_TUNE_HOD_CAP]
                : ((size_t)L.NK * n * ng <= 768 ? kHodCapLevel : kNodeLevel);
  if (hod_cap < 6) hod_cap = 6;
  // chomp_set_tuning: the checker (every listed knot by literal evaluation) and the two
  // thresholds at which a knot leaves the fast path by itself
  const int all_literal = ctx->tune[CHOMP_TUNE_DEEP_LITERAL] > 0 ? 1 : 0;
  const double deep_tol = ctx->tune[CHOMP_TUNE_DEEP_TOL] >= 0
                              ? (double)ctx->tune[CHOMP_TUNE_DEEP_TOL] * 1e-15 : kDeepTol;
  int max_rough = ctx->tune[CHOMP_TUNE_DEEP_MAX_BREAKS] >= 0 ? (int)ctx->tune[CHOMP_TUNE_DEEP_MAX_BREAKS]
                                                             : kDeepMaxRough;
  if (max_rough > kDeepMaxRough) max_rough = kDeepMaxRough;
  int max_fine = ctx->tune[CHOMP_TUNE_DEEP_MAX_FINE] >= 0 ? (int)ctx->tune[CHOMP_TUNE_DEEP_MAX_FINE]
                                                          : kDeepMaxFine;
  if (max_fine > kDeepMaxFine) max_fine = kDeepMaxFine;
  // (the break-point plans of the (epoch, group)s: an extra block row of k_halo_knots)
  const int want_plan = deep_route ? 1 : 0;
  if (want_plan) {
    const int rcp = ensure(ctx, &ctx->d_plan, &ctx->cap_plan, n * 3 * sizeof(DeepPlan));
    if (rcp) return rcp;
  }
#define CHOMP_KNOTS(KNW)                                                                          \
  hipLaunchKernelGGL((k_halo_knots<KNW>), dim3((unsigned)n, kb + (P.want_nbar ? 1u : 0u) + (unsigned)want_plan, (unsigned)ng), \
                     dim3(KNW == 0 ? 64 : 256), shk, ctx->stream, ctx->cfg, L, ctx->d_epochs, ctx->d_tab,        \
                     ctx->d_profile, ctx->d_hod, ctx->d_sici, ctx->d_nodes, ctx->d_endp,          \
                     P.groups[0], P.groups[1], P.groups[2], P.kmask, P.want_nbar, ctx->d_pending, \
                     ctx->d_npend, ctx->d_status, hod_cap, want_plan, max_rough, max_fine,           \
                     reinterpret_cast<DeepPlan*>(ctx->d_plan))
  if (wide) CHOMP_KNOTS(4); else if (lone) CHOMP_KNOTS(0); else CHOMP_KNOTS(1);
#undef CHOMP_KNOTS
  // blocks 0..n-1 take the epochs' tokens; with integrands that can run beyond the node
  // tables (the HOD ones) enough further blocks to fill the chip draw from the list
  unsigned gd = (unsigned)n;
  if (deep_route) {
    // (as many blocks as are resident at once -- two of 256 threads or one of 512 per CU, 256
    //  CUs: a block loops over the list until it is empty, and one that starts after that only
    //  stages its tables to find nothing left)
    // (three of 256 threads -- the lean instance: 164 registers, 51 KB of LDS -- or one of 512)
    unsigned want = (unsigned)(L.NK * n * ng);
    const unsigned resident = (size_t)L.NK * n * ng <= 768 ? 256u : (P.eval ? 512u : 768u);
    if (want > resident) want = resident;
    if (want > gd) gd = want;
  }
  size_t shf = deep_fast_lds<kDeepCoarse>(L.NM, ctx->cfg.divmax);
  if (shf < (size_t)finalize_lds_doubles(L.NK) * sizeof(double))
    shf = (size_t)finalize_lds_doubles(L.NK) * sizeof(double);
#define CHOMP_KNOTS_FAST(NT, SELF, EVAL, GRID, ROUND, LO, HI, FROM) \
  CHOMP_KNOTS_FAST_L(NT, SELF, EVAL, false, GRID, ROUND, LO, HI, FROM)
#define CHOMP_KNOTS_FAST_L(NT, SELF, EVAL, LIT, GRID, ROUND, LO, HI, FROM)                                         \
  hipLaunchKernelGGL((k_halo_knots_fast<kDeepCoarse, NT, SELF, EVAL, LIT>), dim3(GRID), dim3(NT), shf, ctx->stream, \
                     ctx->cfg, L, ctx->d_epochs, ctx->d_tab, ctx->d_sici, P.groups[0], P.groups[1], \
                     P.groups[2], P.kmask, (int)n, ctx->d_pending, ctx->d_npend, ctx->d_epochs,     \
                     P.fam, ctx->d_status, ctx->d_deepw, all_literal, deep_tol, max_rough, max_fine, \
                     ctx->d_deepstat, ctx->d_samples, ctx->d_psum, parts, ROUND, LO, HI, FROM,      \
                     reinterpret_cast<const DeepPlan*>(ctx->d_plan), ctx->d_profile, ctx->d_hod)
#define CHOMP_KNOTS_SAMPLES(GRID, LO, HI)                                                        \
  hipLaunchKernelGGL((k_halo_knots_samples<kDeepCoarse>), dim3(GRID), dim3(256), 0, ctx->stream,   \
                     ctx->cfg, L, ctx->d_sici, P.groups[0], P.groups[1], P.groups[2], P.kmask,      \
                     (int)n, ctx->d_pending, ctx->d_nodes, ctx->d_endp, ctx->d_samples, ctx->d_psum, \
                     parts, LO, HI)
#define CHOMP_KNOTS_LITERAL(NT, GRID)                                                            \
  hipLaunchKernelGGL((k_halo_knots_literal<NT>), dim3(GRID), dim3(NT), shl, ctx->stream, ctx->cfg,  \
                     L, ctx->d_epochs, ctx->d_tab, ctx->d_profile, ctx->d_hod, ctx->d_sici,         \
                     P.groups[0], P.groups[1], P.groups[2], P.kmask, (int)n, ctx->d_pending,        \
                     ctx->d_npend, ctx->d_epochs, P.fam, ctx->d_status, ctx->d_deepstat)
  if (shf > 64 * 1024) {            // (more than 64 KiB of dynamic LDS: opt in, once per context)
    if (!ctx->lds_knots_set) {       // (per context: the attribute belongs to the current device)
#define CHOMP_LDS_OPT_IN(NT, SELF, EVAL)                                                   \
      HIPCHK(hipFuncSetAttribute(                                                          \
          reinterpret_cast<const void*>(&k_halo_knots_fast<kDeepCoarse, NT, SELF, EVAL>), \
          hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024))
      CHOMP_LDS_OPT_IN(kDeepThreadsFew, false, false);
      CHOMP_LDS_OPT_IN(kDeepThreads, false, false);
      CHOMP_LDS_OPT_IN(kDeepThreads, false, true);
      CHOMP_LDS_OPT_IN(kDeepThreadsFew, false, true);
      CHOMP_LDS_OPT_IN(kDeepThreads, true, true);
      HIPCHK(hipFuncSetAttribute(
          reinterpret_cast<const void*>(&k_halo_knots_fast<kDeepCoarse, kDeepThreads, false, true, true>),
          hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
#undef CHOMP_LDS_OPT_IN
      ctx->lds_knots_set = true;
    }
  }
  const bool few = (size_t)L.NK * n * ng <= 768;
  // (the sampling launch: a listed knot's 2^11 + 1 samples in `parts` work items -- the fewer
  //  knots there can be, the finer, so that a single epoch's handful still spreads over the chip)
  const int parts = few ? 8 : ((size_t)L.NK * n * ng <= 8192 ? 4 : 2);
  if (!hod_groups) {                // (group 0 alone: listed knots are done in the same launch)
    CHOMP_KNOTS_FAST(kDeepThreads, true, true, gd, 0, 0, 0x7fffffff, 0);
  } else if (!deep_route) {         // (divmax within the node tables: nothing is ever listed)
    if (few) CHOMP_KNOTS_FAST(kDeepThreadsFew, false, false, gd, 0, 0, 0, 0);
    else CHOMP_KNOTS_FAST(kDeepThreads, false, false, gd, 0, 0, 0, 0);
  } else {
    for (int r = 0; r < rounds; ++r) {
      const int lo = (int)((size_t)r * slots), hi = (int)((size_t)(r + 1) * slots);
      size_t gs = slots * (size_t)parts;
      if (gs > 1536) gs = 1536;
      CHOMP_KNOTS_SAMPLES((unsigned)gs, lo, hi);
      const unsigned g = r == 0 ? gd : (gd < 512u ? gd : 512u);
      if (P.eval) {
        if (few) CHOMP_KNOTS_FAST(kDeepThreadsFew, false, true, (g < 256u ? g : 256u), r, lo, hi, 0);
        else CHOMP_KNOTS_FAST(kDeepThreads, false, true, (g < 512u ? g : 512u), r, lo, hi, 0);
      } else {
        // (the lean instance, and behind it ONE launch for what it hands on: the knots that
        //  need node evaluations -- few and long: 512 threads each -- and, in the last round,
        //  those that go to the literal evaluation)
        const bool last_round = r == rounds - 1;
        const unsigned ge = all_literal ? g : (g < 256u ? g : 256u);
        if (few) CHOMP_KNOTS_FAST(kDeepThreadsFew, false, false, g, r, lo, hi, 0);
        else CHOMP_KNOTS_FAST(kDeepThreads, false, false, g, r, lo, hi, 0);
        if (last_round) CHOMP_KNOTS_FAST_L(kDeepThreads, false, true, true, ge, r, lo, hi, 1);
        else CHOMP_KNOTS_FAST(kDeepThreads, false, true, ge, r, lo, hi, 1);
      }
    }
  }
  // knots can only be handed on when some HOD Romberg may run beyond the node tables
  if (deep_route && P.eval) {       // (alpha != 1: no launch behind the fast sums that could take the list)
    const size_t shl = deep_literal_lds(L.NM, L.NK);
    // (an empty list is the rule: few blocks, each returns after one read)
    const unsigned gl = all_literal ? gd : (gd < 256u ? gd : 256u);
    CHOMP_KNOTS_LITERAL(kDeepThreads, gl);
  }
#undef CHOMP_KNOTS_FAST
#undef CHOMP_KNOTS_FAST_L
#undef CHOMP_KNOTS_SAMPLES
#undef CHOMP_KNOTS_LITERAL
  HIPCHK(hipGetLastError());
  ctx->have_halo = true;
  ctx->fam_mask |= P.fam;
  ctx->status_mirrored = ctx->L.h_status != nullptr;
  return CHOMP_OK;
}

int chomp_mass_setup(chomp_ctx* ctx, const chomp_halo_par* par, int mf_kind) {
  StageRange range_(ctx, "chomp:mass_setup (Stage K: nu table, mass function)");
  if (!ctx || !par) return fail(ctx, CHOMP_ERR_ARG, "mass_setup: bad args");
  if (!ctx->have_epochs) return fail(ctx, CHOMP_ERR_STATE, "mass_setup before epochs_set");
  if (mf_kind != CHOMP_MF_ST && mf_kind != CHOMP_MF_TINKER)
    return fail(ctx, CHOMP_ERR_ARG, "mass_setup: unknown mass function kind");
  HIPCHK(hipSetDevice(ctx->device));
  int rc = upload(ctx, ctx->d_mass_par, par, ctx->n_epoch * sizeof(chomp_halo_par), ctx->sh_mass);
  if (rc) return rc;
  rc = launch_nu_mass(ctx, mf_kind, nullptr);
  if (rc) return rc;
  ctx->have_mass = true;
  // Knot tables already built stay as they are (the reference's MassFunction.set_halo
  // does not reset Halo._initialized_h_m / _pp_mm, halo.py:220-235).
  return CHOMP_OK;
}

int chomp_halo_setup(chomp_ctx* ctx, const chomp_halo_par* profile,
                     const chomp_hod_par* hod, unsigned tables) {
  StageRange range_(ctx, "chomp:halo_setup (Stage K: node tables, knot integrals)");
  if (!ctx || !profile || !hod) return fail(ctx, CHOMP_ERR_ARG, "halo_setup: bad args");
  if (!ctx->have_mass) return fail(ctx, CHOMP_ERR_STATE, "halo_setup before mass_setup");
  HIPCHK(hipSetDevice(ctx->device));
  HaloPlan P;
  int rc = halo_prepare(ctx, profile, hod, tables, &P);
  if (rc) return rc;
  const size_t n = ctx->n_epoch;
  const TabLayout& L = ctx->L;
  const size_t sh = (size_t)(L.NM + 8 * (L.NM - 1) + kKnotScratch) * sizeof(double);
  // (few epochs: the table's nodes over several blocks each)
  const unsigned ngy = (unsigned)(P.ng > 0 ? P.ng : 1);
  unsigned nchunks = (unsigned)(512 / (n * ngy));
  nchunks = nchunks < 1 ? 1 : (nchunks > 8 ? 8 : nchunks);
  hipLaunchKernelGGL(k_halo_nodes, dim3((unsigned)n, ngy, nchunks), dim3(256), sh,
                     ctx->stream, ctx->cfg, L, ctx->d_epochs, ctx->d_tab, ctx->d_profile,
                     ctx->d_hod, ctx->d_sici, ctx->d_nodes, ctx->d_endp, P.groups[0], P.groups[1],
                     P.groups[2], P.kmask, ctx->d_status, ctx->d_npend, ctx->d_pending);
  return launch_halo_knots(ctx, P);
}

int chomp_stage_k(chomp_ctx* ctx, const chomp_halo_par* mass_par, int mf_kind,
                  const chomp_halo_par* profile, const chomp_hod_par* hod, unsigned tables) {
  StageRange range_(ctx, "chomp:stage_k (Stage K: mass function + halo model)");
  if (!ctx || !mass_par || !profile || !hod) return fail(ctx, CHOMP_ERR_ARG, "stage_k: bad args");
  if (!ctx->have_epochs) return fail(ctx, CHOMP_ERR_STATE, "stage_k before epochs_set");
  if (mf_kind != CHOMP_MF_ST && mf_kind != CHOMP_MF_TINKER)
    return fail(ctx, CHOMP_ERR_ARG, "stage_k: unknown mass function kind");
  HIPCHK(hipSetDevice(ctx->device));
  int rc = upload(ctx, ctx->d_mass_par, mass_par, ctx->n_epoch * sizeof(chomp_halo_par), ctx->sh_mass);
  if (rc) return rc;
  HaloPlan P;
  rc = halo_prepare(ctx, profile, hod, tables, &P);
  if (rc) return rc;
  rc = launch_nu_mass(ctx, mf_kind, &P);
  if (rc) return rc;
  ctx->have_mass = true;
  return launch_halo_knots(ctx, P);
}

int chomp_set_transfer(chomp_ctx* ctx, int kind) {
  if (!ctx) return CHOMP_ERR_ARG;
  if (kind != CHOMP_TRANSFER_EH && kind != CHOMP_TRANSFER_EH_BAO)
    return fail(ctx, CHOMP_ERR_ARG, "set_transfer: unknown transfer function");
  if (ctx->with_bao != (kind == CHOMP_TRANSFER_EH_BAO)) {
    ctx->with_bao = kind == CHOMP_TRANSFER_EH_BAO;
    ctx->have_epochs = ctx->have_mass = ctx->have_halo = false;   // every table depends on T(k)
    ctx->fam_mask = 0;
  }
  return CHOMP_OK;
}

int chomp_hod_stats(chomp_ctx* ctx, size_t epoch0, size_t n, double* out) {
  if (!ctx || !out || n == 0) return fail(ctx, CHOMP_ERR_ARG, "hod_stats: bad args");
  if (!ctx->have_halo) return fail(ctx, CHOMP_ERR_STATE, "hod_stats before halo_setup");
  if (epoch0 + n > ctx->n_epoch) return fail(ctx, CHOMP_ERR_ARG, "hod_stats: epoch range");
  HIPCHK(hipSetDevice(ctx->device));
  int rc = ensure(ctx, &ctx->d_stage_out, &ctx->cap_out, 3 * n);
  if (rc) return rc;
  const TabLayout& L = ctx->L;
  const size_t sh = (size_t)(L.NM + 8 * (L.NM - 1) + kKnotScratch) * sizeof(double);
  hipLaunchKernelGGL(k_hod_stats, dim3(3, (unsigned)n), dim3(256), sh, ctx->stream, ctx->cfg, L,
                     ctx->d_epochs, (int)epoch0, ctx->d_tab, ctx->d_profile, ctx->d_hod,
                     ctx->d_sici, ctx->d_stage_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, ctx->d_stage_out, 3 * n * sizeof(double), hipMemcpyDeviceToHost,
                        ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return CHOMP_OK;
}

static int check_power(chomp_ctx* ctx, int which, size_t epoch0, size_t n) {
  if (!ctx) return CHOMP_ERR_ARG;
  if (!ctx->have_epochs) return fail(ctx, CHOMP_ERR_STATE, "power before epochs_set");
  if (epoch0 + n > ctx->n_epoch || n == 0) return fail(ctx, CHOMP_ERR_ARG, "power: epoch range");
  const int w = which & 15;
  const bool hf = (which & CHOMP_P_HALOFIT) != 0;
  if (w < CHOMP_P_LIN || w > CHOMP_P_GG) return fail(ctx, CHOMP_ERR_ARG, "power: unknown spectrum");
  if (hf && w == CHOMP_P_LIN) return fail(ctx, CHOMP_ERR_ARG, "power: halofit|lin");
  unsigned need = 0;
  if (w == CHOMP_P_MM && !hf) need = (1u << F_HM) | (1u << F_PPMM);
  if (w == CHOMP_P_GM) need = (1u << F_HM) | (1u << F_HG) | (1u << F_PPGM);
  if (w == CHOMP_P_GG) need = (1u << F_HG) | (1u << F_PPGG);
  if ((ctx->fam_mask & need) != need)
    return fail(ctx, CHOMP_ERR_STATE, "power: knot tables of this spectrum were not built "
                                      "(chomp_halo_setup families)");
  if (hf)
    for (size_t i = epoch0; i < epoch0 + n; ++i)
      if (!ctx->have_halofit[i])
        return fail(ctx, CHOMP_ERR_STATE, "power: chomp_halofit_setup not called for epoch");
  return CHOMP_OK;
}

// Halo(extrapolate=True): refresh the constants of the continuation above k_max for the
// epochs about to be evaluated (cheap: one 64-thread block per epoch).
static int prepare_extrapolation(chomp_ctx* ctx, int which, size_t epoch0, size_t n) {
  const int w = which & 15;
  if (!(which & CHOMP_P_EXTRAPOLATE) || (which & CHOMP_P_HALOFIT) || w == CHOMP_P_LIN)
    return CHOMP_OK;
  if (ctx->with_bao)
    hipLaunchKernelGGL(k_power_extrap<true>, dim3((unsigned)n), dim3(64), 0, ctx->stream, ctx->cfg,
                       ctx->L, ctx->d_epochs, ctx->d_tab, w, (int)epoch0);
  else
    hipLaunchKernelGGL(k_power_extrap<false>, dim3((unsigned)n), dim3(64), 0, ctx->stream, ctx->cfg,
                       ctx->L, ctx->d_epochs, ctx->d_tab, w, (int)epoch0);
  HIPCHK(hipGetLastError());
  return CHOMP_OK;
}

// k_power_prep of the streaming shape: the k-only table (d_ktab, d_winfo) and the list of k
// groups for the per-lane pass (d_slow, through counter *parity).
static int stage_e_prep(chomp_ctx* ctx, size_t epoch0, int w, const double* dk, size_t nk,
                        int* parity) {
  const TabLayout& L = ctx->L;
  const unsigned gx = (unsigned)((nk + 511) / 512);
  const size_t groups = (nk + 127) / 128;
  // k groups that cannot take the streaming kernel go on a compact list for the per-lane
  // pass.  Its counters ping-pong on a host-side parity bit (a launch appends through one
  // and clears the other for the next streaming call).  A call captured into a HIP graph
  // would replay ONE parity for ever and run its counter past the list, so under capture
  // the call clears both counters itself (a memset node) and keeps the bit still.
  const size_t had = ctx->cap_slow;
  int rc = ensure(ctx, &ctx->d_slow, &ctx->cap_slow, groups + 2);
  if (rc) return rc;
  if (ctx->cap_slow != had)                      // fresh buffer: clear both counters
    HIPCHK(hipMemsetAsync(ctx->d_slow, 0, 2 * sizeof(int), ctx->stream));
  *parity = ctx->slow_parity;
  if (capturing(ctx)) ctx->slow_by_memset = true;
  if (ctx->slow_by_memset) {     // (sticky: a graph may be replayed between any two calls)
    HIPCHK(hipMemsetAsync(ctx->d_slow, 0, 2 * sizeof(int), ctx->stream));
    *parity = 0;
  } else {
    ctx->slow_parity ^= 1;
  }
  const unsigned gx8 = (gx + 7) / 8 * 8;
  rc = ensure(ctx, &ctx->d_winfo, &ctx->cap_winfo, (size_t)gx8 * 4);
  if (rc) return rc;
  rc = ensure(ctx, &ctx->d_ktab, &ctx->cap_ktab, (size_t)gx8 * 1024);
  if (rc) return rc;
  if (ctx->with_bao)
    hipLaunchKernelGGL(k_power_prep<true>, dim3(gx8), dim3(256), 0, ctx->stream, ctx->cfg, L,
                       ctx->d_epochs, (int)epoch0, w, dk, nk, ctx->d_ktab, ctx->d_winfo,
                       ctx->d_slow, *parity);
  else
    hipLaunchKernelGGL(k_power_prep<false>, dim3(gx8), dim3(256), 0, ctx->stream, ctx->cfg, L,
                       ctx->d_epochs, (int)epoch0, w, dk, nk, ctx->d_ktab, ctx->d_winfo,
                       ctx->d_slow, *parity);
  HIPCHK(hipGetLastError());
  return CHOMP_OK;
}

static bool plan_matches(chomp_ctx* ctx, size_t epoch0, const double* dk, size_t nk) {
  const chomp_ctx::PowerPlan& P = ctx->plan;
  if (!P.valid || P.k != dk || P.nk != nk || P.bao != ctx->with_bao) return false;
  if (ctx->sh_cosmo.shadow.size() < (epoch0 + 1) * sizeof(chomp_cosmo)) return false;
  return std::memcmp(&P.cosmo, ctx->sh_cosmo.shadow.data() + epoch0 * sizeof(chomp_cosmo),
                     sizeof(chomp_cosmo)) == 0;
}

int chomp_power_plan(chomp_ctx* ctx, size_t epoch0, const double* k, size_t nk) {
  if (!ctx || !k || nk == 0) return fail(ctx, CHOMP_ERR_ARG, "power_plan: bad args");
  if (!ctx->have_epochs || epoch0 >= ctx->n_epoch) return fail(ctx, CHOMP_ERR_STATE, "power_plan: epoch");
  if (nk % 2 != 0 || reinterpret_cast<uintptr_t>(k) % 16 != 0)
    return fail(ctx, CHOMP_ERR_ARG, "power_plan: k must be 16-byte aligned with an even length");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->plan.valid = false;
  int parity = 0;
  int rc = stage_e_prep(ctx, epoch0, CHOMP_P_MM, k, nk, &parity);
  if (rc) return rc;
  int n_slow = 0;
  HIPCHK(hipMemcpyAsync(&n_slow, ctx->d_slow + parity, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->plan.k = k;
  ctx->plan.nk = nk;
  std::memcpy(&ctx->plan.cosmo, ctx->sh_cosmo.shadow.data() + epoch0 * sizeof(chomp_cosmo),
              sizeof(chomp_cosmo));
  ctx->plan.bao = ctx->with_bao;
  ctx->plan.parity = parity;
  ctx->plan.n_slow = n_slow;
  ctx->plan.valid = true;
  return CHOMP_OK;
}

int chomp_power_range(chomp_ctx* ctx, int which, size_t epoch0, size_t n, const double* k,
                      size_t nk, double* out, int mem) {
  StageRange range_(ctx, "chomp:power (Stage E)");
  int rc = check_power(ctx, which, epoch0, n);
  if (rc) return rc;
  if (!k || !out || nk == 0) return fail(ctx, CHOMP_ERR_ARG, "power: null buffer");
  HIPCHK(hipSetDevice(ctx->device));
  rc = prepare_extrapolation(ctx, which, epoch0, n);
  if (rc) return rc;
  const bool extrap = (which & CHOMP_P_EXTRAPOLATE) && !(which & CHOMP_P_HALOFIT);
  const double* dk = k;
  double* dout = out;
  if (mem == CHOMP_HOST) {
    rc = ensure(ctx, &ctx->d_stage_out, &ctx->cap_out, nk * n);
    if (rc) return rc;
    // through pinned mirrors: a pageable buffer would be staged by the runtime in small
    // synchronous pieces
    rc = ensure_host(ctx, &ctx->h_stage_in, &ctx->cap_hin, nk);
    if (rc) return rc;
    rc = ensure_host(ctx, &ctx->h_stage_out, &ctx->cap_hout, nk * n);
    if (rc) return rc;
    // The reference-shaped loop asks for the same k array at every redshift: the device copy of
    // the last host k grid is kept (a buffer of its own; 32 KB compared in ~1 us) and the upload
    // -- which would sit behind the set-up on the stream, in front of the evaluation -- skipped.
    rc = ensure(ctx, &ctx->d_kcache, &ctx->cap_kcache, nk);
    if (rc) return rc;
    if (ctx->kcache_ptr != ctx->d_kcache || ctx->kcache_shadow.size() != nk ||
        std::memcmp(ctx->kcache_shadow.data(), k, nk * sizeof(double)) != 0) {
      if (capturing(ctx))
        return fail(ctx, CHOMP_ERR_STATE, "power: a new host k grid while the stream is being captured");
      std::memcpy(ctx->h_stage_in, k, nk * sizeof(double));
      HIPCHK(hipMemcpyAsync(ctx->d_kcache, ctx->h_stage_in, nk * sizeof(double),
                            hipMemcpyHostToDevice, ctx->stream));
      ctx->kcache_shadow.assign(k, k + nk);
      ctx->kcache_ptr = ctx->d_kcache;
    }
    dk = ctx->d_kcache;
    dout = ctx->d_stage_out;
  }
  const TabLayout& L = ctx->L;
  if ((which & CHOMP_P_HALOFIT) == 0 &&
      (reinterpret_cast<uintptr_t>(dk) % 16 == 0) && (reinterpret_cast<uintptr_t>(dout) % 16 == 0)) {
    const unsigned gx = (unsigned)((nk + 511) / 512);
    bool one_cosmology = true, streaming = true, lanes_needed = true;
    for (size_t i = 1; i < n; ++i) one_cosmology &= ctx->slot[epoch0 + i] == ctx->slot[epoch0];
    const int w = which & 15;
    size_t stream_min = (size_t)1 << 22;           // samples; below this the launches dominate
    // (chomp_set_tuning: the tests force either launch shape on small grids)
    if (ctx->tune[CHOMP_TUNE_E_STREAM_MIN] >= 0) stream_min = (size_t)ctx->tune[CHOMP_TUNE_E_STREAM_MIN];
    int parity = 0;
    if (one_cosmology && w != CHOMP_P_LIN && nk % 2 == 0 && nk * n >= stream_min) {
      // (a k grid registered with chomp_power_plan keeps its k-only table: no prep launch)
      const bool planned = mem == CHOMP_DEVICE && plan_matches(ctx, epoch0, dk, nk);
      const bool timed = ctx->timing != 0;
      ctx->timing_valid = false;
      if (timed) HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
      if (planned) {
        parity = ctx->plan.parity;
        lanes_needed = ctx->plan.n_slow != 0;
      } else {
        ctx->plan.valid = false;                   // (this call overwrites the table)
        rc = stage_e_prep(ctx, epoch0, w, dk, nk, &parity);
        if (rc) return rc;
      }
      if (timed) HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
      const unsigned gx8 = (gx + 7) / 8 * 8;
      int per = n % 2 == 0 ? 2 : 1;
      // (chomp_set_tuning: rows per block of the streaming kernel; 2 measured best on MI355X)
      {
        const long long v = ctx->tune[CHOMP_TUNE_E_ROWS];
        if ((v == 1 || v == 2 || v == 4) && n % (size_t)v == 0) per = (int)v;
      }
      const unsigned gy = (unsigned)(n / per);
#define CHOMP_STREAM(P)                                                                   \
      hipLaunchKernelGGL(k_power_stream<P>, dim3(gx8, gy), dim3(256), 0, ctx->stream, L,  \
                         ctx->d_tab, w, (int)epoch0, ctx->d_ktab, ctx->d_winfo, nk, dout)
      switch (per) {
        case 1: CHOMP_STREAM(1); break;
        case 4: CHOMP_STREAM(4); break;
        default: CHOMP_STREAM(2); break;
      }
#undef CHOMP_STREAM
      if (timed) {
        HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
        ctx->timing_valid = true;
      }
    } else {
      // enough blocks to fill 256 CUs: split the epochs over blockIdx.y when nk is small
      unsigned gy = (2048 + gx - 1) / gx;
      if (gy > n) gy = (unsigned)n;
      const int epy = (int)((n + gy - 1) / gy);
      gy = (unsigned)((n + epy - 1) / epy);
      if (ctx->with_bao)
        hipLaunchKernelGGL(k_power_grid<true>, dim3(gx, gy), dim3(256), 0, ctx->stream, ctx->cfg, L,
                           ctx->d_epochs, ctx->d_tab, w, (int)epoch0, (int)n, epy, 1, dk, nk,
                           dout, ctx->d_slow, parity, 1, extrap);
      else
        hipLaunchKernelGGL(k_power_grid<false>, dim3(gx, gy), dim3(256), 0, ctx->stream, ctx->cfg, L,
                           ctx->d_epochs, ctx->d_tab, w, (int)epoch0, (int)n, epy, 1, dk, nk,
                           dout, ctx->d_slow, parity, 1, extrap);
      streaming = false;
    }
    // per-lane pass over the listed k groups (streaming shape only; a planned grid knows
    // whether it has any)
    if (streaming && lanes_needed) {
      if (ctx->with_bao)
        hipLaunchKernelGGL(k_power_grid_lanes<true>, dim3(1024), dim3(256), 0, ctx->stream, ctx->cfg,
                           L, ctx->d_epochs, ctx->d_tab, w, extrap, (int)epoch0, (int)n, dk, nk, dout,
                           ctx->d_slow, parity);
      else
        hipLaunchKernelGGL(k_power_grid_lanes<false>, dim3(1024), dim3(256), 0, ctx->stream, ctx->cfg,
                           L, ctx->d_epochs, ctx->d_tab, w, extrap, (int)epoch0, (int)n, dk, nk, dout,
                           ctx->d_slow, parity);
    }
    if (ctx->timing_valid) HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
  } else {
    unsigned gx = (unsigned)((nk + 255) / 256);
    if (gx > 2048) gx = 2048;
    const size_t sh = (size_t)(12 * (L.NK - 1)) * sizeof(double);
    if (ctx->with_bao)
      hipLaunchKernelGGL(k_power<true>, dim3(gx, (unsigned)n), dim3(256), sh, ctx->stream, ctx->cfg,
                         L, ctx->d_epochs, ctx->d_tab, which, (int)epoch0, dk, nk, dout);
    else
      hipLaunchKernelGGL(k_power<false>, dim3(gx, (unsigned)n), dim3(256), sh, ctx->stream, ctx->cfg,
                         L, ctx->d_epochs, ctx->d_tab, which, (int)epoch0, dk, nk, dout);
  }
  HIPCHK(hipGetLastError());
  if (mem == CHOMP_HOST) {
    HIPCHK(hipMemcpyAsync(ctx->h_stage_out, dout, nk * n * sizeof(double), hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::memcpy(out, ctx->h_stage_out, nk * n * sizeof(double));
  }
  return CHOMP_OK;
}

int chomp_power(chomp_ctx* ctx, int which, const double* k, size_t nk, double* out, int mem) {
  if (!ctx) return CHOMP_ERR_ARG;
  return chomp_power_range(ctx, which, 0, ctx->n_epoch, k, nk, out, mem);
}

int chomp_sigma_r(chomp_ctx* ctx, size_t epoch, const double* scale, size_t n, double* out) {
  if (!ctx || !scale || !out || n == 0) return fail(ctx, CHOMP_ERR_ARG, "sigma_r: bad args");
  if (!ctx->have_epochs || epoch >= ctx->n_epoch) return fail(ctx, CHOMP_ERR_STATE, "sigma_r: epoch");
  HIPCHK(hipSetDevice(ctx->device));
  int rc = ensure(ctx, &ctx->d_stage_in, &ctx->cap_in, n);
  if (rc) return rc;
  rc = ensure(ctx, &ctx->d_stage_out, &ctx->cap_out, n);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(ctx->d_stage_in, scale, n * sizeof(double), hipMemcpyHostToDevice,
                        ctx->stream));
  if (ctx->with_bao)
    hipLaunchKernelGGL(k_sigma_r<true>, dim3((unsigned)n), dim3(256), 0, ctx->stream, ctx->cfg,
                       ctx->d_epochs, (int)epoch, ctx->d_stage_in, ctx->d_snodes, ctx->d_stage_out);
  else
    hipLaunchKernelGGL(k_sigma_r<false>, dim3((unsigned)n), dim3(256), 0, ctx->stream, ctx->cfg,
                       ctx->d_epochs, (int)epoch, ctx->d_stage_in, ctx->d_snodes, ctx->d_stage_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, ctx->d_stage_out, n * sizeof(double), hipMemcpyDeviceToHost,
                        ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return CHOMP_OK;
}

int chomp_y_nfw(chomp_ctx* ctx, size_t epoch, const double* ln_k, const double* mass, size_t n,
                double* out) {
  if (!ctx || !ln_k || !mass || !out || n == 0) return fail(ctx, CHOMP_ERR_ARG, "y_nfw: bad args");
  if (!ctx->have_halo || epoch >= ctx->n_epoch) return fail(ctx, CHOMP_ERR_STATE, "y_nfw before halo_setup");
  HIPCHK(hipSetDevice(ctx->device));
  int rc = ensure(ctx, &ctx->d_stage_in, &ctx->cap_in, n);
  if (rc) return rc;
  rc = ensure(ctx, &ctx->d_stage_in2, &ctx->cap_in2, n);
  if (rc) return rc;
  rc = ensure(ctx, &ctx->d_stage_out, &ctx->cap_out, n);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(ctx->d_stage_in, ln_k, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_stage_in2, mass, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_y_nfw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->d_epochs, (int)epoch, ctx->d_sici, ctx->d_stage_in, ctx->d_stage_in2,
                     (int)n, ctx->d_stage_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, ctx->d_stage_out, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return CHOMP_OK;
}

int chomp_eval(chomp_ctx* ctx, size_t epoch, int what, const double* x, size_t n, double* out,
               int mem) {
  if (!ctx || !x || !out || n == 0) return fail(ctx, CHOMP_ERR_ARG, "eval: bad args");
  if (what < 0 || what > CHOMP_EV_DELTA_K) return fail(ctx, CHOMP_ERR_ARG, "eval: unknown function");
  if (!ctx->have_epochs || epoch >= ctx->n_epoch) return fail(ctx, CHOMP_ERR_STATE, "eval: epoch");
  const bool needs_mass = what <= CHOMP_EV_BIAS_NU;
  const bool needs_halo = what >= CHOMP_EV_HOD_FIRST && what <= CHOMP_EV_CONCENTRATION;
  if (needs_mass && !ctx->have_mass) return fail(ctx, CHOMP_ERR_STATE, "eval before mass_setup");
  if (needs_halo && !ctx->have_halo) return fail(ctx, CHOMP_ERR_STATE, "eval before halo_setup");
  HIPCHK(hipSetDevice(ctx->device));
  const double* dx = x;
  double* dout = out;
  if (mem == CHOMP_HOST) {
    int rc = ensure(ctx, &ctx->d_stage_in, &ctx->cap_in, n);
    if (rc) return rc;
    rc = ensure(ctx, &ctx->d_stage_out, &ctx->cap_out, n);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_stage_in, x, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    dx = ctx->d_stage_in;
    dout = ctx->d_stage_out;
  }
  const TabLayout& L = ctx->L;
  unsigned gx = (unsigned)((n + 255) / 256);
  if (gx > 1024) gx = 1024;
  const size_t sh = (size_t)(L.NM + 8 * (L.NM - 1)) * sizeof(double);
  hipLaunchKernelGGL(k_eval, dim3(gx), dim3(256), sh, ctx->stream, L, ctx->d_epochs, (int)epoch,
                     ctx->d_tab, what, dx, (int)n, dout);
  HIPCHK(hipGetLastError());
  if (mem == CHOMP_HOST) {
    HIPCHK(hipMemcpyAsync(out, dout, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return CHOMP_OK;
}

int chomp_halofit_get(chomp_ctx* ctx, size_t epoch, double* out) {
  if (!ctx || !out) return fail(ctx, CHOMP_ERR_ARG, "halofit_get: bad args");
  if (!ctx->have_epochs || epoch >= ctx->n_epoch) return fail(ctx, CHOMP_ERR_STATE, "halofit_get: epoch");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  Epoch E;
  HIPCHK(hipMemcpy(&E, ctx->d_epochs + epoch, sizeof(Epoch), hipMemcpyDeviceToHost));
  std::memcpy(out, &E.hf_f1, CHOMP_HF_COUNT * sizeof(double));
  return CHOMP_OK;
}

int chomp_halofit_put(chomp_ctx* ctx, size_t epoch, const double* in) {
  if (!ctx || !in) return fail(ctx, CHOMP_ERR_ARG, "halofit_put: bad args");
  if (!ctx->have_epochs || epoch >= ctx->n_epoch) return fail(ctx, CHOMP_ERR_STATE, "halofit_put: epoch");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  char* dst = reinterpret_cast<char*>(ctx->d_epochs + epoch) + offsetof(Epoch, hf_f1);
  HIPCHK(hipMemcpy(dst, in, CHOMP_HF_COUNT * sizeof(double), hipMemcpyHostToDevice));
  ctx->have_halofit[epoch] = 1;
  return CHOMP_OK;
}

int chomp_get_scalars(chomp_ctx* ctx, size_t epoch, double* out) {
  if (!ctx || !out) return fail(ctx, CHOMP_ERR_ARG, "get_scalars: bad args");
  if (!ctx->have_epochs || epoch >= ctx->n_epoch) return fail(ctx, CHOMP_ERR_STATE, "get_scalars: epoch");
  HIPCHK(hipSetDevice(ctx->device));
  Epoch E;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(&E, ctx->d_epochs + epoch, sizeof(Epoch), hipMemcpyDeviceToHost));
  for (int i = 0; i < CHOMP_SC_COUNT; ++i) out[i] = 0.0;
  out[CHOMP_SC_Z] = E.z; out[CHOMP_SC_CHI] = E.chi; out[CHOMP_SC_GROWTH] = E.growth;
  out[CHOMP_SC_OMEGA_M] = E.omega_m_z; out[CHOMP_SC_OMEGA_L] = E.omega_l_z;
  out[CHOMP_SC_DELTA_C] = E.delta_c; out[CHOMP_SC_DELTA_V] = E.delta_v;
  out[CHOMP_SC_RHO_BAR] = E.rho_bar; out[CHOMP_SC_SIGMA_NORM] = E.sigma_norm;
  out[CHOMP_SC_LN_MASS_MIN] = E.ln_mass_min; out[CHOMP_SC_LN_MASS_MAX] = E.ln_mass_max;
  out[CHOMP_SC_NU_MIN] = E.nu_min; out[CHOMP_SC_NU_MAX] = E.nu_max;
  out[CHOMP_SC_M_STAR] = E.m_star; out[CHOMP_SC_F_NORM] = E.f_norm;
  out[CHOMP_SC_BIAS_NORM] = E.bias_norm; out[CHOMP_SC_N_BAR] = E.n_bar;
  out[CHOMP_SC_N_BAR_OVER_RHO_BAR] = E.n_bar_over_rho_bar;
  out[CHOMP_SC_N_SEARCH] = (double)E.n_search; out[CHOMP_SC_MF_DELTA_V] = E.mf_delta_v;
  out[CHOMP_SC_T_ALPHA] = E.t_alpha; out[CHOMP_SC_T_BETA] = E.t_beta;
  out[CHOMP_SC_T_GAMMA] = E.t_gamma; out[CHOMP_SC_T_PHI] = E.t_phi;
  out[CHOMP_SC_T_ETA] = E.t_eta; out[CHOMP_SC_GROWTH_NORM] = E.growth_norm;
  out[CHOMP_SC_DELTA_H] = E.delta_H; out[CHOMP_SC_HF_K_S] = E.hf_k_s;
  out[CHOMP_SC_HF_N_EFF] = E.hf_n_eff; out[CHOMP_SC_HF_C] = E.hf_C;
  return CHOMP_OK;
}

int chomp_get_table(chomp_ctx* ctx, size_t epoch, int table, double* out, size_t n) {
  if (!ctx || !out) return fail(ctx, CHOMP_ERR_ARG, "get_table: bad args");
  if (!ctx->have_mass || epoch >= ctx->n_epoch) return fail(ctx, CHOMP_ERR_STATE, "get_table before mass_setup");
  const TabLayout& L = ctx->L;
  int off = -1;
  size_t len = 0;
  unsigned need = 0;                 // knot family that must have been built
  bool need_hf = false;
  switch (table) {
    case CHOMP_TAB_LN_MASS: off = L.off_ln_mass; len = L.NM; break;
    case CHOMP_TAB_NU: off = L.off_nu; len = L.NM; break;
    case CHOMP_TAB_H_M: off = L.off_knot[F_HM]; len = L.NK; need = 1u << F_HM; break;
    case CHOMP_TAB_PP_MM: off = L.off_knot[F_PPMM]; len = L.NK; need = 1u << F_PPMM; break;
    case CHOMP_TAB_H_G: off = L.off_knot[F_HG]; len = L.NK; need = 1u << F_HG; break;
    case CHOMP_TAB_PP_GM: off = L.off_knot[F_PPGM]; len = L.NK; need = 1u << F_PPGM; break;
    case CHOMP_TAB_PP_GG: off = L.off_knot[F_PPGG]; len = L.NK; need = 1u << F_PPGG; break;
    case CHOMP_TAB_LEVELS:
      off = L.off_levels; len = 5 * (size_t)L.NK;
      if (!ctx->fam_mask) return fail(ctx, CHOMP_ERR_STATE, "get_table: no knot table built yet");
      break;
    case CHOMP_TAB_HF_LN_SIGMA2: off = L.off_hf_lns2; len = L.NK; need_hf = true; break;
    default: return fail(ctx, CHOMP_ERR_ARG, "get_table: unknown table");
  }
  if ((ctx->fam_mask & need) != need)
    return fail(ctx, CHOMP_ERR_STATE, "get_table: this knot table was not built (chomp_halo_setup)");
  if (need_hf && !ctx->have_halofit[epoch])
    return fail(ctx, CHOMP_ERR_STATE, "get_table: chomp_halofit_setup not called for this epoch");
  if (n != len) return fail(ctx, CHOMP_ERR_ARG, "get_table: length mismatch");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, ctx->d_tab + epoch * (size_t)L.stride + off, len * sizeof(double),
                   hipMemcpyDeviceToHost));
  return CHOMP_OK;
}

#include "chomp_capi_proj.inc"

}  // extern "C"
