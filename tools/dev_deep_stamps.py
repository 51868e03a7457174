"""Development aid: s_memtime stamps inside k_cell_deep (one listed multipole).  build / run."""
import os, subprocess, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/deep_stamps.so")


def build():
    H = os.path.join(R, "chomp_amd/csrc/chomp_proj_kernels.h"); C = os.path.join(R, "chomp_amd/csrc/chomp_capi.hip")
    h0, c0 = open(H).read(), open(C).read()
    s = h0

    def rep(old, new):
        nonlocal s
        assert old in s, old[:70]
        s = s.replace(old, new, 1)
    rep("constexpr int kCellDeepThreads = 512;",
        "constexpr int kCellDeepThreads = 512;\n__device__ long long g_ds[128];\n#define DSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_ds[(k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)")
    rep("  const int count = deep[0];\n  if ((int)blockIdx.x >= count) return;                  // (block-uniform)\n",
        "  DSTAMP(0);\n  const int count = deep[0];\n  if ((int)blockIdx.x >= count) return;                  // (block-uniform)\n  DSTAMP(1);\n")
    rep("  P.template finish_t<BAO>();\n  const double px0 = log(cfg.k_min), pdx = (log(cfg.k_max) - px0) / (double)kPTabN;\n  const double a = pd.chi_min, b = pd.chi_max;\n  int flip = 0;",
        "  DSTAMP(2);\n  P.template finish_t<BAO>();\n  const double px0 = log(cfg.k_min), pdx = (log(cfg.k_max) - px0) / (double)kPTabN;\n  const double a = pd.chi_min, b = pd.chi_max;\n  int flip = 0;\n  DSTAMP(3);")
    rep("    RombergResume R;\n    R.load(state + (size_t)il * kRombergDump, split, b - a, cfg.global_precision,\n           cfg.corr_precision);\n",
        "    DSTAMP(4);\n    RombergResume R;\n    R.load(state + (size_t)il * kRombergDump, split, b - a, cfg.global_precision,\n           cfg.corr_precision);\n    DSTAMP(5);\n")
    rep("      double part = 0.0;\n      long j = threadIdx.x;\n      for (; j + (U - 1) * (long)NT < numtosum; j += U * (long)NT) {",
        "      double part = 0.0;\n      long j = threadIdx.x;\n      DSTAMP(4 * lv);\n      for (; j + (U - 1) * (long)NT < numtosum; j += U * (long)NT) {")
    rep("      for (; j < numtosum; j += NT) {\n        double v[1];\n        if (!f.fast(",
        "      DSTAMP(4 * lv + 1);\n      for (; j < numtosum; j += NT) {\n        double v[1];\n        if (!f.fast(")
    rep("      R.advance(lv, group_sum<NW>(part, red, flip));\n",
        "      DSTAMP(4 * lv + 2);\n      R.advance(lv, group_sum<NW>(part, red, flip));\n      DSTAMP(4 * lv + 3);\n")
    rep("    if (threadIdx.x == 0) out[il] = R.value;\n  }\n}", "    if (threadIdx.x == 0) out[il] = R.value;\n    DSTAMP(6);\n  }\n}")
    c = c0.replace('int chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {',
                   'int chomp_debug_ds(long long* out, int n) {\n  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chomp::g_ds), (size_t)n * sizeof(long long));\n}\n\nint chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {')
    try:
        open(H, "w").write(s); open(C, "w").write(c)
        os.makedirs(os.path.join(R, "build_exp"), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-o", SO, "chomp_capi.hip"], cwd=os.path.join(R, "chomp_amd/csrc"))
    finally:
        open(H, "w").write(h0); open(C, "w").write(c0)
    print("built", SO)


def run():
    sys.path.insert(0, R)
    from chomp_amd import _lib as _l
    _l.LIB_PATH = SO          # (the instrumented build instead of the product library ...)
    _l.build = lambda *a, **k: SO      # (... which must not be rebuilt over it)
    sys.argv = [sys.argv[0]]
    sys.path.insert(0, os.path.join(R, "tools"))
    import cell_one      # 20 calls on the deepest multipole
    import numpy
    from chomp_amd import _lib
    L = _lib.lib()
    L.chomp_debug_ds.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    out = (ctypes.c_longlong * 128)()
    L.chomp_debug_ds(out, 128)
    a = numpy.array(out[:], dtype=numpy.int64)
    T = 2.4e3
    t0 = a[0]
    print("start 0 | past count %.2f | staged %.2f | finished %.2f | item %.2f | loaded %.2f | done %.2f us" % tuple((a[i] - t0) / T for i in (1, 2, 3, 4, 5, 6)))
    for lv in range(12, 21):
        r = a[4 * lv:4 * lv + 4]
        if r[0] > t0:
            print("  level %2d: at %.2f  batch loop %.2f  tail loop %.2f  sum+row %.2f us" % (lv, (r[0] - t0) / T, (r[1] - r[0]) / T, (r[2] - r[1]) / T, (r[3] - r[2]) / T))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
