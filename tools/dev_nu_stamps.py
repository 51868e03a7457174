"""Development aid: per-block duration of k_nu_table (s_memtime at entry and exit), by mass index,
on configs[1] (64 redshifts).  build / run.  Not part of the product."""
import os, subprocess, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/nu_stamps.so")


def build():
    H = os.path.join(R, "chomp_amd/csrc/chomp_mass_kernels.h"); C = os.path.join(R, "chomp_amd/csrc/chomp_capi.hip")
    h0, c0 = open(H).read(), open(C).read()
    s = h0

    def rep(old, new):
        nonlocal s
        assert old in s, old[:70]
        s = s.replace(old, new, 1)
    rep("template <bool BAO>\n__global__ __launch_bounds__(64) void k_nu_table(", "__device__ long long g_nu[64 * 64 * 2];\ntemplate <bool BAO>\n__global__ __launch_bounds__(64) void k_nu_table(")
    rep("  __shared__ Epoch E;\n  const int i = blockIdx.x, e = blockIdx.y;\n  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),\n               kEpochDoubles);\n  __syncthreads();\n  const double* snode = snodes + (size_t)E.cosmo_slot * kSigmaStride;",
        "  __shared__ Epoch E;\n  const int i = blockIdx.x, e = blockIdx.y;\n  const long long t_in = (long long)__builtin_amdgcn_s_memtime();\n  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),\n               kEpochDoubles);\n  __syncthreads();\n  const double* snode = snodes + (size_t)E.cosmo_slot * kSigmaStride;")
    rep("    if (!conv) atomicOr(&status[e], kStSigmaDivmax);     // scipy: AccuracyWarning, last row kept\n",
        "    if (!conv) atomicOr(&status[e], kStSigmaDivmax);     // scipy: AccuracyWarning, last row kept\n    if (e < 64 && i < 64) { g_nu[(e * 64 + i) * 2] = t_in; g_nu[(e * 64 + i) * 2 + 1] = (long long)__builtin_amdgcn_s_memtime(); }\n")
    c = c0.replace('int chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {',
                   'int chomp_debug_nu(long long* out, int n) {\n  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chomp::g_nu), (size_t)n * sizeof(long long));\n}\n\nint chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {')
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
    sys.path.insert(0, R)
    import numpy, torch
    from chomp_amd import grid, _lib
    hg = grid.HaloGrid(numpy.linspace(0.0, 1.5, 64))
    for _ in range(4):
        hg.setup("power_mm")
    torch.cuda.synchronize()
    L = _lib.lib()
    L.chomp_debug_nu.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    out = (ctypes.c_longlong * (64 * 64 * 2))()
    L.chomp_debug_nu(out, 64 * 64 * 2)
    a = numpy.array(out[:], dtype=numpy.int64).reshape(64, 64, 2)[:, :50]
    T = 2.4e3
    d = (a[:, :, 1] - a[:, :, 0]) / T
    t0 = a[:, :, 0].min()
    numpy.set_printoptions(linewidth=200, precision=1, suppress=True)
    print("kernel span %.1f us; block durations: mean %.1f  max %.1f us" % ((a[:, :, 1].max() - t0) / T, d.mean(), d.max()))
    print("mean duration by mass index (50):"); print(d.mean(axis=0))
    print("max duration by mass index:"); print(d.max(axis=0))
    print("start time of blocks (us after first): min/mean/max by mass index 0, 25, 49:", [(float((a[:, i, 0].min() - t0) / T), float((a[:, i, 0].mean() - t0) / T), float((a[:, i, 0].max() - t0) / T)) for i in (0, 25, 49)])
    print("mean duration by epoch (64):"); print(d.mean(axis=1))
    print("max duration by epoch:"); print(d.max(axis=1))
    print("the 12 longest: (epoch, mass index, us)", sorted(((float(d[e, i]), e, i) for e in range(64) for i in range(50)), reverse=True)[:12])
    print("histogram of durations (us, 5-us bins from 0):", numpy.histogram(d, bins=numpy.arange(0, 60, 5))[0])
    print("sum of durations %.0f us -> /1024 SIMDs = %.1f us of perfectly packed single-wave time" % (d.sum(), d.sum() / 1024))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
