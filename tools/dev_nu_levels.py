"""Development aid: s_memtime at every Romberg level of ONE integral of k_nu_table (epoch 57, mass
index 46 of configs[1]: among the longest).  build / run.  Not part of the product."""
import os, subprocess, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/nu_levels.so")


def build():
    H = os.path.join(R, "chomp_amd/csrc/chomp_romberg.h"); C = os.path.join(R, "chomp_amd/csrc/chomp_capi.hip")
    h0, c0 = open(H).read(), open(C).read()
    s = h0

    def rep(old, new):
        nonlocal s
        assert old in s, old[:60]
        s = s.replace(old, new, 1)
    rep("namespace detail {\n// Integrands may take",
        "__device__ long long g_rs[4 * 256];\n"
        "#define RSTAMP(k) do { if (gridDim.x == 50 && blockDim.x == 64 && gridDim.y == 64 && blockIdx.y == 57 && (blockIdx.x == 46 || blockIdx.x == 10 || blockIdx.x == 30) && threadIdx.x == 0 && (k) < 256) g_rs[(blockIdx.x == 46 ? 0 : blockIdx.x == 10 ? 256 : 512) + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)\n"
        "namespace detail {\n// Integrands may take")
    rep("  double n = (double)(1L << (i0 - 1));\n  for (int i = i0; i <= divmax && !all_done; ++i) {\n    const double c_il = CHOMP_ROMBERG_C[i][cl];          // latency hidden by the nodes\n",
        "  double n = (double)(1L << (i0 - 1));\n  RSTAMP(0);\n  for (int i = i0; i <= divmax && !all_done; ++i) {\n    RSTAMP(4 * i);\n    const double c_il = CHOMP_ROMBERG_C[i][cl];          // latency hidden by the nodes\n")
    rep("    all_done = true;\n#pragma unroll\n    for (int q = 0; q < NF; ++q) {\n      const double S = group_sum<NW>(part[q], red, flip);\n      if (!done[q]) advance(q, i, S, n, c_il);\n      all_done = all_done && done[q];\n    }\n  }\n",
        "    RSTAMP(4 * i + 2);\n    all_done = true;\n#pragma unroll\n    for (int q = 0; q < NF; ++q) {\n      const double S = group_sum<NW>(part[q], red, flip);\n      if (!done[q]) advance(q, i, S, n, c_il);\n      all_done = all_done && done[q];\n    }\n    RSTAMP(4 * i + 3);\n  }\n  RSTAMP(1);\n")
    c = c0.replace('int chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {',
                   'int chomp_debug_rs(long long* out, int n) {\n  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chomp::g_rs), (size_t)n * sizeof(long long));\n}\n\nint chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {')
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
    L.chomp_debug_rs.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    out = (ctypes.c_longlong * 1024)()
    L.chomp_debug_rs(out, 1024)
    T = 2.4e3
    for name, off in (("mass 46", 0), ("mass 10", 256), ("mass 30", 512)):
        a = numpy.array(out[off:off + 256], dtype=numpy.int64)
        t0 = a[0]
        print(name, "epoch 57: level loop %.2f us" % ((a[1] - t0) / T))
        for i in range(1, 21):
            r = a[4 * i:4 * i + 4]
            if r[0] >= t0 and r[0] > 0:
                print("  level %2d: at %6.2f  nodes %6.2f  sum+row %5.2f us" % (i, (r[0] - t0) / T, (r[2] - r[0]) / T, (r[3] - r[2]) / T))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
