"""Development aid: build build_exp/cell_stamps.so = the library with s_memtime stamps in the
level loop of romberg_group, recorded by a one-block 256-thread launch (chomp_cell with one
multipole) once chomp_debug_rs(1) armed them; `run` (GPU box) prints the per-level phases of
the deepest multipole of C4.  Not part of the product."""
import os, subprocess, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/cell_stamps.so")


def build():
    H = os.path.join(R, "chomp_amd/csrc/chomp_romberg.h"); C = os.path.join(R, "chomp_amd/csrc/chomp_capi.hip")
    h0, c0 = open(H).read(), open(C).read()
    s = h0

    def rep(old, new):
        nonlocal s
        assert old in s, old[:60]
        s = s.replace(old, new, 1)
    rep("namespace detail {\n// Integrands may take",
        "__device__ long long g_rs[256];\n__device__ int g_rs_on;\n"
        "#define RSTAMP(k) do { if (g_rs_on && gridDim.x == 1 && blockDim.x == 256 && threadIdx.x == 0 && (k) < 256) g_rs[(k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)\n"
        "namespace detail {\n// Integrands may take")
    rep("  double n = (double)(1L << (i0 - 1));\n  for (int i = i0; i <= divmax && !all_done; ++i) {\n    const double c_il = CHOMP_ROMBERG_C[i][cl];          // latency hidden by the nodes\n",
        "  double n = (double)(1L << (i0 - 1));\n  RSTAMP(0);\n  for (int i = i0; i <= divmax && !all_done; ++i) {\n    RSTAMP(4 * i);\n    const double c_il = CHOMP_ROMBERG_C[i][cl];          // latency hidden by the nodes\n")
    rep("    for (; j < numtosum; j += NT) {\n      double v[NF];\n      detail::call_f<F, NF>(f, lox + h * (double)j, v, i, j, 0);",
        "    RSTAMP(4 * i + 1);\n    for (; j < numtosum; j += NT) {\n      double v[NF];\n      detail::call_f<F, NF>(f, lox + h * (double)j, v, i, j, 0);")
    rep("    all_done = true;\n#pragma unroll\n    for (int q = 0; q < NF; ++q) {\n      const double S = group_sum<NW>(part[q], red, flip);\n      if (!done[q]) advance(q, i, S, n, c_il);\n      all_done = all_done && done[q];\n    }\n  }\n",
        "    RSTAMP(4 * i + 2);\n    all_done = true;\n#pragma unroll\n    for (int q = 0; q < NF; ++q) {\n      const double S = group_sum<NW>(part[q], red, flip);\n      if (!done[q]) advance(q, i, S, n, c_il);\n      all_done = all_done && done[q];\n    }\n    RSTAMP(4 * i + 3);\n  }\n  RSTAMP(1);\n")
    c = c0.replace('int chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {',
                   'int chomp_debug_rs(int on, long long* out, int n) {\n  if (out) return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chomp::g_rs), (size_t)n * sizeof(long long));\n'
                   '  return (int)hipMemcpyToSymbol(HIP_SYMBOL(chomp::g_rs_on), &on, sizeof(int));\n}\n\nint chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {')
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
    import contextlib, warnings
    import numpy, torch
    from chomp_amd import cosmology, correlation, halo, kernel, _lib
    d2r = numpy.pi / 180.0
    cm = cosmology.MultiEpoch(0.0, 5.0)
    with contextlib.redirect_stdout(sys.stderr):
        wa = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
        wb = kernel.WindowFunctionGalaxy(kernel.dNdzMagLim(0.0, 2.0, 2.0, 0.3, 2.0), cm)
        kern = kernel.Kernel(1e-6 * d2r, 100.0 * d2r, wa, wb, cm)
    h = halo.Halo(0.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        corr = correlation.Correlation(0.001, 1.0, kern, input_halo=h, power_spec="power_gg")
        ctx, code = corr._prepare()
    ell = torch.logspace(1, 4, 2048, dtype=torch.float64, device="cuda")
    L = _lib.lib()
    L.chomp_debug_rs.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    for which in (-1, 0, 1500):
        for _ in range(5):
            ctx.cell(code, 0, corr.D_z, ell[which:which + 1] if which >= 0 else ell[-1:])
        torch.cuda.synchronize()
        L.chomp_debug_rs(1, None, 0)
        ctx.cell(code, 0, corr.D_z, ell[which:which + 1] if which >= 0 else ell[-1:])
        torch.cuda.synchronize()
        L.chomp_debug_rs(0, None, 0)
        out = (ctypes.c_longlong * 256)()
        L.chomp_debug_rs(0, out, 256)
        a = numpy.array(out[:], dtype=numpy.int64)
        print("ell index", which, " (stamps in 100 MHz ticks x 10 = ns if s_memtime is the 100 MHz counter)")
        t0 = a[0]
        print("  loop start 0, loop end %d ticks" % (a[1] - t0))
        for i in range(8, 21):
            r = a[4 * i:4 * i + 4]
            if r[0] > 0 and r[0] >= t0:
                print("  level %2d: start %7d  batch %6d  tail %6d  sum+row %6d" % (i, r[0] - t0, r[1] - r[0], r[2] - r[1], r[3] - r[2]))
        # clear
    return 0


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
