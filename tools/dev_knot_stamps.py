"""Development aid: build build_exp/knot_stamps.so = the library with s_memtime stamps at the
phase boundaries of k_halo_knots_fast (first knot each block draws); `run` (GPU box) prints the
mean phase durations on configs[2] (64 epochs, power_gm, Tinker10) and on one epoch.
Not part of the product."""
import os, subprocess, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(R, "build_exp/knot_stamps.so")
NB, NS = 2048, 24


def build():
    H = os.path.join(R, "chomp_amd/csrc/chomp_halo_kernels.h"); C = os.path.join(R, "chomp_amd/csrc/chomp_capi.hip")
    h0, c0 = open(H).read(), open(C).read()
    s = h0

    def rep(old, new):
        nonlocal s
        assert old in s, old[:70]
        s = s.replace(old, new, 1)
    rep("template <int LC, int NT>\n// (eight wavefronts per CU",
        "__device__ long long g_ks[%d * %d];\n#define KSTAMP(k) do { if (first_item && threadIdx.x == 0 && blockIdx.x < %d) g_ks[blockIdx.x * %d + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)\n"
        "template <int LC, int NT>\n// (eight wavefronts per CU" % (NB, NS, NB, NS))
    rep("  if (count == 0) return;          // nothing listed: no traffic on the queue head\n  for (;;) {\n",
        "  if (count == 0) return;          // nothing listed: no traffic on the queue head\n  bool first_item = true; int n_items = 0;\n  for (;;) {\n    first_item = (n_items++ == 0);\n")
    rep("    HaloLds H;\n    // (the weights of every deep level", "    KSTAMP(0);\n    HaloLds H;\n    // (the weights of every deep level")
    rep("      // ---- coarse samples: from the", "      __syncthreads(); KSTAMP(1);\n      // ---- coarse samples: from the")
    rep("      if (tid == 0) { n_rough_sh = 0; n_fine_sh = 0; n_seg_sh = 0; }\n      __syncthreads();\n",
        "      if (tid == 0) { n_rough_sh = 0; n_fine_sh = 0; n_seg_sh = 0; }\n      __syncthreads();\n      KSTAMP(2);\n")
    rep("      if (!R.all_done()) {\n        // ---- break points", "      KSTAMP(3);\n      if (!R.all_done()) {\n        // ---- break points")
    rep("            // ---- self-check: the same machinery one level up.", "            KSTAMP(4);\n            // ---- self-check: the same machinery one level up.")
    rep("      if (!literal) {\n        // ---- deeper levels, kDeepRound at a time",
        "      KSTAMP(5);\n      if (!literal) {\n        if (first_item && tid == 0 && blockIdx.x < %d) g_ks[blockIdx.x * %d + 22] = n_fine_sh;\n        // ---- deeper levels, kDeepRound at a time" % (NB, NS))
    rep("          // the break-point intervals: level lv0 + g has n0 << g nodes in each\n", "          if (lv0 == LC + 1) KSTAMP(10);\n          // the break-point intervals: level lv0 + g has n0 << g nodes in each\n")
    rep("          // one exchange for the 2 ng sums\n", "          if (lv0 == LC + 1) KSTAMP(11);\n          // one exchange for the 2 ng sums\n")
    rep("            flip ^= 1;\n          }\n#pragma unroll\n          for (int g = 0; g < kDeepRound; ++g)\n            if (g < ng && !R.all_done()) R.advance(lv0 + g, s0[g], s1[g]);", "            flip ^= 1;\n          }\n          if (lv0 == LC + 1) KSTAMP(12);\n#pragma unroll\n          for (int g = 0; g < kDeepRound; ++g)\n            if (g < ng && !R.all_done()) R.advance(lv0 + g, s0[g], s1[g]);")
    rep("            if (g < ng && !R.all_done()) R.advance(lv0 + g, s0[g], s1[g]);\n        }\n",
        "            if (g < ng && !R.all_done()) R.advance(lv0 + g, s0[g], s1[g]);\n          KSTAMP(5 + (lv0 - LC - 1) / kDeepRound + 1);\n        }\n")
    rep("    arrive(e, false);\n  }   // next item", "    KSTAMP(16);\n    if (first_item && tid == 0 && blockIdx.x < %d) g_ks[blockIdx.x * %d + 23] = lev[0] > lev[1] ? lev[0] : lev[1];\n    arrive(e, false);\n    KSTAMP(17);\n  }   // next item" % (NB, NS))
    c = c0.replace('int chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {',
                   'int chomp_debug_ks(long long* out, int n, int clear) {\n  if (clear) { static long long z[%d]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(chomp::g_ks), z, sizeof(z)); }\n'
                   '  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chomp::g_ks), (size_t)n * sizeof(long long));\n}\n\nint chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {' % (NB * NS))
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
    _l.LIB_PATH = SO          # (the instrumented build instead of the product library)
    sys.path.insert(0, R)
    import numpy, torch
    from chomp_amd import grid, _lib
    L = _lib.lib()
    L.chomp_debug_ks.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, ctypes.c_int]
    numpy.set_printoptions(linewidth=220, precision=1, suppress=True)
    for which, z, mf in (("power_gm", numpy.linspace(0.0, 1.5, 64), "tinker"), ("power_gg", numpy.array([0.3]), "st")):
        hg = grid.HaloGrid(z, mass_function=mf)
        for _ in range(3):
            hg.setup(which)
        torch.cuda.synchronize()
        L.chomp_debug_ks(None, 0, 1)
        hg.setup(which)
        torch.cuda.synchronize()
        out = (ctypes.c_longlong * (NB * NS))()
        L.chomp_debug_ks(out, NB * NS, 0)
        a = numpy.array(out[:], dtype=numpy.int64).reshape(NB, NS)
        a = a[a[:, 0] > 0]
        T = 2.4e3       # ticks per us (shader clock)
        print(which, mf, "blocks with a first knot:", len(a))
        names = ["stage", "coarse", "levels<=LC", "breaks+segments", "self-check"]
        for i, nm in enumerate(names):
            ok = a[:, i + 1] > 0
            d = (a[ok, i + 1] - a[ok, i]) / T
            print("  %-16s n %4d  mean %6.2f  max %6.2f us" % (nm, ok.sum(), d.mean() if ok.sum() else 0, d.max() if ok.sum() else 0))
        for lv in range(1, 5):
            ok = (a[:, 5 + lv] > 0) & (a[:, 4 + lv] > 0)
            if ok.sum():
                d = (a[ok, 5 + lv] - a[ok, 4 + lv]) / T
                print("  round %2d         n %4d  mean %6.2f  max %6.2f us   (nf mean %.1f)" % (lv, ok.sum(), d.mean(), d.max(), a[ok, 22].mean()))
        ok = (a[:, 10] > 0) & (a[:, 12] > 0) & (a[:, 6] > 0)
        if ok.sum():
            print("  inside round 1: stencil pass %.2f  node-by-node %.2f  exchange %.2f  rows %.2f us (mean)" % (
                ((a[ok, 10] - a[ok, 5]) / T).mean(), ((a[ok, 11] - a[ok, 10]) / T).mean(),
                ((a[ok, 12] - a[ok, 11]) / T).mean(), ((a[ok, 6] - a[ok, 12]) / T).mean()))
        ok = a[:, 16] > 0
        tot = (a[ok, 16] - a[ok, 0]) / T
        print("  whole knot       n %4d  mean %6.2f  max %6.2f us;  arrive %5.2f us" % (ok.sum(), tot.mean(), tot.max(), ((a[ok, 17] - a[ok, 16]) / T).mean()))
        print("  final level histogram 11..20:", numpy.bincount(a[ok, 23].astype(int), minlength=21)[11:])
        print("  launch span (first stamp -> last stamp) %.1f us" % ((a[:, 17].max() - a[:, 0].min()) / T))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
