"""Development aid: build build_exp/stamps.so = the library with s_memtime stamps at the phase
boundaries of k_epoch_probe (read back by tools/stamps.py).  Not part of the product."""
import os, subprocess, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = os.path.join(R, "chomp_amd/csrc/chomp_mass_kernels.h"); C = os.path.join(R, "chomp_amd/csrc/chomp_capi.hip")
m0, c0 = open(M).read(), open(C).read()
s = m0
def rep(old, new, count=1):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, count)
rep('template <bool BAO>\n__global__ __launch_bounds__(64 * kInitNW) void k_epoch_probe(',
    '__device__ long long g_stamps[4096 * 8];\n#define STAMP(i) do { if (threadIdx.x == 0) g_stamps[((size_t)blockIdx.x * 8 + blockIdx.y) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)\ntemplate <bool BAO>\n__global__ __launch_bounds__(64 * kInitNW) void k_epoch_probe(')
rep('  if (fixed && !chi_role) return;\n  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),',
    '  if (fixed && !chi_role) return;\n  STAMP(0);\n  copy_doubles(reinterpret_cast<double*>(&E), reinterpret_cast<const double*>(&epochs[e]),')
rep('  {\n    // ---- this block\'s probe: candidate j - 2 + p of its side\n    const SidePlan plan = plan_side(E, lns, side, cand, &sh_j);',
    '  STAMP(1);\n  {\n    // ---- this block\'s probe: candidate j - 2 + p of its side\n    const SidePlan plan = plan_side(E, lns, side, cand, &sh_j);\n    STAMP(2);')
rep('  if (threadIdx.x == 0) {\n    __threadfence();               // results visible before the arrival is counted\n    last = atomicAdd(&count[e], 1) == 2 * kProbes - 1 ? 1 : 0;\n  }\n  __syncthreads();\n  if (!last) return;',
    '  STAMP(3);\n  if (threadIdx.x == 0) {\n    __threadfence();               // results visible before the arrival is counted\n    last = atomicAdd(&count[e], 1) == 2 * kProbes - 1 ? 1 : 0;\n  }\n  __syncthreads();\n  STAMP(4);\n  if (!last) return;')
rep('  __syncthreads();\n  for (int sd = 0; sd < 2; ++sd) {\n    if (!open_side[sd]) continue;                // block-uniform',
    '  __syncthreads();\n  STAMP(5);\n  for (int sd = 0; sd < 2; ++sd) {\n    if (!open_side[sd]) continue;                // block-uniform')
i = s.index('  STAMP(5);')
j = s.index('  copy_doubles(reinterpret_cast<double*>(&epochs[e]), reinterpret_cast<const double*>(&E),\n               kEpochDoubles);\n}', i)
s = s[:j] + '  STAMP(6);\n' + s[j:]
j = s.index('               kEpochDoubles);\n}', j) + len('               kEpochDoubles);\n')
s = s[:j] + '  STAMP(7);\n' + s[j:]
c = c0.replace('int chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {',
               'int chomp_debug_stamps(long long* out, int n) {\n  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chomp::g_stamps), (size_t)n * sizeof(long long));\n}\n\nint chomp_set_tuning(chomp_ctx* ctx, int what, long long value) {')
try:
    open(M, "w").write(s); open(C, "w").write(c)
    os.makedirs(os.path.join(R, "build_exp"), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-o", os.path.join(R, "build_exp/stamps.so"), "chomp_capi.hip"], cwd=os.path.join(R, "chomp_amd/csrc"))
finally:
    open(M, "w").write(m0); open(C, "w").write(c0)
print("built build_exp/stamps.so")
