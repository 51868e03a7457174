// chomp_probe.hip -- the translation unit of k_epoch_probe (compiled with machine LICM; the
// rest of the library without: chomp_amd/_lib.py).
#include <hip/hip_runtime.h>

#include "../../include/chomp_mi355x.h"
#include "chomp_probe_kernel.h"

namespace chomp {

void launch_epoch_probe(bool bao, unsigned n_epoch, hipStream_t stream, const chomp_config& cfg,
                        Epoch* epochs, double* search, const double* cand, const double* snodes,
                        double* probe, int* count, unsigned* status) {
  const dim3 grid(n_epoch, 2 * kProbes), block(64 * kInitNW);
#define CHOMP_PROBE(BAO, PHASE, GRID)                                                         \
  hipLaunchKernelGGL((k_epoch_probe<BAO, PHASE>), GRID, block, 0, stream, cfg, epochs, search, \
                     cand, snodes, probe, count, status)
  if (n_epoch >= 128) {            // a large batch: probed by the caller (k_epoch_probe<., 1, 1>,
                                   // instantiated in chomp_capi.hip, the unit without machine LICM,
                                   // capped at 128 registers: 120), certified here, behind the
                                   // kernel boundary
    if (bao) CHOMP_PROBE(true, 2, dim3(n_epoch)); else CHOMP_PROBE(false, 2, dim3(n_epoch));
  } else {
    if (bao) CHOMP_PROBE(true, 0, grid); else CHOMP_PROBE(false, 0, grid);
  }
#undef CHOMP_PROBE
}

}  // namespace chomp

#if defined(CHOMP_STAMPS) && CHOMP_STAMPS == 4
// (development builds only: the stamps of k_epoch_probe, tools/dev_probe_stamps4.py)
extern "C" int chomp_debug_ps(long long* out, int n, int clear) {
  if (clear) {
    static long long z[64 * 16 * chomp::kMStampSlots];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(chomp::g_ms), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chomp::g_ms), (size_t)n * sizeof(long long));
}
#endif
