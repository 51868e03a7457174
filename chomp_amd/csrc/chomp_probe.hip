// chomp_probe.hip -- the translation unit of k_epoch_probe (compiled with machine LICM; the
// rest of the library without: chomp_amd/_lib.py).
#include <hip/hip_runtime.h>

#include "../../include/chomp_mi355x.h"
#include "chomp_probe_kernel.h"

namespace chomp {

void launch_epoch_probe(bool bao, unsigned n_epoch, hipStream_t stream, const chomp_config& cfg,
                        Epoch* epochs, double* search, const double* cand, const double* snodes,
                        double* probe, int* count, unsigned* status) {
  const dim3 grid(n_epoch, 2 * kProbes);
  if (bao)
    hipLaunchKernelGGL(k_epoch_probe<true>, grid, dim3(64 * kInitNW), 0, stream, cfg, epochs, search,
                       cand, snodes, probe, count, status);
  else
    hipLaunchKernelGGL(k_epoch_probe<false>, grid, dim3(64 * kInitNW), 0, stream, cfg, epochs, search,
                       cand, snodes, probe, count, status);
}

}  // namespace chomp
