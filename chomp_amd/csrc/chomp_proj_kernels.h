// chomp_proj_kernels.h -- projection-side state and kernels (placeholder until the
// MultiEpoch / window / kernel / correlation kernels land).
#pragma once
#include <hip/hip_runtime.h>
#include "chomp_math.h"
#include "chomp_romberg.h"

namespace chomp {
struct ProjState {
  bool ready = false;
};
inline void proj_free(ProjState&) {}
}  // namespace chomp
