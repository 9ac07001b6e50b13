// Four-step path for transforms that do not fit one workgroup's LDS (N > 16384).  Placeholder: not
// implemented yet -- creation fails loudly so that nothing silently falls back.
#pragma once
#include <hip/hip_runtime.h>

namespace ksa {
struct SpecParams;
struct FourStep {
  int threads = 0, lds_bytes = 0, vgprs = 0;
};
inline int fourstep_create(FourStep&, int, int, int, int) { return 1; }
inline void fourstep_destroy(FourStep&) {}
inline int fourstep_run(FourStep&, const SpecParams&, int, hipStream_t, int) { return 1; }
}  // namespace ksa
