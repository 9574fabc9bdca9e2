// fccqp_fused.hip -- shape-specialised fused kernels (placeholder: none compiled yet).
#include "fccqp_device.h"

namespace fccqp {
bool fused_supported(int, int, int, int) { return false; }
hipError_t launch_fused(const KernelArgs &, hipStream_t) { return hipErrorNotSupported; }
}  // namespace fccqp
