// fccqp_fused.hip -- instantiations and launcher of the shape-specialised kernel (fccqp_fused_impl.h) for the
// shapes listed in fused_shapes.def (an X-macro list: one FCCQP_FUSED_SHAPE(n, m, nc, s) per line).
// Compiled twice: as is (Q / A_eq stored as double) and through fccqp_fused_f32.hip (as float).
#include "fccqp_fused_impl.h"

namespace fccqp {
// the three tile-edge cases of the template are always among the compiled shapes (tests)
static_assert(fused::Shape<60, 38, 12, 38>::kBorder == 2 && fused::Shape<12, 5, 6, 3>::kBorder == 1 && fused::Shape<12, 7, 6, 3>::kBorder == 3 &&
                  fused::Shape<12, 8, 6, 3>::kBorder == 0,
              "two bordered rows, one, identity padding");

namespace {
// One storage type per translation unit.  With the float and double instantiations in the same
// module the compiler emitted a measurably worse double kernel (more SGPR spill traffic in the
// ADMM loop, +3% run time) than when it is compiled alone, so fccqp_fused_f32.hip re-includes
// this file with FCCQP_FUSED_STORAGE_F32 defined and only contributes launch_fused_f32().
#ifdef FCCQP_FUSED_STORAGE_F32
using Storage = float;
#else
using Storage = double;
#endif

template <class SH>
bool matches(int n, int m, int nc, int s) {
  return n == SH::n && m == SH::m && nc == SH::nc && s == SH::s;
}
template <class SH>
hipError_t launch(const KernelArgs &a, hipStream_t stream) {
  KernelArgs k = a;
  k.m = one_wave_resident_workgroups();        // (staggered start of fused_kernel; the kernel's sizes are template parameters)
  // (a.inst_list: the caller's block -> instance order, served by an instantiation of its own)
  if (a.inst_list) hipLaunchKernelGGL((fused::fused_kernel<SH, Storage, false, true>), dim3(a.batch), dim3(fused::kWave), 0, stream, k);
  else hipLaunchKernelGGL((fused::fused_kernel<SH, Storage>), dim3(a.batch), dim3(fused::kWave), 0, stream, k);
  return hipGetLastError();
}
hipError_t launch_by_shape(const KernelArgs &a, hipStream_t stream) {
  if (a.batch <= 0) return hipSuccess;
#define FCCQP_FUSED_SHAPE(N_, M_, NC_, S_) \
  if (matches<fused::Shape<N_, M_, NC_, S_>>(a.n, a.m, a.nc, a.s)) return launch<fused::Shape<N_, M_, NC_, S_>>(a, stream);
#include "fused_shapes.def"
#undef FCCQP_FUSED_SHAPE
  return hipErrorNotSupported;
}
}  // namespace

#ifdef FCCQP_FUSED_STORAGE_F32
hipError_t launch_fused_f32(const KernelArgs &a, hipStream_t stream) { return launch_by_shape(a, stream); }
#else
hipError_t launch_fused_f32(const KernelArgs &a, hipStream_t stream);   // fccqp_fused_f32.hip

bool fused_supported(int n, int m, int nc, int s) {
#define FCCQP_FUSED_SHAPE(N_, M_, NC_, S_) \
  if (matches<fused::Shape<N_, M_, NC_, S_>>(n, m, nc, s)) return true;
#include "fused_shapes.def"
#undef FCCQP_FUSED_SHAPE
  return false;
}

bool fused_pointers_ok(const KernelArgs &) {
  // Q and A_eq are read one element per lane (issue_kkt_loads): natural alignment is all that is needed.
  // (Rounds 1-2c fetched them 16 bytes at a time and sent misaligned arrays to the generic kernel.)
  return true;
}

hipError_t launch_fused(const KernelArgs &a, hipStream_t stream, bool qa_f32, int split) {
  if (split != 0) return launch_fused_split(a, stream, qa_f32, split);     // fccqp_fused_split.hip
  return qa_f32 ? launch_fused_f32(a, stream) : launch_by_shape(a, stream);
}
#endif
}  // namespace fccqp
