// fccqp_fused.hip -- instantiations and launcher of the shape-specialised kernel
// (fccqp_fused_impl.h).  Shapes compiled in:
//   walking  FCCQP(60, 38, 12, 38)   reference harness shape, /root/reference/fcc_qp_test.py:77
//   mini     FCCQP(12, 5, 6, 3)      small shape that exercises the same template in the tests (N = 17: one
//                                    bordered row, like the walking shape's two)
//   padded   FCCQP(12, 8, 6, 3)      N = 20: four rows past the last full tile, more than the border takes --
//                                    the identity-padded last tile row of the same template
// Compiled twice: as is (Q / A_eq stored as double) and through fccqp_fused_f32.hip (as float).
#include "fccqp_fused_impl.h"

namespace fccqp {
using Walking = fused::Shape<60, 38, 12, 38>;
using Mini = fused::Shape<12, 5, 6, 3>;
using Padded = fused::Shape<12, 8, 6, 3>;
static_assert(Walking::kBorder == 2 && Mini::kBorder == 1 && Padded::kBorder == 0, "the three tile-edge cases");


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
  hipLaunchKernelGGL((fused::fused_kernel<SH, Storage>), dim3(a.batch), dim3(fused::kWave), 0, stream, a);
  return hipGetLastError();
}
hipError_t launch_by_shape(const KernelArgs &a, hipStream_t stream) {
  if (a.batch <= 0) return hipSuccess;
  if (matches<Walking>(a.n, a.m, a.nc, a.s)) return launch<Walking>(a, stream);
  if (matches<Mini>(a.n, a.m, a.nc, a.s)) return launch<Mini>(a, stream);
  if (matches<Padded>(a.n, a.m, a.nc, a.s)) return launch<Padded>(a, stream);
  return hipErrorNotSupported;
}
}  // namespace

#ifdef FCCQP_FUSED_STORAGE_F32
hipError_t launch_fused_f32(const KernelArgs &a, hipStream_t stream) { return launch_by_shape(a, stream); }
#else
hipError_t launch_fused_f32(const KernelArgs &a, hipStream_t stream);   // fccqp_fused_f32.hip

bool fused_supported(int n, int m, int nc, int s) {
  return matches<Walking>(n, m, nc, s) || matches<Mini>(n, m, nc, s) || matches<Padded>(n, m, nc, s);
}

bool fused_pointers_ok(const KernelArgs &) {
  // Q and A_eq are read one element per lane (issue_kkt_loads): natural alignment is all that is needed.
  // (Rounds 1-2c fetched them 16 bytes at a time and sent misaligned arrays to the generic kernel.)
  return true;
}

hipError_t launch_fused(const KernelArgs &a, hipStream_t stream, bool qa_f32) {
  return qa_f32 ? launch_fused_f32(a, stream) : launch_by_shape(a, stream);
}
#endif
}  // namespace fccqp
