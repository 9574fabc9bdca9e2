// fccqp_coop.hip -- instantiations and launcher of the cooperative kernel (fccqp_coop_impl.h: one workgroup of W
// wavefronts per QP) for the shapes listed in coop_shapes.def.  Both storage types of Q / A_eq (double, float) are
// compiled here.  Reference work replaced: FCCQP::Solve / DoADMM, /root/reference/src/fcc_qp.cpp:57-191.
#include "fccqp_coop_impl.h"

namespace fccqp {
namespace {
template <class SH>
bool matches(int n, int m, int nc, int s) {
  return n == SH::n && m == SH::m && nc == SH::nc && s == SH::s;
}
template <class SH>
hipError_t launch(const KernelArgs &a, hipStream_t stream, bool qa_f32) {
  const dim3 grid(a.batch), block(fused::kWave * SH::W);
  if (qa_f32) hipLaunchKernelGGL((coop::coop_kernel<SH, float>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((coop::coop_kernel<SH, double>), grid, block, 0, stream, a);
  return hipGetLastError();
}
}  // namespace

bool coop_supported(int n, int m, int nc, int s) {
#define FCCQP_COOP_SHAPE(N_, M_, NC_, S_, W_) \
  if (matches<coop::Shape<N_, M_, NC_, S_, W_>>(n, m, nc, s)) return true;
#include "coop_shapes.def"
#undef FCCQP_COOP_SHAPE
  return false;
}

// workgroups of this shape that are resident at once on a chip of `cus` compute units (LDS and wave slots)
int coop_resident_workgroups(int n, int m, int nc, int s, int cus) {
#define FCCQP_COOP_SHAPE(N_, M_, NC_, S_, W_)                                                    \
  if (matches<coop::Shape<N_, M_, NC_, S_, W_>>(n, m, nc, s)) {                                   \
    using SH = coop::Shape<N_, M_, NC_, S_, W_>;                                                  \
    const int by_lds = (160 * 1024) / (8 * SH::kLdsDoubles), by_waves = 4 / ((W_ + 3) / 4);       \
    return cus * (by_lds < by_waves ? (by_lds < 1 ? 1 : by_lds) : by_waves);                      \
  }
#include "coop_shapes.def"
#undef FCCQP_COOP_SHAPE
  return 0;
}

hipError_t launch_coop(const KernelArgs &a, hipStream_t stream, bool qa_f32) {
  if (a.batch <= 0) return hipSuccess;
#define FCCQP_COOP_SHAPE(N_, M_, NC_, S_, W_) \
  if (matches<coop::Shape<N_, M_, NC_, S_, W_>>(a.n, a.m, a.nc, a.s)) return launch<coop::Shape<N_, M_, NC_, S_, W_>>(a, stream, qa_f32);
#include "coop_shapes.def"
#undef FCCQP_COOP_SHAPE
  return hipErrorNotSupported;
}
}  // namespace fccqp
