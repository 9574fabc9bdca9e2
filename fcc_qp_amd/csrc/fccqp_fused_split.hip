// fccqp_fused_split.hip -- the two-launch form of the shape-specialised kernel (FCCQP_KERNEL_FUSED_SPLIT):
//   launch 1  fused::fused_kernel<SH, TIN, /*SPLIT=*/true>   P0 .. Pw (load, factorise, Gram product, presolve) at one
//             wave per SIMD; leaves each lane's row of rho G and x_c in a workspace, x0 in the warm-start state;
//   launch 2  fused::admm_kernel<SH, W>                      P4 + P5 (ADMM iterations, metrics, write-back) at W = 2
//             or 3 waves per SIMD.
// A translation unit of its own: the whole-solve kernel of fccqp_fused.hip is sensitive to what else is compiled
// in its module (see the note there).  Reference work replaced: FCCQP::DoADMM's loop, src/fcc_qp.cpp:79-110, and
// the epilogue of FCCQP::Solve, :184-186.
#include "fccqp_fused_impl.h"

namespace fccqp {
namespace {
template <class SH>
bool matches(int n, int m, int nc, int s) {
  return n == SH::n && m == SH::m && nc == SH::nc && s == SH::s;
}
template <class SH>
hipError_t launch(const KernelArgs &a, hipStream_t stream, bool qa_f32, int waves) {
  const dim3 grid(a.batch), block(fused::kWave);
  KernelArgs k = a;
  k.m = one_wave_resident_workgroups();        // (staggered start of fused_kernel, see KernelArgs)
  if (qa_f32) hipLaunchKernelGGL((fused::fused_kernel<SH, float, true>), grid, block, 0, stream, k);
  else hipLaunchKernelGGL((fused::fused_kernel<SH, double, true>), grid, block, 0, stream, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (waves >= 3) hipLaunchKernelGGL((fused::admm_kernel<SH, 3>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((fused::admm_kernel<SH, 2>), grid, block, 0, stream, a);
  return hipGetLastError();
}
}  // namespace

long long fused_split_workspace_doubles(int n) { return (long long)(n + 1) * fused::kWave; }

hipError_t launch_fused_split(const KernelArgs &a, hipStream_t stream, bool qa_f32, int waves) {
  if (a.batch <= 0) return hipSuccess;
  if (!a.workspace || a.workspace_stride < fused_split_workspace_doubles(a.n)) return hipErrorInvalidValue;
#define FCCQP_FUSED_SHAPE(N_, M_, NC_, S_) \
  if (matches<fused::Shape<N_, M_, NC_, S_>>(a.n, a.m, a.nc, a.s)) return launch<fused::Shape<N_, M_, NC_, S_>>(a, stream, qa_f32, waves);
#include "fused_shapes.def"
#undef FCCQP_FUSED_SHAPE
  return hipErrorNotSupported;
}
}  // namespace fccqp
