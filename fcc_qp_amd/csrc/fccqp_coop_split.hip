// fccqp_coop_split.hip -- the two-launch form of the cooperative kernel (FCCQP_KERNEL_COOP_SPLIT) for the shapes of
// coop_shapes.def with more than 64 variables (the row-split ADMM layout of fccqp_coop_impl.h):
//   launch 1  coop::coop_kernel<SH, TIN, 1>     load, K_rho = U^T D U with the forward solve, Gram product: one QP per
//             CU (8 x 252 registers, 145 KB of LDS); leaves the ADMM waves' rows of rho G, x_c and diag(rho G) in a
//             workspace;
//   launch 2  coop::coop_admm_kernel<SH, WGS>   presolve by conjugate gradients, ADMM iterations, metrics, write-back:
//             four waves and 2.7 KB of LDS per QP, WGS = 2 or 3 QPs per CU, which fill each other's waits;
//   launch 3  coop::coop_kernel<SH, TIN, 2>     the whole solve again for the instances whose conjugate gradients
//             declined in launch 2 (W = I - rho G singular or ill-conditioned: the direct factorisation, its
//             singularity rule and the COD hand-off live there); every other workgroup leaves at once.
// A translation unit of its own (compile time; and the whole-solve kernels are sensitive to what else is compiled in
// their module, see fccqp_fused.hip).  Reference work replaced: FCCQP::Solve / DoADMM, /root/reference/src/fcc_qp.cpp:57-191.
#include "fccqp_coop_impl.h"

namespace fccqp {
namespace {
template <class SH>
bool matches(int n, int m, int nc, int s) {
  return n == SH::n && m == SH::m && nc == SH::nc && s == SH::s;
}
template <class SH>
hipError_t launch_split(const KernelArgs &a, hipStream_t stream, bool qa_f32, int wgs) {
  if constexpr (SH::kRowSplit) {
    const dim3 grid(a.batch), block(fused::kWave * SH::W), block2(fused::kWave * 4);
    if (qa_f32) hipLaunchKernelGGL((coop::coop_kernel<SH, float, 1>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((coop::coop_kernel<SH, double, 1>), grid, block, 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (wgs >= 3) hipLaunchKernelGGL((coop::coop_admm_kernel<SH, 3>), grid, block2, 0, stream, a);
    else hipLaunchKernelGGL((coop::coop_admm_kernel<SH, 2>), grid, block2, 0, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (!a.warm_start) {        // (conjugate gradients run on a cold start only)
      if (qa_f32) hipLaunchKernelGGL((coop::coop_kernel<SH, float, 2>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((coop::coop_kernel<SH, double, 2>), grid, block, 0, stream, a);
      e = hipGetLastError();
    }
    return e;
  } else {
    return hipErrorNotSupported;
  }
}
template <class SH>
long long split_doubles() {
  if constexpr (SH::kRowSplit) return coop::SplitWorkspace<SH>::kDoubles;
  else return 0;
}
}  // namespace

// doubles of workspace per instance the two-launch form of this shape needs; 0: the shape has none
long long coop_split_workspace_doubles(int n, int m, int nc, int s) {
#define FCCQP_COOP_SHAPE(N_, M_, NC_, S_, W_) \
  if (matches<coop::Shape<N_, M_, NC_, S_, W_>>(n, m, nc, s)) return split_doubles<coop::Shape<N_, M_, NC_, S_, W_>>();
#include "coop_shapes.def"
#undef FCCQP_COOP_SHAPE
  return 0;
}

hipError_t launch_coop_split(const KernelArgs &a, hipStream_t stream, bool qa_f32, int wgs_per_cu) {
  if (a.batch <= 0) return hipSuccess;
  if (!a.workspace || a.workspace_stride < coop_split_workspace_doubles(a.n, a.m, a.nc, a.s)) return hipErrorInvalidValue;
#define FCCQP_COOP_SHAPE(N_, M_, NC_, S_, W_) \
  if (matches<coop::Shape<N_, M_, NC_, S_, W_>>(a.n, a.m, a.nc, a.s)) return launch_split<coop::Shape<N_, M_, NC_, S_, W_>>(a, stream, qa_f32, wgs_per_cu);
#include "coop_shapes.def"
#undef FCCQP_COOP_SHAPE
  return hipErrorNotSupported;
}
}  // namespace fccqp
