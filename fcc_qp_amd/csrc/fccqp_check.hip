// fccqp_check.hip -- self-check of a finished solve: does the returned z satisfy the equality constraints?
//
// Every primal iterate of the path is the head of a KKT solve (reference /root/reference/src/fcc_qp.cpp:86-87 and, for
// the presolve, :174-178), so A_eq z = b_eq holds at exit to rounding whatever the ADMM iteration count -- unless the
// solve itself went wrong.  The kernels for 256 < N <= 512 (generic kernel, KKT matrix in a global workspace) once
// returned wrong iterates on some boxes for reasons of code generation; the cause was found
// and is guarded by code-object tests, but a silent wrong z is the one failure a drop-in must not have, so for those
// shapes (and on request for any shape: fccqp_set_self_check) this kernel runs behind every solve and raises
// FCCQP_FLAG_EQ_RESIDUAL on an instance whose z is not finite or leaves
//     |A_eq z - b_eq|_i  >  tol * (|b_eq,i| + sum_j |A_ij z_j|)      for some row i      (tol = 1e-6).
// One wavefront per instance, rows one after the other, lanes along a row (coalesced); HBM-bound: m n words per
// instance, a few per cent of what the solve itself reads and writes for these shapes.
#include "fccqp_device.h"

namespace fccqp {
namespace {

constexpr int kWave = 64;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

template <class TIN>
__global__ __launch_bounds__(kWave) void eq_residual_kernel(const KernelArgs a, const double tol) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  if (inst >= a.batch) return;
  const int n = a.n, m = a.m;
  const TIN *__restrict__ A = reinterpret_cast<const TIN *>(a.A) + (size_t)inst * m * n;
  const double *__restrict__ beq = a.b_eq + (size_t)inst * m;
  const double *__restrict__ z = a.x + (size_t)inst * n;
  bool bad = false;
  for (int j = lane; j < n; j += kWave) bad = bad || !(fabs(z[j]) <= 1.79769313486231570815e+308);   // NaN or infinity
  for (int i = 0; i < m; ++i) {
    double acc = 0.0, mag = 0.0;
    for (int j = lane; j < n; j += kWave) {
      const double t = (double)A[(size_t)i * n + j] * z[j];
      acc += t;
      mag += fabs(t);
    }
    acc = wave_sum_f64(acc);
    mag = wave_sum_f64(mag);
    const double b = beq[i];
    // (written so that a NaN on either side counts as a failure)
    bad = bad || !(fabs(acc - b) <= tol * (fabs(b) + mag));
  }
  const bool any = __builtin_amdgcn_ballot_w64(bad) != 0;
  if (lane == 0 && any) a.flags[inst] |= kFlagEqResidual;
}

}  // namespace

hipError_t launch_eq_residual_check(const KernelArgs &a, hipStream_t stream, bool qa_f32, double tol) {
  if (a.batch <= 0) return hipSuccess;
  if (qa_f32) hipLaunchKernelGGL((eq_residual_kernel<float>), dim3(a.batch), dim3(kWave), 0, stream, a, tol);
  else hipLaunchKernelGGL((eq_residual_kernel<double>), dim3(a.batch), dim3(kWave), 0, stream, a, tol);
  return hipGetLastError();
}

}  // namespace fccqp
